// placeholder until the subtraction kernels land
#include "zm_internal.h"
extern "C" void zm_hp_params_default(zm_hp_params* p) { if (p) memset(p, 0, sizeof(*p)); }
extern "C" int zm_subtract(zm_ctx*, const float*, const float*, const float*, const float*,
                           const uint8_t*, int, int, const zm_hp_params*, float*, float*,
                           zm_hp_info*) { zm_set_error("zm_subtract: not built yet"); return 4; }
extern "C" int zm_subtract_dev(zm_ctx*, const float*, const float*, const float*, const float*,
                               const uint8_t*, int, int, const zm_hp_params*, float*, float*,
                               zm_hp_info*) { zm_set_error("zm_subtract_dev: not built yet"); return 4; }
extern "C" int zm_median_mad(zm_ctx*, const float*, const int32_t*, int64_t, double*, double*) {
    zm_set_error("zm_median_mad: not built yet"); return 4; }
