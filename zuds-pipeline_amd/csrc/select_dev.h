// Shared by the two forms of the exact median / MAD select (api_subtract.hip: the three-pass radix select;
// select_bracket.hip: the sample-bracketed select of round 6).  quick_background_estimate, zuds/utils.py:32-53.
#pragma once
#include "zm_internal.h"

#define ZM_RS_MAXIMG 4

// order-preserving integer key of a float (every NaN sorts above +inf or below -inf by its sign bit; NaN pixels
// never reach a key - they are not valid - but |v - centre| may be one: it is counted where it sorts)
__device__ inline uint32_t f2key(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ inline float key2f_dev(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

struct rs_image {
    const float* img;
    const int32_t* mask;
};

struct rs_batch {
    rs_image im[ZM_RS_MAXIMG];
    int64_t n;
};

// select_bracket.hip: median, 1.4826 MAD and count of each image -> out_dev[3 * nimg] (device), enqueued on the
// context's stream; `vbits` as in median_mad_batch (the validity bit plane, or nullptr).  Requires 16-byte aligned
// planes and n >= ZM_RS2_MIN_N.
#define ZM_RS2_MIN_N (1 << 20)
int zm_rs2_median_mad(zm_ctx* ctx, int nimg, const rs_batch& B, unsigned long long* d_vbits, double* out_dev);
