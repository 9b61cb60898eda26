// Internal definitions shared by the libzudsmi translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include <stdexcept>

#include "../../include/zudsmi.h"

#define ZM_BIGVAR 1e30f      // variance of a bad pixel (finite: 0 * BIG == 0)
#define ZM_BADVAR_TEST 1e14f // |interpolated variance| above this => a bad tap was hit
#define ZM_SNAP 1e-5f        // SWarp: |frac| < 1e-5 => delta kernel

void zm_set_error(const char* fmt, ...);

// Developer switches (ablations, phase clocks, A / B forms that lost) are read from the environment only in a
// -DZM_DEV build (ZM_HIPCC_FLAGS=-DZM_DEV python -m zuds-pipeline_amd.build --force); the shipped library reads the
// handful of switches that select between forms it actually carries (README.md) and nothing else.
#ifdef ZM_DEV
#define ZM_DEVENV(name) getenv(name)
#define ZM_DEV_BUILD 1
#else
#define ZM_DEVENV(name) ((const char*)nullptr)
#define ZM_DEV_BUILD 0
#endif

#define ZM_HIP(call)                                                         \
    do {                                                                     \
        hipError_t e_ = (call);                                              \
        if (e_ != hipSuccess) {                                              \
            zm_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                         __FILE__, __LINE__);                                \
            return 1;                                                        \
        }                                                                    \
    } while (0)

#define ZM_CHECK(cond, ...)                                                  \
    do {                                                                     \
        if (!(cond)) {                                                       \
            zm_set_error(__VA_ARGS__);                                       \
            return 2;                                                        \
        }                                                                    \
    } while (0)

#define ZM_TRY(expr)                                                         \
    do {                                                                     \
        int r_ = (expr);                                                     \
        if (r_) return r_;                                                   \
    } while (0)

struct zm_timer_slot {
    double total_ms = 0.0;
    int64_t launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

struct zm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = true;
    // grow-only named scratch buffers on the device
    std::map<std::string, std::pair<void*, size_t>> scratch;
    // pinned host staging
    std::map<std::string, std::pair<void*, size_t>> pinned;
    hipStream_t aux = nullptr;                 // second stream (zm_ctx_aux: made on first use): box-OR planes, lattices, item headers, batch convolutions
    std::vector<hipEvent_t> sync_events;       // cross-stream ordering (no timing)
    // set by zm_launch_prep when it also produced the box-OR plane of a mask (scratch slot
    // "mask_box"): zm_launch_resample then skips its own k_mask_box launch
    const void* box_ready_for = nullptr;
    int box_ready_nt = 0;
    // How many contexts run subtractions on this GPU at the same time (zm_ctx_set_share): the
    // fused Cholesky keeps every workgroup of a launch resident behind in-kernel barriers, so
    // concurrent jobs split the CUs between them instead of each claiming all of them.
    int share = 1;
    // per-context launch plans of the subtraction kernels (function attributes, resident grid):
    // per context, not per process - contexts may sit on different devices
    int hp_wg_cap = 0;
    bool hp_rset = false, hp_bset = false;
    // a subtraction that returned before its convolution had run (zm_hp_params.async_info): what zm_subtract_info
    // needs to finish the job
    hipEvent_t hp_done = nullptr;
    // the words the rejection rounds signal the host through (hotpants.hip: hp_round_signal)
    unsigned long long *hp_sig_h = nullptr, *hp_sig_d = nullptr;
    unsigned hp_seq = 0;
    bool hp_pending = false;
    int hp_pend_rounds = 0, hp_pend_retries = 0, hp_pend_nreg = 0, hp_pend_nunk = 0;
    std::map<int, size_t> hp_set_max;          // LDS opt-in of k_hp_apply<half width>
    std::vector<double> hp_filt_host;          // the 1-D filter table the device copy (scratch slot "hp_filt") holds
    const void* hp_filt_dev = nullptr;
    // item headers of a fused coadd made ahead of the launch, on the second stream (zm_launch_fused_headers_early)
    bool ff_pre_valid = false;
    int ff_pre_nfr = 0, ff_pre_onx = 0, ff_pre_ony = 0, ff_pre_lds = 0;
    bool ff_pre_own = false;
    // SWarp's own edge / mask conventions as options (zm_ctx_set_conventions; csrc/resample_opts.hip)
    int edge = 0;                              // ZM_EDGE_ZERO / ZM_EDGE_TRUNCATE
    int mask_resample = 0;                     // ZM_MASKRES_OR / ZM_MASKRES_LANCZOS_ROUND
    int ff_last_form = 0;                      // the fused kernel the last zm_launch_coadd_fused ran: 1 _dma, 2 _own (zm_ctx_query)
    bool bk_stats_set = false, bk_filter_set = false;   // LDS opt-in of the background kernels
    // recorded behind the last k_mesh_stats_fast launch of a stack (round 6): the box-OR pre-pass of a fused coadd waits
    // for it and runs beside the small kernels that follow the statistics instead of beside the statistics themselves
    hipEvent_t bk_stats_event = nullptr;
    bool bk_stats_event_valid = false;
    bool timing = false;
    std::string timing_only;                   // non-empty: only this scope is timed
    std::map<std::string, zm_timer_slot> timers;
    std::vector<hipEvent_t> event_pool;

    int get(const char* name, size_t bytes, void** out);
    int get_pinned(const char* name, size_t bytes, void** out);
    void release_all();
};

// RAII-ish helper: record start/stop events around a launch when timing is on.
struct zm_scope_timer {
    zm_ctx* ctx;
    const char* name;
    hipEvent_t a = nullptr, b = nullptr;
    zm_scope_timer(zm_ctx* c, const char* n);
    ~zm_scope_timer();
};

static inline int zm_div_up(int a, int b) { return (a + b - 1) / b; }

// ---- device-side plans -------------------------------------------------------
#define ZM_LATTICE_STEP 16

struct zm_map_params {   // everything a lattice node needs, fp64
    zm_wcs wout;
    zm_wcs win;
    double rot[9];       // in-frame axes expressed in the out-frame basis
};

// host helpers (wcs_host.cpp)
void zm_wcs_frame(const zm_wcs* w, double fr[9]);
void zm_make_map(const zm_wcs* wout, const zm_wcs* win, zm_map_params* mp);
void zm_map_point(const zm_map_params* mp, double xo, double yo, double* xi, double* yi);
double zm_pixel_area(const zm_wcs* w, double x, double y);

// kernels / launchers (each in its own .hip)
int zm_launch_lattice(zm_ctx* ctx, const zm_map_params* mp, int lnx, int lny, double2* lat_dev);
int zm_launch_lattice_batch(zm_ctx* ctx, const zm_map_params* mp_host, int n, int lnx, int lny,
                            double2* lat_dev, hipEvent_t after = nullptr);
int zm_launch_prep(zm_ctx* ctx, const float* img, const float* wgt, int nx, int ny,
                   const float* bknodes, int nbx, int nby, int mesh,
                   const float* var_scale_dev, float wthresh, float2* dst, int spitch,
                   const int32_t* mask_for_box = nullptr, int box_nt = 0, uint16_t* mbox_out = nullptr,
                   int mbox_pitch = 0);
struct zm_ff {                       // one input frame of a fused coadd (device memory; read through the scalar cache)
    const float* img;                // raw planes: staged with background, variance and threshold applied on the way
    const float* wgt;                // ... or NULL (unit weights)
    const float4* ytab;              // y part of the background spline per (row, mesh column) (k_bk_rows), or NULL
    const float4* xtab;              // ... and the x weights per pixel column (k_bk_cols); set with ytab
    const float* vscale;             // device scalar: variance scale (RESCALE_WEIGHTS), or NULL
    const float2* src;               // prepped {value, variance} plane: frames that cannot be staged raw (footprints
                                     // beyond the LDS tile, BACK_SIZE not a multiple of 8, ZM_FF_RAW=0), else NULL
    const double2* lat;              // lattice of this frame
    const void* mask;                // raw mask (int32, or int16 when mask16) or NULL
    const uint16_t* mbox;            // box-OR plane of the mask
    const float* bk;                 // spline nodes (4 planes [nby][nbx]) or NULL: the per-tap path of the generic code
    const int* mboxflag;             // device word: != 0 when the box-OR plane holds ZM_BOX_RAW entries (NULL: assume so)
    int nx, ny, spitch, nbx, nby, ytp;
    float invmesh, wthresh, fscale, fscale2;
    int vec_ok, mpitch;              // mpitch: pixels per row of the box-OR plane (a multiple of 4)
    int mask16, pad_;                // the raw mask is an int16 plane (ZM_MASKTYPE_I16)
};
// jobs of the pre-pass of a fused coadd (resample.hip: k_bk_rows, k_mask_box_batch)
struct zm_bkrows {
    const float* bk;                 // spline nodes of the frame (4 planes [nby][nbx])
    float4* out;                     // [ny][ytp]
    int nbx, nby, ny, ytp;
    float invmesh;
    int nx;
    float4* xout;                    // [nx]: the x weights of every pixel column (k_bk_cols)
};
struct zm_boxjob {
    const void* m;                   // int32 plane, or int16 when is16
    uint16_t* B;
    int* rawflag;                    // set when an entry defers to the raw mask (or NULL)
    int nx, ny, pitch, is16;
};
// the fused coadd's tile rows and LDS capacity in staged pixels (resample.hip: FT_H, FF_LDS_CAP)
void zm_fused_geometry(int* tile_h, int* lds_cap);
int zm_launch_fused_prepass(zm_ctx* ctx, const zm_bkrows* rows, int nrows);
int zm_launch_mask_boxes(zm_ctx* ctx, const zm_boxjob* boxes, int nboxes, hipEvent_t after, hipEvent_t* joined,
                         hipEvent_t after2 = nullptr);
int zm_launch_fused_headers_early(zm_ctx* ctx, const zm_ff* frames_host, int nfr, int lnx, int lny, int onx, int ony,
                                  int lds_elems, bool fits_own);
int zm_launch_coadd_fused(zm_ctx* ctx, const zm_ff* frames_host, int nfr, int lnx, int lny, int onx, int ony,
                          int lds_elems, int combine, int mask_kind, float* out_img, float* out_wgt,
                          int32_t* out_mask, float* out_cov, int partial, int32_t* unmasked_out,
                          float2* stack = nullptr, int64_t fstride = 0, bool fits_own = false);
int zm_get_lanczos_table(zm_ctx* ctx, const float** out);
int zm_frame_background(zm_ctx* ctx, const float* img, const float* wgt, int nx, int ny,
                        int mesh, int fsize, float wthresh, int mode0, int nmode,
                        float** nodes_dev, float** stats_dev, int* nbx_out, int* nby_out,
                        const char* slot, int index = 0, int count = 1);
int zm_batch_stats(zm_ctx* ctx, int nf, const float* const* imgs, const float* const* wgts, int nx,
                   int ny, int mesh, float wthresh, int mode0, int nmode, const char* slot, int index0,
                   int count, int nslot);
int zm_batch_filter(zm_ctx* ctx, int nf, int nx, int ny, int mesh, int fsize, int nmode,
                    const char* slot, int index0, int count, int nslot);
int zm_frame_products(zm_ctx* ctx, int nx, int ny, int mesh, const char* slot, int index, int count,
                      int nslot, float** nodes_dev, float** stats_dev, int* nbx_out, int* nby_out);
int zm_batch_var_scale(zm_ctx* ctx, int nf, const float* stats, float* out);
int zm_frame_stats(zm_ctx* ctx, const float* img, const float* wgt, int nx, int ny, int mesh,
                   float wthresh, int mode0, int nmode, const char* slot, int index, int count);
int zm_frame_filter(zm_ctx* ctx, int nx, int ny, int mesh, int fsize, int nmode, float** nodes_dev,
                    float** stats_dev, int* nbx_out, int* nby_out, const char* slot, int index,
                    int count);
int zm_get_sync_events(zm_ctx* ctx, int n, hipEvent_t** out);
hipStream_t zm_ctx_aux(zm_ctx* ctx);          // the context's second stream, created on first use (may return nullptr)
int zm_launch_var_scale(zm_ctx* ctx, const float* bstats, const float* vstats, float* out);
int zm_launch_resample(zm_ctx* ctx, const float2* src, int nx, int ny, int spitch,
                       const double2* lat, int lnx, int lny, int kernel, float fscale,
                       float2* dst, int onx, int ony, int lds_elems, const int32_t* mask,
                       int32_t* macc, int mop, int mkind, int mfirst,
                       float* plane_a = nullptr, float* plane_b = nullptr, float pair_scale = 0.f);
int zm_launch_prep_pair(zm_ctx* ctx, const float* a, const float* b, int nx, int ny, float2* dst, int spitch);
int zm_launch_resample_mask(zm_ctx* ctx, const int32_t* mask, int nx, int ny,
                            const double2* lat, int lnx, int lny, int kernel,
                            int32_t* dst, int onx, int ony, int32_t fill);
int zm_launch_combine(zm_ctx* ctx, int n, const float2* stack, int64_t frame_stride,
                      int64_t npix, int kind, float clip_sigma, float clip_ampfrac,
                      float* out_img, float* out_wgt, int partial);
int zm_launch_resample_rim(zm_ctx* ctx, const float2* src, int nx, int ny, int spitch, const double2* lat, int lnx,
                           int lny, int kernel, float fscale, float2* dst, float* plane_a, float* plane_b, int onx,
                           int ony, const int32_t* mask, int32_t* macc, int mop, int mkind);
int zm_launch_resample_mask_opts(zm_ctx* ctx, const int32_t* mask, int nx, int ny, const double2* lat, int lnx, int lny,
                                 int kernel, int32_t* dst, int onx, int ony, int32_t fill, int lanczos, int trunc);
int zm_launch_mask_accum(zm_ctx* ctx, int32_t* acc, const int32_t* m, int64_t npix, int kind,
                         int first);
int zm_launch_mask_finalize(zm_ctx* ctx, int32_t* acc, float* cov, int64_t npix);
int zm_launch_split_pairs(zm_ctx* ctx, const float2* src, int64_t npix, float* a, float* b);
int zm_launch_mask_widen(zm_ctx* ctx, const int16_t* in, int64_t n, int32_t* out);
