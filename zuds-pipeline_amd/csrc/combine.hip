// Per-pixel stack combine on gfx950: WEIGHTED / AVERAGE / MEDIAN / CLIPPED over a
// resident stack of resampled {value, weight} planes, and AND / OR for masks.
//
// Replaces the co-addition loop of the SWarp runs launched from
// zuds/coadd.py:133,156 (COMBINE_TYPE CLIPPED, CLIP_SIGMA 4.0, CLIP_AMPFRAC 0.3:
// zuds/astromatic/makecoadd/default.swarp:24-31; AND: mask.swarp:25; OR:
// zuds/swarp.py:141).  Arithmetic conventions: oracle/combine.py.
//
// Layout: stack float2 [n][frame_stride] (frame_stride >= npix), one thread per
// output pixel, the n samples of a pixel live in registers (n <= 64) and the
// median comes from a fully unrolled bitonic network; lanes read consecutive
// pixels of one frame, so every load is a coalesced 512-B wave transaction.
// Deeper stacks (<= 512) spread a pixel over 2, 4 or 8 lanes (k_combine_wide).
#include "zm_internal.h"

template <int N>
__device__ inline void bitonic_sort(float (&a)[N]) {
#pragma unroll
    for (int k = 2; k <= N; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const int l = i ^ j;
                if (l > i) {
                    const bool up = ((i & k) == 0);
                    float lo = fminf(a[i], a[l]), hi = fmaxf(a[i], a[l]);
                    a[i] = up ? lo : hi;
                    a[l] = up ? hi : lo;
                }
            }
        }
    }
}

// The median of the valid samples without a data-dependent pick: an invalid sample's key is an infinity, minus
// and plus by turns (the first invalid sample gets minus), so that ceil(m / 2) of the m pads sort in front of the
// valid keys and the rest behind them.  The valid samples' middle then sits at fixed positions of the sorted
// array: N / 2 - 1 and N / 2 for an even count, N / 2 (twice) for an odd one - the elements (nv - 1) >> 1 and
// nv >> 1 of the valid keys in both cases, i.e. the value oracle/combine.py takes.  (The select chain over a
// lane-varying index that used to fetch them was turned into a stack array by the compiler: the sorted keys
// written to scratch, 136 B per pixel and launch of HBM writes)
__device__ __forceinline__ float pad_key(bool ok, float v, unsigned& par) {
    const float pad = __uint_as_float(0x7f800000u | (par << 31));     // par = 1: -inf, 0: +inf
    par ^= ok ? 0u : 1u;
    return ok ? v : pad;
}

template <int NMAX>
__global__ __launch_bounds__(256, NMAX <= 32 ? 4 : 2) void k_combine(const float2* __restrict__ stack,
                                                 int64_t fstride, int n, int64_t npix, int kind,
                                                 float clip_sigma, float clip_ampfrac,
                                                 float* __restrict__ out_img,
                                                 float* __restrict__ out_wgt, int partial) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    float v[NMAX], w[NMAX];
#pragma unroll
    for (int i = 0; i < NMAX; ++i) {
        float2 s = make_float2(0.f, 0.f);
        if (i < n) s = stack[(int64_t)i * fstride + p];
        v[i] = s.x;
        w[i] = s.y > 0.f ? s.y : 0.f;
    }
    float s0 = 0.f, s1 = 0.f;
    if (kind == ZM_COMBINE_WEIGHTED || kind == ZM_COMBINE_AVERAGE) {
        float sw = 0.f;
#pragma unroll
        for (int i = 0; i < NMAX; ++i) {
            float ww = (kind == ZM_COMBINE_WEIGHTED) ? w[i] : (w[i] > 0.f ? 1.f : 0.f);
            s1 = fmaf(ww, v[i], s1);
            s0 += ww;
            sw += w[i];
        }
        if (partial) {
            out_img[p] = s1;
            out_wgt[p] = s0;
        } else {
            out_img[p] = s0 > 0.f ? s1 / s0 : 0.f;
            out_wgt[p] = sw;
        }
        return;
    }
    float key[NMAX];
    int nv = 0;
    float sw = 0.f;
    unsigned par = 1u;
#pragma unroll
    for (int i = 0; i < NMAX; ++i) {
        const bool ok = w[i] > 0.f;
        key[i] = pad_key(ok, v[i], par);
        nv += ok ? 1 : 0;
        sw += w[i];
    }
    bitonic_sort<NMAX>(key);
    float med = 0.f;
    if (nv > 0) med = 0.5f * (((nv & 1) ? key[NMAX / 2] : key[NMAX / 2 - 1]) + key[NMAX / 2]);
    if (kind == ZM_COMBINE_MEDIAN) {
        out_img[p] = med;
        out_wgt[p] = sw;
        return;
    }
    // CLIPPED
    const float amp = clip_ampfrac * fabsf(med);
    // (branch-free: the sums of the kept samples in frame order, the same operations as the nested conditions)
#pragma unroll
    for (int i = 0; i < NMAX; ++i) {
        float wi = w[i];
        // (opaque, and behind the median: carried over from the key loop a validity predicate is held in a scalar
        // register pair across the sort; hoisted above it, every sample's sigma costs a register)
        asm volatile("" : "+v"(wi) : "v"(med), "v"(s0));
        // (an invalid sample - weight 0, sigma infinite - passes the test and adds exact zeros: one predicate)
        const float sig = rsqrtf(wi), vi = wi > 0.f ? v[i] : 0.f;
        const bool keep = fabsf(vi - med) <= clip_sigma * sig + amp;
        const float t1 = fmaf(wi, vi, s1), t0 = s0 + wi;
        s1 = keep ? t1 : s1;
        s0 = keep ? t0 : s0;
        __builtin_amdgcn_sched_barrier(0);          // (one sample at a time: hoisted, the 32 - 64 sigmas cost a register each)
    }
    out_img[p] = s0 > 0.f ? s1 / s0 : 0.f;
    out_wgt[p] = s0;
}

// The bitonic network over the 64 LPP keys of a pixel held by LPP lanes (element index of register r
// of lane `sub`: e = 64 sub + r), expanded at compile time level by level (K) and step by step (J).
template <int LPP, int K, int J>
__device__ __forceinline__ void wide_step(float (&key)[64], const int sub) {
    constexpr int NL = 64, PPW = 64 / LPP;
    if constexpr (J >= NL) {
        // the partner lives in the lane J / 64 subs away
        const bool up = ((sub * NL) & K) == 0, lower = ((sub * NL) & J) == 0;
        const bool keepmin = lower == up;
#pragma unroll
        for (int r = 0; r < NL; ++r) {
            const float other = __shfl_xor(key[r], (J / NL) * PPW);
            key[r] = keepmin ? fminf(key[r], other) : fmaxf(key[r], other);
        }
    } else {
        const bool upl = ((sub * NL) & K) == 0;                 // the lane's part of the direction (K >= 64)
#pragma unroll
        for (int r = 0; r < NL; ++r) {
            if ((r ^ J) > r) {
                const bool up = K >= NL ? upl : ((r & K) == 0);
                const float lo = fminf(key[r], key[r ^ J]), hi = fmaxf(key[r], key[r ^ J]);
                key[r] = up ? lo : hi;
                key[r ^ J] = up ? hi : lo;
            }
        }
    }
    if constexpr (J > 1) wide_step<LPP, K, J / 2>(key, sub);
}
template <int LPP, int K>
__device__ __forceinline__ void wide_level(float (&key)[64], const int sub) {
    wide_step<LPP, K, K / 2>(key, sub);
    if constexpr (K < 64 * LPP) wide_level<LPP, 2 * K>(key, sub);
}

// Deep stacks (64 < n <= 512: the row bands of a multi-GPU CLIPPED / MEDIAN stack hold every frame of
// the stack, 256 at BASELINE config 4): LPP = 2, 4 or 8 lanes share a pixel, each with 64 of its
// samples in registers - lane (sub, px) of a wave holds frames sub, sub + LPP, ... of pixel px, so
// a load instruction still reads whole 128-B lines (64 / LPP consecutive pixels of LPP frames).  The
// bitonic network runs over the 64 LPP keys of a pixel: steps with a partner distance below 64 are
// compare-exchanges between registers of a lane, the others exchange registers with the lane that
// holds the partner (one cross-lane read each).  (Its predecessor kept a column of n keys per lane
// in LDS and Shell-sorted it: 64 KB of LDS per wave at n = 256, two waves per CU, 29.6 ms for the
// 256 x 384 x 3072 band of an 8-GPU stack against 0.7 ms for the same samples at n = 32.)
template <int LPP>
__global__ __launch_bounds__(256, 2) void k_combine_wide(const float2* __restrict__ stack, int64_t fstride, int n,
                                                      int64_t npix, int kind, float clip_sigma,
                                                      float clip_ampfrac, float* __restrict__ out_img,
                                                      float* __restrict__ out_wgt) {
    constexpr int NL = 64, PPW = 64 / LPP, N = NL * LPP;
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int px = lane % PPW, sub = lane / PPW;
    const int64_t p = wave * PPW + px;
    const bool live = p < npix;                         // (no early exit: every lane takes part in the exchanges)
    float v[NL], w[NL];
#pragma unroll
    for (int r = 0; r < NL; ++r) {
        const int f = r * LPP + sub;
        float2 s = make_float2(0.f, 0.f);
        if (live && f < n) s = stack[(int64_t)f * fstride + p];
        v[r] = s.x;
        w[r] = s.y > 0.f ? s.y : 0.f;
    }
    // the pads of the pixel's invalid samples alternate over the whole pixel (pad_key): lane `sub` starts with the
    // parity of the invalid samples of the lanes before it
    int nv = 0;
    float sw = 0.f;
#pragma unroll
    for (int r = 0; r < NL; ++r) {
        nv += w[r] > 0.f ? 1 : 0;
        sw += w[r];
    }
    unsigned par = 1u;
#pragma unroll
    for (int s = 0; s < LPP - 1; ++s) {
        const int o = __shfl(NL - nv, s * PPW + px);
        par ^= (s < sub) ? ((unsigned)o & 1u) : 0u;
    }
    float key[NL];
#pragma unroll
    for (int r = 0; r < NL; ++r) key[r] = pad_key(w[r] > 0.f, v[r], par);
#pragma unroll
    for (int o = PPW; o < 64; o <<= 1) {                 // over the lanes of the pixel, fixed order
        nv += __shfl_xor(nv, o);
        sw += __shfl_xor(sw, o);
    }
    wide_level<LPP, 2>(key, sub);
    float med = 0.f;
    {
        // sorted positions N / 2 - 1 and N / 2: the last register of lane LPP / 2 - 1, the first of lane LPP / 2
        const float m1 = __shfl(key[NL - 1], (LPP / 2 - 1) * PPW + px), m2 = __shfl(key[0], (LPP / 2) * PPW + px);
        if (nv > 0) med = 0.5f * (((nv & 1) ? m2 : m1) + m2);
    }
    if (kind == ZM_COMBINE_MEDIAN) {
        if (live && sub == 0) {
            out_img[p] = med;
            out_wgt[p] = sw;
        }
        return;
    }
    // CLIPPED
    const float amp = clip_ampfrac * fabsf(med);
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int r = 0; r < NL; ++r) {
        float wi = w[r];
        asm volatile("" : "+v"(wi) : "v"(med), "v"(s0));
        // (an invalid sample - weight 0, sigma infinite - passes the test and adds exact zeros: one predicate)
        const float sig = rsqrtf(wi), vi = wi > 0.f ? v[r] : 0.f;
        const bool keep = fabsf(vi - med) <= clip_sigma * sig + amp;
        const float t1 = fmaf(wi, vi, s1), t0 = s0 + wi;
        s1 = keep ? t1 : s1;
        s0 = keep ? t0 : s0;
        __builtin_amdgcn_sched_barrier(0);          // (one sample at a time: hoisted, the 32 - 64 sigmas cost a register each)
    }
#pragma unroll
    for (int o = PPW; o < 64; o <<= 1) {
        s1 += __shfl_xor(s1, o);
        s0 += __shfl_xor(s0, o);
    }
    if (live && sub == 0) {
        out_img[p] = s0 > 0.f ? s1 / s0 : 0.f;
        out_wgt[p] = s0;
    }
}

// WEIGHTED / AVERAGE for any depth: running sums, no sample storage
__global__ __launch_bounds__(256) void k_combine_sum(const float2* __restrict__ stack,
                                                     int64_t fstride, int n, int64_t npix,
                                                     int kind, float* __restrict__ out_img,
                                                     float* __restrict__ out_wgt, int partial) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    float s0 = 0.f, s1 = 0.f, sw = 0.f;
    for (int i = 0; i < n; ++i) {
        float2 s = stack[(int64_t)i * fstride + p];
        float w = s.y > 0.f ? s.y : 0.f;
        float ww = (kind == ZM_COMBINE_WEIGHTED) ? w : (w > 0.f ? 1.f : 0.f);
        s1 = fmaf(ww, s.x, s1);
        s0 += ww;
        sw += w;
    }
    if (partial) {
        out_img[p] = s1;
        out_wgt[p] = s0;
    } else {
        out_img[p] = s0 > 0.f ? s1 / s0 : 0.f;
        out_wgt[p] = sw;
    }
}

int zm_launch_combine(zm_ctx* ctx, int n, const float2* stack, int64_t frame_stride,
                      int64_t npix, int kind, float clip_sigma, float clip_ampfrac,
                      float* out_img, float* out_wgt, int partial) {
    ZM_CHECK(n >= 1, "combine: empty stack");
    ZM_CHECK(kind == ZM_COMBINE_WEIGHTED || kind == ZM_COMBINE_AVERAGE ||
             kind == ZM_COMBINE_MEDIAN || kind == ZM_COMBINE_CLIPPED,
             "combine: unknown COMBINE_TYPE %d", kind);
    ZM_CHECK(!partial || kind == ZM_COMBINE_WEIGHTED || kind == ZM_COMBINE_AVERAGE,
             "combine: partial sums exist only for WEIGHTED / AVERAGE");
    dim3 blk(256, 1, 1), grd((unsigned)((npix + 255) / 256), 1, 1);
    zm_scope_timer t(ctx, "combine");
    if (kind == ZM_COMBINE_WEIGHTED || kind == ZM_COMBINE_AVERAGE) {
        hipLaunchKernelGGL(k_combine_sum, grd, blk, 0, ctx->stream, stack, frame_stride, n, npix,
                           kind, out_img, out_wgt, partial);
        ZM_HIP(hipGetLastError());
        return 0;
    }
#define ZM_COMBINE_CASE(NM)                                                              \
    hipLaunchKernelGGL(k_combine<NM>, grd, blk, 0, ctx->stream, stack, frame_stride, n,   \
                       npix, kind, clip_sigma, clip_ampfrac, out_img, out_wgt, partial)
    if (n <= 4) ZM_COMBINE_CASE(4);
    else if (n <= 8) ZM_COMBINE_CASE(8);
    else if (n <= 16) ZM_COMBINE_CASE(16);
    else if (n <= 32) ZM_COMBINE_CASE(32);
    else if (n <= 64) ZM_COMBINE_CASE(64);
    else {
        ZM_CHECK(n <= 512, "combine: stack depth %d > 512 not supported", n);
        const int lpp = n <= 128 ? 2 : n <= 256 ? 4 : 8;
        const int64_t waves = (npix + 64 / lpp - 1) / (64 / lpp);
        dim3 g2((unsigned)((waves + 3) / 4), 1, 1);
#define ZM_WIDE_CASE(L) hipLaunchKernelGGL(k_combine_wide<L>, g2, blk, 0, ctx->stream, stack, frame_stride, n, \
                                           npix, kind, clip_sigma, clip_ampfrac, out_img, out_wgt)
        if (lpp == 2) ZM_WIDE_CASE(2);
        else if (lpp == 4) ZM_WIDE_CASE(4);
        else ZM_WIDE_CASE(8);
#undef ZM_WIDE_CASE
    }
#undef ZM_COMBINE_CASE
    ZM_HIP(hipGetLastError());
    return 0;
}

// ---- masks -------------------------------------------------------------------
// acc (init: AND -> 0xFFFFFFFF, OR -> 0) op= m where the frame covers the pixel
// (m == -1 marks "not covered"); finalize turns an untouched AND accumulator to 0
// and writes the coverage plane.
__global__ void k_mask_accum(int32_t* __restrict__ acc, const int32_t* __restrict__ m,
                             int64_t npix, int kind, int first) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    int32_t v = m[p];
    int32_t a = first ? -1 : acc[p];       // -1: nothing covered yet
    if (v != -1) {
        if (a == -1) a = v;
        else a = (kind == ZM_MASK_AND) ? (a & v) : (a | v);
    }
    acc[p] = a;
}

__global__ void k_mask_finalize(int32_t* __restrict__ acc, float* __restrict__ cov,
                                int64_t npix) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    int32_t a = acc[p];
    bool none = (a == -1);
    acc[p] = none ? 0 : a;
    if (cov) cov[p] = none ? 0.f : 1.f;
}

int zm_launch_mask_accum(zm_ctx* ctx, int32_t* acc, const int32_t* m, int64_t npix, int kind,
                         int first) {
    dim3 blk(256, 1, 1), grd((unsigned)((npix + 255) / 256), 1, 1);
    hipLaunchKernelGGL(k_mask_accum, grd, blk, 0, ctx->stream, acc, m, npix, kind, first);
    ZM_HIP(hipGetLastError());
    return 0;
}

int zm_launch_mask_finalize(zm_ctx* ctx, int32_t* acc, float* cov, int64_t npix) {
    dim3 blk(256, 1, 1), grd((unsigned)((npix + 255) / 256), 1, 1);
    hipLaunchKernelGGL(k_mask_finalize, grd, blk, 0, ctx->stream, acc, cov, npix);
    ZM_HIP(hipGetLastError());
    return 0;
}

// ---- planes ---------------------------------------------------------------------
__global__ void k_split_pairs(const float2* __restrict__ src, int64_t npix,
                              float* __restrict__ a, float* __restrict__ b) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    float2 s = src[p];
    if (a) a[p] = s.x;
    if (b) b[p] = s.y;
}

int zm_launch_split_pairs(zm_ctx* ctx, const float2* src, int64_t npix, float* a, float* b) {
    dim3 blk(256, 1, 1), grd((unsigned)((npix + 255) / 256), 1, 1);
    hipLaunchKernelGGL(k_split_pairs, grd, blk, 0, ctx->stream, src, npix, a, b);
    ZM_HIP(hipGetLastError());
    return 0;
}

__global__ void k_finalize_weighted(float* __restrict__ s1, const float* __restrict__ s0,
                                    int64_t npix) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    float w = s0[p];
    s1[p] = w > 0.f ? s1[p] / w : 0.f;
}

extern "C" int zm_coadd_finalize_dev(zm_ctx* ctx, float* s1_to_img, const float* s0,
                                     int64_t npix) {
    ZM_CHECK(ctx && s1_to_img && s0, "zm_coadd_finalize_dev: null argument");
    dim3 blk(256, 1, 1), grd((unsigned)((npix + 255) / 256), 1, 1);
    hipLaunchKernelGGL(k_finalize_weighted, grd, blk, 0, ctx->stream, s1_to_img, s0, npix);
    ZM_HIP(hipGetLastError());
    return 0;
}
