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
// Deeper stacks use one LDS column per lane (k_combine_deep).
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

template <int N>
__device__ inline float pick(const float (&a)[N], int idx) {
    float r = a[0];
#pragma unroll
    for (int i = 1; i < N; ++i) r = (i == idx) ? a[i] : r;
    return r;
}

template <int NMAX>
__global__ __launch_bounds__(256) void k_combine(const float2* __restrict__ stack,
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
#pragma unroll
    for (int i = 0; i < NMAX; ++i) {
        bool ok = w[i] > 0.f;
        key[i] = ok ? v[i] : __builtin_inff();
        nv += ok ? 1 : 0;
        sw += w[i];
    }
    bitonic_sort<NMAX>(key);
    float med = 0.f;
    if (nv > 0) med = 0.5f * (pick<NMAX>(key, (nv - 1) >> 1) + pick<NMAX>(key, nv >> 1));
    if (kind == ZM_COMBINE_MEDIAN) {
        out_img[p] = med;
        out_wgt[p] = sw;
        return;
    }
    // CLIPPED
    const float amp = clip_ampfrac * fabsf(med);
#pragma unroll
    for (int i = 0; i < NMAX; ++i) {
        if (w[i] > 0.f) {
            float sig = rsqrtf(w[i]);
            if (fabsf(v[i] - med) <= clip_sigma * sig + amp) {
                s1 = fmaf(w[i], v[i], s1);
                s0 += w[i];
            }
        }
    }
    out_img[p] = s0 > 0.f ? s1 / s0 : 0.f;
    out_wgt[p] = s0;
}

// Deep stacks (n > 64): each lane owns an LDS column of n keys (stride 64 words,
// bank = lane, conflict free) and Shell-sorts it in place.
__global__ __launch_bounds__(64) void k_combine_deep(const float2* __restrict__ stack,
                                                     int64_t fstride, int n, int64_t npix,
                                                     int kind, float clip_sigma,
                                                     float clip_ampfrac,
                                                     float* __restrict__ out_img,
                                                     float* __restrict__ out_wgt) {
    extern __shared__ float col[];   // [n][64]
    const int lane = threadIdx.x;
    int64_t p = (int64_t)blockIdx.x * 64 + lane;
    const bool live = p < npix;
    int nv = 0;
    float sw = 0.f;
    for (int i = 0; i < n; ++i) {
        float2 s = live ? stack[(int64_t)i * fstride + p] : make_float2(0.f, 0.f);
        bool ok = s.y > 0.f;
        col[i * 64 + lane] = ok ? s.x : __builtin_inff();
        nv += ok ? 1 : 0;
        sw += ok ? s.y : 0.f;
    }
    // Shell sort (Ciura gaps), per lane, no cross-lane traffic
    const int gaps[8] = {701, 301, 132, 57, 23, 10, 4, 1};
    for (int g = 0; g < 8; ++g) {
        const int gap = gaps[g];
        if (gap >= n) continue;
        for (int i = gap; i < n; ++i) {
            float t = col[i * 64 + lane];
            int j = i;
            while (j >= gap && col[(j - gap) * 64 + lane] > t) {
                col[j * 64 + lane] = col[(j - gap) * 64 + lane];
                j -= gap;
            }
            col[j * 64 + lane] = t;
        }
    }
    if (!live) return;
    float med = 0.f;
    if (nv > 0) med = 0.5f * (col[((nv - 1) >> 1) * 64 + lane] + col[(nv >> 1) * 64 + lane]);
    if (kind == ZM_COMBINE_MEDIAN) {
        out_img[p] = med;
        out_wgt[p] = sw;
        return;
    }
    const float amp = clip_ampfrac * fabsf(med);
    float s0 = 0.f, s1 = 0.f;
    for (int i = 0; i < n; ++i) {
        float2 s = stack[(int64_t)i * fstride + p];
        if (s.y > 0.f) {
            float sig = rsqrtf(s.y);
            if (fabsf(s.x - med) <= clip_sigma * sig + amp) {
                s1 = fmaf(s.y, s.x, s1);
                s0 += s.y;
            }
        }
    }
    out_img[p] = s0 > 0.f ? s1 / s0 : 0.f;
    out_wgt[p] = s0;
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
        dim3 b2(64, 1, 1), g2((unsigned)((npix + 63) / 64), 1, 1);
        size_t shmem = (size_t)n * 64 * sizeof(float);
        if (shmem > 65536)
            ZM_HIP(hipFuncSetAttribute((const void*)k_combine_deep,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        hipLaunchKernelGGL(k_combine_deep, g2, b2, shmem, ctx->stream, stack, frame_stride, n,
                           npix, kind, clip_sigma, clip_ampfrac, out_img, out_wgt);
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
