// Robust statistics of the C-ABI: exact median and 1.4826 x MAD of the unmasked
// pixels (quick_background_estimate, zuds/utils.py:32-53), by a three-pass
// radix select on order-preserving integer keys (11 + 11 + 10 bits).
#include "zm_internal.h"

__device__ inline uint32_t f2key(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

static inline float key2f(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// mode 0: v = img[p]; mode 1: v = |img[p] - centre| (float32 arithmetic, as numpy)
__global__ __launch_bounds__(256) void k_radix_hist(const float* __restrict__ img,
                                                    const int32_t* __restrict__ mask, int64_t n,
                                                    int mode, float centre, uint32_t prefix,
                                                    uint32_t prefix_mask, int shift, int nbins,
                                                    unsigned int* __restrict__ hist) {
    __shared__ unsigned int lh[2048];
    for (int k = threadIdx.x; k < nbins; k += 256) lh[k] = 0;
    __syncthreads();
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n; p += (int64_t)gridDim.x * 256) {
        if (mask && mask[p] != 0) continue;
        float v = img[p];
        if (!(v == v)) continue;
        if (mode == 1) v = fabsf(v - centre);
        uint32_t key = f2key(v);
        if ((key & prefix_mask) != prefix) continue;
        atomicAdd(&lh[(key >> shift) & (nbins - 1)], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nbins; k += 256)
        if (lh[k]) atomicAdd(&hist[k], lh[k]);
}

// k-th smallest (0-based) of the selected values; *count receives their number
static int radix_select(zm_ctx* ctx, const float* img, const int32_t* mask, int64_t n, int mode,
                        float centre, int64_t kth, float* out, int64_t* count) {
    unsigned int* d_hist = nullptr;
    ZM_TRY(ctx->get("rs_hist", sizeof(unsigned int) * 2048, (void**)&d_hist));
    unsigned int h[2048];
    uint32_t prefix = 0, pmask = 0;
    const int shifts[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
    int grid = (int)std::min<int64_t>((n + 255) / 256, 2048);
    if (grid < 1) grid = 1;
    int64_t k = kth;
    for (int pass = 0; pass < 3; ++pass) {
        int nb = 1 << bits[pass];
        ZM_HIP(hipMemsetAsync(d_hist, 0, sizeof(unsigned int) * 2048, ctx->stream));
        hipLaunchKernelGGL(k_radix_hist, dim3(grid), dim3(256), 0, ctx->stream, img, mask, n, mode,
                           centre, prefix, pmask, shifts[pass], nb, d_hist);
        ZM_HIP(hipGetLastError());
        ZM_HIP(hipMemcpyAsync(h, d_hist, sizeof(unsigned int) * nb, hipMemcpyDeviceToHost, ctx->stream));
        ZM_HIP(hipStreamSynchronize(ctx->stream));
        int64_t tot = 0;
        for (int b = 0; b < nb; ++b) tot += h[b];
        if (pass == 0) {
            if (count) *count = tot;
            if (tot == 0) { *out = 0.f; return 0; }
            if (k < 0) k = 0;       // caller asked for the count only
            if (k >= tot) k = tot - 1;
        }
        int b = 0;
        int64_t run = 0;
        for (; b < nb; ++b) {
            if (run + h[b] > k) break;
            run += h[b];
        }
        k -= run;
        prefix |= (uint32_t)b << shifts[pass];
        pmask |= (uint32_t)(nb - 1) << shifts[pass];
    }
    *out = key2f(prefix);
    return 0;
}

static int median_of(zm_ctx* ctx, const float* img, const int32_t* mask, int64_t n, int mode,
                     float centre, float* med, int64_t* count) {
    int64_t cnt = 0;
    float a = 0.f, b = 0.f;
    ZM_TRY(radix_select(ctx, img, mask, n, mode, centre, -1, &a, &cnt));
    if (count) *count = cnt;
    if (cnt == 0) { *med = 0.f; return 0; }
    ZM_TRY(radix_select(ctx, img, mask, n, mode, centre, (cnt - 1) / 2, &a, nullptr));
    b = a;
    if ((cnt & 1) == 0) ZM_TRY(radix_select(ctx, img, mask, n, mode, centre, cnt / 2, &b, nullptr));
    *med = 0.5f * (a + b);
    return 0;
}

extern "C" int zm_median_mad_dev(zm_ctx* ctx, const float* img, const int32_t* mask, int64_t n,
                                 double* out_median, double* out_mad_sigma) {
    ZM_CHECK(ctx && img && out_median && out_mad_sigma, "zm_median_mad_dev: null argument");
    ZM_CHECK(n > 0, "zm_median_mad_dev: empty image");
    ZM_HIP(hipSetDevice(ctx->device));
    float med = 0.f, mad = 0.f;
    int64_t cnt = 0;
    ZM_TRY(median_of(ctx, img, mask, n, 0, 0.f, &med, &cnt));
    ZM_CHECK(cnt > 0, "zm_median_mad: every pixel is masked");
    ZM_TRY(median_of(ctx, img, mask, n, 1, med, &mad, nullptr));
    *out_median = med;
    *out_mad_sigma = 1.4826 * (double)mad;
    return 0;
}

extern "C" int zm_median_mad(zm_ctx* ctx, const float* img, const int32_t* mask, int64_t n,
                             double* out_median, double* out_mad_sigma) {
    ZM_CHECK(ctx && img && out_median && out_mad_sigma, "zm_median_mad: null argument");
    ZM_CHECK(n > 0, "zm_median_mad: empty image");
    ZM_HIP(hipSetDevice(ctx->device));
    float* d_img = nullptr;
    int32_t* d_mask = nullptr;
    ZM_TRY(ctx->get("h_img", (size_t)n * 4, (void**)&d_img));
    ZM_HIP(hipMemcpyAsync(d_img, img, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    if (mask) {
        ZM_TRY(ctx->get("h_mask", (size_t)n * 4, (void**)&d_mask));
        ZM_HIP(hipMemcpyAsync(d_mask, mask, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    return zm_median_mad_dev(ctx, d_img, d_mask, n, out_median, out_mad_sigma);
}
