// Per-pixel bookkeeping kernels of the device-resident pipelines: weight <-> rms
// maps, mask algebra, pedestal.  They restate numpy one-liners of the reference
// (cited per entry point) so that a coadd -> subtract chain never leaves HBM.
#include <algorithm>
#include "zm_internal.h"

#define EW_GRID(n) dim3((unsigned)(((n) + 255) / 256)), dim3(256), 0, ctx->stream

// rms = 1 / sqrt(w); BIG_RMS where the pixel is bad or w <= 0
// (CalibratableImageBase.rms_image, zuds/image.py:173-208)
__global__ void k_rms_from_weight(const float* __restrict__ w, const uint8_t* __restrict__ bad,
                                  int64_t n, float big, float* __restrict__ out) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    // (a weight that carries no information gets BIG_RMS too - also on a pixel the mask does not flag, where the
    // reference's numpy gives 1 / sqrt(0) = inf: callers that must reproduce that put it back, subtraction.py)
    float ww = w[p];
    bool b = !(ww > 0.f) || (bad && bad[p]);
    out[p] = b ? big : 1.0f / sqrtf(ww);
}

// w = 1 / rms^2; 0 where bad or data >= satur (zuds/image.py:136-171)
__global__ void k_weight_from_rms(const float* __restrict__ rms, const uint8_t* __restrict__ bad,
                                  const float* __restrict__ img, float satur, int64_t n,
                                  float* __restrict__ out) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    float r = rms[p];
    bool b = (bad && bad[p]) || (img && satur > 0.f && img[p] >= satur);
    out[p] = b ? 0.f : 1.0f / (r * r);
}

// The "false weight map" of a background run without weights (zuds/sextractor.py:80-96): 1, and 0 where the mask
// carries a bad bit or - raw science frames - inside the 10-pixel border; beside it the boolean bad-pixel map of the
// mask alone (MaskImageBase.boolean, zuds/mask.py:42-72).  T: int16 (a ZTF mask as its file holds it) or int32.
template <typename T>
__global__ void k_false_weight(const T* __restrict__ m, int32_t badsum, int border, int nx, int ny,
                               float* __restrict__ out_w, uint8_t* __restrict__ out_bpm) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (int64_t)nx * ny) return;
    const int y = (int)(p / nx), x = (int)(p - (int64_t)y * nx);
    const bool bad = ((int32_t)m[p] & badsum) != 0;
    const bool edge = x < border || x >= nx - border || y < border || y >= ny - border;
    if (out_w) out_w[p] = (bad || edge) ? 0.f : 1.f;
    if (out_bpm) out_bpm[p] = bad ? 1 : 0;
}

extern "C" int zm_false_weight_dev(zm_ctx* ctx, const void* mask, int mask_type, int32_t badsum, int border,
                                   int nx, int ny, float* out_wgt, uint8_t* out_bpm) {
    ZM_CHECK(ctx && mask && nx > 0 && ny > 0 && border >= 0 && (out_wgt || out_bpm), "zm_false_weight_dev: bad argument");
    ZM_CHECK(mask_type == ZM_MASKTYPE_I32 || mask_type == ZM_MASKTYPE_I16, "zm_false_weight_dev: unknown mask_type %d", mask_type);
    ZM_HIP(hipSetDevice(ctx->device));
    const int64_t n = (int64_t)nx * ny;
    if (mask_type == ZM_MASKTYPE_I16)
        hipLaunchKernelGGL(k_false_weight<int16_t>, EW_GRID(n), (const int16_t*)mask, badsum, border, nx, ny, out_wgt, out_bpm);
    else
        hipLaunchKernelGGL(k_false_weight<int32_t>, EW_GRID(n), (const int32_t*)mask, badsum, border, nx, ny, out_wgt, out_bpm);
    ZM_HIP(hipGetLastError());
    return 0;
}

// out_or = a | b; out_bpm = ((a | b) & badsum) > 0
// (zuds/subtraction.py:135-142; MaskImageBase.boolean zuds/mask.py:42-72)
__global__ void k_mask_bad(const int32_t* __restrict__ a, const int32_t* __restrict__ b,
                           int32_t badsum, int64_t n, int32_t* __restrict__ out_or,
                           uint8_t* __restrict__ out_bpm) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    int32_t m = a[p] | (b ? b[p] : 0);
    if (out_or) out_or[p] = m;
    if (out_bpm) out_bpm[p] = (m & badsum) != 0;
}

// mask += bit where img == value (bit 16: weight == 0, zuds/mask.py:26-33;
// bit 17: diff == 1e-30, zuds/subtraction.py:170-171)
__global__ void k_mask_flag(int32_t* __restrict__ mask, const float* __restrict__ img, float value,
                            int32_t bit, int64_t n) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    if (img[p] == value) mask[p] |= bit;
}

// img += v (the 150-count pedestal, zuds/coadd.py:205-206, zuds/hotpants.py:29)
__global__ void k_add_scalar(float* __restrict__ img, float v, int64_t n) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    img[p] += v;
}

extern "C" int zm_rms_from_weight_dev(zm_ctx* ctx, const float* wgt, const uint8_t* bad, int64_t n,
                                      float big_rms, float* out) {
    ZM_CHECK(ctx && wgt && out && n > 0, "zm_rms_from_weight_dev: bad argument");
    ZM_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_rms_from_weight, EW_GRID(n), wgt, bad, n, big_rms, out);
    ZM_HIP(hipGetLastError());
    return 0;
}

extern "C" int zm_weight_from_rms_dev(zm_ctx* ctx, const float* rms, const uint8_t* bad,
                                      const float* img, float satur, int64_t n, float* out) {
    ZM_CHECK(ctx && rms && out && n > 0, "zm_weight_from_rms_dev: bad argument");
    ZM_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_weight_from_rms, EW_GRID(n), rms, bad, img, satur, n, out);
    ZM_HIP(hipGetLastError());
    return 0;
}

extern "C" int zm_mask_bad_dev(zm_ctx* ctx, const int32_t* a, const int32_t* b, int32_t badsum,
                               int64_t n, int32_t* out_or, uint8_t* out_bpm) {
    ZM_CHECK(ctx && a && n > 0 && (out_or || out_bpm), "zm_mask_bad_dev: bad argument");
    ZM_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_mask_bad, EW_GRID(n), a, b, badsum, n, out_or, out_bpm);
    ZM_HIP(hipGetLastError());
    return 0;
}

extern "C" int zm_mask_flag_dev(zm_ctx* ctx, int32_t* mask, const float* img, float value,
                                int32_t bit, int64_t n) {
    ZM_CHECK(ctx && mask && img && n > 0, "zm_mask_flag_dev: bad argument");
    ZM_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_mask_flag, EW_GRID(n), mask, img, value, bit, n);
    ZM_HIP(hipGetLastError());
    return 0;
}

// float4 copy: the rate a plain streaming kernel reaches on this GPU, read + write ("an achieved-copy
// ceiling with a float4 copy kernel").  A workgroup walks slabs of 4 x 256 float4 (16 KB): four independent
// 16-byte loads in flight per thread, stores non-temporal (written once, never read back by the launch); eight
// workgroups per CU.  (Round 4's form - one float4 per thread and iteration, plain stores - reported 4.9 TB/s
// where MI355X_MICROARCH.md measures 6.29 for a float4 copy: a ceiling that flattered every kernel held
// against it)
template <int U, bool NTL>
__global__ __launch_bounds__(256) void k_copy4(const float4* __restrict__ src, float4* __restrict__ dst, int64_t n4) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f* s = reinterpret_cast<const v4f*>(src);
    v4f* d = reinterpret_cast<v4f*>(dst);
    const int64_t nslab = n4 / (256 * U);
    for (int64_t b = blockIdx.x; b < nslab; b += gridDim.x) {
        const int64_t o = b * (256 * U) + threadIdx.x;
        v4f r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = NTL ? __builtin_nontemporal_load(s + o + 256 * u) : s[o + 256 * u];
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_nontemporal_store(r[u], d + o + 256 * u);
    }
    for (int64_t p = nslab * (256 * U) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n4;
         p += (int64_t)gridDim.x * blockDim.x)
        d[p] = s[p];
}

extern "C" int zm_copy_probe_dev(zm_ctx* ctx, const void* src, void* dst, int64_t nbytes) {
    ZM_CHECK(ctx && src && dst && nbytes > 0 && nbytes % 16 == 0 && ((uintptr_t)src & 15) == 0 &&
             ((uintptr_t)dst & 15) == 0, "zm_copy_probe_dev: bad argument");
    ZM_HIP(hipSetDevice(ctx->device));
    int ncu = 256;
    ZM_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device));
    const int64_t n4 = nbytes / 16;
    const unsigned grid = (unsigned)std::min<int64_t>((n4 + 255) / 256, (int64_t)ncu * 8);
    // (developer: ZM_COPY_FORM = loads in flight per thread (4, 8) + 'n' for non-temporal loads, e.g. "8n")
    // (measured, GB/s read + write on one box: 4: 5558, 4n: 5865, 8: 5323, 8n: 5681 - the default is 4n)
    const char* cf = ZM_DEVENV("ZM_COPY_FORM");
    const int u = cf ? atoi(cf) : 4;
    const bool ntl = cf ? strchr(cf, 'n') != nullptr : true;
    if (u == 8 && ntl) hipLaunchKernelGGL((k_copy4<8, true>), dim3(grid), dim3(256), 0, ctx->stream, (const float4*)src, (float4*)dst, n4);
    else if (u == 8) hipLaunchKernelGGL((k_copy4<8, false>), dim3(grid), dim3(256), 0, ctx->stream, (const float4*)src, (float4*)dst, n4);
    else if (ntl) hipLaunchKernelGGL((k_copy4<4, true>), dim3(grid), dim3(256), 0, ctx->stream, (const float4*)src, (float4*)dst, n4);
    else hipLaunchKernelGGL((k_copy4<4, false>), dim3(grid), dim3(256), 0, ctx->stream, (const float4*)src, (float4*)dst, n4);
    ZM_HIP(hipGetLastError());
    return 0;
}

extern "C" int zm_add_scalar_dev(zm_ctx* ctx, float* img, float v, int64_t n) {
    ZM_CHECK(ctx && img && n > 0, "zm_add_scalar_dev: bad argument");
    ZM_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_add_scalar, EW_GRID(n), img, v, n);
    ZM_HIP(hipGetLastError());
    return 0;
}

// int16 mask plane -> int32 (sign extension: what numpy's astype(int32) gives the host path).  For the
// paths that read int32 masks only (k_resample, k_prep_box, k_resample_mask); the fused coadd reads int16.
__global__ __launch_bounds__(256) void k_mask_widen(const int16_t* __restrict__ in, int64_t n, int32_t* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (int64_t p = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; p < n; p += stride) {
        if (p + 4 <= n && ((uintptr_t)in & 7) == 0 && ((uintptr_t)out & 15) == 0) {
            const short4 v = *reinterpret_cast<const short4*>(in + p);
            *reinterpret_cast<int4*>(out + p) = make_int4(v.x, v.y, v.z, v.w);
        } else {
            for (int64_t q = p; q < n && q < p + 4; ++q) out[q] = in[q];
        }
    }
}

int zm_launch_mask_widen(zm_ctx* ctx, const int16_t* in, int64_t n, int32_t* out) {
    int ncu = 256;
    ZM_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device));
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((n / 4 + 255) / 256, (int64_t)ncu * 8));
    hipLaunchKernelGGL(k_mask_widen, dim3(grid), dim3(256), 0, ctx->stream, in, n, out);
    ZM_HIP(hipGetLastError());
    return 0;
}

extern "C" int zm_mask_widen_dev(zm_ctx* ctx, const int16_t* in, int64_t n, int32_t* out) {
    ZM_CHECK(ctx && in && out && n > 0, "zm_mask_widen_dev: bad argument");
    ZM_HIP(hipSetDevice(ctx->device));
    return zm_launch_mask_widen(ctx, in, n, out);
}
