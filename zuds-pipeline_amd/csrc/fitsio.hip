// FITS data blocks <-> native planes on the device: the byte swap and BSCALE / BZERO
// arithmetic that astropy / fitsio do on the host in the reference's
// FITSFile.load_data / save (zuds/fitsfile.py:69-94,146-206).  The raw big-endian data
// block of a primary HDU goes over PCIe as it lies on disk (pinned staging, async copy);
// decoding to the float32 / int32 / uint8 planes the kernels take, and encoding the
// products back, are streaming kernels (HBM bound, one pass).
#include "zm_internal.h"

__device__ inline uint32_t bswap32(uint32_t v) { return __builtin_bswap32(v); }
__device__ inline uint16_t bswap16(uint16_t v) { return (uint16_t)((v << 8) | (v >> 8)); }

// BITPIX: 8, 16, 32, -32, -64.  out_kind: 0 float32, 1 int32, 2 uint8, 3 int16 (a BITPIX 16 mask kept at 16 bits).
// Physical value = bzero + bscale * stored (FITS 4.0, 5.3); integer outputs take the
// integer BZERO exactly (unsigned 16 / 32 bit conventions), NaN stays NaN.
__global__ __launch_bounds__(256) void k_fits_decode(const uint8_t* __restrict__ raw, int bitpix,
                                                     double bscale, double bzero, int64_t n,
                                                     int out_kind, void* __restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const bool scaled = (bscale != 1.0) || (bzero != 0.0);
    double v;
    int64_t iv = 0;
    bool is_int = true;
    switch (bitpix) {
        case 8: iv = raw[p]; break;
        case 16: iv = (int16_t)bswap16(reinterpret_cast<const uint16_t*>(raw)[p]); break;
        case 32: iv = (int32_t)bswap32(reinterpret_cast<const uint32_t*>(raw)[p]); break;
        case -32: {
            is_int = false;
            v = (double)__uint_as_float(bswap32(reinterpret_cast<const uint32_t*>(raw)[p]));
            break;
        }
        default: {   // -64
            is_int = false;
            const uint32_t hi = bswap32(reinterpret_cast<const uint32_t*>(raw)[2 * p]);
            const uint32_t lo = bswap32(reinterpret_cast<const uint32_t*>(raw)[2 * p + 1]);
            v = __longlong_as_double(((long long)hi << 32) | lo);
            break;
        }
    }
    if (is_int) {
        if (out_kind != 0 && bscale == 1.0 && bzero == floor(bzero)) {
            iv += (int64_t)bzero;
            if (out_kind == 1) reinterpret_cast<int32_t*>(out)[p] = (int32_t)iv;
            else if (out_kind == 3) reinterpret_cast<int16_t*>(out)[p] = (int16_t)iv;
            else reinterpret_cast<uint8_t*>(out)[p] = (uint8_t)iv;
            return;
        }
        v = (double)iv;
    }
    if (scaled) v = bzero + bscale * v;
    if (out_kind == 0) reinterpret_cast<float*>(out)[p] = (float)v;
    else if (out_kind == 1) reinterpret_cast<int32_t*>(out)[p] = (int32_t)v;
    else if (out_kind == 3) reinterpret_cast<int16_t*>(out)[p] = (int16_t)v;
    else reinterpret_cast<uint8_t*>(out)[p] = (uint8_t)v;
}

// in_kind: 0 float32 -> BITPIX -32, 1 int32 -> BITPIX 32, 2 uint8 -> BITPIX 8,
// 3 int32 -> BITPIX 16 (values must fit)
__global__ __launch_bounds__(256) void k_fits_encode(const void* __restrict__ in, int in_kind,
                                                     int64_t n, uint8_t* __restrict__ raw) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    switch (in_kind) {
        case 0: reinterpret_cast<uint32_t*>(raw)[p] = bswap32(__float_as_uint(reinterpret_cast<const float*>(in)[p])); break;
        case 1: reinterpret_cast<uint32_t*>(raw)[p] = bswap32((uint32_t)reinterpret_cast<const int32_t*>(in)[p]); break;
        case 2: raw[p] = reinterpret_cast<const uint8_t*>(in)[p]; break;
        default: reinterpret_cast<uint16_t*>(raw)[p] = bswap16((uint16_t)(int16_t)reinterpret_cast<const int32_t*>(in)[p]); break;
    }
}

extern "C" int zm_fits_decode_dev(zm_ctx* ctx, const void* raw_dev, int bitpix, double bscale,
                                  double bzero, int64_t n, int out_kind, void* out_dev) {
    ZM_CHECK(ctx && raw_dev && out_dev && n > 0, "zm_fits_decode_dev: bad argument");
    ZM_CHECK(bitpix == 8 || bitpix == 16 || bitpix == 32 || bitpix == -32 || bitpix == -64,
             "zm_fits_decode_dev: unsupported BITPIX %d", bitpix);
    ZM_CHECK(out_kind >= 0 && out_kind <= 3, "zm_fits_decode_dev: unknown output kind %d", out_kind);
    ZM_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_fits_decode, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const uint8_t*)raw_dev, bitpix, bscale, bzero, n, out_kind, out_dev);
    ZM_HIP(hipGetLastError());
    return 0;
}

extern "C" int zm_fits_encode_dev(zm_ctx* ctx, const void* in_dev, int in_kind, int64_t n,
                                  void* raw_dev) {
    ZM_CHECK(ctx && in_dev && raw_dev && n > 0, "zm_fits_encode_dev: bad argument");
    ZM_CHECK(in_kind >= 0 && in_kind <= 3, "zm_fits_encode_dev: unknown input kind %d", in_kind);
    ZM_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_fits_encode, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       in_dev, in_kind, n, (uint8_t*)raw_dev);
    ZM_HIP(hipGetLastError());
    return 0;
}
