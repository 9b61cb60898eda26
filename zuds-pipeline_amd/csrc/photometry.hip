// Forced circular-aperture photometry on resident planes.
//
// Replaces photutils.aperture_photometry(..., method='exact') + the bounding-box
// flag OR of zuds/photometry.py:61-113,116-249 (r = APERTURE_RADIUS = 3 px,
// zuds/constants.py:14).  One wave per position: lanes walk the pixels of the
// aperture's bounding box, the exact circle / pixel overlap is the closed-form
// quarter-box area (oracle/photometry.py), sums are wave reductions in fp64.
#include "zm_internal.h"

__device__ inline double ap_P(double u, double r) {
    double v = fmax(r * r - u * u, 0.0);
    double t = fmin(fmax(u / r, -1.0), 1.0);
    return 0.5 * (u * sqrt(v) + r * r * asin(t));
}

__device__ inline double ap_quarter(double x, double y, double r) {
    x = fmin(x, r);
    y = fmin(y, r);
    if (x * x + y * y <= r * r) return x * y;
    double xc = sqrt(fmax(r * r - y * y, 0.0));
    double xm = fmin(x, xc);
    return y * xm + ap_P(x, r) - ap_P(xm, r);
}

__device__ inline double ap_signed(double x, double y, double r) {
    double s = ((x > 0) - (x < 0)) * ((y > 0) - (y < 0));
    return s * ap_quarter(fabs(x), fabs(y), r);
}

__global__ __launch_bounds__(64) void k_aperture(const float* __restrict__ img,
                                                 const float* __restrict__ rms,
                                                 const int32_t* __restrict__ mask, int nx, int ny,
                                                 int npos, const double* __restrict__ xs,
                                                 const double* __restrict__ ys, double r,
                                                 double* __restrict__ flux, double* __restrict__ err,
                                                 int32_t* __restrict__ flags) {
    const int k = blockIdx.x, lane = threadIdx.x;
    if (k >= npos) return;
    const double xc = xs[k], yc = ys[k];
    // photutils BoundingBox.from_float(x - r, x + r, y - r, y + r)
    int ixmin = (int)floor(xc - r + 0.5), ixmax = (int)ceil(xc + r + 0.5);
    int iymin = (int)floor(yc - r + 0.5), iymax = (int)ceil(yc + r + 0.5);
    ixmin = max(ixmin, 0); ixmax = min(ixmax, nx);
    iymin = max(iymin, 0); iymax = min(iymax, ny);
    const int bw = ixmax - ixmin, bh = iymax - iymin;
    double f = 0.0, v = 0.0;
    int fl = 0;
    if (bw > 0 && bh > 0 && isfinite(xc) && isfinite(yc)) {
        for (int e = lane; e < bw * bh; e += 64) {
            const int j = iymin + e / bw, i = ixmin + e % bw;
            const double x0 = i - 0.5 - xc, x1 = i + 0.5 - xc, y0 = j - 0.5 - yc, y1 = j + 0.5 - yc;
            const double frac = ap_signed(x1, y1, r) - ap_signed(x0, y1, r) - ap_signed(x1, y0, r) +
                                ap_signed(x0, y0, r);
            const size_t idx = (size_t)j * nx + i;
            f += (double)img[idx] * frac;
            if (rms) { double s = rms[idx]; v += s * s * frac; }
            if (mask) fl |= mask[idx];
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        f += __shfl_xor(f, o);
        v += __shfl_xor(v, o);
        fl |= __shfl_xor(fl, o);
    }
    if (lane == 0) {
        flux[k] = f;
        err[k] = sqrt(fmax(v, 0.0));
        flags[k] = fl;
    }
}

extern "C" int zm_aperture_photometry_dev(zm_ctx* ctx, const float* img, const float* rms,
                                          const int32_t* mask, int nx, int ny, int npos,
                                          const double* x, const double* y, double radius,
                                          double* out_flux, double* out_err, int32_t* out_flags) {
    ZM_CHECK(ctx && img && x && y && out_flux && out_err && out_flags,
             "zm_aperture_photometry_dev: null argument");
    ZM_CHECK(nx > 0 && ny > 0 && npos >= 0, "zm_aperture_photometry_dev: bad sizes");
    ZM_CHECK(radius > 0 && radius < 512, "zm_aperture_photometry_dev: radius %g outside (0, 512)", radius);
    if (npos == 0) return 0;
    ZM_HIP(hipSetDevice(ctx->device));
    zm_scope_timer t(ctx, "aperture");
    hipLaunchKernelGGL(k_aperture, dim3(npos), dim3(64), 0, ctx->stream, img, rms, mask, nx, ny, npos, x,
                       y, radius, out_flux, out_err, out_flags);
    ZM_HIP(hipGetLastError());
    return 0;
}

extern "C" int zm_aperture_photometry(zm_ctx* ctx, const float* img, const float* rms,
                                      const int32_t* mask, int nx, int ny, int npos, const double* x,
                                      const double* y, double radius, double* out_flux,
                                      double* out_err, int32_t* out_flags) {
    ZM_CHECK(ctx && img && x && y && out_flux && out_err && out_flags,
             "zm_aperture_photometry: null argument");
    ZM_CHECK(nx > 0 && ny > 0 && npos >= 0, "zm_aperture_photometry: bad sizes");
    if (npos == 0) return 0;
    ZM_HIP(hipSetDevice(ctx->device));
    const size_t np = (size_t)nx * ny;
    float *d_img = nullptr, *d_rms = nullptr;
    int32_t* d_mask = nullptr;
    double* d_pos = nullptr;
    ZM_TRY(ctx->get("h_img", np * 4, (void**)&d_img));
    ZM_HIP(hipMemcpyAsync(d_img, img, np * 4, hipMemcpyHostToDevice, ctx->stream));
    if (rms) {
        ZM_TRY(ctx->get("h_wgt", np * 4, (void**)&d_rms));
        ZM_HIP(hipMemcpyAsync(d_rms, rms, np * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    if (mask) {
        ZM_TRY(ctx->get("h_mask", np * 4, (void**)&d_mask));
        ZM_HIP(hipMemcpyAsync(d_mask, mask, np * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    ZM_TRY(ctx->get("ap_pos", sizeof(double) * 4 * (size_t)npos + sizeof(int32_t) * npos, (void**)&d_pos));
    double *d_x = d_pos, *d_y = d_pos + npos, *d_f = d_pos + 2 * (size_t)npos, *d_e = d_pos + 3 * (size_t)npos;
    int32_t* d_fl = reinterpret_cast<int32_t*>(d_pos + 4 * (size_t)npos);
    ZM_HIP(hipMemcpyAsync(d_x, x, sizeof(double) * npos, hipMemcpyHostToDevice, ctx->stream));
    ZM_HIP(hipMemcpyAsync(d_y, y, sizeof(double) * npos, hipMemcpyHostToDevice, ctx->stream));
    ZM_TRY(zm_aperture_photometry_dev(ctx, d_img, d_rms, d_mask, nx, ny, npos, d_x, d_y, radius, d_f, d_e,
                                      d_fl));
    ZM_HIP(hipMemcpyAsync(out_flux, d_f, sizeof(double) * npos, hipMemcpyDeviceToHost, ctx->stream));
    ZM_HIP(hipMemcpyAsync(out_err, d_e, sizeof(double) * npos, hipMemcpyDeviceToHost, ctx->stream));
    ZM_HIP(hipMemcpyAsync(out_flags, d_fl, sizeof(int32_t) * npos, hipMemcpyDeviceToHost, ctx->stream));
    ZM_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}
