// Pixel-only pieces of the reference's seeing estimate and detection filter
// (SURVEY.md 8(f) row 4), so that `calculate_seeing=True` and the post-subtraction cuts
// run without SExtractor catalogs or the Gaia / Kowalski network queries:
//
//   zm_find_stars    isolated, unsaturated, unmasked local maxima above a threshold
//                    (stands in for "catalog sources matched to Gaia stars",
//                    zuds/seeing.py:10-103)
//   zm_star_fwhm     FWHM of each star from adaptive Gaussian-weighted second moments
//                    (stands in for SExtractor's FWHM_IMAGE, zuds/seeing.py:105-118)
//   zm_negpix_test   the "negative pixel next to a positive one" dipole cut of
//                    filter_sexcat (zuds/filterobjects.py:155-195)
//
// Conventions (chosen, stated in oracle/detect.py): see there.
#include "zm_internal.h"

// out: candidates in no particular order (the host sorts by peak value, then y, then x)
__global__ __launch_bounds__(256) void k_find_stars(const float* __restrict__ img,
                                                    const uint8_t* __restrict__ bad, int nx, int ny,
                                                    float lo, float hi, int iso, int border,
                                                    int max_out, int* __restrict__ count,
                                                    int* __restrict__ ox, int* __restrict__ oy,
                                                    float* __restrict__ opeak) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x < border || x >= nx - border || y < border || y >= ny - border) return;
    const float v = img[(size_t)y * nx + x];
    if (!(v > lo) || !(v < hi)) return;
    for (int dy = -iso; dy <= iso; ++dy)
        for (int dx = -iso; dx <= iso; ++dx) {
            const size_t q = (size_t)(y + dy) * nx + (x + dx);
            const float n = img[q];
            if (!(n == n) || (bad && bad[q])) return;
            if (dx == 0 && dy == 0) continue;
            // strict maximum; ties go to the pixel that comes first in raster order
            const bool earlier = dy < 0 || (dy == 0 && dx < 0);
            if (earlier ? n >= v : n > v) return;
        }
    const int k = atomicAdd(count, 1);
    if (k < max_out) { ox[k] = x; oy[k] = y; opeak[k] = v; }
}

__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// one wave per star; window (2 half + 1)^2 around the peak pixel
__global__ __launch_bounds__(64) void k_star_fwhm(const float* __restrict__ img, int nx, int ny, int nstar,
                                                  const int* __restrict__ sx, const int* __restrict__ sy,
                                                  int half, int maxit, double* __restrict__ fwhm,
                                                  double* __restrict__ cxo, double* __restrict__ cyo) {
    const int k = blockIdx.x, lane = threadIdx.x;
    if (k >= nstar) return;
    const int px = sx[k], py = sy[k], side = 2 * half + 1;
    double cx = px, cy = py, s2 = 4.0;                   // sigma_w^2 of the weight
    double out = __builtin_nan("");
    for (int it = 0; it < maxit; ++it) {
        double s0 = 0, s1x = 0, s1y = 0;
        for (int e = lane; e < side * side; e += 64) {
            const int j = py - half + e / side, i = px - half + e % side;
            if (i < 0 || i >= nx || j < 0 || j >= ny) continue;
            const double dx = i - cx, dy = j - cy;
            const double w = exp(-0.5 * (dx * dx + dy * dy) / s2) * (double)img[(size_t)j * nx + i];
            s0 += w; s1x += w * i; s1y += w * j;
        }
        s0 = wave_sum(s0); s1x = wave_sum(s1x); s1y = wave_sum(s1y);
        if (!(s0 > 0)) break;
        cx = s1x / s0; cy = s1y / s0;
        double sxx = 0, syy = 0, t0 = 0;
        for (int e = lane; e < side * side; e += 64) {
            const int j = py - half + e / side, i = px - half + e % side;
            if (i < 0 || i >= nx || j < 0 || j >= ny) continue;
            const double dx = i - cx, dy = j - cy;
            const double w = exp(-0.5 * (dx * dx + dy * dy) / s2) * (double)img[(size_t)j * nx + i];
            t0 += w; sxx += w * dx * dx; syy += w * dy * dy;
        }
        t0 = wave_sum(t0); sxx = wave_sum(sxx); syy = wave_sum(syy);
        if (!(t0 > 0)) break;
        const double m2 = 0.5 * (sxx + syy) / t0;        // measured sigma^2 under the weight
        const double inv = 1.0 / m2 - 1.0 / s2;
        if (!(m2 > 0) || !(inv > 0)) break;
        const double ns2 = 1.0 / inv;                    // de-weighted sigma^2
        const bool done = fabs(ns2 - s2) <= 1e-8 * ns2;
        s2 = ns2;
        out = 2.3548200450309493 * sqrt(s2);
        if (done) break;
    }
    if (lane == 0) { fwhm[k] = out; cxo[k] = cx; cyo[k] = cy; }
}

// x, y: SExtractor X_IMAGE / Y_IMAGE (1-based, float).  bad[k] = 1 when a pixel of the
// 11 x 11 cutout below -5 sigma has a neighbour (3 x 3) above +5 sigma.
__global__ void k_negpix(const float* __restrict__ img, int nx, int ny, int npos,
                         const double* __restrict__ xs, const double* __restrict__ ys, float med,
                         float sig, int half, int32_t* __restrict__ bad) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= npos) return;
    const int xc = (int)rint(xs[k]) - 1, yc = (int)rint(ys[k]) - 1;     // np.round: half to even
    int b = 0;
    for (int j = yc - half; j <= yc + half && !b; ++j)
        for (int i = xc - half; i <= xc + half && !b; ++i) {
            if (i < 0 || i >= nx || j < 0 || j >= ny) continue;
            if (!((img[(size_t)j * nx + i] - med) / sig < -5.f)) continue;
            for (int dj = -1; dj <= 1 && !b; ++dj)
                for (int di = -1; di <= 1; ++di) {
                    const int ii = i + di, jj = j + dj;
                    // the 13 x 13 "big" cutout: neighbours outside it (or the frame) do not count
                    if (ii < xc - half - 1 || ii > xc + half + 1 || jj < yc - half - 1 || jj > yc + half + 1) continue;
                    if (ii < 0 || ii >= nx || jj < 0 || jj >= ny) continue;
                    if ((img[(size_t)jj * nx + ii] - med) / sig > 5.f) { b = 1; break; }
                }
        }
    bad[k] = b;
}

// ---- host-pointer entry points (small outputs; the image is uploaded once) --------------
// img / bad: device planes; the candidate lists come back to host arrays (at most max_out entries)
extern "C" int zm_find_stars_dev(zm_ctx* ctx, const float* d_img, const uint8_t* d_bad, int nx, int ny,
                                 float thresh_lo, float thresh_hi, int isolation, int border, int max_out,
                                 int* out_x, int* out_y, float* out_peak, int* out_n) {
    ZM_CHECK(ctx && d_img && out_x && out_y && out_peak && out_n, "zm_find_stars: null argument");
    ZM_CHECK(nx > 0 && ny > 0 && max_out > 0, "zm_find_stars: bad sizes");
    ZM_CHECK(isolation >= 1 && isolation <= 32 && border >= isolation, "zm_find_stars: border >= isolation >= 1");
    ZM_HIP(hipSetDevice(ctx->device));
    char* d_out = nullptr;
    ZM_TRY(ctx->get("det_out", 16 + (size_t)max_out * 12, (void**)&d_out));
    int* d_n = (int*)d_out;
    int* d_x = (int*)(d_out + 16);
    int* d_y = d_x + max_out;
    float* d_p = (float*)(d_y + max_out);
    ZM_HIP(hipMemsetAsync(d_n, 0, 16, ctx->stream));
    hipLaunchKernelGGL(k_find_stars, dim3(zm_div_up(nx, 256), ny), dim3(256), 0, ctx->stream, d_img, d_bad, nx,
                       ny, thresh_lo, thresh_hi, isolation, border, max_out, d_n, d_x, d_y, d_p);
    ZM_HIP(hipGetLastError());
    int n = 0;
    ZM_HIP(hipMemcpyAsync(&n, d_n, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    ZM_HIP(hipStreamSynchronize(ctx->stream));
    *out_n = n;                                           // may exceed max_out: caller raises the threshold
    const int m = n < max_out ? n : max_out;
    if (m > 0) {
        ZM_HIP(hipMemcpyAsync(out_x, d_x, sizeof(int) * m, hipMemcpyDeviceToHost, ctx->stream));
        ZM_HIP(hipMemcpyAsync(out_y, d_y, sizeof(int) * m, hipMemcpyDeviceToHost, ctx->stream));
        ZM_HIP(hipMemcpyAsync(out_peak, d_p, sizeof(float) * m, hipMemcpyDeviceToHost, ctx->stream));
        ZM_HIP(hipStreamSynchronize(ctx->stream));
    }
    return 0;
}

extern "C" int zm_find_stars(zm_ctx* ctx, const float* img, const uint8_t* bad, int nx, int ny,
                             float thresh_lo, float thresh_hi, int isolation, int border, int max_out,
                             int* out_x, int* out_y, float* out_peak, int* out_n) {
    ZM_CHECK(ctx && img, "zm_find_stars: null argument");
    ZM_CHECK(nx > 0 && ny > 0, "zm_find_stars: bad sizes");
    ZM_HIP(hipSetDevice(ctx->device));
    const size_t np = (size_t)nx * ny;
    float* d_img = nullptr;
    uint8_t* d_bad = nullptr;
    ZM_TRY(ctx->get("h_img", np * 4, (void**)&d_img));
    ZM_HIP(hipMemcpyAsync(d_img, img, np * 4, hipMemcpyHostToDevice, ctx->stream));
    if (bad) {
        ZM_TRY(ctx->get("h_bpm", np, (void**)&d_bad));
        ZM_HIP(hipMemcpyAsync(d_bad, bad, np, hipMemcpyHostToDevice, ctx->stream));
    }
    return zm_find_stars_dev(ctx, d_img, d_bad, nx, ny, thresh_lo, thresh_hi, isolation, border, max_out, out_x, out_y,
                             out_peak, out_n);
}

extern "C" int zm_star_fwhm(zm_ctx* ctx, const float* img, int nx, int ny, int nstar, const int* x,
                            const int* y, int half, double* out_fwhm, double* out_cx, double* out_cy) {
    ZM_CHECK(ctx && img, "zm_star_fwhm: null argument");
    ZM_CHECK(nx > 0 && ny > 0, "zm_star_fwhm: bad sizes");
    if (nstar == 0) return 0;
    ZM_HIP(hipSetDevice(ctx->device));
    const size_t np = (size_t)nx * ny;
    float* d_img = nullptr;
    ZM_TRY(ctx->get("h_img", np * 4, (void**)&d_img));
    ZM_HIP(hipMemcpyAsync(d_img, img, np * 4, hipMemcpyHostToDevice, ctx->stream));
    return zm_star_fwhm_dev(ctx, d_img, nx, ny, nstar, x, y, half, out_fwhm, out_cx, out_cy);
}

// img: a device plane; star positions in, widths and centroids out: host arrays
extern "C" int zm_star_fwhm_dev(zm_ctx* ctx, const float* d_img, int nx, int ny, int nstar, const int* x,
                                const int* y, int half, double* out_fwhm, double* out_cx, double* out_cy) {
    ZM_CHECK(ctx && d_img && x && y && out_fwhm && out_cx && out_cy, "zm_star_fwhm: null argument");
    ZM_CHECK(nx > 0 && ny > 0 && nstar >= 0 && half >= 2 && half <= 64, "zm_star_fwhm: bad sizes");
    if (nstar == 0) return 0;
    ZM_HIP(hipSetDevice(ctx->device));
    char* d_s = nullptr;
    ZM_TRY(ctx->get("det_star", (size_t)nstar * 32, (void**)&d_s));
    double* d_f = (double*)d_s;
    double* d_cx = d_f + nstar;
    double* d_cy = d_cx + nstar;
    int* d_x = (int*)(d_cy + nstar);
    int* d_y = d_x + nstar;
    ZM_HIP(hipMemcpyAsync(d_x, x, sizeof(int) * nstar, hipMemcpyHostToDevice, ctx->stream));
    ZM_HIP(hipMemcpyAsync(d_y, y, sizeof(int) * nstar, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_star_fwhm, dim3(nstar), dim3(64), 0, ctx->stream, d_img, nx, ny, nstar, d_x, d_y, half,
                       25, d_f, d_cx, d_cy);
    ZM_HIP(hipGetLastError());
    ZM_HIP(hipMemcpyAsync(out_fwhm, d_f, sizeof(double) * nstar, hipMemcpyDeviceToHost, ctx->stream));
    ZM_HIP(hipMemcpyAsync(out_cx, d_cx, sizeof(double) * nstar, hipMemcpyDeviceToHost, ctx->stream));
    ZM_HIP(hipMemcpyAsync(out_cy, d_cy, sizeof(double) * nstar, hipMemcpyDeviceToHost, ctx->stream));
    ZM_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int zm_negpix_test(zm_ctx* ctx, const float* img, int nx, int ny, int npos, const double* x,
                              const double* y, double median, double sigma, int32_t* out_bad) {
    ZM_CHECK(ctx && img && x && y && out_bad, "zm_negpix_test: null argument");
    ZM_CHECK(nx > 0 && ny > 0 && npos >= 0, "zm_negpix_test: bad sizes");
    ZM_CHECK(sigma > 0, "zm_negpix_test: sigma must be positive");
    if (npos == 0) return 0;
    ZM_HIP(hipSetDevice(ctx->device));
    const size_t np = (size_t)nx * ny;
    float* d_img = nullptr;
    char* d_s = nullptr;
    ZM_TRY(ctx->get("h_img", np * 4, (void**)&d_img));
    ZM_HIP(hipMemcpyAsync(d_img, img, np * 4, hipMemcpyHostToDevice, ctx->stream));
    ZM_TRY(ctx->get("det_neg", (size_t)npos * 20, (void**)&d_s));
    double* d_x = (double*)d_s;
    double* d_y = d_x + npos;
    int32_t* d_b = (int32_t*)(d_y + npos);
    ZM_HIP(hipMemcpyAsync(d_x, x, sizeof(double) * npos, hipMemcpyHostToDevice, ctx->stream));
    ZM_HIP(hipMemcpyAsync(d_y, y, sizeof(double) * npos, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_negpix, dim3(zm_div_up(npos, 64)), dim3(64), 0, ctx->stream, d_img, nx, ny, npos, d_x,
                       d_y, (float)median, (float)sigma, 5, d_b);
    ZM_HIP(hipGetLastError());
    ZM_HIP(hipMemcpyAsync(out_bad, d_b, sizeof(int32_t) * npos, hipMemcpyDeviceToHost, ctx->stream));
    ZM_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}
