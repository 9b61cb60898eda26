// Resample-to-reference-WCS on gfx950: lattice of exact fp64 inverse-map nodes,
// prep pass (background / variance algebra -> interleaved {value, variance}
// plane) and the LDS-tiled Lanczos-3 / bilinear / nearest gather.
//
// Replaces the per-pixel inverse WCS + interpolation loop of the SWarp runs the
// reference launches from zuds/coadd.py:133,156 and zuds/swarp.py:175 (flags:
// zuds/astromatic/makecoadd/default.swarp:42-67).  Arithmetic conventions are
// stated in oracle/resample.py.
//
// Layout in HBM
//   src  : float2 [ny][spitch]  {value, variance}, spitch = nx rounded up to even;
//          bad pixel = {v, 1e30}; pad column = {0, 1e30}
//   lat  : double2 [lny][lnx]   0-based input position of output pixel
//          (gx * 16, gy * 16); lnx = (onx - 1) / 16 + 2
//   dst  : float2 [ony][onx]    {value, weight}; weight 0 = no data
//
// One workgroup = one 64 x 16 output tile = 4 x 1 lattice cells; its input
// footprint (bounding box of the 10 tile nodes + kernel support) is staged in
// LDS with 16-byte loads and read back as 8-byte {value, variance} pairs.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "zm_internal.h"
#include "wcs_math.h"

#define TW 64
#define TH 16
#define LSTEP ZM_LATTICE_STEP
#define RTH 32          // output rows of a k_resample tile (two lattice cells: the other kernels keep TH)
#define HDR_FLOATS 128  // LDS header ring of k_resample: 3 x (30 node floats + bbox ints + flags) in 512 B

// ---------------------------------------------------------------------------
// The map travels as a by-value kernel argument (1.5 KB): no host staging buffer
// whose lifetime would have to outlive the enqueue.
__global__ void k_lattice(const zm_map_params mp, int lnx, int lny, double2* __restrict__ lat) {
    int gx = blockIdx.x * blockDim.x + threadIdx.x;
    int gy = blockIdx.y * blockDim.y + threadIdx.y;
    if (gx >= lnx || gy >= lny) return;
    double xi, yi;
    zm_map_out_to_in(&mp.wout, &mp.win, mp.rot, 1.0 + (double)gx * LSTEP,
                     1.0 + (double)gy * LSTEP, &xi, &yi);
    lat[(size_t)gy * lnx + gx] = make_double2(xi - 1.0, yi - 1.0);
}

// every frame of a stack in one launch: blockIdx.z = frame, maps read from device memory
__global__ void k_lattice_batch(const zm_map_params* __restrict__ mps, int lnx, int lny,
                                double2* __restrict__ lat) {
    int gx = blockIdx.x * blockDim.x + threadIdx.x;
    int gy = blockIdx.y * blockDim.y + threadIdx.y;
    if (gx >= lnx || gy >= lny) return;
    const zm_map_params* mp = mps + blockIdx.z;
    double xi, yi;
    zm_map_out_to_in(&mp->wout, &mp->win, mp->rot, 1.0 + (double)gx * LSTEP,
                     1.0 + (double)gy * LSTEP, &xi, &yi);
    lat[((size_t)blockIdx.z * lny + gy) * lnx + gx] = make_double2(xi - 1.0, yi - 1.0);
}

// mp_host: n maps (any host memory); staged through a pinned buffer guarded by an event, so
// that a later call cannot overwrite the staging area of a copy still in flight
// after != NULL: the lattices depend on the WCS only, so the kernel goes to the second stream,
// ordered after `after` (an event recorded on the main stream before the caller enqueued the
// mesh statistics - nothing older may still read the lattice buffer) and therefore free to run
// beside those statistics; the main stream resumes behind it.
int zm_launch_lattice_batch(zm_ctx* ctx, const zm_map_params* mp_host, int n, int lnx, int lny,
                            double2* lat_dev, hipEvent_t after) {
    zm_map_params *pin = nullptr, *dev = nullptr;
    hipEvent_t* ev = nullptr;
    ZM_TRY(zm_get_sync_events(ctx, 5, &ev));
    ZM_HIP(hipEventSynchronize(ev[0]));
    ZM_TRY(ctx->get_pinned("map_params_h", sizeof(zm_map_params) * (size_t)n, (void**)&pin));
    ZM_TRY(ctx->get("map_params", sizeof(zm_map_params) * (size_t)n, (void**)&dev));
    memcpy(pin, mp_host, sizeof(zm_map_params) * (size_t)n);
    // (the scope timers record on the main stream: when this scope is being timed the kernel stays there)
    const bool timed = ctx->timing && (ctx->timing_only.empty() || ctx->timing_only == "lattice");
    const bool side = after != nullptr && ctx->aux != nullptr && !timed;
    hipStream_t s = side ? ctx->aux : ctx->stream;
    if (side) ZM_HIP(hipStreamWaitEvent(s, after, 0));
    ZM_HIP(hipMemcpyAsync(dev, pin, sizeof(zm_map_params) * (size_t)n, hipMemcpyHostToDevice, s));
    ZM_HIP(hipEventRecord(ev[0], s));
    dim3 blk(16, 16, 1), grd(zm_div_up(lnx, 16), zm_div_up(lny, 16), n);
    {
        zm_scope_timer t(ctx, "lattice");
        hipLaunchKernelGGL(k_lattice_batch, grd, blk, 0, s, dev, lnx, lny, lat_dev);
    }
    ZM_HIP(hipGetLastError());
    if (side) {
        ZM_HIP(hipEventRecord(ev[4], s));
        ZM_HIP(hipStreamWaitEvent(ctx->stream, ev[4], 0));
    }
    return 0;
}

int zm_launch_lattice(zm_ctx* ctx, const zm_map_params* mp, int lnx, int lny, double2* lat_dev) {
    dim3 blk(16, 16, 1), grd(zm_div_up(lnx, 16), zm_div_up(lny, 16), 1);
    zm_scope_timer t(ctx, "lattice");
    hipLaunchKernelGGL(k_lattice, grd, blk, 0, ctx->stream, *mp, lnx, lny, lat_dev);
    ZM_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// Bicubic-spline background from mesh nodes.  bk holds 4 planes [nby][nbx]:
// value V, d2/dy2 / 6 (DY), d2/dx2 / 6 of V (A), d2/dx2 / 6 of DY (B); the
// tensor-product natural spline is then a 16-term combination (equivalent to
// SExtractor's "spline along y per node column, then along x per line").
//
// The arithmetic is PINNED (explicit fused multiply-adds, contraction off): three kernels evaluate
// it - k_prep (the prepped plane k_resample reads), k_bk_rows (the y part, once per frame row and
// mesh column) and the staging of k_coadd_fused (the x part, per staged pixel) - and the fused
// coadd must equal the k_resample path bit for bit (tests/test_fused_coadd_gpu.py).
//   y part of (y, mesh column i0): {r0, r1, e0, e1} - the spline along y through the node columns
//   i0 and i0 + 1 and through their d2/dx2 columns;  x part: dx1 r0 + dx r1 + cdx1 e0 + cdx e1.
#pragma clang fp contract(off)
__device__ inline int bk_col(int nbx, float invmesh, int x) {
    if (nbx <= 1) return 0;
    const float tx = __builtin_fmaf((float)x + 0.5f, invmesh, -0.5f);
    return min(max((int)floorf(tx), 0), nbx - 2);
}
__device__ inline float bk_dx(int nbx, float invmesh, int x, int i0) {
    if (nbx <= 1) return 0.f;
    return __builtin_fmaf((float)x + 0.5f, invmesh, -0.5f) - (float)i0;
}
__device__ inline float4 bk_ypart(const float* __restrict__ bk, int nbx, int nby, float invmesh, int y, int i0) {
    const size_t pl = (size_t)nbx * nby;
    int j0 = 0;
    float dy = 0.f;
    if (nby > 1) {
        const float ty = __builtin_fmaf((float)y + 0.5f, invmesh, -0.5f);
        j0 = min(max((int)floorf(ty), 0), nby - 2);
        dy = ty - (float)j0;
    }
    const int j1 = nby > 1 ? j0 + 1 : j0, i1 = nbx > 1 ? i0 + 1 : i0;
    const float dy1 = 1.f - dy;
    const float cdy = __builtin_fmaf(dy * dy, dy, -dy), cdy1 = __builtin_fmaf(dy1 * dy1, dy1, -dy1);
    const float* V = bk;
    const float* DY = bk + pl;
    const float* A = bk + 2 * pl;
    const float* B = bk + 3 * pl;
    const int a00 = j0 * nbx + i0, a01 = j0 * nbx + i1, a10 = j1 * nbx + i0, a11 = j1 * nbx + i1;
    // (all sixteen node loads first: a load inside an expression is waited for on the spot)
    const float v00 = V[a00], v10 = V[a10], d00 = DY[a00], d10 = DY[a10];
    const float v01 = V[a01], v11 = V[a11], d01 = DY[a01], d11 = DY[a11];
    const float p00 = A[a00], p10 = A[a10], q00 = B[a00], q10 = B[a10];
    const float p01 = A[a01], p11 = A[a11], q01 = B[a01], q11 = B[a11];
    float4 r;
    r.x = __builtin_fmaf(cdy, d10, __builtin_fmaf(cdy1, d00, __builtin_fmaf(dy, v10, dy1 * v00)));
    r.y = __builtin_fmaf(cdy, d11, __builtin_fmaf(cdy1, d01, __builtin_fmaf(dy, v11, dy1 * v01)));
    r.z = __builtin_fmaf(cdy, q10, __builtin_fmaf(cdy1, q00, __builtin_fmaf(dy, p10, dy1 * p00)));
    r.w = __builtin_fmaf(cdy, q11, __builtin_fmaf(cdy1, q01, __builtin_fmaf(dy, p11, dy1 * p01)));
    return r;
}
// the four x weights of a pixel column: {dx1, dx, cdx1, cdx}
__device__ inline float4 bk_xweights(float dx) {
    const float dx1 = 1.f - dx;
    return make_float4(dx1, dx, __builtin_fmaf(dx1 * dx1, dx1, -dx1), __builtin_fmaf(dx * dx, dx, -dx));
}
__device__ inline float bk_xpart(float4 yp, float4 xw) {
    return __builtin_fmaf(xw.w, yp.w, __builtin_fmaf(xw.z, yp.z, __builtin_fmaf(xw.y, yp.y, xw.x * yp.x)));
}
__device__ inline float bk_eval(const float* __restrict__ bk, int nbx, int nby, float invmesh,
                                int x, int y) {
    const int i0 = bk_col(nbx, invmesh, x);
    return bk_xpart(bk_ypart(bk, nbx, nby, invmesh, y, i0), bk_xweights(bk_dx(nbx, invmesh, x, i0)));
}

// Background of four consecutive pixels of a row (x a multiple of 4).  The y part is shared when
// the four pixels lie in one mesh column, which they always do when BACK_SIZE is a multiple of 8.
__device__ inline void bk_eval4(const float* __restrict__ bk, int nbx, int nby, float invmesh, int x,
                                int y, float out[4]) {
    const int i0 = bk_col(nbx, invmesh, x), i3 = bk_col(nbx, invmesh, x + 3);
    if (i0 != i3) {
#pragma unroll
        for (int k = 0; k < 4; ++k) out[k] = bk_eval(bk, nbx, nby, invmesh, x + k, y);
        return;
    }
    const float4 yp = bk_ypart(bk, nbx, nby, invmesh, y, i0);
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = bk_xpart(yp, bk_xweights(bk_dx(nbx, invmesh, x + k, i0)));
}

// One prepped pixel {value, variance}: background off, variance = var_scale / weight (a weight at
// or below the threshold, a NaN pixel: bad = {., BIGVAR}).  The quotient is a reciprocal estimate
// and a multiply (1 ulp; the parity tolerance of a resampled weight is 5e-5): the staging of the
// fused coadd evaluates this once per staged pixel.
__device__ inline float2 prep_pixel(float v, float w, bool has_w, float bg, float var_scale, float wthresh) {
    const float val = v - bg;
    const bool ok = (val == val);                         // NaN pixels are bad
    // (one select per plane: the weight test and the NaN test meet in the scalar condition)
    const bool good = has_w ? (ok && w > wthresh) : ok;
    const float var = has_w ? var_scale * __builtin_amdgcn_rcpf(w) : var_scale;
    return make_float2(ok ? val : 0.f, good ? var : ZM_BIGVAR);
}

// four prepped pixels (x a multiple of 4): two float4 {value, variance, value, variance}
__device__ inline void prep_quad(const float* __restrict__ img, const float* __restrict__ wgt, int nx,
                                 const float* __restrict__ bk, int nbx, int nby, float invmesh,
                                 float var_scale, float wthresh, int vec_ok, int x, int y, float4 o[2]) {
    float v[4] = {0.f, 0.f, 0.f, 0.f}, w[4] = {1.f, 1.f, 1.f, 1.f};
    const size_t idx = (size_t)y * nx + x;
    if (vec_ok && x + 3 < nx) {
        const float4 a = *reinterpret_cast<const float4*>(img + idx);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        if (wgt) {
            const float4 b = *reinterpret_cast<const float4*>(wgt + idx);
            w[0] = b.x; w[1] = b.y; w[2] = b.z; w[3] = b.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (x + k < nx) {
                v[k] = img[idx + k];
                if (wgt) w[k] = wgt[idx + k];
            }
    }
    float bg[4] = {0.f, 0.f, 0.f, 0.f};
    if (bk) bk_eval4(bk, nbx, nby, invmesh, x, y, bg);
    float r[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float2 p = make_float2(0.f, ZM_BIGVAR);
        if (x + k < nx) p = prep_pixel(v[k], w[k], wgt != nullptr, bg[k], var_scale, wthresh);
        r[2 * k] = p.x;
        r[2 * k + 1] = p.y;
    }
    o[0] = make_float4(r[0], r[1], r[2], r[3]);
    o[1] = make_float4(r[4], r[5], r[6], r[7]);
}
#pragma clang fp contract(fast)

__global__ __launch_bounds__(256) void k_prep(const float* __restrict__ img,
                                              const float* __restrict__ wgt, int nx, int ny,
                                              const float* __restrict__ bk, int nbx, int nby,
                                              float invmesh,
                                              const float* __restrict__ var_scale_dev,
                                              float wthresh, int vec_ok, float2* __restrict__ dst,
                                              int spitch) {
    const int x = (blockIdx.x * blockDim.x + threadIdx.x) * 4;   // pixel quad
    const int y = blockIdx.y;
    if (x >= spitch) return;
    const float var_scale = var_scale_dev ? var_scale_dev[0] : 1.0f;
    float4 o[2];
    prep_quad(img, wgt, nx, bk, nbx, nby, invmesh, var_scale, wthresh, vec_ok, x, y, o);
    float4* d4 = reinterpret_cast<float4*>(dst + (size_t)y * spitch + x);
    d4[0] = o[0];
    if (x + 2 < spitch) d4[1] = o[1];
}

// ---------------------------------------------------------------------------
// Unit-sum Lanczos-3 taps for d in [SNAP, 1 - SNAP]; offsets k = -2..3.
//   t_k ~ n_k / x_k^2,  x_k = d + 2 - k,  n = {n1, n2, n3, n1, n2, n3}
// with the sin recurrence of SWarp's make_kernel (sin(a +- 2 pi / 3) expanded,
// a = pi d / 3 in [0, pi/3], sin / cos by polynomial).  Since the taps are
// normalised anyway they are evaluated over the common denominator
//   prod x_k^2 = (p1 p2 p3)^2,  p1 = x0 x5 = q - 6, p2 = x1 x4 = q - 2, p3 = x2 x3 = q,
//   q = d^2 - d,
// i.e. t_0 ~ n1 (x5 p2 p3)^2, t_5 ~ n3 (x0 p2 p3)^2, ... : no per-tap reciprocal,
// one reciprocal for the sum.  Every operation is written on a 2-vector holding
// the x axis in lane 0 and the y axis in lane 1, which maps onto the packed fp32
// VALU (v_pk_mul / v_pk_add / v_pk_fma): a VALU instruction costs 4 cycles per
// wave whatever it computes, so the instruction count is what bounds this kernel.
typedef float zm_v2f __attribute__((ext_vector_type(2)));

__host__ __device__ inline zm_v2f zm_rcp2(zm_v2f v) {
#ifdef __HIP_DEVICE_COMPILE__
    return (zm_v2f){__builtin_amdgcn_rcpf(v.x), __builtin_amdgcn_rcpf(v.y)};
#else
    return (zm_v2f){1.f / v.x, 1.f / v.y};
#endif
}

__host__ __device__ inline void zm_lanczos3_pair(zm_v2f d, zm_v2f t[6]) {
    const zm_v2f a = d * 1.0471975511965976f;
    const zm_v2f a2 = a * a;
    const zm_v2f s = a * (1.f + a2 * (-1.6666667e-1f + a2 * (8.3333333e-3f + a2 * (-1.98412698e-4f
                     + a2 * (2.7557319e-6f + a2 * -2.5052108e-8f)))));
    const zm_v2f c = 1.f + a2 * (-0.5f + a2 * (4.1666667e-2f + a2 * (-1.3888889e-3f
                     + a2 * (2.4801587e-5f + a2 * (-2.7557319e-7f + a2 * 2.0876757e-9f)))));
    const zm_v2f hs = 0.5f * s, hc = 0.8660254037844386f * c;
    const zm_v2f n1 = hs - hc, n2 = hs + hc, n3 = -s;
    const zm_v2f x0 = d + 2.f, x1 = d + 1.f, x2 = d, x3 = d - 1.f, x4 = d - 2.f, x5 = d - 3.f;
    const zm_v2f q = d * d - d;
    const zm_v2f p1 = q - 6.f, p2 = q - 2.f, p3 = q;
    const zm_v2f p23 = p2 * p3, p13 = p1 * p3, p12 = p1 * p2;
    zm_v2f u0 = x5 * p23, u5 = x0 * p23, u1 = x4 * p13, u4 = x1 * p13, u2 = x3 * p12, u3 = x2 * p12;
    t[0] = n1 * (u0 * u0); t[1] = n2 * (u1 * u1); t[2] = n3 * (u2 * u2);
    t[3] = n1 * (u3 * u3); t[4] = n2 * (u4 * u4); t[5] = n3 * (u5 * u5);
    const zm_v2f inv = zm_rcp2(((t[0] + t[1]) + (t[2] + t[3])) + (t[4] + t[5]));
#pragma unroll
    for (int k = 0; k < 6; ++k) t[k] *= inv;
}

__host__ __device__ inline void zm_lanczos3(float d, float t[6]) {
    zm_v2f tt[6];
    zm_lanczos3_pair((zm_v2f){d, d}, tt);
    for (int k = 0; k < 6; ++k) t[k] = tt[k].x;
}

// ---------------------------------------------------------------------------
// Tabulated taps.  The evaluation above costs ~60 packed VALU instructions per pixel (both
// axes), a third of the resample kernel's vector work, and that kernel is bound by vector
// issue (tools/valu_rate.hip: v_pk_fma_f32 5.1, v_fma_f32 3.4 cycles per instruction and SIMD
// at 4 waves per SIMD).  The six unit-sum taps are smooth functions of d, so the kernel reads
// them from a table in LDS instead: LZ_N + 1 nodes d_i = i / LZ_N, per node and tap the
// quadratic through the three Chebyshev points of [d_i - h/2, d_i + h/2] (h = 1 / LZ_N),
//   t_k(d) ~ c0 + (d - d_i) (c1 + (d - d_i) c2),
// 2.4e-7 from the exact taps at LZ_N = 64 (the direct fp32 evaluation: 1e-7; the parity
// tolerance of the resampled pixels is 2e-5).  Neighbouring lanes look at the same or the
// next node (d moves by ~1e-3 per output pixel), so the five 16-byte LDS reads of a lookup
// are broadcasts.  The coefficients of taps (k, k + 1) sit side by side: one packed FMA
// updates two taps.  Entry layout (20 floats):
//   {c0_0 c0_1 c1_0 c1_1} {c2_0 c2_1 c0_2 c0_3} {c1_2 c1_3 c2_2 c2_3} {c0_4 c0_5 c1_4 c1_5} {c2_4 c2_5 - -}
#define LZ_N 64
#define LZ_ENTRY 20
#define LZ_FLOATS ((LZ_N + 1) * LZ_ENTRY)

static void lz3_exact(double d, double t[6]) {
    const double PI = 3.14159265358979323846;
    double sum = 0.0;
    for (int k = 0; k < 6; ++k) {
        const double x = d - (double)(k - 2);
        t[k] = (x == 0.0) ? PI * PI / 3.0 : sin(PI * x) * sin(PI * x / 3.0) / (x * x);
        sum += t[k];
    }
    for (int k = 0; k < 6; ++k) t[k] /= sum;
}

// [LZ_N + 1][LZ_ENTRY] floats
void zm_lanczos_table(float* tab) {
    const double h = 1.0 / LZ_N, a = 0.5 * h * 0.86602540378443864676;
    for (int i = 0; i <= LZ_N; ++i) {
        double f0[6], fp[6], fm[6];
        lz3_exact(i * h, f0);
        lz3_exact(i * h + a, fp);
        lz3_exact(i * h - a, fm);
        float c0[6], c1[6], c2[6];
        for (int k = 0; k < 6; ++k) {
            c0[k] = (float)f0[k];
            c1[k] = (float)((fp[k] - fm[k]) / (2 * a));
            c2[k] = (float)((fp[k] - 2 * f0[k] + fm[k]) / (2 * a * a));
        }
        float* e = tab + (size_t)i * LZ_ENTRY;
        const float v[LZ_ENTRY] = {c0[0], c0[1], c1[0], c1[1], c2[0], c2[1], c0[2], c0[3], c1[2], c1[3],
                                   c2[2], c2[3], c0[4], c0[5], c1[4], c1[5], c2[4], c2[5], 0.f, 0.f};
        for (int q = 0; q < LZ_ENTRY; ++q) e[q] = v[q];
    }
}

// t[j] = {tap 2j, tap 2j + 1}; tab: the table (LDS on the device), d in [SNAP, 1 - SNAP]
__host__ __device__ inline void zm_lz3_lookup(const float* tab, float d, zm_v2f t[3]) {
    const float fi = __builtin_rintf(d * (float)LZ_N);
    const float dl = __builtin_fmaf(fi, -1.0f / LZ_N, d);          // exact: fi / LZ_N is a dyadic rational
    const float4* e = reinterpret_cast<const float4*>(tab) + 5 * (int)fi;
    const float4 a = e[0], b = e[1], c = e[2], g = e[3], h = e[4];
    const zm_v2f dd = (zm_v2f){dl, dl};
    t[0] = __builtin_elementwise_fma(dd, __builtin_elementwise_fma(dd, (zm_v2f){b.x, b.y}, (zm_v2f){a.z, a.w}),
                                     (zm_v2f){a.x, a.y});
    t[1] = __builtin_elementwise_fma(dd, __builtin_elementwise_fma(dd, (zm_v2f){c.z, c.w}, (zm_v2f){c.x, c.y}),
                                     (zm_v2f){b.z, b.w});
    t[2] = __builtin_elementwise_fma(dd, __builtin_elementwise_fma(dd, (zm_v2f){h.x, h.y}, (zm_v2f){g.z, g.w}),
                                     (zm_v2f){g.x, g.y});
}

// the table in device memory (one per context, filled on first use)
int zm_get_lanczos_table(zm_ctx* ctx, const float** out) {
    float* dev = nullptr;
    auto it = ctx->scratch.find("lz3_table");
    if (it != ctx->scratch.end()) {
        *out = (const float*)it->second.first;
        return 0;
    }
    ZM_TRY(ctx->get("lz3_table", sizeof(float) * LZ_FLOATS, (void**)&dev));
    std::vector<float> host(LZ_FLOATS);
    zm_lanczos_table(host.data());
    ZM_HIP(hipMemcpy(dev, host.data(), sizeof(float) * LZ_FLOATS, hipMemcpyHostToDevice));   // blocking, once
    *out = dev;
    return 0;
}

// what the kernels evaluate (table path), on the host: tests/test_abi.py pins it to the oracle
extern "C" void zm_debug_lanczos3(float d, float* out6) {
    static std::vector<float> tab;
    if (tab.empty()) {
        tab.resize(LZ_FLOATS);
        zm_lanczos_table(tab.data());
    }
    zm_v2f t[3];
    zm_lz3_lookup(tab.data(), d, t);
    for (int k = 0; k < 6; ++k) out6[k] = (k & 1) ? t[k >> 1].y : t[k >> 1].x;
}
// the direct evaluation (the nearest / fallback paths and the record of what the table replaced)
extern "C" void zm_debug_lanczos3_direct(float d, float* out6) { zm_lanczos3(d, out6); }

// floor / fraction with the snap rule of oracle/resample.py::split_position
__device__ inline void split_pos(float p, int* i, float* d, bool* delta) {
    float f = floorf(p);
    float fr = p - f;
    int ii = (int)f;
    if (fr > 1.f - ZM_SNAP) { ii += 1; fr = 0.f; }
    bool dl = fr < ZM_SNAP;
    *i = ii;
    *d = dl ? 0.f : fr;
    *delta = dl;
}

template <int KIND> struct taps_traits;
template <> struct taps_traits<ZM_RESAMPLE_LANCZOS3> { enum { N = 6, OFF = -2 }; };
template <> struct taps_traits<ZM_RESAMPLE_BILINEAR> { enum { N = 2, OFF = 0 }; };

// taps of both axes at once: t[k].x along x, t[k].y along y
template <int KIND>
__device__ inline void make_taps2(float dx, float dy, bool ddx, bool ddy, zm_v2f* t) {
    if (KIND == ZM_RESAMPLE_LANCZOS3) {
        zm_lanczos3_pair((zm_v2f){ddx ? 0.5f : dx, ddy ? 0.5f : dy}, t);
        // delta kernels are rare (aligned grids): patch them under a wave-uniform test
        if (__any(ddx || ddy)) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const float dl = (k == 2) ? 1.f : 0.f;
                t[k].x = ddx ? dl : t[k].x;
                t[k].y = ddy ? dl : t[k].y;
            }
        }
    } else {
        t[0] = (zm_v2f){1.f - dx, 1.f - dy};
        t[1] = (zm_v2f){dx, dy};
    }
}

// Tile header shared by the image and mask kernels: bbox of the input footprint
// and the 10 tile nodes relative to its origin, in fp32.
struct tile_hdr {
    float nrel[2][5][2];
    int bx0, by0, bw, bh;
};

__device__ inline void build_tile_header(const double2* __restrict__ lat, int lnx, int lny,
                                         int cx0, int cy0, int support_lo, int support_hi,
                                         tile_hdr* h) {
    // executed by the first wave; lanes 0..9 own one node each
    int lane = threadIdx.x & 63;
    int ngx = min(cx0 + (lane % 5), lnx - 1);
    int ngy = min(cy0 + (lane / 5), lny - 1);
    double2 nd = make_double2(0.0, 0.0);
    double mnx = 1e300, mxx = -1e300, mny = 1e300, mxy = -1e300;
    if (lane < 10) {
        nd = lat[(size_t)ngy * lnx + ngx];
        mnx = mxx = nd.x;
        mny = mxy = nd.y;
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) {
        mnx = fmin(mnx, __shfl_xor(mnx, o));
        mxx = fmax(mxx, __shfl_xor(mxx, o));
        mny = fmin(mny, __shfl_xor(mny, o));
        mxy = fmax(mxy, __shfl_xor(mxy, o));
    }
    // clamp wild positions (frames far off the grid) so the int conversion is safe
    mnx = fmax(fmin(mnx, 1e8), -1e8); mxx = fmax(fmin(mxx, 1e8), -1e8);
    mny = fmax(fmin(mny, 1e8), -1e8); mxy = fmax(fmin(mxy, 1e8), -1e8);
    int bx0 = ((int)floor(mnx) + support_lo - 1) & ~1;
    int by0 = (int)floor(mny) + support_lo - 1;
    int bx1 = (int)floor(mxx) + support_hi + 2;
    int by1 = (int)floor(mxy) + support_hi + 2;
    int bw = (bx1 - bx0 + 2) & ~1;
    int bh = by1 - by0 + 1;
    if (lane < 10) {
        h->nrel[lane / 5][lane % 5][0] = (float)(nd.x - bx0);
        h->nrel[lane / 5][lane % 5][1] = (float)(nd.y - by0);
    }
    if (lane == 0) { h->bx0 = bx0; h->by0 = by0; h->bw = bw; h->bh = bh; }
}

// the same for the 64 x 32 tiles of k_resample: 3 x 5 nodes
struct tile_hdr3 {
    float nrel[3][5][2];
    int bx0, by0, bw, bh;
};

__device__ inline double2 zm_lat_load(const double2* p) { return *p; }
__device__ inline double2 zm_lat_load(const double2 __attribute__((address_space(1)))* p) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    const v2d v = *(const v2d __attribute__((address_space(1)))*)p;
    return make_double2(v.x, v.y);
}

template <typename LatPtr>
__device__ inline void build_tile_header3(LatPtr lat, int lnx, int lny,
                                          int cx0, int cy0, int support_lo, int support_hi,
                                          tile_hdr3* h) {
    // executed by the first wave; lanes 0..14 own one node each
    int lane = threadIdx.x & 63;
    int ngx = min(cx0 + (lane % 5), lnx - 1);
    int ngy = min(cy0 + (lane / 5), lny - 1);
    double2 nd = make_double2(0.0, 0.0);
    double mnx = 1e300, mxx = -1e300, mny = 1e300, mxy = -1e300;
    if (lane < 15) {
        nd = zm_lat_load(lat + ((size_t)ngy * lnx + ngx));
        mnx = mxx = nd.x;
        mny = mxy = nd.y;
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) {
        mnx = fmin(mnx, __shfl_xor(mnx, o));
        mxx = fmax(mxx, __shfl_xor(mxx, o));
        mny = fmin(mny, __shfl_xor(mny, o));
        mxy = fmax(mxy, __shfl_xor(mxy, o));
    }
    mnx = fmax(fmin(mnx, 1e8), -1e8); mxx = fmax(fmin(mxx, 1e8), -1e8);
    mny = fmax(fmin(mny, 1e8), -1e8); mxy = fmax(fmin(mxy, 1e8), -1e8);
    // the box starts on a multiple of 4 pixels and is a multiple of 4 wide: rows of the prepped
    // plane start 32-byte aligned, rows of the 16-bit box-OR plane 8-byte aligned
    int bx0 = ((int)floor(mnx) + support_lo - 1) & ~3;
    int by0 = (int)floor(mny) + support_lo - 1;
    int bx1 = (int)floor(mxx) + support_hi + 2;
    int by1 = (int)floor(mxy) + support_hi + 2;
    int bw = (bx1 - bx0 + 4) & ~3;
    int bh = by1 - by0 + 1;
    if (lane < 15) {
        h->nrel[lane / 5][lane % 5][0] = (float)(nd.x - bx0);
        h->nrel[lane / 5][lane % 5][1] = (float)(nd.y - by0);
    }
    if (lane == 0) { h->bx0 = bx0; h->by0 = by0; h->bw = bw; h->bh = bh; }
}

__device__ inline void tile_position(const tile_hdr* h, int tx, int ty, float* px, float* py) {
    int cell = tx >> 4;
    float fx = (float)(tx & 15) * (1.f / LSTEP);
    float fy = (float)ty * (1.f / LSTEP);
    float x00 = h->nrel[0][cell][0], x10 = h->nrel[0][cell + 1][0];
    float x01 = h->nrel[1][cell][0], x11 = h->nrel[1][cell + 1][0];
    float y00 = h->nrel[0][cell][1], y10 = h->nrel[0][cell + 1][1];
    float y01 = h->nrel[1][cell][1], y11 = h->nrel[1][cell + 1][1];
    float xa = x00 + fx * (x10 - x00), xb = x01 + fx * (x11 - x01);
    float ya = y00 + fx * (y10 - y00), yb = y01 + fx * (y11 - y01);
    *px = xa + fy * (xb - xa);
    *py = ya + fy * (yb - ya);
}

// LDS row reads as single ds_read_b64 instructions.  Left to the compiler, the six
// adjacent {value, variance} pairs of a tap row become ds_read2_b64, which moves
// half the bytes per LDS cycle (MI355X_MICROARCH.md, LDS table).  The wait is part
// of the same statement sequence and carries the values, so no consumer can be
// scheduled above it.
template <int NT> struct lds_row;
template <> struct lds_row<6> {
    static __device__ inline void read(const float2* p, float2 (&s)[6]) {
        const unsigned a = (unsigned)(size_t)p;
        unsigned long long r0, r1, r2, r3, r4, r5;
        asm volatile("ds_read_b64 %0, %6\n\t"
                     "ds_read_b64 %1, %6 offset:8\n\t"
                     "ds_read_b64 %2, %6 offset:16\n\t"
                     "ds_read_b64 %3, %6 offset:24\n\t"
                     "ds_read_b64 %4, %6 offset:32\n\t"
                     "ds_read_b64 %5, %6 offset:40\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5)
                     : "v"(a)
                     : "memory");
        const unsigned long long r[6] = {r0, r1, r2, r3, r4, r5};
#pragma unroll
        for (int c = 0; c < 6; ++c)
            s[c] = make_float2(__uint_as_float((unsigned)r[c]), __uint_as_float((unsigned)(r[c] >> 32)));
    }
};
template <> struct lds_row<2> {
    static __device__ inline void read(const float2* p, float2 (&s)[2]) {
        const unsigned a = (unsigned)(size_t)p;
        unsigned long long r0, r1;
        asm volatile("ds_read_b64 %0, %2\n\t"
                     "ds_read_b64 %1, %2 offset:8\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(r0), "=&v"(r1)
                     : "v"(a)
                     : "memory");
        s[0] = make_float2(__uint_as_float((unsigned)r0), __uint_as_float((unsigned)(r0 >> 32)));
        s[1] = make_float2(__uint_as_float((unsigned)r1), __uint_as_float((unsigned)(r1 >> 32)));
    }
};

// MASKOP 0: no mask; 1: store the resampled mask (0 where not covered);
// 2: accumulate into macc with `mkind` (AND / OR), -1 = "no frame covered yet".
//
// Persistent, software-pipelined: a workgroup walks tiles t, t + G, t + 2G, ...
// While it interpolates tile k out of LDS, the global loads of tile k + 1 are in
// flight into registers and wave 0 builds the header of tile k + 2, so neither
// the lattice / pixel load latency nor the block launch cost sits on the critical
// path.  PF float4 (+ int2) registers per thread bound the staged tile; larger
// footprints (strong rotation / scale change) gather from global memory instead.
#define RS_PF 7                      // prefetch slots per thread: 7 x 256 float4 = 3584 px
#define RS_PFCAP (RS_PF * 256 * 2)

struct rs_hdr {
    tile_hdr3 h;
    int use_lds, touches;
};

template <int KIND>
__device__ inline void rs_build_header(const double2* __restrict__ lat, int lnx, int lny, int t,
                                       int ntx, int nx, int ny, int lds_cap, rs_hdr* H) {
    constexpr int NT = taps_traits<KIND>::N;
    constexpr int OFF = taps_traits<KIND>::OFF;
    const int tyi = t / ntx, txi = t - tyi * ntx;
    build_tile_header3(lat, lnx, lny, txi * (TW / LSTEP), tyi * (RTH / LSTEP), OFF, OFF + NT - 1, &H->h);
    if ((threadIdx.x & 63) == 0) {
        const int bx0 = H->h.bx0, by0 = H->h.by0, bw = H->h.bw, bh = H->h.bh;
        const int touches = (bx0 < nx) && (bx0 + bw > 0) && (by0 < ny) && (by0 + bh > 0);
        const long long area = (long long)bw * bh;
        H->touches = touches;
        H->use_lds = touches && area <= (long long)lds_cap && area <= RS_PFCAP;
    }
}

// Box OR of an integer mask: B[y][x] = OR of m[y .. y + NT - 1][x .. x + NT - 1] where the
// whole window lies on the frame (other entries are never read).  One 64 x 16 tile per
// workgroup, separable in LDS.  The resample kernel then needs a single gather per pixel.
// The plane holds 16 bits per pixel (ZTF masks are 16-bit: 2 B written here and gathered by the
// resample kernel instead of 4).  Where the OR has a bit above 15 - a reference mask carrying
// bit 16 - the entry is ZM_BOX_RAW and the resample kernel ORs the raw mask under that footprint
// itself (the path it already has for delta kernels); a genuine 0xffff takes that path too.
#define ZM_BOX_RAW 0xffffu
__device__ inline uint16_t box_entry(int32_t o) {
    return ((uint32_t)o >> 16) ? (uint16_t)ZM_BOX_RAW : (uint16_t)o;
}

template <int NT>
__global__ __launch_bounds__(256) void k_mask_box(const int32_t* __restrict__ m, int nx, int ny,
                                                  uint16_t* __restrict__ B) {
    constexpr int TWB = 64, THB = 16, IW = TWB + NT - 1, IH = THB + NT - 1, IP = IW + 1;
    __shared__ int32_t t0[IH * IP];
    __shared__ int32_t h[IH * TWB];
    const int x0 = blockIdx.x * TWB, y0 = blockIdx.y * THB, tid = threadIdx.x;
    constexpr int NLD = (IH * IW + 255) / 256;          // loads first, LDS stores after: one latency
    int32_t mm[NLD];
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
        const int e = tid + 256 * q;
        const int r = e / IW, c = e - r * IW;
        const int x = x0 + c, y = y0 + r;
        mm[q] = (e < IH * IW && x < nx && y < ny) ? m[(size_t)y * nx + x] : 0;
    }
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
        const int e = tid + 256 * q;
        if (e < IH * IW) t0[(e / IW) * IP + (e % IW)] = mm[q];
    }
    __syncthreads();
    for (int e = tid; e < IH * TWB; e += 256) {
        const int r = e / TWB, c = e - r * TWB;
        int32_t o = 0;
#pragma unroll
        for (int k = 0; k < NT; ++k) o |= t0[r * IP + c + k];
        h[e] = o;
    }
    __syncthreads();
    for (int e = tid; e < THB * TWB; e += 256) {
        const int r = e / TWB, c = e - r * TWB;
        const int x = x0 + c, y = y0 + r;
        if (x + NT <= nx && y + NT <= ny) {
            int32_t o = 0;
#pragma unroll
            for (int k = 0; k < NT; ++k) o |= h[(r + k) * TWB + c];
            B[(size_t)y * nx + x] = box_entry(o);
        }
    }
}

// k_prep and k_mask_box of one frame in one launch (one 64 x 16 tile of input pixels per
// workgroup): a stack is launch-gap bound between its three per-frame kernels.
template <int NT>
__global__ __launch_bounds__(256) void k_prep_box(const float* __restrict__ img,
                                                  const float* __restrict__ wgt, int nx, int ny,
                                                  const float* __restrict__ bk, int nbx, int nby,
                                                  float invmesh, const float* __restrict__ var_scale_dev,
                                                  float wthresh, int vec_ok, float2* __restrict__ dst,
                                                  int spitch, const int32_t* __restrict__ m,
                                                  uint16_t* __restrict__ B, int bpitch) {
    constexpr int TWB = 64, THB = 16, IW = TWB + NT - 1, IH = THB + NT - 1, IP = IW + 1;
    __shared__ int32_t t0[IH * IP];
    __shared__ int32_t h[IH * TWB];
    const int x0 = blockIdx.x * TWB, y0 = blockIdx.y * THB, tid = threadIdx.x;
    // the mask loads of this thread go out together and land in LDS after the prep below (a loop
    // of load -> LDS store is waited for load by load)
    constexpr int NLD = (IH * IW + 255) / 256;
    int32_t mm[NLD];
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
        const int e = tid + 256 * q;
        const int r = e / IW, c = e - r * IW;
        const int x = x0 + c, y = y0 + r;
        mm[q] = (e < IH * IW && x < nx && y < ny) ? m[(size_t)y * nx + x] : 0;
    }
    // the prep of this tile while the mask loads are in flight: one pixel quad per thread
    {
        const float var_scale = var_scale_dev ? var_scale_dev[0] : 1.0f;
        const int r = tid / (TWB / 4), x = x0 + 4 * (tid - r * (TWB / 4)), y = y0 + r;
        if (y < ny && x < spitch) {
            float4 o[2];
            prep_quad(img, wgt, nx, bk, nbx, nby, invmesh, var_scale, wthresh, vec_ok, x, y, o);
            typedef float zm_nt4 __attribute__((ext_vector_type(4)));
            zm_nt4* d4 = reinterpret_cast<zm_nt4*>(dst + (size_t)y * spitch + x);
            __builtin_nontemporal_store((zm_nt4){o[0].x, o[0].y, o[0].z, o[0].w}, &d4[0]);
            if (x + 2 < spitch) __builtin_nontemporal_store((zm_nt4){o[1].x, o[1].y, o[1].z, o[1].w}, &d4[1]);
        }
    }
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
        const int e = tid + 256 * q;
        if (e < IH * IW) t0[(e / IW) * IP + (e % IW)] = mm[q];
    }
    __syncthreads();
    for (int e = tid; e < IH * TWB; e += 256) {
        const int r = e / TWB, c = e - r * TWB;
        int32_t o = 0;
#pragma unroll
        for (int k = 0; k < NT; ++k) o |= t0[r * IP + c + k];
        h[e] = o;
    }
    __syncthreads();
    for (int e = tid; e < THB * TWB; e += 256) {
        const int r = e / TWB, c = e - r * TWB;
        const int x = x0 + c, y = y0 + r;
        if (x + NT <= nx && y + NT <= ny) {
            int32_t o = 0;
#pragma unroll
            for (int k = 0; k < NT; ++k) o |= h[(r + k) * TWB + c];
            B[(size_t)y * bpitch + x] = box_entry(o);
        }
    }
}

// mask_for_box != NULL (box_nt = taps of the resampling kernel, 6 or 2): also fill the
// "mask_box" scratch plane for the zm_launch_resample call that follows
int zm_launch_prep(zm_ctx* ctx, const float* img, const float* wgt, int nx, int ny,
                   const float* bknodes, int nbx, int nby, int mesh, const float* var_scale_dev,
                   float wthresh, float2* dst, int spitch, const int32_t* mask_for_box, int box_nt,
                   uint16_t* mbox_out, int mbox_pitch) {
    if (mbox_pitch <= 0) mbox_pitch = nx;
    const float invmesh = mesh > 0 ? 1.0f / mesh : 0.f;
    const int vec_ok = (nx % 4 == 0) && (((uintptr_t)img & 15) == 0) && (((uintptr_t)wgt & 15) == 0);
    ctx->box_ready_for = nullptr;
    if (mask_for_box && (box_nt == 6 || box_nt == 2)) {
        uint16_t* mbox = mbox_out;      // caller's plane (fused coadd: one per frame) or the ctx scratch
        if (!mbox) ZM_TRY(ctx->get("mask_box", sizeof(uint16_t) * (size_t)nx * ny, (void**)&mbox));
        dim3 grd(zm_div_up(spitch, 64), zm_div_up(ny, 16), 1);
        zm_scope_timer t(ctx, "prep");
        if (box_nt == 6)
            hipLaunchKernelGGL(k_prep_box<6>, grd, dim3(256), 0, ctx->stream, img, wgt, nx, ny, bknodes, nbx,
                               nby, invmesh, var_scale_dev, wthresh, vec_ok, dst, spitch, mask_for_box, mbox, mbox_pitch);
        else
            hipLaunchKernelGGL(k_prep_box<2>, grd, dim3(256), 0, ctx->stream, img, wgt, nx, ny, bknodes, nbx,
                               nby, invmesh, var_scale_dev, wthresh, vec_ok, dst, spitch, mask_for_box, mbox, mbox_pitch);
        ZM_HIP(hipGetLastError());
        if (!mbox_out) {
            ctx->box_ready_for = mask_for_box;
            ctx->box_ready_nt = box_nt;
        }
        return 0;
    }
    dim3 blk(256, 1, 1), grd(zm_div_up(zm_div_up(spitch, 4), 256), ny, 1);
    zm_scope_timer t(ctx, "prep");
    hipLaunchKernelGGL(k_prep, grd, blk, 0, ctx->stream, img, wgt, nx, ny, bknodes, nbx, nby, invmesh,
                       var_scale_dev, wthresh, vec_ok, dst, spitch);
    ZM_HIP(hipGetLastError());
    return 0;
}

template <int KIND, int MASKOP>
__global__ __launch_bounds__(256, 4) void k_resample(
    const float2* __restrict__ src, int nx, int ny, int spitch, const double2* __restrict__ lat,
    int lnx, int lny, float fscale, float2* __restrict__ dst, int onx, int ony, int lds_cap,
    const int32_t* __restrict__ mask, const uint16_t* __restrict__ mbox, int32_t* __restrict__ macc,
    int mkind, int mfirst, int ntx, int ntiles, const float* __restrict__ taptab,
    float* __restrict__ plane_a, float* __restrict__ plane_b) {
    // (plane_a / plane_b: value and weight as two planes - what an alignment hands back - instead of the pair
    // plane `dst` that a stack keeps: no pass to split them afterwards)
    extern __shared__ float4 smem4[];
    rs_hdr* HR = reinterpret_cast<rs_hdr*>(smem4);                 // ring of 3 headers
    // Lanczos-3: the tap table sits between the headers and the pixel tile
    constexpr int TABF = (KIND == ZM_RESAMPLE_LANCZOS3) ? LZ_FLOATS : 0;
    const float* ltab = reinterpret_cast<const float*>(smem4) + HDR_FLOATS;
    float2* tile = reinterpret_cast<float2*>(smem4) + (HDR_FLOATS + TABF) / 2;
    constexpr int NT = taps_traits<KIND>::N;
    constexpr int OFF = taps_traits<KIND>::OFF;
    constexpr int CI = -OFF;                                       // tap index of a delta kernel
    const int tid = threadIdx.x;
    const int G = gridDim.x;
    const float fscale2 = fscale * fscale;
    const float4 fill = make_float4(0.f, ZM_BIGVAR, 0.f, ZM_BIGVAR);

    float4 pf[RS_PF];
    // issue the loads of tile `t` (header H) into the prefetch registers
    auto prefetch = [&](const rs_hdr* H) {
        if (!H->use_lds) return;
        const int bx0 = H->h.bx0, by0 = H->h.by0, bw2 = H->h.bw >> 1;
        const int n4 = bw2 * H->h.bh;
        const float inv = 1.0f / (float)bw2;
#pragma unroll
        for (int k = 0; k < RS_PF; ++k) {
            const int e = tid + 256 * k;
            pf[k] = fill;
            if (e < n4) {
                const int r = (int)(((float)e + 0.5f) * inv);
                const int c2 = e - r * bw2;
                const int gy = by0 + r, gx = bx0 + 2 * c2;
                if (gy >= 0 && gy < ny && gx >= 0 && gx < spitch)
                    pf[k] = *reinterpret_cast<const float4*>(src + (size_t)gy * spitch + gx);
            }
        }
    };
    auto store = [&](const rs_hdr* H) {
        if (!H->use_lds) return;
        const int n4 = (H->h.bw >> 1) * H->h.bh;
#pragma unroll
        for (int k = 0; k < RS_PF; ++k) {
            const int e = tid + 256 * k;
            if (e < n4)
                *reinterpret_cast<float4*>(tile + 2 * e) = pf[k];      // rows are bw = 2 bw2 wide: linear
        }
    };

    int t = blockIdx.x;
    if (t >= ntiles) return;
    if (KIND == ZM_RESAMPLE_LANCZOS3) {
        for (int e = tid; e < LZ_FLOATS / 4; e += 256)
            smem4[HDR_FLOATS / 4 + e] = reinterpret_cast<const float4*>(taptab)[e];
    }
    if (tid < 64) {
        rs_build_header<KIND>(lat, lnx, lny, t, ntx, nx, ny, lds_cap, &HR[0]);
        if (t + G < ntiles) rs_build_header<KIND>(lat, lnx, lny, t + G, ntx, nx, ny, lds_cap, &HR[1]);
    }
    __syncthreads();
    prefetch(&HR[0]);
    int slot = 0;
    for (; t < ntiles; t += G) {
        const rs_hdr* H = &HR[slot];
        const int nslot = slot == 2 ? 0 : slot + 1;
        const int nnslot = nslot == 2 ? 0 : nslot + 1;
        store(H);
        __syncthreads();
        const bool use_lds = H->use_lds, touches = H->touches;
        const int bx0 = H->h.bx0, by0 = H->h.by0, bw = H->h.bw, bh = H->h.bh;
        const int tyi = t / ntx, txi = t - tyi * ntx;
        const int ox0 = txi * TW, oy0 = tyi * RTH;
        const int tx = tid & 63, tyb = tid >> 6;
        const int ox = ox0 + tx;
        // the running mask coadd of this thread's four pixels: loaded before the prefetch
        // so that waiting for it (vmcnt is in order) does not wait for the next tile
        int32_t aprev[RTH / 4];
#pragma unroll
        for (int q = 0; q < RTH / 4; ++q) aprev[q] = -1;
        if (MASKOP == 2 && !mfirst) {
#pragma unroll
            for (int q = 0; q < RTH / 4; ++q) {
                const int oy = oy0 + tyb + 4 * q;
                if (ox < onx && oy < ony) aprev[q] = macc[(size_t)oy * onx + ox];
            }
        }
        // next tile's pixels into registers, the tile after that gets its header
        if (t + G < ntiles) prefetch(&HR[nslot]);
        if (tid < 64 && t + 2 * G < ntiles)
            rs_build_header<KIND>(lat, lnx, lny, t + 2 * G, ntx, nx, ny, lds_cap, &HR[nnslot]);

        // the four pixels of a thread share their column: the x part of the bilinear lattice
        // interpolation is done once, each pixel adds its row fraction (same operations and
        // order as tile_position)
        // (a tile spans two lattice cells in y: node rows 0 / 1 for its upper 16 rows, 1 / 2 below)
        float xr[3], yr[3];
        {
            const int cell = tx >> 4;
            const float fx = (float)(tx & 15) * (1.f / LSTEP);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float x0 = H->h.nrel[r][cell][0], x1 = H->h.nrel[r][cell + 1][0];
                const float y0 = H->h.nrel[r][cell][1], y1 = H->h.nrel[r][cell + 1][1];
                xr[r] = x0 + fx * (x1 - x0);
                yr[r] = y0 + fx * (y1 - y0);
            }
        }
#pragma unroll 1
        for (int q = 0; q < RTH / 4; ++q) {
            const int ty = tyb + 4 * q;
            const int oy = oy0 + ty;
            if (ox >= onx || oy >= ony) continue;
            const int cr = q >> 2;                                   // ty >> 4: the lattice cell row
            const float fy = (float)(ty & 15) * (1.f / LSTEP);
            const float xa = cr ? xr[1] : xr[0], xb = cr ? xr[2] : xr[1];
            const float ya = cr ? yr[1] : yr[0], yb = cr ? yr[2] : yr[1];
            const float px = xa + fy * (xb - xa), py = ya + fy * (yb - ya);
            int ixr, iyr;
            float dx, dy;
            bool ddx, ddy;
            split_pos(px, &ixr, &dx, &ddx);
            split_pos(py, &iyr, &dy, &ddy);
            const int ix = bx0 + ixr + OFF, iy = by0 + iyr + OFF;   // first tap, absolute
            // the non-zero taps must lie on the frame: per axis the whole footprint, or only the
            // centre pixel of a delta kernel (oracle/resample.py::on_frame)
            const bool inbx = ddx ? (ix + CI >= 0 && ix + CI < nx) : (ix >= 0 && ix + NT <= nx);
            const bool inby = ddy ? (iy + CI >= 0 && iy + CI < ny) : (iy >= 0 && iy + NT <= ny);
            const bool inb = touches && inbx && inby;
            float2 res = make_float2(0.f, 0.f);
            int32_t mres = 0;
            uint32_t m16 = 0;                     // box-OR entry of a non-delta footprint
            if (inb) {
                if (MASKOP) {
                    // every tap of a non-delta axis is non-zero: the OR over the NT x NT footprint
                    // is one gather from the box-OR plane (k_mask_box); delta kernels (aligned
                    // grids) only touch the centre tap of that axis and read the raw mask
                    // (the entry is only looked at after the interpolation below: its latency
                    // hides behind the taps)
                    if (!(ddx || ddy)) {
                        m16 = mbox[(size_t)iy * nx + ix];
                    } else {
                        const int c0 = ddx ? CI : 0, c1 = ddx ? CI + 1 : NT;
                        const int r0 = ddy ? CI : 0, r1 = ddy ? CI + 1 : NT;
                        for (int r = r0; r < r1; ++r) {
                            const int32_t* mp = mask + (size_t)(iy + r) * nx + ix;
                            for (int c = c0; c < c1; ++c) mres |= mp[c];
                        }
                    }
                }
                zm_v2f tw[NT];                     // tw[k] = {x tap k, y tap k}
                float acc = 0.f, vacc = 0.f;
                if (KIND == ZM_RESAMPLE_LANCZOS3) {
                    // taps from the LDS table, two per register pair: txp[j] = {x tap 2j, x tap 2j + 1}
                    zm_v2f txp[3], typ[3];
                    zm_lz3_lookup(ltab, ddx ? 0.5f : dx, txp);
                    zm_lz3_lookup(ltab, ddy ? 0.5f : dy, typ);
                    // delta kernels are rare (aligned grids): patch them under a wave-uniform test
                    if (__any(ddx || ddy)) {
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            const zm_v2f dl = (zm_v2f){j == 1 ? 1.f : 0.f, 0.f};
                            txp[j] = ddx ? dl : txp[j];
                            typ[j] = ddy ? dl : typ[j];
                        }
                    }
#pragma unroll
                    for (int k = 0; k < NT; ++k)
                        tw[k] = (zm_v2f){(k & 1) ? txp[k >> 1].y : txp[k >> 1].x,
                                         (k & 1) ? typ[k >> 1].y : typ[k >> 1].x};
                } else {
                    make_taps2<KIND>(dx, dy, ddx, ddy, tw);
                }
                if (use_lds) {
                    const float2* p = tile + (iyr + OFF) * bw + (ixr + OFF);
                    zm_v2f av = (zm_v2f){0.f, 0.f};     // {value, variance} accumulators
#pragma unroll
                    for (int r = 0; r < NT; ++r) {
                        float2 s[NT];
                        lds_row<NT>::read(p, s);
                        zm_v2f rv2 = (zm_v2f){0.f, 0.f};
#pragma unroll
                        for (int c = 0; c < NT; ++c)
                            rv2 = __builtin_elementwise_fma((zm_v2f){tw[c].x, tw[c].x},
                                                            (zm_v2f){s[c].x, s[c].y}, rv2);
                        av = __builtin_elementwise_fma((zm_v2f){tw[r].y, tw[r].y}, rv2, av);
                        p += bw;
                    }
                    acc = av.x;
                    vacc = av.y;
                } else {
                    const float2* p = src + (size_t)iy * spitch + ix;
#pragma unroll
                    for (int r = 0; r < NT; ++r) {
                        float ra = 0.f, rv = 0.f;
                        // (zero taps of a delta axis may lie off the frame: not read)
                        if (tw[r].y != 0.f) {
#pragma unroll
                            for (int c = 0; c < NT; ++c) {
                                if (tw[c].x != 0.f) {
                                    float2 s = p[c];
                                    ra = fmaf(tw[c].x, s.x, ra);
                                    rv = fmaf(tw[c].x, s.y, rv);
                                }
                            }
                        }
                        acc = fmaf(tw[r].y, ra, acc);
                        vacc = fmaf(tw[r].y, rv, vacc);
                        p += spitch;
                    }
                }
                if (vacc > 0.f && vacc < ZM_BADVAR_TEST) {
                    res.x = acc * fscale;
                    res.y = __builtin_amdgcn_rcpf(vacc * fscale2);      // 1 ulp: one instruction
                }
                if (MASKOP && !(ddx || ddy)) {
                    if (m16 != ZM_BOX_RAW) {
                        mres = (int32_t)m16;
                    } else {                        // bits above 15 somewhere under the footprint
                        for (int r = 0; r < NT; ++r) {
                            const int32_t* mp = mask + (size_t)(iy + r) * nx + ix;
                            for (int c = 0; c < NT; ++c) mres |= mp[c];
                        }
                    }
                }
            }
            const size_t oidx = (size_t)oy * onx + ox;
            if (plane_a) {
                plane_a[oidx] = res.x;
                plane_b[oidx] = res.y;
            } else {
                dst[oidx] = res;
            }
            if (MASKOP == 1) {
                macc[oidx] = mres;
            } else if (MASKOP == 2) {
                int32_t a = aprev[0];
#pragma unroll
                for (int k = 1; k < RTH / 4; ++k) a = (q == k) ? aprev[k] : a;
                if (inb) {
                    if (a == -1) a = mres;
                    else a = (mkind == ZM_MASK_AND) ? (a & mres) : (a | mres);
                }
                macc[oidx] = a;
            }
        }
        __syncthreads();          // everyone is done with the LDS tile and with header `slot`
        slot = nslot;
    }
}

// nearest neighbour: no footprint, plain gather
__global__ __launch_bounds__(256) void k_resample_nearest(const float2* __restrict__ src, int nx,
                                                          int ny, int spitch,
                                                          const double2* __restrict__ lat,
                                                          int lnx, int lny, float fscale,
                                                          float2* __restrict__ dst, int onx,
                                                          int ony) {
    __shared__ tile_hdr hdr;
    const int tid = threadIdx.x;
    const int ox0 = blockIdx.x * TW, oy0 = blockIdx.y * TH;
    if (tid < 64) build_tile_header(lat, lnx, lny, ox0 / LSTEP, oy0 / LSTEP, 0, 0, &hdr);
    __syncthreads();
    const int tx = tid & 63, tyb = tid >> 6;
    const int ox = ox0 + tx;
    for (int q = 0; q < 4; ++q) {
        const int ty = tyb + 4 * q, oy = oy0 + ty;
        if (ox >= onx || oy >= ony) continue;
        float px, py;
        tile_position(&hdr, tx, ty, &px, &py);
        int ix = hdr.bx0 + (int)floorf(px + 0.5f), iy = hdr.by0 + (int)floorf(py + 0.5f);
        float2 res = make_float2(0.f, 0.f);
        if (ix >= 0 && ix < nx && iy >= 0 && iy < ny) {
            float2 s = src[(size_t)iy * spitch + ix];
            if (s.y > 0.f && s.y < ZM_BADVAR_TEST) {
                res.x = s.x * fscale;
                res.y = 1.f / (s.y * fscale * fscale);
            }
        }
        dst[(size_t)oy * onx + ox] = res;
    }
}

template <int KIND>
static int launch_resample_kind(zm_ctx* ctx, dim3 grd, size_t shmem, const float2* src, int nx, int ny,
                                int spitch, const double2* lat, int lnx, int lny, float fscale,
                                float2* dst, int onx, int ony, int lds_elems, const int32_t* mask,
                                int32_t* macc, int mop, int mkind, int mfirst, float* plane_a, float* plane_b) {
    dim3 blk(256, 1, 1);
    const int ntx = grd.x, ntiles = grd.x * grd.y;
    uint16_t* mbox = nullptr;
    if (mop) {
        constexpr int NT = taps_traits<KIND>::N;
        ZM_TRY(ctx->get("mask_box", sizeof(uint16_t) * (size_t)nx * ny, (void**)&mbox));
        if (ctx->box_ready_for != (const void*)mask || ctx->box_ready_nt != NT) {
            zm_scope_timer tb(ctx, "mask_box");
            hipLaunchKernelGGL(k_mask_box<NT>, dim3(zm_div_up(nx, 64), zm_div_up(ny, 16)), blk, 0, ctx->stream,
                               mask, nx, ny, mbox);
        }
        ctx->box_ready_for = nullptr;
    }
    const float* taptab = nullptr;
    if (KIND == ZM_RESAMPLE_LANCZOS3) {
        ZM_TRY(zm_get_lanczos_table(ctx, &taptab));
        shmem += sizeof(float) * LZ_FLOATS;
    }
    zm_scope_timer t(ctx, "resample");
    // persistent grid: a few workgroups per CU, each walking ntiles / G tiles
    dim3 pgrd(std::min(ntiles, 256 * 4), 1, 1);
    if (mop == 0)
        hipLaunchKernelGGL((k_resample<KIND, 0>), pgrd, blk, shmem, ctx->stream, src, nx, ny, spitch, lat,
                           lnx, lny, fscale, dst, onx, ony, lds_elems, mask, mbox, macc, mkind, mfirst, ntx, ntiles, taptab, plane_a, plane_b);
    else if (mop == 1)
        hipLaunchKernelGGL((k_resample<KIND, 1>), pgrd, blk, shmem, ctx->stream, src, nx, ny, spitch, lat,
                           lnx, lny, fscale, dst, onx, ony, lds_elems, mask, mbox, macc, mkind, mfirst, ntx, ntiles, taptab, plane_a, plane_b);
    else
        hipLaunchKernelGGL((k_resample<KIND, 2>), pgrd, blk, shmem, ctx->stream, src, nx, ny, spitch, lat,
                           lnx, lny, fscale, dst, onx, ony, lds_elems, mask, mbox, macc, mkind, mfirst, ntx, ntiles, taptab, plane_a, plane_b);
    ZM_HIP(hipGetLastError());
    return 0;
}

// mop 0: image only; 1: also store the resampled mask into macc; 2: accumulate it
// into macc with mkind (ZM_MASK_AND / ZM_MASK_OR), mfirst = first frame of the stack.
int zm_launch_resample(zm_ctx* ctx, const float2* src, int nx, int ny, int spitch,
                       const double2* lat, int lnx, int lny, int kernel, float fscale,
                       float2* dst, int onx, int ony, int lds_elems, const int32_t* mask,
                       int32_t* macc, int mop, int mkind, int mfirst, float* plane_a, float* plane_b) {
    dim3 blk(256, 1, 1), grd(zm_div_up(onx, TW), zm_div_up(ony, TH), 1);
    dim3 rgrd(zm_div_up(onx, TW), zm_div_up(ony, RTH), 1);     // k_resample: 64 x 32 tiles
    if (!mask || !macc) mop = 0;
    if (lds_elems > RS_PFCAP) lds_elems = RS_PFCAP;      // what the prefetch registers can stage
    size_t shmem = (size_t)HDR_FLOATS * 4 + (size_t)lds_elems * sizeof(float2);
    if (kernel == ZM_RESAMPLE_LANCZOS3)
        return launch_resample_kind<ZM_RESAMPLE_LANCZOS3>(ctx, rgrd, shmem, src, nx, ny, spitch, lat, lnx,
                                                          lny, fscale, dst, onx, ony, lds_elems, mask,
                                                          macc, mop, mkind, mfirst, plane_a, plane_b);
    if (kernel == ZM_RESAMPLE_BILINEAR)
        return launch_resample_kind<ZM_RESAMPLE_BILINEAR>(ctx, rgrd, shmem, src, nx, ny, spitch, lat, lnx,
                                                          lny, fscale, dst, onx, ony, lds_elems, mask,
                                                          macc, mop, mkind, mfirst, plane_a, plane_b);
    if (kernel == ZM_RESAMPLE_NEAREST) {
        zm_scope_timer t(ctx, "resample");
        hipLaunchKernelGGL(k_resample_nearest, grd, blk, 0, ctx->stream, src, nx, ny, spitch, lat, lnx,
                           lny, fscale, dst, onx, ony);
        ZM_HIP(hipGetLastError());
        if (plane_a) ZM_TRY(zm_launch_split_pairs(ctx, dst, (int64_t)onx * ony, plane_a, plane_b));
        if (mop) {
            // nearest neighbour has no footprint: the stand-alone mask kernel + accumulate
            int32_t* tmp = nullptr;
            const int64_t opix = (int64_t)onx * ony;
            if (mop == 1) return zm_launch_resample_mask(ctx, mask, nx, ny, lat, lnx, lny, kernel, macc,
                                                         onx, ony, 0);
            ZM_TRY(ctx->get("mask_tmp", sizeof(int32_t) * opix, (void**)&tmp));
            ZM_TRY(zm_launch_resample_mask(ctx, mask, nx, ny, lat, lnx, lny, kernel, tmp, onx, ony, -1));
            return zm_launch_mask_accum(ctx, macc, tmp, opix, mkind, mfirst);
        }
        return 0;
    }
    zm_set_error("zm_launch_resample: unknown kernel %d", kernel);
    return 2;
}

// ---------------------------------------------------------------------------
// Integer masks: OR of every input pixel under a non-zero tap (oracle/resample.py).
template <int KIND>
__global__ __launch_bounds__(256) void k_resample_mask(const int32_t* __restrict__ mask, int nx,
                                                       int ny, const double2* __restrict__ lat,
                                                       int lnx, int lny,
                                                       int32_t* __restrict__ dst, int onx,
                                                       int ony, int32_t fill) {
    __shared__ tile_hdr hdr;
    constexpr int NT = taps_traits<KIND>::N;
    constexpr int OFF = taps_traits<KIND>::OFF;
    const int tid = threadIdx.x;
    const int ox0 = blockIdx.x * TW, oy0 = blockIdx.y * TH;
    if (tid < 64) build_tile_header(lat, lnx, lny, ox0 / LSTEP, oy0 / LSTEP, OFF, OFF + NT - 1, &hdr);
    __syncthreads();
    const int tx = tid & 63, tyb = tid >> 6;
    const int ox = ox0 + tx;
    for (int q = 0; q < 4; ++q) {
        const int ty = tyb + 4 * q, oy = oy0 + ty;
        if (ox >= onx || oy >= ony) continue;
        float px, py;
        tile_position(&hdr, tx, ty, &px, &py);
        int ixr, iyr;
        float dx, dy;
        bool ddx, ddy;
        split_pos(px, &ixr, &dx, &ddx);
        split_pos(py, &iyr, &dy, &ddy);
        const int ix = hdr.bx0 + ixr + OFF, iy = hdr.by0 + iyr + OFF;
        int32_t m = fill;
        const bool inbx = ddx ? (ix - OFF >= 0 && ix - OFF < nx) : (ix >= 0 && ix + NT <= nx);
        const bool inby = ddy ? (iy - OFF >= 0 && iy - OFF < ny) : (iy >= 0 && iy + NT <= ny);
        if (inbx && inby) {
            m = 0;
            // delta kernels only touch the centre tap (index -OFF)
            int c0 = ddx ? -OFF : 0, c1 = ddx ? -OFF + 1 : NT;
            int r0 = ddy ? -OFF : 0, r1 = ddy ? -OFF + 1 : NT;
            if (KIND == ZM_RESAMPLE_BILINEAR) {   // taps (1-d, d): d == 0 kills the second
                c0 = 0; c1 = ddx ? 1 : 2; r0 = 0; r1 = ddy ? 1 : 2;
            }
            for (int r = r0; r < r1; ++r) {
                const int32_t* p = mask + (size_t)(iy + r) * nx + ix;
                for (int c = c0; c < c1; ++c) m |= p[c];
            }
        }
        dst[(size_t)oy * onx + ox] = m;
    }
}

__global__ __launch_bounds__(256) void k_resample_mask_nearest(const int32_t* __restrict__ mask,
                                                               int nx, int ny,
                                                               const double2* __restrict__ lat,
                                                               int lnx, int lny,
                                                               int32_t* __restrict__ dst, int onx,
                                                               int ony, int32_t fill) {
    __shared__ tile_hdr hdr;
    const int tid = threadIdx.x;
    const int ox0 = blockIdx.x * TW, oy0 = blockIdx.y * TH;
    if (tid < 64) build_tile_header(lat, lnx, lny, ox0 / LSTEP, oy0 / LSTEP, 0, 0, &hdr);
    __syncthreads();
    const int tx = tid & 63, tyb = tid >> 6;
    const int ox = ox0 + tx;
    for (int q = 0; q < 4; ++q) {
        const int ty = tyb + 4 * q, oy = oy0 + ty;
        if (ox >= onx || oy >= ony) continue;
        float px, py;
        tile_position(&hdr, tx, ty, &px, &py);
        int ix = hdr.bx0 + (int)floorf(px + 0.5f), iy = hdr.by0 + (int)floorf(py + 0.5f);
        int32_t m = fill;
        if (ix >= 0 && ix < nx && iy >= 0 && iy < ny) m = mask[(size_t)iy * nx + ix];
        dst[(size_t)oy * onx + ox] = m;
    }
}

int zm_launch_resample_mask(zm_ctx* ctx, const int32_t* mask, int nx, int ny, const double2* lat,
                            int lnx, int lny, int kernel, int32_t* dst, int onx, int ony,
                            int32_t fill) {
    dim3 blk(256, 1, 1), grd(zm_div_up(onx, TW), zm_div_up(ony, TH), 1);
    zm_scope_timer t(ctx, "resample_mask");
    if (kernel == ZM_RESAMPLE_LANCZOS3) {
        hipLaunchKernelGGL(k_resample_mask<ZM_RESAMPLE_LANCZOS3>, grd, blk, 0, ctx->stream, mask,
                           nx, ny, lat, lnx, lny, dst, onx, ony, fill);
    } else if (kernel == ZM_RESAMPLE_BILINEAR) {
        hipLaunchKernelGGL(k_resample_mask<ZM_RESAMPLE_BILINEAR>, grd, blk, 0, ctx->stream, mask,
                           nx, ny, lat, lnx, lny, dst, onx, ony, fill);
    } else if (kernel == ZM_RESAMPLE_NEAREST) {
        hipLaunchKernelGGL(k_resample_mask_nearest, grd, blk, 0, ctx->stream, mask, nx, ny, lat,
                           lnx, lny, dst, onx, ony, fill);
    } else {
        zm_set_error("zm_launch_resample_mask: unknown kernel %d", kernel);
        return 2;
    }
    ZM_HIP(hipGetLastError());
    return 0;
}

// ===========================================================================
// Fused resample -> coadd (+ mask coadd): the frames of a stack are looped INSIDE the output
// tile, and the frames are read RAW - background, variance and the weight threshold are applied
// while a tile is staged, so no prepped plane is ever written (round 3; SURVEY.md section 7
// step 5 / 8(d): "background fused into the resample read, 0 extra").
//
// A workgroup (512 threads, one per CU: 2 waves per SIMD, 256 registers per lane) owns a
// 64 x 64 output tile, walks the N frames of this rank, resamples each one out of LDS exactly as
// k_resample does and keeps the running sums
//   S1 = sum(w v), S0 = sum(w)   (and the AND / OR mask coadd)
// of its 8 pixels per thread in registers; the coadd (or the partial sums of a multi-GPU stack)
// is written once per tile.  This removes what SWarp does through `.resamp.fits` files
// (zuds/coadd.py:126-140) and what the materialised path does through HBM: the prepped plane
// (8 B / px written and read back), the N-deep {value, weight} stack, its re-read by
// k_combine_sum and the per-frame read-modify-write of the mask accumulator.  HBM traffic per
// frame and output pixel: img + wgt (8 B x tile halo 1.35) + the 16-bit box-OR entry.  The sums
// run in frame order with the operations of k_combine_sum (fmaf(w, v, s1); s0 += w), and every
// sample is the one k_resample computes, so the result is bit-identical to the materialised path.
//
// Why bit-identical although the tile is twice k_resample's: a fused tile is two STACKED
// k_resample tiles (64 x 32), each with its own header - lattice nodes relative to its own box
// origin - so a pixel's position is computed with k_resample's very operands; the two boxes are
// staged as one (their union), the sub-box offsets enter as integers.  A wave (64 columns x 8
// rows) lies in one sub-tile and one lattice cell row: sub-tile, node rows and row fractions are
// wave-uniform.
//
// LDS diet (round 2 measured the LDS pipe as the first bound: 36 ds_read_b64 per pixel): a thread
// owns 4 vertically ADJACENT pixels twice.  At near-unit scale their 6 x 6 windows are rows
// iy .. iy + 8 of the same six columns: 9 x 6 reads serve 4 pixels (13.5 per pixel), each row
// read once and used by up to four pixels with their own taps - the arithmetic per pixel is
// unchanged.  A wave whose lanes do not all have that shape (rotations of degrees, scale
// changes, a floor boundary between the rows) takes the generic per-pixel code.
//
// The staging is double-buffered in LDS (one workgroup per CU leaves 160 KB): the raw planes of
// item i + 1 are requested into registers before the pixels of item i are computed, prepped and
// written to the other buffer after them - one barrier per item.
//
// Pointers that arrive through the descriptor array are generic to the compiler: it would
// emit flat_load, which counts on lgkmcnt as well as vmcnt - every LDS wait of the tap rows
// would then also wait for the prefetch of the next tile and for the mask gathers.  Casting
// to the global address space gives global_load (vmcnt only).
#define ZM_GLOBAL __attribute__((address_space(1)))
template <typename T> __device__ inline const T ZM_GLOBAL* zm_gptr(const T* p) { return (const T ZM_GLOBAL*)p; }
// (HIP's float2 / float4 / double2 classes have no constructors from address-space qualified
// references: loads through such pointers use the plain vector types)
typedef float zm_v4f __attribute__((ext_vector_type(4)));
typedef double zm_v2d __attribute__((ext_vector_type(2)));
typedef unsigned zm_v2u __attribute__((ext_vector_type(2)));
__device__ inline float2 zm_gload2(const float2 ZM_GLOBAL* p) {
    const zm_v2f v = *(const zm_v2f ZM_GLOBAL*)p;
    return make_float2(v.x, v.y);
}
__device__ inline float4 zm_gload4f(const float ZM_GLOBAL* p) {
    const zm_v4f v = *(const zm_v4f ZM_GLOBAL*)p;
    return make_float4(v.x, v.y, v.z, v.w);
}

// compile-time loop: the index is a constant in the front end, so register arrays indexed by it are
// scalarised at once (with `#pragma unroll` the staging arrays of k_coadd_fused went to scratch)
template <int I, int N, typename Fn>
__device__ __forceinline__ void zm_static_for(Fn&& fn) {
    if constexpr (I < N) {
        fn(std::integral_constant<int, I>{});
        zm_static_for<I + 1, N>(fn);
    }
}

// Two shapes (FF_TALL): 1 = 64 x 64 tiles, one workgroup of 512 threads per CU (halo 1.35, every wave of a
// CU in the same phase); 0 = 64 x 32 tiles, two workgroups of 256 threads per CU (halo 1.52; while one
// stages and waits at its barrier the other computes).  Two waves per SIMD either way.
#ifndef FF_TALL
#define FF_TALL 0
#endif
#define FF_NSUB (FF_TALL ? 2 : 1)    // k_resample tiles (64 x 32) stacked in a fused tile
#define FT_H (RTH * FF_NSUB)         // output rows of a fused tile
#define FF_THREADS (256 * FF_NSUB)
#define FF_WG_PER_CU (FF_TALL ? 1 : 2)
#define FF_NSLOT 4                   // staging slots per thread; a slot = 4 consecutive pixels of one box row
#define FF_NPX 8                     // output pixels per thread: rows 8 wave .. 8 wave + 7 of column (tid & 63)
#define FF_HDR_WORDS (FF_TALL ? 80 : 48)
#define FF_LDS_HDR 1152              // bytes: 3 headers, tile ring, raw-mask flags
#define FF_LDS_TAB ((LZ_FLOATS * 4 + 127) & ~127)
// staged pixels per buffer: 2 x (8 + 2) B x cap + table + headers < 160 KB / workgroups per CU
#define FF_LDS_CAP (FF_TALL ? 7800 : 3700)
// the DMA-staged kernel (k_coadd_fused_dma below)
#define FD_THREADS 512
#define FD_YROWS 64                  // box rows the y table holds
#define FD_YCOLS 2                   // mesh columns a box may span (BACK_SIZE >= FD_XCOLS)
#define FD_XCOLS 96                  // box columns the x-weight table holds
#define FD_XQ (FD_XCOLS / 4)          // ... as [pixel of the quad][quad column]: conflict-free b128 reads
#define FD_LDS_CAP 3480              // staged pixels (a 64 x 32 tile at unit scale stages at most 80 x 43)

struct ff_hdr {
    tile_hdr3 sub[FF_NSUB];          // the headers k_resample would build for the stacked tiles
    int bx0, by0, bw, bh;            // the staged box: union of the two sub-boxes
    int use_lds, touches, fast;      // (edge item = use_lds && !fast)
    float vscale;                    // the frame's variance scale (a device scalar: fetched here, by the pre-pass)
    int sdx[FF_NSUB], sdy[FF_NSUB];  // sub-box origin minus union origin
    int frame_raw;                   // the frame's box-OR plane has entries that defer to the raw mask (pre-pass flag)
    // background columns under the box (raw-staged frames with a background; round 4): the mesh column of the
    // box's first pixel column and the first pixel column (frame coordinates, a multiple of 4) that lies in
    // the next mesh column - INT_MAX when the box stays inside one.  The staging of k_coadd_fused_dma picks
    // the y-table column of a quad with one comparison instead of evaluating bk_col per quad.
    int ia, xb;
    int pad[FF_HDR_WORDS - 34 * FF_NSUB - 11 - 2 * FF_NSUB];
};
static_assert(sizeof(ff_hdr) == FF_HDR_WORDS * 4, "ff_hdr is not its record");
static_assert(3 * sizeof(ff_hdr) + 4 * 4 + 2 * 8 * 4 <= FF_LDS_HDR, "LDS header area too small");

__device__ inline void ff_build_header(const zm_ff* __restrict__ fr, int f, int lnx, int lny, int t, int ntx,
                                       int onx, int ony, int lds_cap, int dma, ff_hdr* H, int skip_vscale = 0) {
    constexpr int NT = 6, OFF = -2;
    const int tyi = t / ntx, txi = t - tyi * ntx;
    const zm_ff* F = fr + f;
    int bx0 = 0, by0 = 0, bx1 = 0, by1 = 0;
#pragma unroll
    for (int u = 0; u < FF_NSUB; ++u) {
        // (a last tile row whose lower half lies off the grid: the upper header twice)
        const int live = (u == 0 || tyi * FT_H + u * RTH < ony) ? u : 0;
        build_tile_header3(zm_gptr(F->lat), lnx, lny, txi * (TW / LSTEP), tyi * (FT_H / LSTEP) + live * (RTH / LSTEP), OFF,
                           OFF + NT - 1, &H->sub[u]);
    }
    if ((threadIdx.x & 63) == 0) {
        const int nx = F->nx, ny = F->ny;
#pragma unroll
        for (int u = 0; u < FF_NSUB; ++u) {
            const tile_hdr3& a = H->sub[u];
            bx0 = u ? min(bx0, a.bx0) : a.bx0;
            by0 = u ? min(by0, a.by0) : a.by0;
            bx1 = u ? max(bx1, a.bx0 + a.bw) : a.bx0 + a.bw;
            by1 = u ? max(by1, a.by0 + a.bh) : a.by0 + a.bh;
        }
        const int bw = bx1 - bx0, bh = by1 - by0;              // bw: a multiple of 4, like its parts
        const int touches = (bx0 < nx) && (bx1 > 0) && (by0 < ny) && (by1 > 0);
        const long long area = (long long)bw * bh;
        // what the staging registers hold: FF_NSLOT rows per thread of a [512 / (bw / 4)] x [bw / 4] arrangement
        // (the DMA-staged kernel: rows of the y table, columns of the x-weight table)
        // (dma == 2: the owner-staged kernel, whose slots hold a box of at most 80 x 42 pixels at a fixed pitch)
        const int use_lds = touches && area <= (long long)lds_cap && bw >= 8 && bh >= 1 &&
                            (dma == 2 ? (bw <= 80 && bh <= 42)
                             : dma ? (bw <= FD_XCOLS && bh <= FD_YROWS) : (bw <= 256 && bh <= FF_NSLOT * (FF_THREADS / (bw >> 2))));
        const int inside = bx0 >= 0 && by0 >= 0 && bx1 <= nx && by1 <= ny && (txi + 1) * TW <= onx &&
                           (tyi + 1) * FT_H <= ony;
        H->bx0 = bx0; H->by0 = by0; H->bw = bw; H->bh = bh;
        H->touches = touches;
        H->use_lds = use_lds;
        // fast: the box lies on the frame and the tile on the grid - no bounds test anywhere.  edge (staged,
        // not fast): what lies off the frame becomes {0, BIGVAR} at the store.
        H->fast = use_lds && inside;
        H->vscale = (F->vscale && !skip_vscale) ? *F->vscale : 1.f;     // (skip_vscale: k_ff_vscale fills it in later)
        H->frame_raw = F->mboxflag ? *F->mboxflag : 1;
#pragma unroll
        for (int u = 0; u < FF_NSUB; ++u) {
            H->sdx[u] = H->sub[u].bx0 - bx0;
            H->sdy[u] = H->sub[u].by0 - by0;
        }
    }
    {
        // ia / xb: lane l looks at quad column l of the box (boxes staged in LDS are at most FD_XCOLS = 96 wide)
        const int ubx0 = __shfl(bx0, 0), ubw = __shfl(bx1 - bx0, 0);
        int ia = 0, xb = 0x7fffffff;
        if (F->ytab) {
            const int nxm1 = F->nx - 1, l = threadIdx.x & 63;
            ia = bk_col(F->nbx, F->invmesh, min(max(ubx0, 0), nxm1));
            const bool beyond = l < (ubw >> 2) && bk_col(F->nbx, F->invmesh, min(max(ubx0 + 4 * l, 0), nxm1)) > ia;
            const unsigned long long bal = __ballot(beyond);
            if (bal) xb = ubx0 + 4 * (__ffsll((long long)bal) - 1);
        }
        if ((threadIdx.x & 63) == 0) { H->ia = ia; H->xb = xb; }
    }
}

// One prepped pixel straight from the raw planes (frames staged raw keep no prepped plane): the
// global-gather path of a footprint that exceeds the LDS tile - rare, slow, correct.
__device__ inline float2 ff_raw_pixel(const zm_ff* __restrict__ F, int x, int y) {
    const size_t idx = (size_t)y * F->nx + x;
    const float v = zm_gptr(F->img)[idx];
    const float w = F->wgt ? zm_gptr(F->wgt)[idx] : 1.f;
    const float bg = F->bk ? bk_eval(F->bk, F->nbx, F->nby, F->invmesh, x, y) : 0.f;
    const float vs = F->vscale ? *F->vscale : 1.f;
    return prep_pixel(v, w, F->wgt != nullptr, bg, vs, F->wthresh);
}

// one word of a frame's raw mask: an int16 plane (ZM_MASKTYPE_I16) means what its sign extension means
__device__ inline int32_t ff_mask_at(const zm_ff* __restrict__ F, size_t idx) {
    if (F->mask16) return (int32_t)((const int16_t ZM_GLOBAL*)F->mask)[idx];
    return ((const int32_t ZM_GLOBAL*)F->mask)[idx];
}

// result of one generic pixel: {value, weight, mask bits, inb}
struct ff_px {
    float v, w;
    int32_t m;
    int inb;
};

// The general per-pixel code (k_resample's): bounds tests, delta kernels, global gather for
// footprints that do not fit the LDS tile, raw-mask OR where the box-OR plane defers.
// tile: the staged box shifted to the sub-box origin; bx0 / by0: the sub-box origin; bw: the pitch.
template <int MOP>
__device__ inline ff_px ff_generic_pixel(const zm_ff* __restrict__ F, const float2* tile, const float* ltab,
                                         bool use_lds, bool touches, int bx0, int by0, int bw, float px,
                                         float py) {
    constexpr int NT = 6, OFF = -2, CI = 2;
    const int nx = F->nx, ny = F->ny, spitch = F->spitch;
    int ixr, iyr;
    float dx, dy;
    bool ddx, ddy;
    split_pos(px, &ixr, &dx, &ddx);
    split_pos(py, &iyr, &dy, &ddy);
    const int ix = bx0 + ixr + OFF, iy = by0 + iyr + OFF;
    const bool inbx = ddx ? (ix + CI >= 0 && ix + CI < nx) : (ix >= 0 && ix + NT <= nx);
    const bool inby = ddy ? (iy + CI >= 0 && iy + CI < ny) : (iy >= 0 && iy + NT <= ny);
    const bool inb = touches && inbx && inby;
    ff_px r;
    r.v = 0.f; r.w = 0.f; r.m = 0; r.inb = inb;
    if (!inb) return r;
    const bool with_mask = MOP && F->mask != nullptr;
    int32_t mres = 0;
    uint32_t m16 = 0;
    if (with_mask) {
        if (!(ddx || ddy)) {
            m16 = zm_gptr(F->mbox)[(size_t)iy * F->mpitch + ix];
        } else {
            const int c0 = ddx ? CI : 0, c1 = ddx ? CI + 1 : NT;
            const int r0 = ddy ? CI : 0, r1 = ddy ? CI + 1 : NT;
#pragma unroll 1
            for (int rr = r0; rr < r1; ++rr) {
                const size_t mo = (size_t)(iy + rr) * nx + ix;
#pragma unroll 1
                for (int c = c0; c < c1; ++c) mres |= ff_mask_at(F, mo + c);
            }
        }
    }
    zm_v2f txp[3], typ[3];
    zm_lz3_lookup(ltab, ddx ? 0.5f : dx, txp);
    zm_lz3_lookup(ltab, ddy ? 0.5f : dy, typ);
    if (__any(ddx || ddy)) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const zm_v2f dl = (zm_v2f){j == 1 ? 1.f : 0.f, 0.f};
            txp[j] = ddx ? dl : txp[j];
            typ[j] = ddy ? dl : typ[j];
        }
    }
    float tx[NT], ty[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        tx[k] = (k & 1) ? txp[k >> 1].y : txp[k >> 1].x;
        ty[k] = (k & 1) ? typ[k >> 1].y : typ[k >> 1].x;
    }
    float acc = 0.f, vacc = 0.f;
    if (use_lds) {
        const float2* p = tile + (iyr + OFF) * bw + (ixr + OFF);
        zm_v2f av = (zm_v2f){0.f, 0.f};
#pragma unroll
        for (int rr = 0; rr < NT; ++rr) {
            float2 s[NT];
            lds_row<NT>::read(p, s);
            zm_v2f rv2 = (zm_v2f){0.f, 0.f};
#pragma unroll
            for (int c = 0; c < NT; ++c)
                rv2 = __builtin_elementwise_fma((zm_v2f){tx[c], tx[c]}, (zm_v2f){s[c].x, s[c].y}, rv2);
            av = __builtin_elementwise_fma((zm_v2f){ty[rr], ty[rr]}, rv2, av);
            p += bw;
        }
        acc = av.x;
        vacc = av.y;
    } else {
        const float2 ZM_GLOBAL* p = F->src ? zm_gptr(F->src) + (size_t)iy * spitch + ix : nullptr;
#pragma unroll
        for (int rr = 0; rr < NT; ++rr) {
            float ra = 0.f, rv = 0.f;
            if (ty[rr] != 0.f) {                // (zero taps of a delta axis may lie off the frame: not read)
#pragma unroll
                for (int c = 0; c < NT; ++c) {
                    if (tx[c] != 0.f) {
                        const float2 s = p ? zm_gload2(p + c) : ff_raw_pixel(F, ix + c, iy + rr);
                        ra = fmaf(tx[c], s.x, ra);
                        rv = fmaf(tx[c], s.y, rv);
                    }
                }
            }
            acc = fmaf(ty[rr], ra, acc);
            vacc = fmaf(ty[rr], rv, vacc);
            if (p) p += spitch;
        }
    }
    if (vacc > 0.f && vacc < ZM_BADVAR_TEST) {
        r.v = acc * F->fscale;
        r.w = __builtin_amdgcn_rcpf(vacc * F->fscale2);
    }
    if (with_mask && !(ddx || ddy)) {
        if (m16 != ZM_BOX_RAW) {
            mres = (int32_t)m16;
        } else {
#pragma unroll 1
            for (int rr = 0; rr < NT; ++rr) {
                const size_t mo = (size_t)(iy + rr) * nx + ix;
#pragma unroll 1
                for (int c = 0; c < NT; ++c) mres |= ff_mask_at(F, mo + c);
            }
        }
    }
    r.m = mres;
    return r;
}

// Mask coadd of a pixel in registers: one AND per sample for both kinds.  AND: the accumulator
// starts at -1 ("no frame covered the pixel yet", k_mask_accum's marker) and -1 & m == m.
// OR: by De Morgan on the complement - the accumulator holds ~(OR so far) in bits 0 .. 30 and
// "never covered" in bit 31 (masks carry their flags in bits 0 .. 30): it starts at -1, a sample
// ANDs in m ^ 0x7fffffff (bit 31 clear, the other bits complemented).  (k_mask_accum's literal
// `a == -1 ? m : a | m` in the unrolled pixel loop made the compiler spill 250 registers.)
template <int MOP>
__device__ inline int32_t ff_mask_term(int32_t m) { return MOP == 1 ? m : (m ^ 0x7fffffff); }
template <int MOP>
__device__ inline int32_t ff_mask_fold(int32_t a, int32_t m) { return a & ff_mask_term<MOP>(m); }
template <int MOP>
__device__ inline int32_t ff_mask_result(int32_t a) {       // k_mask_accum's convention: -1 = never covered
    if (MOP == 1) return a;
    return a < 0 ? -1 : (a ^ 0x7fffffff);
}

// ---- item headers, precomputed ------------------------------------------------------------
// An item = (output tile, frame).  Its header (the two sub-tile headers: box of the input
// footprint, 15 lattice nodes relative to the box origin; the union box; the path flags) needs
// fp64 loads and wave reductions: a pre-pass builds all of them, one wave per item; the
// persistent kernel fetches a header two items ahead with one 4-byte load per lane.
__global__ __launch_bounds__(256) void k_ff_headers(const zm_ff* __restrict__ fr, int nfr, int lnx, int lny,
                                                    int onx, int ony, int lds_cap, int dma, int ntx, int ntiles,
                                                    int* __restrict__ out, int* __restrict__ tilectr, int ctr0,
                                                    int skip_vscale) {
    __shared__ ff_hdr H[4];
    if (blockIdx.x == 0 && threadIdx.x == 0) *tilectr = ctr0;     // k_coadd_fused's tile queue starts behind its first wave of tiles
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long item = (long long)blockIdx.x * 4 + w;
    const bool live = item < (long long)ntiles * nfr;
    if (live) {
        const int t = (int)(item / nfr), f = (int)(item - (long long)t * nfr);
        ff_build_header(fr, f, lnx, lny, t, ntx, onx, ony, lds_cap, dma, &H[w], skip_vscale);
    }
    __syncthreads();
    if (live) {
        if (lane < FF_HDR_WORDS) out[item * FF_HDR_WORDS + lane] = ((const int*)&H[w])[lane];
        if (lane + 64 < FF_HDR_WORDS) out[item * FF_HDR_WORDS + 64 + lane] = ((const int*)&H[w])[64 + lane];
    }
}

// The variance scale of a frame comes out of the background chain (k_var_scale_batch), the rest of a header
// does not: round 4 builds the headers on the second stream BESIDE the mesh statistics (k_ff_headers with
// skip_vscale) and this pass drops the one word into every header once the scales exist.
__global__ __launch_bounds__(256) void k_ff_vscale(const zm_ff* __restrict__ fr, int nfr, long long items,
                                                   int* __restrict__ out) {
    const long long item = (long long)blockIdx.x * 256 + threadIdx.x;
    if (item >= items) return;
    const int f = (int)(item % nfr);
    const float* vs = fr[f].vscale;
    reinterpret_cast<float*>(out)[item * FF_HDR_WORDS + offsetof(ff_hdr, vscale) / 4] = vs ? *vs : 1.f;
}

// ---- the y part of the background spline, once per frame row and mesh column ---------------
// T[y][i0] = {r0, r1, e0, e1} (bk_ypart): what the staging of k_coadd_fused combines with a
// pixel's four x weights.  1.1 MB per 3072^2 frame against the 75 MB of a prepped plane.
__global__ __launch_bounds__(256) void k_bk_rows(const zm_bkrows* __restrict__ jobs) {
    const zm_bkrows J = jobs[blockIdx.y];
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= J.ny * J.ytp) return;
    const int y = e / J.ytp, i0 = e - y * J.ytp;
    J.out[e] = bk_ypart(J.bk, J.nbx, J.nby, J.invmesh, y, i0);
}

// ... and the x part: the four x weights {dx1, dx, cdx1, cdx} of every pixel column of a frame (a function of
// the column alone).  The staging of k_coadd_fused_dma fetches the columns of a box with the LDS-DMA engine
// instead of computing them per item (round 4).  Layout: [weight k][quad column] float4 = weight k of the four
// pixels of a quad (nx a multiple of 4: only frames that are staged raw get a table) - the prep pass then
// evaluates the background of two pixels per packed FMA without moving registers around.
__global__ __launch_bounds__(256) void k_bk_cols(const zm_bkrows* __restrict__ jobs) {
    const zm_bkrows J = jobs[blockIdx.y];
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= J.nx || !J.xout) return;
    const float4 w = bk_xweights(bk_dx(J.nbx, J.invmesh, x, bk_col(J.nbx, J.invmesh, x)));
    float* xo = reinterpret_cast<float*>(J.xout);
    const size_t nq4 = (size_t)(J.nx >> 2), e = (size_t)(x >> 2) * 4 + (x & 3);
    xo[e] = w.x;
    xo[nq4 * 4 + e] = w.y;
    xo[nq4 * 8 + e] = w.z;
    xo[nq4 * 12 + e] = w.w;
}

// box-OR planes of all masks of a stack in one launch (k_mask_box per frame: 32 launches)
template <int NT>
__global__ __launch_bounds__(256) void k_mask_box_batch(const zm_boxjob* __restrict__ jobs) {
    constexpr int TWB = 64, THB = 16, IW = TWB + NT - 1, IH = THB + NT - 1, IP = IW + 1;
    __shared__ int32_t t0[IH * IP];
    __shared__ int32_t h[IH * TWB];
    const zm_boxjob J = jobs[blockIdx.z];
    const int nx = J.nx, ny = J.ny;
    const int x0 = blockIdx.x * TWB, y0 = blockIdx.y * THB, tid = threadIdx.x;
    if (x0 >= nx || y0 >= ny) return;                        // (the grid covers the largest frame)
    const int32_t* __restrict__ m = static_cast<const int32_t*>(J.m);
    constexpr int NLD = (IH * IW + 255) / 256;          // loads first, LDS stores after: one latency
    int32_t mm[NLD];
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
        const int e = tid + 256 * q;
        const int r = e / IW, c = e - r * IW;
        const int x = x0 + c, y = y0 + r;
        mm[q] = (e < IH * IW && x < nx && y < ny) ? m[(size_t)y * nx + x] : 0;
    }
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
        const int e = tid + 256 * q;
        if (e < IH * IW) t0[(e / IW) * IP + (e % IW)] = mm[q];
    }
    __syncthreads();
    for (int e = tid; e < IH * TWB; e += 256) {
        const int r = e / TWB, c = e - r * TWB;
        int32_t o = 0;
#pragma unroll
        for (int k = 0; k < NT; ++k) o |= t0[r * IP + c + k];
        h[e] = o;
    }
    __syncthreads();
    for (int e = tid; e < THB * TWB; e += 256) {
        const int r = e / TWB, c = e - r * TWB;
        const int x = x0 + c, y = y0 + r;
        if (x + NT <= nx && y + NT <= ny) {
            int32_t o = 0;
#pragma unroll
            for (int k = 0; k < NT; ++k) o |= h[(r + k) * TWB + c];
            const uint16_t en = box_entry(o);
            J.B[(size_t)y * J.pitch + x] = en;
            if (en == ZM_BOX_RAW && J.rawflag) atomicOr(J.rawflag, 1);
        }
    }
}

// The same planes from a streaming kernel (round 4): the tiled kernel above reads a 69 x 21 halo box per
// 64 x 16 tile (1.41 x the mask) through LDS; here a WAVE owns a strip of columns and walks down a band of
// rows, every lane holding CPL consecutive columns: the horizontal OR of a row comes from the next lane(s)
// (cross-lane moves, no LDS tile, no barrier), the vertical OR from a ring of the last NT row results in
// registers - the mask is read once (+ NT - 1 rows per band, + one or two lanes per strip: 1.06 x) with
// 16-byte loads, NT rows in flight.  T = int16_t: a ZTF mask as it lies on disk (ZM_MASKTYPE_I16), half the
// bytes; a negative word stands for its sign extension, i.e. bits above 15: ZM_BOX_RAW.
#define MB_ROWS 121                   // output rows per band: 121 + NT - 1 = 126 = 21 x NT input rows
// CPL columns per lane, one 16-byte load per lane and row: 8 for int16, 4 for int32.  The horizontal OR
// reaches NT - 1 columns ahead: into the next lane (CPL = 8), into the next two (CPL = 4); the last one /
// two lanes of a wave only supply that halo, the next strip owns their columns.
template <typename T> struct mb_row;
template <> struct mb_row<int32_t> {
    enum { CPL = 4, HALO_LANES = 2 };
    static __device__ inline void load(const int32_t* p, bool vec, int nvalid, int32_t v[4]) {
        if (vec && nvalid == 4) {
            const int4 q = *reinterpret_cast<const int4*>(p);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = k < nvalid ? p[k] : 0;
        }
    }
};
template <> struct mb_row<int16_t> {
    enum { CPL = 8, HALO_LANES = 1 };
    static __device__ inline void load(const int16_t* p, bool vec, int nvalid, int32_t v[8]) {
        if (vec && nvalid == 8) {
            const int4 q = *reinterpret_cast<const int4*>(p);          // eight words; sign extension below
            const int w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[2 * k] = (int32_t)(int16_t)(w[k] & 0xffff);
                v[2 * k + 1] = w[k] >> 16;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = k < nvalid ? (int32_t)p[k] : 0;
        }
    }
};
template <typename T, int NT>
__global__ __launch_bounds__(256) void k_mask_box_rows(const zm_boxjob* __restrict__ jobs) {
    constexpr int CPL = mb_row<T>::CPL, HL = mb_row<T>::HALO_LANES, OWN = 64 - HL;
    static_assert(NT >= 2 && NT - 1 <= CPL * HL, "the horizontal OR reaches into HALO_LANES lanes");
    const zm_boxjob J = jobs[blockIdx.z];
    const int nx = J.nx, ny = J.ny;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = blockIdx.x * (CPL * OWN) + CPL * lane;             // this lane's columns
    const int y0 = (blockIdx.y * 4 + wv) * MB_ROWS;                  // first output row of this wave's band
    if (blockIdx.x * (CPL * OWN) >= nx || y0 >= ny) return;          // (wave-uniform: the grid covers the largest frame)
    const T* __restrict__ m = reinterpret_cast<const T*>(J.m);
    const bool vec = (nx % CPL) == 0 && (reinterpret_cast<uintptr_t>(m) & 15) == 0;
    const int nvalid = min(max(nx - x, 0), CPL);
    const bool owner = lane < OWN;
    int32_t ring[NT][CPL];
#pragma unroll
    for (int k = 0; k < NT; ++k)
#pragma unroll
        for (int e = 0; e < CPL; ++e) ring[k][e] = 0;
    bool raw_seen = false;
#pragma unroll 1
    for (int r0 = 0; r0 < MB_ROWS + NT - 1; r0 += NT) {
        if (y0 + r0 >= ny) break;                                    // nothing below the frame contributes
        int32_t a[NT][CPL];
#pragma unroll
        for (int k = 0; k < NT; ++k) {                               // NT rows requested together
            const int y = y0 + r0 + k;
            if (y < ny && nvalid > 0) {
                mb_row<T>::load(m + (size_t)y * nx + x, vec, nvalid, a[k]);
            } else {
#pragma unroll
                for (int e = 0; e < CPL; ++e) a[k][e] = 0;
            }
        }
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const int y = y0 + r0 + k;                               // input row; it completes output row y - NT + 1
            int32_t win[CPL + NT - 1];
#pragma unroll
            for (int e = 0; e < CPL; ++e) win[e] = a[k][e];
#pragma unroll
            for (int e = 0; e < NT - 1; ++e)                         // columns x + CPL + e: lane + 1 (+ 2 beyond its CPL)
                win[CPL + e] = __shfl_down(a[k][e % CPL], 1 + e / CPL);
            // h[e] = OR of columns x + e .. x + e + NT - 1 of this row
#pragma unroll
            for (int e = 0; e < CPL; ++e) {
                int32_t o = 0;
#pragma unroll
                for (int t = 0; t < NT; ++t) o |= win[e + t];
                ring[k][e] = o;
            }
            const int yo = y - (NT - 1);
            if (yo >= y0 && yo < y0 + MB_ROWS && y < ny && owner) {
                uint16_t en[CPL];
                bool all_in = true;
#pragma unroll
                for (int e = 0; e < CPL; ++e) {
                    int32_t o = 0;
#pragma unroll
                    for (int t = 0; t < NT; ++t) o |= ring[t][e];
                    en[e] = box_entry(o);
                    const bool in = x + e + NT <= nx;
                    all_in = all_in && in;
                    raw_seen = raw_seen || (in && en[e] == ZM_BOX_RAW);
                }
                uint16_t* dst = J.B + (size_t)yo * J.pitch + x;
                if (all_in) {
                    unsigned pk[CPL / 2];
#pragma unroll
                    for (int e = 0; e < CPL / 2; ++e) pk[e] = (unsigned)en[2 * e] | ((unsigned)en[2 * e + 1] << 16);
                    if constexpr (CPL == 8) *reinterpret_cast<uint4*>(dst) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                    else *reinterpret_cast<uint2*>(dst) = make_uint2(pk[0], pk[1]);
                } else {
#pragma unroll
                    for (int e = 0; e < CPL; ++e)
                        if (x + e + NT <= nx) dst[e] = en[e];
                }
            }
        }
    }
    if (J.rawflag && __any(raw_seen) && lane == 0) atomicOr(J.rawflag, 1);
}

// int16 masks whose rows are whole 16-byte pieces (nx a multiple of 8, aligned planes - every ZTF mask): the same
// walk on PACKED words.  A lane keeps its eight columns as the four dwords it loaded (two mask words each), a
// strip is 64 lanes x 8 columns = 512 columns = whole 128-byte lines (the kernel above gives a lane to the halo:
// 504-column strips, a seventh strip of 48 columns at 3072, every row load straddling two lines).  The columns
// right of the lane come from the next lane by a DPP wave shift (lane 63: from the first piece of the next strip,
// loaded once per NT rows by NT lanes and handed over by v_readlane as the shift's fill value).  A negative
// int16 word is the sign extension box_entry() turns into ZM_BOX_RAW, i.e. bit 15 of the 16-bit OR: the sliding
// OR works on halves of dwords (f = lo | hi of a pair; even columns: OR of whole pairs; odd columns: hi of the
// first, whole pairs, lo of the last - v_or3_b32 / v_and_or_b32 / v_lshl_or_b32), the vertical OR on the packed
// results, and the entry is o | 0xffff per half whose bit 15 is set.  ~75 vector instructions per row of eight
// columns instead of ~150, no LDS cross-lane traffic.  Same plane, bit for bit (tests/test_mask_i16_gpu.py).
__device__ __forceinline__ uint32_t mb_shl1(uint32_t v, uint32_t fill) {
    // lane i <- lane i + 1 (DPP wave_shl:1); lane 63 keeps `fill`
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x130, 0xf, 0xf, false);
}
template <int NT>
__global__ __launch_bounds__(256) void k_mask_box_rows16(const zm_boxjob* __restrict__ jobs) {
    static_assert(NT % 2 == 0 && NT >= 2 && NT <= 6, "pairs of columns; the halo is at most three dwords");
    constexpr int HP = NT / 2;                                        // whole pairs in a window
    const zm_boxjob J = jobs[blockIdx.z];
    const int nx = J.nx, ny = J.ny;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int xs = blockIdx.x * 512, x = xs + 8 * lane;              // this lane's columns
    const int y0 = (blockIdx.y * 4 + wv) * MB_ROWS;                  // first output row of this wave's band
    if (xs >= nx || y0 >= ny) return;                                // (wave-uniform: the grid covers the largest frame)
    const int16_t* __restrict__ m = reinterpret_cast<const int16_t*>(J.m);
    const bool mine = x < nx;                                        // (nx % 8 == 0: a lane's piece is whole or absent)
    const int xh = xs + 512;                                         // the piece right of the strip
    const bool halo = xh < nx;
    const bool all_in = x + 7 + NT <= nx;                            // every window of this lane lies on the frame
    uint32_t ring[NT][4];
#pragma unroll
    for (int k = 0; k < NT; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) ring[k][i] = 0u;
    uint32_t rawm = 0u;
#pragma unroll 1
    for (int r0 = 0; r0 < MB_ROWS + NT - 1; r0 += NT) {
        if (y0 + r0 >= ny) break;                                    // nothing below the frame contributes
        uint4 a[NT];
#pragma unroll
        for (int k = 0; k < NT; ++k) {                               // NT rows requested together
            const int y = y0 + r0 + k;
            a[k] = make_uint4(0u, 0u, 0u, 0u);
            if (y < ny && mine) a[k] = *reinterpret_cast<const uint4*>(m + (size_t)y * nx + x);
        }
        uint4 hp = make_uint4(0u, 0u, 0u, 0u);                       // lane k: the halo piece of row r0 + k
        if (lane < NT && y0 + r0 + lane < ny && halo)
            hp = *reinterpret_cast<const uint4*>(m + (size_t)(y0 + r0 + lane) * nx + xh);
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const int y = y0 + r0 + k;                               // input row; it completes output row y - NT + 1
            uint32_t P[7] = {a[k].x, a[k].y, a[k].z, a[k].w, 0u, 0u, 0u};
            const uint32_t hs[3] = {(uint32_t)__builtin_amdgcn_readlane((int)hp.x, k),
                                    (uint32_t)__builtin_amdgcn_readlane((int)hp.y, k),
                                    (uint32_t)__builtin_amdgcn_readlane((int)hp.z, k)};
#pragma unroll
            for (int i = 0; i < HP; ++i) P[4 + i] = mb_shl1(P[i], hs[i]);
            uint32_t hi[7], f[7];
#pragma unroll
            for (int i = 0; i < 4 + HP; ++i) {
                hi[i] = P[i] >> 16;
                f[i] = (P[i] & 0xffffu) | hi[i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t he = f[i], ho = hi[i];
#pragma unroll
                for (int t = 1; t < HP; ++t) { he |= f[i + t]; ho |= f[i + t]; }
                ho |= P[i + HP] & 0xffffu;
                ring[k][i] = he | (ho << 16);
            }
            const int yo = y - (NT - 1);
            if (yo >= y0 && yo < y0 + MB_ROWS && y < ny && mine) {
                uint32_t en[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    uint32_t o = ring[0][i];
#pragma unroll
                    for (int t = 1; t < NT; ++t) o |= ring[t][i];
                    const uint32_t neg = (o >> 15) & 0x00010001u;    // halves with bit 15: a negative word in the window
                    en[i] = o | (neg * 0xffffu);
                    if (all_in) rawm |= neg;
                }
                uint16_t* dst = J.B + (size_t)yo * J.pitch + x;
                if (all_in) {
                    *reinterpret_cast<uint4*>(dst) = make_uint4(en[0], en[1], en[2], en[3]);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (x + e + NT <= nx) {
                            const uint16_t v = (uint16_t)(en[e >> 1] >> (16 * (e & 1)));
                            dst[e] = v;
                            rawm |= v == ZM_BOX_RAW ? 1u : 0u;
                        }
                }
            }
        }
    }
    if (J.rawflag && __any(rawm != 0u) && lane == 0) atomicOr(J.rawflag, 1);
}

// jobs: host arrays (staged through pinned memory behind an event, like the frame descriptors)
int zm_launch_fused_prepass(zm_ctx* ctx, const zm_bkrows* rows, int nrows) {
    if (nrows == 0) return 0;
    hipEvent_t* ev = nullptr;
    ZM_TRY(zm_get_sync_events(ctx, 9, &ev));
    ZM_HIP(hipEventSynchronize(ev[6]));
    const size_t rb = sizeof(zm_bkrows) * (size_t)nrows;
    char *pin = nullptr, *dev = nullptr;
    ZM_TRY(ctx->get_pinned("ff_pre_h", rb, (void**)&pin));
    ZM_TRY(ctx->get("ff_pre", rb, (void**)&dev));
    memcpy(pin, rows, rb);
    ZM_HIP(hipMemcpyAsync(dev, pin, rb, hipMemcpyHostToDevice, ctx->stream));
    ZM_HIP(hipEventRecord(ev[6], ctx->stream));
    int most = 1, mostx = 1;
    for (int i = 0; i < nrows; ++i) {
        most = std::max(most, rows[i].ny * rows[i].ytp);
        mostx = std::max(mostx, rows[i].nx);
    }
    zm_scope_timer t(ctx, "bk_rows");
    hipLaunchKernelGGL(k_bk_rows, dim3(zm_div_up(most, 256), nrows), dim3(256), 0, ctx->stream, (const zm_bkrows*)dev);
    hipLaunchKernelGGL(k_bk_cols, dim3(zm_div_up(mostx, 256), nrows), dim3(256), 0, ctx->stream, (const zm_bkrows*)dev);
    ZM_HIP(hipGetLastError());
    return 0;
}

// 2: int16, rows and planes in whole 16-byte pieces (k_mask_box_rows16; ZM_MASK_BOX=lanes keeps such planes on
// the unpacked kernel: developer A / B); 1: other int16 planes; 0: int32
static int box_kind(const zm_boxjob& b) {
    static const bool unpacked = getenv("ZM_MASK_BOX") && !strcmp(getenv("ZM_MASK_BOX"), "lanes");
    if (!b.is16) return 0;
    const bool pieces = b.nx % 8 == 0 && b.pitch % 8 == 0 && (reinterpret_cast<uintptr_t>(b.m) & 15) == 0 &&
                        (reinterpret_cast<uintptr_t>(b.B) & 15) == 0;
    return pieces && !unpacked ? 2 : 1;
}

// The box-OR planes depend on the masks only.  after != NULL: the launch goes to the second stream, ordered
// after `after` (an event recorded on the main stream before the caller enqueues the mesh statistics: nothing
// older may still read the planes), and runs BESIDE those statistics - they are bound by their own moment /
// histogram work at 2.5 TB/s, this kernel streams.  *joined receives the event the main stream has to wait
// for before the planes are read (NULL: same stream, nothing to wait for).
int zm_launch_mask_boxes(zm_ctx* ctx, const zm_boxjob* boxes, int nboxes, hipEvent_t after, hipEvent_t* joined) {
    if (joined) *joined = nullptr;
    if (nboxes == 0) return 0;
    hipEvent_t* ev = nullptr;
    ZM_TRY(zm_get_sync_events(ctx, 9, &ev));
    ZM_HIP(hipEventSynchronize(ev[7]));
    const size_t bb = sizeof(zm_boxjob) * (size_t)nboxes;
    char *pin = nullptr, *dev = nullptr;
    ZM_TRY(ctx->get_pinned("ff_box_h", bb, (void**)&pin));
    ZM_TRY(ctx->get("ff_box", bb, (void**)&dev));
    {
        // int16 jobs of whole 16-byte pieces first (the packed kernel), then the other int16 ones, then int32
        zm_boxjob* pj = reinterpret_cast<zm_boxjob*>(pin);
        int k = 0;
        for (int pass = 2; pass >= 0; --pass)
            for (int i = 0; i < nboxes; ++i)
                if (box_kind(boxes[i]) == pass) pj[k++] = boxes[i];
    }
    // (the scope timers record on the main stream: when this scope is being timed the kernel stays there)
    const bool timed = ctx->timing && (ctx->timing_only.empty() || ctx->timing_only == "mask_box");
    static const bool fork_off = getenv("ZM_FF_FORK") && getenv("ZM_FF_FORK")[0] == '0';
    const bool side = after != nullptr && joined != nullptr && ctx->aux != nullptr && !timed && !fork_off;
    hipStream_t s = side ? ctx->aux : ctx->stream;
    if (side) ZM_HIP(hipStreamWaitEvent(s, after, 0));
    ZM_HIP(hipMemcpyAsync(dev, pin, bb, hipMemcpyHostToDevice, s));
    ZM_HIP(hipEventRecord(ev[7], s));
    int mx = 1, my = 1;
    bool any16 = false, any32 = false;
    for (int i = 0; i < nboxes; ++i) {
        mx = std::max(mx, boxes[i].nx);
        my = std::max(my, boxes[i].ny);
        (boxes[i].is16 ? any16 : any32) = true;
    }
    {
        // ZM_MASK_BOX=tile: the LDS-tiled kernel of round 3 (developer A / B; int32 planes only)
        static const bool tiled = getenv("ZM_MASK_BOX") && !strcmp(getenv("ZM_MASK_BOX"), "tile");
        zm_scope_timer t(ctx, "mask_box");
        if (tiled && !any16) {
            hipLaunchKernelGGL(k_mask_box_batch<6>, dim3(zm_div_up(mx, 64), zm_div_up(my, 16), nboxes), dim3(256), 0, s,
                               (const zm_boxjob*)dev);
        } else {
            // one launch per mask type over the jobs of that type (a stack normally has one): the jobs are
            // sorted by type in the staging copy, a launch covers a contiguous range of them
            const unsigned gy = zm_div_up(zm_div_up(my, MB_ROWS), 4);
            int n16 = 0, npk = 0;
            for (int i = 0; i < nboxes; ++i) {
                n16 += boxes[i].is16 ? 1 : 0;
                npk += box_kind(boxes[i]) == 2 ? 1 : 0;
            }
            const zm_boxjob* d = (const zm_boxjob*)dev;
            if (npk)
                hipLaunchKernelGGL((k_mask_box_rows16<6>), dim3(zm_div_up(mx, 512), gy, npk), dim3(256), 0, s, d);
            if (n16 - npk)
                hipLaunchKernelGGL((k_mask_box_rows<int16_t, 6>), dim3(zm_div_up(mx, 8 * 63), gy, n16 - npk), dim3(256), 0, s,
                                   d + npk);
            if (nboxes - n16)
                hipLaunchKernelGGL((k_mask_box_rows<int32_t, 6>), dim3(zm_div_up(mx, 4 * 62), gy, nboxes - n16), dim3(256), 0,
                                   s, d + n16);
        }
    }
    ZM_HIP(hipGetLastError());
    if (side) {
        ZM_HIP(hipEventRecord(ev[8], s));
        *joined = ev[8];
    }
    return 0;
}

// LDS row reads, software-pipelined by hand: the six ds_read_b64 of tap row r + 1 are issued
// before the packed FMAs of row r; `lds_wait` then waits until at most N reads are outstanding
// (LDS returns in order) and carries the registers, so no consumer can be scheduled above it.
struct lds_row6 {
    unsigned long long r0, r1, r2, r3, r4, r5;
};
__device__ inline void lds_issue6(unsigned a, lds_row6& o);
__device__ inline void lds_issue6(const float2* p, lds_row6& o) { lds_issue6((unsigned)(size_t)p, o); }
// (a: the 32-bit LDS address - row arithmetic on a generic 64-bit pointer costs 64-bit multiply-adds)
__device__ inline void lds_issue6(unsigned a, lds_row6& o) {
    asm volatile("ds_read_b64 %0, %6\n\t"
                 "ds_read_b64 %1, %6 offset:8\n\t"
                 "ds_read_b64 %2, %6 offset:16\n\t"
                 "ds_read_b64 %3, %6 offset:24\n\t"
                 "ds_read_b64 %4, %6 offset:32\n\t"
                 "ds_read_b64 %5, %6 offset:40"
                 : "=&v"(o.r0), "=&v"(o.r1), "=&v"(o.r2), "=&v"(o.r3), "=&v"(o.r4), "=&v"(o.r5)
                 : "v"(a)
                 : "memory");
}
__device__ inline zm_v2f lds_pair(unsigned long long r) {
    return (zm_v2f){__uint_as_float((unsigned)r), __uint_as_float((unsigned)(r >> 32))};
}
template <int N>
__device__ inline void lds_wait_n(lds_row6& o) {
    static_assert(N >= 0 && N <= 15, "lgkmcnt has four bits");
    asm volatile("s_waitcnt lgkmcnt(%6)"
                 : "+v"(o.r0), "+v"(o.r1), "+v"(o.r2), "+v"(o.r3), "+v"(o.r4), "+v"(o.r5)
                 : "n"(N)
                 : "memory");
}

// one node of the tap table (zm_lz3_lookup's five reads), issued without waiting
struct lz3_node {
    zm_v4f a, b, c, g;
    zm_v2f h;
};
__device__ inline void lz3_issue(const float* tab, float d, lz3_node& n, float& dl) {
    // fi = rint(d LZ_N), the node, as zm_lz3_lookup computes it - here through the magic-number addition:
    // d LZ_N is exact (a power of two), 1.5 x 2^23 + it rounds to the nearest integer, ties to even like
    // rint (the magic number is even), and the integer sits in the low mantissa bits: the node address is one
    // 24-bit multiply-add on the bit pattern (the 24-bit operand ignores the exponent bits above), no
    // float -> int conversion.  Four instructions instead of five, the same node and the same dl.
    const float magic = 12582912.0f;                                  // 0x4B400000: mantissa field 0x400000 + fi
    const float tm = __builtin_fmaf(d, (float)LZ_N, magic);
    const float fi = tm - magic;
    dl = __builtin_fmaf(fi, -1.0f / LZ_N, d);
    const unsigned a = __umul24(__float_as_uint(tm), LZ_ENTRY * 4) + ((unsigned)(size_t)tab - 0x400000u * (LZ_ENTRY * 4));
    asm volatile("ds_read_b128 %0, %5\n\t"
                 "ds_read_b128 %1, %5 offset:16\n\t"
                 "ds_read_b128 %2, %5 offset:32\n\t"
                 "ds_read_b128 %3, %5 offset:48\n\t"
                 "ds_read_b64 %4, %5 offset:64"
                 : "=&v"(n.a), "=&v"(n.b), "=&v"(n.c), "=&v"(n.g), "=&v"(n.h)
                 : "v"(a)
                 : "memory");
}
template <int N>
__device__ inline void lz3_wait(lz3_node& n) {
    asm volatile("s_waitcnt lgkmcnt(%5)"
                 : "+v"(n.a), "+v"(n.b), "+v"(n.c), "+v"(n.g), "+v"(n.h)
                 : "n"(N)
                 : "memory");
}
// the arithmetic of zm_lz3_lookup on a node that has arrived
__device__ inline void lz3_eval(const lz3_node& n, float dl, zm_v2f t[3]) {
    const zm_v2f dd = (zm_v2f){dl, dl};
    t[0] = __builtin_elementwise_fma(dd, __builtin_elementwise_fma(dd, (zm_v2f){n.b.x, n.b.y}, (zm_v2f){n.a.z, n.a.w}),
                                     (zm_v2f){n.a.x, n.a.y});
    t[1] = __builtin_elementwise_fma(dd, __builtin_elementwise_fma(dd, (zm_v2f){n.c.z, n.c.w}, (zm_v2f){n.c.x, n.c.y}),
                                     (zm_v2f){n.b.z, n.b.w});
    t[2] = __builtin_elementwise_fma(dd, __builtin_elementwise_fma(dd, (zm_v2f){n.h.x, n.h.y}, (zm_v2f){n.g.z, n.g.w}),
                                     (zm_v2f){n.g.x, n.g.y});
}
__device__ inline void lds_issue_u16(const uint16_t* p, uint32_t& o) {
    asm volatile("ds_read_u16 %0, %1" : "=v"(o) : "v"((unsigned)(size_t)p) : "memory");
}

// LDS: [3 headers, tile ring, raw flags][tap table][pixel tile x 2][mask tile x 2].
// Everything a pixel of a staged item touches is in LDS - also its box-OR mask entry: a per-pixel
// global gather would be waited for with vmcnt(0), and vmcnt retires in order, so it would
// drain the register prefetch of the next item at the first pixel.  In the fast path the only
// vector-memory instructions between two barriers are that prefetch, the 4-byte header fetch and
// the tile-queue atomic issued before it, and the stores of a finished tile behind it.
// STACK: the same machinery as a resampler - nothing is summed, every item's samples {value, weight}
// go to its frame's plane of a resident stack (the CLIPPED / MEDIAN path), the mask coadd still
// accumulates in registers.  An item's samples wait in the sum registers and are stored when the
// next item starts, ahead of its prefetch: stores issued behind the prefetch would sit in front of
// it in the (in-order) vmcnt queue of the wait that ends the item.
template <int MOP, bool AVG, bool STACK>
__global__ __launch_bounds__(FF_THREADS, FF_WG_PER_CU) void k_coadd_fused(
    const zm_ff* __restrict__ fr, int nfr, int onx, int ony, int lds_cap, int ntx, int ntiles,
    const int* __restrict__ ghdr, float* __restrict__ out_img, float* __restrict__ out_wgt,
    int32_t* __restrict__ out_mask, float* __restrict__ out_cov, int partial,
    const float* __restrict__ taptab, int* __restrict__ tilectr, float2* __restrict__ stack, long long fstride,
    int dbg, long long* __restrict__ prof) {
    extern __shared__ float4 smem4[];
    char* smem = reinterpret_cast<char*>(smem4);
    ff_hdr* HR = reinterpret_cast<ff_hdr*>(smem);                  // ring of 3 headers
    int* tring = reinterpret_cast<int*>(smem + 3 * sizeof(ff_hdr));   // tiles held, by ordinal & 3
    int* rawflag = tring + 4;                                       // [buffer][wave]
    const float* ltab = reinterpret_cast<const float*>(smem + FF_LDS_HDR);
    float2* tile0 = reinterpret_cast<float2*>(smem + FF_LDS_HDR + FF_LDS_TAB);
    uint16_t* mtile0 = reinterpret_cast<uint16_t*>(tile0 + 2 * (size_t)lds_cap);   // lds_cap is a multiple of 8
    constexpr int NT = 6, OFF = -2;
    const int tid = threadIdx.x;

    // ---- staging: thread (r0, c) of a [RP rows][bw / 4 quads] arrangement holds the quad c of rows
    // r0, r0 + RP, ... - one column position per thread, a uniform row stride per slot.
    float4 pi[FF_NSLOT], pw[FF_NSLOT], py4[FF_NSLOT];     // raw image / weight quads (or two prepped pairs), y part
    uint2 pm[FF_NSLOT];                                   // box-OR entries
    // One straight-line sequence of loads for every kind of item and frame, from addresses clamped
    // onto the frame; what a quad is worth is decided at the store.  (Branches here - fast / edge,
    // with / without weights - end in a join where the compiler copies the loaded registers: a use,
    // i.e. a vmcnt(0) wait that drains the prefetch right after it was issued.  Measured: the
    // staging loads and the pixel work did not overlap at all.)  Frames staged raw have 16-byte rows
    // (vec_ok: the host preps the others into a plane); the prepped plane has an even pitch; the
    // box-OR plane a pitch that is a multiple of 4.
    auto prefetch = [&](const ff_hdr* H, int f, auto lo_tag, auto hi_tag) __attribute__((always_inline)) {
        constexpr int KLO = decltype(lo_tag)::value, KHI = decltype(hi_tag)::value;
        if (dbg & 4) return;
        const zm_ff* F = fr + f;
        const int nx = F->nx, ny = F->ny;
        const int bx0 = H->bx0, by0 = H->by0, bh = max(H->bh, 1), bw4 = max(H->bw >> 2, 1);
        const int r0 = (int)(((float)tid + 0.5f) * (1.0f / (float)bw4));
        const int c = tid - r0 * bw4;
        const int RP = FF_THREADS / bw4;                  // (uniform) rows per slot
        const bool prepped = F->src != nullptr;
        const int gx = bx0 + 4 * c;
        // raw: one quad of image, one of weights, the y part of its mesh column.  prepped: the two pairs.
        const int sp = F->spitch;
        const int xa = prepped ? min(max(gx, 0), sp - 2) : min(max(gx, 0), nx - 4);
        const int xb = prepped ? min(max(gx + 2, 0), sp - 2) : xa;
        const int xm = min(max(gx, 0), F->mpitch - 4);
        const int i0 = bk_col(F->nbx, F->invmesh, min(max(gx, 0), nx - 1));
        const float ZM_GLOBAL* pa0 = prepped ? (const float ZM_GLOBAL*)zm_gptr(F->src) + 2 * (size_t)xa : zm_gptr(F->img) + xa;
        const float ZM_GLOBAL* pb0 = prepped ? (const float ZM_GLOBAL*)zm_gptr(F->src) + 2 * (size_t)xb
                                             : (F->wgt ? zm_gptr(F->wgt) : zm_gptr(F->img)) + xb;
        const size_t rowf = prepped ? 2 * (size_t)sp : (size_t)nx;       // floats per row of that plane
        // (frames without a background / a mask: a harmless load of image pixels, ignored at the store)
        const bool has_y = F->ytab != nullptr && !prepped, has_m = MOP && F->mask != nullptr;
        const char ZM_GLOBAL* py0 = has_y ? (const char ZM_GLOBAL*)(zm_gptr(F->ytab) + i0) : (const char ZM_GLOBAL*)pa0;
        const size_t rowy = has_y ? sizeof(float4) * (size_t)F->ytp : sizeof(float) * rowf;
        const char ZM_GLOBAL* pm0 = has_m ? (const char ZM_GLOBAL*)(zm_gptr(F->mbox) + xm) : (const char ZM_GLOBAL*)pa0;
        const size_t rowm = has_m ? sizeof(uint16_t) * (size_t)F->mpitch : sizeof(float) * rowf;
        // (measured and dropped: not loading a slot none of this wave's lanes needs - the last one, for
        // most waves - by a wave-uniform branch: 2.38 -> 2.46 ms)
        zm_static_for<KLO, KHI>([&](auto K) {
            constexpr int k = decltype(K)::value;
            const int row = min(min(r0, RP - 1) + k * RP, bh - 1);
            const size_t gy = (size_t)min(max(by0 + row, 0), ny - 1);
            pi[k] = zm_gload4f(pa0 + gy * rowf);
            pw[k] = zm_gload4f(pb0 + gy * rowf);
            const zm_v4f vy = *(const zm_v4f ZM_GLOBAL*)(py0 + gy * rowy);
            py4[k] = make_float4(vy.x, vy.y, vy.z, vy.w);
            const zm_v2u vm = *(const zm_v2u ZM_GLOBAL*)(pm0 + gy * rowm);
            pm[k] = make_uint2(vm.x, vm.y);
        });
    };
    // prep the staged quads (background off, variance, bad pixels) and write them to LDS buffer `b`
    // (two instantiations: `fast` items - nine in ten - carry no bounds test; written as one body with run-time
    // tests the compiler built a branch per staged pixel around the fill)
    auto store_impl = [&](const ff_hdr* H, int f, int b, auto fast_tag) __attribute__((always_inline)) {
        constexpr bool FAST = decltype(fast_tag)::value;
        const zm_ff* F = fr + f;
        const int nx = F->nx, ny = F->ny, sp = F->spitch;
        const int bx0 = H->bx0, by0 = H->by0, bh = H->bh, bw = H->bw, bw4 = bw >> 2;
        const int r0 = (int)(((float)tid + 0.5f) * (1.0f / (float)bw4));
        const int c = tid - r0 * bw4;
        const int RP = FF_THREADS / bw4;
        float2* tile = tile0 + (size_t)b * lds_cap;
        uint16_t* mtile = mtile0 + (size_t)b * lds_cap;
        const bool prepped = F->src != nullptr;
        const bool has_w = F->wgt != nullptr, has_y = F->ytab != nullptr && !prepped, has_m = MOP && F->mask != nullptr;
        const int gx = bx0 + 4 * c;
        const float vs = H->vscale, wth = F->wthresh;
        const float4 fill = make_float4(0.f, ZM_BIGVAR, 0.f, ZM_BIGVAR);
        // columns of this thread that lie on the frame (the loads came from clamped addresses)
        const bool cok = FAST || (gx >= 0 && gx + 4 <= nx);                         // raw quad (nx % 4 == 0)
        const bool cpa = FAST || (gx >= 0 && gx <= sp - 2), cpb = FAST || (gx + 2 >= 0 && gx + 2 <= sp - 2);
        const bool cm = FAST || (gx >= 0 && gx + 4 <= F->mpitch);
        // the four x weights of this thread's four columns: once per item
        float4 xw[4];
        if (has_y) {
            const int i0 = bk_col(F->nbx, F->invmesh, min(max(gx, 0), nx - 1));
#pragma unroll
            for (int e = 0; e < 4; ++e) xw[e] = bk_xweights(bk_dx(F->nbx, F->invmesh, gx + e, i0));
        }
        bool raw = false;
        if (r0 < RP) {
            zm_static_for<0, FF_NSLOT>([&](auto K) {
                constexpr int k = decltype(K)::value;
                const int row = r0 + k * RP;
                if (row >= bh) return;
                const bool rowok = FAST || (unsigned)(by0 + row) < (unsigned)ny;
                float4 o0, o1;
                if (prepped) {
                    o0 = (rowok && cpa) ? pi[k] : fill;
                    o1 = (rowok && cpb) ? pw[k] : fill;
                } else {
                    const float v[4] = {pi[k].x, pi[k].y, pi[k].z, pi[k].w};
                    const float w[4] = {pw[k].x, pw[k].y, pw[k].z, pw[k].w};
                    float2 p[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float bg = has_y ? bk_xpart(py4[k], xw[e]) : 0.f;
                        p[e] = prep_pixel(v[e], w[e], has_w, bg, vs, wth);
                        if (!FAST) {
                            const bool ok = rowok && cok;
                            p[e].x = ok ? p[e].x : 0.f;
                            p[e].y = ok ? p[e].y : ZM_BIGVAR;
                        }
                    }
                    o0 = make_float4(p[0].x, p[0].y, p[1].x, p[1].y);
                    o1 = make_float4(p[2].x, p[2].y, p[3].x, p[3].y);
                }
                float4* d = reinterpret_cast<float4*>(tile + (size_t)row * bw + 4 * c);
                d[0] = o0;
                d[1] = o1;
                if (MOP) {
                    // (entries whose 6 x 6 footprint leaves the frame are never written by the box
                    // pre-pass and never folded: the pixel loop tests the footprint)
                    uint2 mv = pm[k];
                    if (!(has_m && rowok && cm)) mv = make_uint2(0u, 0u);
                    *reinterpret_cast<uint2*>(mtile + (size_t)row * bw + 4 * c) = mv;
                    const uint32_t a = mv.x, bb = mv.y;
                    raw |= (a & 0xffffu) == ZM_BOX_RAW || (a >> 16) == ZM_BOX_RAW ||
                           (bb & 0xffffu) == ZM_BOX_RAW || (bb >> 16) == ZM_BOX_RAW;
                }
            });
        }
        if (MOP) {
            // does any entry of the box defer to the raw mask (ZM_BOX_RAW: bits above 15)?  Decided
            // here, once per item, so that the pixel loop carries no vote and no branch for it
            const bool wraw = __any(raw);
            if ((tid & 63) == 0) rawflag[b * (FF_THREADS / 64) + (tid >> 6)] = wraw;
        }
    };
    auto store = [&](const ff_hdr* H, int f, int b) __attribute__((always_inline)) {
        if (!H->use_lds || (dbg & 2)) return;
        if (H->fast) store_impl(H, f, b, std::true_type{});
        else store_impl(H, f, b, std::false_type{});
    };
    const int nty = ntiles / ntx;
    // queue position -> tile: the top and bottom rows of tiles (edge items: the slow ones) go first
    auto tile_of = [&](int s) -> int {
        if (s >= ntiles) return s;
        const int r = s / ntx, c = s - r * ntx;
        return (r == 0 ? 0 : r == 1 ? nty - 1 : r - 1) * ntx + c;
    };
    auto next_item = [&](int& tt, int& ff, int& kk) {
        if (++ff == nfr) { ff = 0; ++kk; tt = tring[kk & 3]; }
    };
    auto hdr_word = [&](int tt, int ff) -> int {          // this thread's word of the header of item (tt, ff)
        return tid < FF_HDR_WORDS ? ghdr[((size_t)tt * nfr + ff) * FF_HDR_WORDS + tid] : 0;
    };
    auto hdr_put = [&](int sl, int wv) {
        if (tid < FF_HDR_WORDS) reinterpret_cast<int*>(&HR[sl])[tid] = wv;
    };

    if ((int)blockIdx.x >= ntiles) return;
    int t0 = tile_of(blockIdx.x), f0 = 0, k2 = 0;
    for (int e = tid; e < LZ_FLOATS / 4; e += FF_THREADS)
        reinterpret_cast<float4*>(smem + FF_LDS_HDR)[e] = reinterpret_cast<const float4*>(taptab)[e];
    if (tid == 0) {
        // stacks of one or two frames look two items = up to two tiles ahead
        tring[0] = t0;
        if (nfr <= 2) tring[1] = tile_of(atomicAdd(tilectr, 1));
        if (nfr == 1) tring[2] = tile_of(atomicAdd(tilectr, 1));
    }
    __syncthreads();
    int t1 = t0, f1 = f0;
    next_item(t1, f1, k2);
    int t2 = t1, f2 = f1;
    next_item(t2, f2, k2);
    hdr_put(0, hdr_word(t0, f0));
    if (t1 < ntiles) hdr_put(1, hdr_word(t1, f1));
    __syncthreads();
    prefetch(&HR[0], f0, std::integral_constant<int, 0>{}, std::integral_constant<int, FF_NSLOT>{});
    store(&HR[0], f0, 0);
    __syncthreads();

    // ---- this thread's pixels: column tx, rows 8 wv .. 8 wv + 7 of the tile.  The wave lies in
    // sub-tile `sub` (k_resample's tile), lattice cell row `cr` of it; all wave-uniform.
    const int tx = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sub = (FF_NSUB > 1) ? wv >> 2 : 0, cr = (wv & 3) >> 1;
    const int cell = tx >> 4;
    const float fx = (float)(tx & 15) * (1.f / LSTEP);
    const float fyb = (float)((wv & 1) * 8) * (1.f / LSTEP);
    float S1[FF_NPX], S0[FF_NPX], SW[FF_NPX];
    int32_t MK[FF_NPX];
#pragma unroll
    for (int q = 0; q < FF_NPX; ++q) { S1[q] = 0.f; S0[q] = 0.f; SW[q] = 0.f; MK[q] = -1; }

    // STACK: the samples of item (pt, pfr), held in S1 / S0, to plane pfr of the stack
    int pt = -1, pfr = 0;
    auto flush = [&]() {
        if (pt < 0) return;
        const int ptyi = pt / ntx, ptxi = pt - ptyi * ntx;
        const int pox = ptxi * TW + tx, poy0 = ptyi * FT_H + wv * FF_NPX;
        float2* plane = stack + (size_t)pfr * (size_t)fstride;
#pragma unroll
        for (int q = 0; q < FF_NPX; ++q) {
            const int oy = poy0 + q;
            // (read back by the combine kernel, 2.4 GB later: non-temporal)
            if (pox < onx && oy < ony)
                __builtin_nontemporal_store((zm_v2f){S1[q], S0[q]}, reinterpret_cast<zm_v2f*>(plane + (size_t)oy * onx + pox));
        }
    };
    // developer (ZM_FF_PROF=1): shader-clock sums per phase of this wave
    long long ptk[5] = {0, 0, 0, 0, 0}, tc = 0;
#define FF_TICK(k) do { if (prof) { const long long t_ = __builtin_amdgcn_s_memtime(); ptk[k] += t_ - tc; tc = t_; } } while (0)
    if (prof) tc = __builtin_amdgcn_s_memtime();
    int slot = 0, buf = 0;
    for (;;) {
        const ff_hdr* H = &HR[slot];
        const int nslot = slot == 2 ? 0 : slot + 1;
        const int nnslot = nslot == 2 ? 0 : nslot + 1;
        const zm_ff* F = fr + f0;
        const bool use_lds = H->use_lds, touches = H->touches, fast = H->fast;
        const tile_hdr3* SH = &H->sub[sub];
        const int bw = H->bw;
        const int sbx0 = SH->bx0, sby0 = SH->by0;                 // the sub-box origin (k_resample's box)
        const int soff = H->sdy[sub] * bw + H->sdx[sub];          // ... inside the staged box
        const float2* tile = tile0 + (size_t)buf * lds_cap;
        const uint16_t* mtile = mtile0 + (size_t)buf * lds_cap;
        const int tyi = t0 / ntx, txi = t0 - tyi * ntx;
        const int ox0 = txi * TW, oy0 = tyi * FT_H + wv * FF_NPX;
        const int ox = ox0 + tx;
        // in this order: the header word and the queue atomic are older than the prefetch, so using
        // them at the end of the item waits with vmcnt(prefetch loads), not vmcnt(0)
        if (STACK) {
            flush();
#pragma unroll
            for (int q = 0; q < FF_NPX; ++q) { S1[q] = 0.f; S0[q] = 0.f; }
            pt = t0;
            pfr = f0;
        }
        int hw2 = 0;
        if (t2 < ntiles) hw2 = hdr_word(t2, f2);
        // the item two ahead is the last of its tile: take the tile after it from the queue (its ring
        // slot, ordinal k2 + 1, is not in use)
        const bool grab = f2 == nfr - 1;
        int gnext = 0;
        if (grab && tid == 0) gnext = atomicAdd(tilectr, 1);
        // (unconditional: past the last item the current one is fetched again, into registers nobody stores)
        const bool more = t1 < ntiles;
        // The staging loads of the next item go out in two bursts, one ahead of each pixel group: sixteen
        // loads per thread at once back up the vector-memory pipe (3 700 cycles per item spent issuing them)
        const ff_hdr* HN = more ? &HR[nslot] : H;
        const int fn = more ? f1 : f0;
        prefetch(HN, fn, std::integral_constant<int, 0>{}, std::integral_constant<int, FF_NSLOT / 2>{});
        FF_TICK(0);

        const bool do_px = touches && !(dbg & 1);
        {
            // x part of the bilinear lattice interpolation, once per item (k_resample's operations)
            const float x0a = SH->nrel[cr][cell][0], x1a = SH->nrel[cr][cell + 1][0];
            const float y0a = SH->nrel[cr][cell][1], y1a = SH->nrel[cr][cell + 1][1];
            const float x0b = SH->nrel[cr + 1][cell][0], x1b = SH->nrel[cr + 1][cell + 1][0];
            const float y0b = SH->nrel[cr + 1][cell][1], y1b = SH->nrel[cr + 1][cell + 1][1];
            const float xa = __builtin_fmaf(fx, x1a - x0a, x0a), ya = __builtin_fmaf(fx, y1a - y0a, y0a);
            const float xb = __builtin_fmaf(fx, x1b - x0b, x0b), yb = __builtin_fmaf(fx, y1b - y0b, y0b);
            const float xd = xb - xa, yd = yb - ya;
            const bool with_mask = MOP && F->mask != nullptr;
            const bool staged = use_lds;
            // wave-uniform: pixels left to the generic code (items that do not go through LDS: all)
            unsigned slow = !do_px ? 0u : staged ? 0u : 0xffu;
            bool any_raw = false;
            if (MOP && with_mask && staged) {
                const int* rf = rawflag + buf * (FF_THREADS / 64);
                int any = 0;
#pragma unroll
                for (int u = 0; u < FF_THREADS / 64; ++u) any |= rf[u];
                any_raw = any != 0;
            }
            const float fscale = F->fscale, fscale2 = F->fscale2;
            const float2* tbase = tile + (soff + OFF * bw + OFF);
            const uint16_t* mbase = mtile + (soff + OFF * bw + OFF);
            const int enx = F->nx, eny = F->ny;
            // four vertically adjacent pixels (rows 4 g .. 4 g + 3 of the thread) out of one 9 x 6 window
            auto group = [&](auto edge_tag, auto g_tag) __attribute__((always_inline)) {
                constexpr bool EDGE = decltype(edge_tag)::value;
                constexpr int G = decltype(g_tag)::value;
                float fxf0 = 0.f, fyf0 = 0.f, dxs[4], dys[4];
                bool shape = true;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float fy = fyb;
                    // (an empty volatile asm keeps the row fraction from being hoisted out of the item
                    // loop as eight live constants)
                    asm volatile("" : "+v"(fy));
                    fy += (float)(4 * G + j) * (1.f / LSTEP);
                    const float px = __builtin_fmaf(fy, xd, xa), py = __builtin_fmaf(fy, yd, ya);
                    const float fxf = floorf(px), fyf = floorf(py);
                    const float dx = px - fxf, dy = py - fyf;
                    dxs[j] = dx;
                    dys[j] = dy;
                    if (j == 0) { fxf0 = fxf; fyf0 = fyf; }
                    // snap rule: a fraction within 1e-5 of 0 or 1 on either axis -> generic code;
                    // window shape: the same six columns, rows one apart
                    const float edge = fminf(fminf(dx, 1.f - dx), fminf(dy, 1.f - dy));
                    shape = shape && !(edge < ZM_SNAP) && fxf == fxf0 && fyf == fyf0 + (float)j;
                }
                if (!__all(shape)) {
                    slow |= 0xfu << (4 * G);
                    return;
                }
                const int ix0 = (int)fxf0, iy0 = (int)fyf0;
                const int lo = iy0 * bw + ix0;                       // element offset in both tiles
                const float2* p = tbase + lo;
                // edge items: is the footprint on the frame (decides the mask fold; value and weight
                // follow from the {0, BIGVAR} fill), is the pixel on the output grid.  (Decided here,
                // as four bits: computed after the row pipeline these tests cost 300 spilled registers)
                unsigned inbm = 0xfu;
                if (EDGE && MOP) {
                    const int ix = sbx0 + OFF + ix0, iy = sby0 + OFF + iy0;
                    const bool xin = ix >= 0 && ix + NT <= enx && ox < onx;
                    inbm = 0u;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        inbm |= (xin && iy + j >= 0 && iy + j + NT <= eny && oy0 + 4 * G + j < ony) ? (1u << j) : 0u;
                }
                // The mask term of each pixel, as plain data flow: what the accumulator is ANDed with
                // (-1: nothing to fold - no mask, footprint off the frame).  Conditional code in the
                // epilogue below is cloned per pixel by the compiler and costs 250 spilled registers.
                int32_t mterm[4] = {-1, -1, -1, -1};
                if (MOP) {
                    uint32_t m16[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) m16[j] = mbase[lo + j * bw];          // (the tile exists, mask or not)
                    // bits above 15 somewhere in this box (a reference mask with bit 16): a footprint whose
                    // box-OR entry defers to the raw mask is left to the generic code, which ORs it
                    if (any_raw) {
                        bool defer = false;
#pragma unroll
                        for (int j = 0; j < 4; ++j) defer |= m16[j] == ZM_BOX_RAW && ((inbm >> j) & 1u);
                        if (__any(defer)) {
                            slow |= 0xfu << (4 * G);
                            return;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int32_t t = ff_mask_term<MOP>((int32_t)m16[j]);
                        mterm[j] = (with_mask && ((inbm >> j) & 1u)) ? t : -1;
                    }
                }
                // the eight tap-table nodes (x and y of four pixels), two in flight: the node of lookup i + 1 is
                // requested before lookup i is evaluated (left to the compiler, every lookup waits for its own
                // five reads right after issuing them: eight exposed LDS round trips per group)
                zm_v2f txp[4][3], typ[4][3];
                {
                    lz3_node na, nb;
                    float dla, dlb;
                    asm volatile("; ZM_LGKM_BEGIN" ::: "memory");   // (tests/test_isa_lint.py: no compiler-made lgkm operation up to ZM_LGKM_END)
                    lz3_issue(ltab, dxs[0], na, dla);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        lz3_node& cur = (i & 1) ? nb : na;
                        lz3_node& nxt = (i & 1) ? na : nb;
                        float& dlc = (i & 1) ? dlb : dla;
                        float& dln = (i & 1) ? dla : dlb;
                        if (i + 1 < 8) {
                            lz3_issue(ltab, ((i + 1) & 1) ? dys[(i + 1) >> 1] : dxs[(i + 1) >> 1], nxt, dln);
                            lz3_wait<5>(cur);
                        } else {
                            lz3_wait<0>(cur);
                        }
                        if (i & 1) lz3_eval(cur, dlc, typ[i >> 1]);
                        else lz3_eval(cur, dlc, txp[i >> 1]);
                    }
                }
                // rows 0 .. 8 of the window; row rho is tap row rho - j of pixel j.  Row rho + 1 is read
                // while the packed FMAs of row rho run (two row buffers).
                zm_v2f av[4];
                // three row buffers: rows rho + 1 and rho + 2 are in flight while row rho is applied (two
                // waves per SIMD do not cover an LDS round trip with the 24 packed FMAs of one row)
                lds_row6 rbuf[3];
                lds_issue6(p, rbuf[0]);
                lds_issue6(p + bw, rbuf[1]);
#pragma unroll
                for (int rho = 0; rho < NT + 3; ++rho) {
                    lds_row6& cur = rbuf[rho % 3];
                    if (rho + 2 < NT + 3) {
                        lds_issue6(p + (rho + 2) * bw, rbuf[(rho + 2) % 3]);
                        lds_wait_n<12>(cur);
                    } else if (rho + 1 < NT + 3) {
                        lds_wait_n<6>(cur);
                    } else {
                        lds_wait_n<0>(cur);
                    }
                    const unsigned long long rr[NT] = {cur.r0, cur.r1, cur.r2, cur.r3, cur.r4, cur.r5};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = rho - j;
                        if (r < 0 || r >= NT) continue;
                        zm_v2f rv2 = (zm_v2f){0.f, 0.f};
#pragma unroll
                        for (int c = 0; c < NT; ++c) {
                            const float tc = (c & 1) ? txp[j][c >> 1].y : txp[j][c >> 1].x;
                            rv2 = __builtin_elementwise_fma((zm_v2f){tc, tc}, lds_pair(rr[c]), rv2);
                        }
                        const float tr = (r & 1) ? typ[j][r >> 1].y : typ[j][r >> 1].x;
                        av[j] = __builtin_elementwise_fma((zm_v2f){tr, tr}, rv2, r == 0 ? (zm_v2f){0.f, 0.f} : av[j]);
                    }
                }
                asm volatile("; ZM_LGKM_END" ::: "memory");
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    constexpr int Q0 = 4 * G;
                    const int q = Q0 + j;
                    const float acc = av[j].x, vacc = av[j].y;
                    const bool ok = vacc > 0.f && vacc < ZM_BADVAR_TEST;
                    const float v = ok ? acc * fscale : 0.f;
                    const float w = ok ? __builtin_amdgcn_rcpf(vacc * fscale2) : 0.f;
                    const float ww = AVG ? (w > 0.f ? 1.f : 0.f) : w;
                    if (STACK) {
                        S1[q] = v;
                        S0[q] = w;
                    } else {
                        S1[q] = fmaf(ww, v, S1[q]);
                        S0[q] += ww;
                    }
                    if (AVG) SW[q] += w;
                    if (MOP) MK[q] &= mterm[j];
                }
            };
            if (do_px) {
                if (fast) group(std::false_type{}, std::integral_constant<int, 0>{});
                else if (use_lds) group(std::true_type{}, std::integral_constant<int, 0>{});
            }
            prefetch(HN, fn, std::integral_constant<int, FF_NSLOT / 2>{}, std::integral_constant<int, FF_NSLOT>{});
            if (do_px) {
                if (fast) group(std::false_type{}, std::integral_constant<int, 1>{});
                else if (use_lds) group(std::true_type{}, std::integral_constant<int, 1>{});
            }
            // the generic code, once: delta kernels, windows of another shape, footprints beyond the LDS tile
#pragma unroll 1
            while (slow) {
                const int q = __builtin_ctz(slow);
                slow &= slow - 1;
                const int oy = oy0 + q;
                if (ox >= onx || oy >= ony) continue;
                const float fy = fyb + (float)q * (1.f / LSTEP);
                const float px = __builtin_fmaf(fy, xd, xa), py = __builtin_fmaf(fy, yd, ya);
                const ff_px r = ff_generic_pixel<MOP>(F, tile + soff, ltab, use_lds, touches, sbx0, sby0, bw, px, py);
                const float ww = AVG ? (r.w > 0.f ? 1.f : 0.f) : r.w;
#pragma unroll
                for (int k = 0; k < FF_NPX; ++k) {
                    const bool me = (k == q);
                    S1[k] = me ? (STACK ? r.v : fmaf(ww, r.v, S1[k])) : S1[k];
                    S0[k] = me ? (STACK ? r.w : S0[k] + ww) : S0[k];
                    if (AVG) SW[k] = me ? SW[k] + r.w : SW[k];
                    if (MOP) MK[k] = (me && with_mask && r.inb) ? ff_mask_fold<MOP>(MK[k], r.m) : MK[k];
                }
            }
        }

        if (f0 == nfr - 1) {
            // the tile is complete: coadd (or partial sums) and mask coadd, once
#pragma unroll
            for (int q = 0; q < FF_NPX; ++q) {
                const int oy = oy0 + q;
                if (ox < onx && oy < ony) {
                    const size_t o = (size_t)oy * onx + ox;
                    const float s1 = S1[q], s0 = S0[q];
                    if (STACK) {
                        // (the samples of this item are flushed with the others)
                    } else if (partial) {
                        out_img[o] = s1;
                        out_wgt[o] = s0;
                    } else {
                        out_img[o] = s0 > 0.f ? s1 / s0 : 0.f;
                        out_wgt[o] = AVG ? SW[q] : s0;
                    }
                    if (MOP) {
                        const int32_t a = ff_mask_result<MOP>(MK[q]);
                        if (partial) {
                            out_mask[o] = a;
                        } else {
                            out_mask[o] = a == -1 ? 0 : a;
                            if (out_cov) out_cov[o] = a == -1 ? 0.f : 1.f;
                        }
                    }
                }
                if (!STACK) { S1[q] = 0.f; S0[q] = 0.f; }
                SW[q] = 0.f; MK[q] = -1;
            }
        }
        FF_TICK(1);
        if (prof) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); FF_TICK(2); }
        // the next item: prepped into the other LDS buffer (the one the pixels of the item before
        // this one were read from - every wave is past that since the last barrier)
        if (more) store(&HR[nslot], f1, buf ^ 1);
        FF_TICK(3);
        if (t2 < ntiles) hdr_put(nnslot, hw2);       // slot nnslot was last read an item ago
        if (grab && tid == 0) tring[(k2 + 1) & 3] = tile_of(gnext);
        __syncthreads();
        FF_TICK(4);
        t0 = t1; f0 = f1;
        t1 = t2; f1 = f2;
        next_item(t2, f2, k2);
        slot = nslot;
        buf ^= 1;
        if (t0 >= ntiles) break;
    }
    if (STACK) flush();
    if (prof && (tid & 63) == 0)
        for (int k = 0; k < 5; ++k) prof[((size_t)blockIdx.x * (FF_THREADS / 64) + (tid >> 6)) * 5 + k] = ptk[k];
#undef FF_TICK
}

// ===========================================================================
// The same fused coadd with the staging done by the LDS-DMA engine (global_load_lds): the raw
// planes of the next item go from HBM straight into LDS - no staging registers - and are prepped
// LDS -> LDS behind the pixel phase.  Without the 56 staging registers and with four output
// pixels per thread (one vertical group) a wave needs <= 128 registers: two workgroups of 512
// threads per CU = FOUR waves per SIMD instead of two.  (The register-staged kernel above was
// measured bound by vector issue at ~50 % utilisation: two waves per SIMD do not cover each
// other's staging, barrier and LDS phases.)
//
// LDS per workgroup (80 KB): [headers][tap table][x weights of the box columns][y table: the y part of
// the background per box row and mesh column][raw image quads][raw weight quads][box-OR tile x 2]
// [prepped tile].  The DMA writes lane-linear (wave-uniform base + lane x 16 B), so the raw tiles
// are the box in row-major quads; the per-lane SOURCE address carries the row / column split.
// Per item: DMA of item i + 1 issued -> pixels of item i -> wait for the DMA, barrier -> prep
// pass raw -> prepped tile (item i + 1) -> barrier.  Results: bit-identical to the register-staged
// kernel and to k_resample (the same prep_pixel / bk_* functions, the same pixel group code).
#define FD_OFF_XW (FF_LDS_HDR + FF_LDS_TAB)
#define FD_OFF_YT (FD_OFF_XW + FD_XCOLS * 16)
#define FD_OFF_RAW (FD_OFF_YT + FD_YCOLS * FD_YROWS * 16)
static_assert(FD_OFF_RAW + 20 * FD_LDS_CAP + 4 * 8 * FD_YROWS <= 80 * 1024, "DMA-staged kernel: LDS budget of half a CU");

__device__ inline void ff_glds16(const void ZM_GLOBAL* src, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// DEV: the developer instance (ZM_FF_PROF phase clocks, ZM_FF_DBG ablations).  The production instances
// carry neither: the five phase counters and their clock alone held 12 SGPRs through the whole item loop of
// a kernel that spills SGPRs into VGPR lanes (every spill slot costs v_readlane / v_writelane on the vector
// pipe, and the lanes' registers count against the 128 of a wave at four waves per SIMD).
template <int MOP, bool AVG, bool STACK, bool DEV = false>
__global__ __launch_bounds__(FD_THREADS, 2) void k_coadd_fused_dma(
    const zm_ff* __restrict__ fr, int nfr, int onx, int ony, int lds_cap, int ntx, int ntiles,
    const int* __restrict__ ghdr, float* __restrict__ out_img, float* __restrict__ out_wgt,
    int32_t* __restrict__ out_mask, float* __restrict__ out_cov, int partial,
    const float* __restrict__ taptab, int* __restrict__ tilectr, float2* __restrict__ stack, long long fstride,
    int dbg_arg, long long* __restrict__ prof_arg) {
    long long* const prof = DEV ? prof_arg : nullptr;
    const int dbg = DEV ? dbg_arg : (dbg_arg & ~255);            // (bits 8 ..: the tile budget of the yield mode)
    extern __shared__ float4 smem4[];
    char* smem = reinterpret_cast<char*>(smem4);
    ff_hdr* HR = reinterpret_cast<ff_hdr*>(smem);                  // ring of 3 headers
    int* tring = reinterpret_cast<int*>(smem + 3 * sizeof(ff_hdr));   // tiles held, by ordinal & 3
    const float* ltab = reinterpret_cast<const float*>(smem + FF_LDS_HDR);
    float4* XW = reinterpret_cast<float4*>(smem + FD_OFF_XW);       // per box column: {dx1, dx, cdx1, cdx}
    float4* YT = reinterpret_cast<float4*>(smem + FD_OFF_YT);       // [mesh column][box row]
    const int mcap = lds_cap + 8 * FD_YROWS;                        // box-OR tile: rows padded to 8 pixels
    char* RAWI = smem + FD_OFF_RAW;
    char* RAWW = RAWI + 4 * (size_t)lds_cap;
    uint16_t* MSK0 = reinterpret_cast<uint16_t*>(RAWW + 4 * (size_t)lds_cap);
    float2* PREP = reinterpret_cast<float2*>(reinterpret_cast<char*>(MSK0) + 4 * (size_t)mcap);
    constexpr int NT = 6, OFF = -2, NW = FD_THREADS / 64, NPX = 4;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- staging, part 1: the DMA of an item's raw planes (every wave takes chunks of 64 pieces of 16 B)
    // Round 4: the instruction diet of the staging.  What an item needs of the background geometry comes with
    // its header (ia / xb) instead of two bk_col evaluations per item and one per quad; the x weights of the
    // box columns are fetched from the frame's table (k_bk_cols) by the DMA engine - the xweights pass is gone;
    // frame fields are read one by one (the descriptor by value cost ~30 SGPRs per section, spilled to lanes).
    auto dma = [&](const ff_hdr* H, int f, int mb) __attribute__((always_inline)) {
        const zm_ff* F = fr + f;
        const int use_lds = H->use_lds;
        const int bx0 = H->bx0, by0 = H->by0, bw = H->bw, bh = H->bh, bw4 = bw >> 2;
        const int hia = H->ia, hxb = H->xb;
        if (!use_lds || (dbg & 4)) return;
        const int nx = F->nx, ny = F->ny;
        const int nq = bw4 * bh;
        const float2* fsrc = F->src;
        const bool prepped = fsrc != nullptr;
        const float inv4 = __builtin_amdgcn_rcpf((float)bw4) * 1.0000002f;   // (p + 0.5) / bw4 floors right for p < 2^12
        if (prepped) {
            const int sp = F->spitch;
            const float ZM_GLOBAL* gS = (const float ZM_GLOBAL*)zm_gptr(fsrc);
#pragma unroll 1
            for (int chunk = wv; chunk * 64 < nq; chunk += NW) {
                const int p = chunk * 64 + lane;
                if (p < nq) {
                    const int row = (int)(((float)p + 0.5f) * inv4), c = p - row * bw4;
                    const unsigned gy = (unsigned)min(max(by0 + row, 0), ny - 1);
                    const int gx = bx0 + 4 * c;
                    const unsigned oa = (gy * (unsigned)sp + (unsigned)min(max(gx, 0), sp - 2)) * 2u;
                    const unsigned ob = (gy * (unsigned)sp + (unsigned)min(max(gx + 2, 0), sp - 2)) * 2u;
                    ff_glds16(gS + oa, RAWI + (size_t)chunk * 1024);
                    ff_glds16(gS + ob, RAWW + (size_t)chunk * 1024);
                }
            }
        } else {
            // (element offsets of a plane fit 32 bits: one uniform base + an unsigned lane offset per load)
            const float* fw = F->wgt;
            const float ZM_GLOBAL* gI = zm_gptr(F->img);
            const float ZM_GLOBAL* gW = fw ? zm_gptr(fw) : gI;
#pragma unroll 1
            for (int chunk = wv; chunk * 64 < nq; chunk += NW) {
                const int p = chunk * 64 + lane;
                if (p < nq) {
                    const int row = (int)(((float)p + 0.5f) * inv4), c = p - row * bw4;
                    const unsigned gy = (unsigned)min(max(by0 + row, 0), ny - 1);
                    const unsigned o = gy * (unsigned)nx + (unsigned)min(max(bx0 + 4 * c, 0), nx - 4);
                    ff_glds16(gI + o, RAWI + (size_t)chunk * 1024);
                    ff_glds16(gW + o, RAWW + (size_t)chunk * 1024);
                }
            }
        }
        const uint16_t* fmb = F->mbox;
        if (MOP && F->mask) {
            // the box-OR tile starts on a multiple of 8 pixels (16-byte pieces of a plane with such a pitch)
            const int mpitch = F->mpitch;
            const int mx0 = bx0 & ~7, bwm8 = ((bx0 + bw - mx0) + 7) >> 3, nm = bwm8 * bh;
            const float inv8 = __builtin_amdgcn_rcpf((float)bwm8) * 1.0000002f;
            char* M = reinterpret_cast<char*>(MSK0) + (size_t)mb * 2 * mcap;
            const uint16_t ZM_GLOBAL* gM = zm_gptr(fmb);
            // (the waves with the fewest image chunks first: chunk k goes to wave NW - 1 - k)
#pragma unroll 1
            for (int chunk = NW - 1 - wv; chunk * 64 < nm; chunk += NW) {
                const int p = chunk * 64 + lane;
                if (p < nm) {
                    const int row = (int)(((float)p + 0.5f) * inv8), c8 = p - row * bwm8;
                    const unsigned gy = (unsigned)min(max(by0 + row, 0), ny - 1);
                    const int gxm = min(max(mx0 + 8 * c8, 0), mpitch - 8);
                    ff_glds16(gM + (gy * (unsigned)mpitch + (unsigned)gxm), M + (size_t)chunk * 1024);
                }
            }
        }
        const float4* fyt = F->ytab;
        if (fyt && !prepped) {
            // the y part of the background for the box rows, one table column per mesh column under the box
            // (waves 4, 5), and the x weights of the box columns as [pixel of the quad][quad column] (waves 6, 7)
            if (wv >= 4 && wv < 6) {
                const int col = wv - 4;
                if ((col == 0 || hxb != 0x7fffffff) && lane < bh) {
                    const int ytp = F->ytp;
                    const unsigned gy = (unsigned)min(max(by0 + lane, 0), ny - 1);
                    ff_glds16(zm_gptr(fyt) + (gy * (unsigned)ytp + (unsigned)min(hia + col, ytp - 1)),
                              reinterpret_cast<char*>(YT) + (size_t)col * FD_YROWS * 16);
                }
            } else if (wv >= 6) {
                // slot k FD_XQ + c of the LDS table: weight k of the four pixels of quad column c of the box
                const int slot = (wv - 6) * 64 + lane;
                if (slot < FD_XCOLS) {
                    const int k = (slot * 2731) >> 16, c = slot - k * FD_XQ;          // slot / 24 for slot < 96
                    const int nq4 = nx >> 2;
                    const int gq = min(max((bx0 >> 2) + c, 0), nq4 - 1);
                    ff_glds16(zm_gptr(F->xtab) + (k * nq4 + gq), reinterpret_cast<char*>(XW) + (size_t)(wv - 6) * 1024);
                }
            }
        }
    };
    // ---- staging, part 2: raw quads -> prepped tile (background off, variance, bad pixels, fill)
    // A thread takes quads tid and tid + FD_THREADS of the box.  Raw and prepped tiles are linear in the quad
    // index (the DMA wrote quad q at 16 q, the pair plane holds it at 32 q): no row / column arithmetic for
    // the addresses; the row and quad column are needed for the background and for the frame edge only.
    // Straight-line per quad: every LDS read of both quads first, then the arithmetic (a read inside a
    // condition is waited for on the spot).  Conditions are item-uniform branches, never per pixel.
    auto prep_raw = [&](const ff_hdr* H, int f, auto fast_tag) __attribute__((always_inline)) {
        constexpr bool FAST = decltype(fast_tag)::value;
        const zm_ff* F = fr + f;
        const int bx0 = H->bx0, by0 = H->by0, bw = H->bw, bh = H->bh, bw4 = bw >> 2, hxb = H->xb;
        const float vs = H->vscale;
        const float* fw = F->wgt;
        const float4* fyt = F->ytab;
        const float fwth = F->wthresh;
        const int nx = F->nx, ny = F->ny;
        const int nq = bw4 * bh;
        const bool has_w = fw != nullptr, has_y = fyt != nullptr;
        const float inv4 = __builtin_amdgcn_rcpf((float)bw4) * 1.0000002f;   // (q + 0.5) / bw4 floors right for q < 2^12
        // (the second quad exists for the first waves only: a wave-uniform count)
        const int nk = (FD_THREADS + 64 * wv < nq) ? 2 : 1;
        float4 ra[2], rb[2], ry[2], xw[2][4];
        int rows[2], cs[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (k >= nk) break;
            const int q = min(tid + FD_THREADS * k, nq - 1);
            ra[k] = reinterpret_cast<const float4*>(RAWI)[q];
            rb[k] = reinterpret_cast<const float4*>(RAWW)[q];
            rows[k] = 0;
            cs[k] = 0;
            if (has_y || !FAST) {
                rows[k] = (int)(((float)q + 0.5f) * inv4);
                cs[k] = q - rows[k] * bw4;
            }
            if (has_y) {
                const int ysel = (bx0 + 4 * cs[k] >= hxb) ? FD_YROWS : 0;
                ry[k] = YT[ysel + rows[k]];
#pragma unroll
                for (int e = 0; e < 4; ++e) xw[k][e] = XW[e * FD_XQ + cs[k]];      // weight e of the quad's four pixels
            }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (k >= nk) break;
            const float v[4] = {ra[k].x, ra[k].y, ra[k].z, ra[k].w};
            const float w[4] = {rb[k].x, rb[k].y, rb[k].z, rb[k].w};
            float bg[4] = {0.f, 0.f, 0.f, 0.f};
            if (has_y) {
                // bk_xpart of the four pixels, two per packed instruction: xw[k][j] holds weight j of the four
                // pixels, ry[k] the y part {r0, r1, e0, e1}; the same products and fused multiply-adds in the
                // same order as bk_xpart (a product does not depend on the order of its factors)
                const float4 *X = xw[k], Y = ry[k];
                zm_v2f lo = (zm_v2f){X[0].x, X[0].y} * (zm_v2f){Y.x, Y.x};
                zm_v2f hi = (zm_v2f){X[0].z, X[0].w} * (zm_v2f){Y.x, Y.x};
                lo = __builtin_elementwise_fma((zm_v2f){X[1].x, X[1].y}, (zm_v2f){Y.y, Y.y}, lo);
                hi = __builtin_elementwise_fma((zm_v2f){X[1].z, X[1].w}, (zm_v2f){Y.y, Y.y}, hi);
                lo = __builtin_elementwise_fma((zm_v2f){X[2].x, X[2].y}, (zm_v2f){Y.z, Y.z}, lo);
                hi = __builtin_elementwise_fma((zm_v2f){X[2].z, X[2].w}, (zm_v2f){Y.z, Y.z}, hi);
                lo = __builtin_elementwise_fma((zm_v2f){X[3].x, X[3].y}, (zm_v2f){Y.w, Y.w}, lo);
                hi = __builtin_elementwise_fma((zm_v2f){X[3].z, X[3].w}, (zm_v2f){Y.w, Y.w}, hi);
                bg[0] = lo.x; bg[1] = lo.y; bg[2] = hi.x; bg[3] = hi.y;
            }
            float2 p[4];
            if (has_w) {
#pragma unroll
                for (int e = 0; e < 4; ++e) p[e] = prep_pixel(v[e], w[e], true, bg[e], vs, fwth);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) p[e] = prep_pixel(v[e], 1.f, false, bg[e], vs, fwth);
            }
            if (!FAST) {
                const int gx = bx0 + 4 * cs[k];
                const bool ok = (unsigned)(by0 + rows[k]) < (unsigned)ny && gx >= 0 && gx + 4 <= nx;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    p[e].x = ok ? p[e].x : 0.f;
                    p[e].y = ok ? p[e].y : ZM_BIGVAR;
                }
            }
            const int q = tid + FD_THREADS * k;
            if (q < nq) {
                float4* d = reinterpret_cast<float4*>(PREP) + 2 * q;
                d[0] = make_float4(p[0].x, p[0].y, p[1].x, p[1].y);
                d[1] = make_float4(p[2].x, p[2].y, p[3].x, p[3].y);
            }
        }
    };
    // frames that could not be staged raw arrive prepped (zm_ff.src): pairs as they are, fill at the frame edge
    auto prep_src = [&](const ff_hdr* H, int f, bool fast) __attribute__((always_inline)) {
        const zm_ff* F = fr + f;
        const int bx0 = H->bx0, by0 = H->by0, bw = H->bw, bh = H->bh, bw4 = bw >> 2;
        const int ny = F->ny, sp = F->spitch;
        const int nq = bw4 * bh;
        const float inv4 = __builtin_amdgcn_rcpf((float)bw4) * 1.0000002f;
#pragma unroll 1
        for (int q = tid; q < nq; q += FD_THREADS) {
            const float4 a = reinterpret_cast<const float4*>(RAWI)[q], b = reinterpret_cast<const float4*>(RAWW)[q];
            const int row = (int)(((float)q + 0.5f) * inv4), c = q - row * bw4;
            const int gx = bx0 + 4 * c;
            const bool rowok = fast || (unsigned)(by0 + row) < (unsigned)ny;
            const bool cpa = fast || (gx >= 0 && gx <= sp - 2), cpb = fast || (gx + 2 >= 0 && gx + 2 <= sp - 2);
            // (component-wise: a select between whole float4 values is lowered through a stack array)
            const bool oka = rowok && cpa, okb = rowok && cpb;
            float4* d = reinterpret_cast<float4*>(PREP) + 2 * q;
            d[0] = make_float4(oka ? a.x : 0.f, oka ? a.y : ZM_BIGVAR, oka ? a.z : 0.f, oka ? a.w : ZM_BIGVAR);
            d[1] = make_float4(okb ? b.x : 0.f, okb ? b.y : ZM_BIGVAR, okb ? b.z : 0.f, okb ? b.w : ZM_BIGVAR);
        }
    };
    auto prep = [&](const ff_hdr* H, int f, int mb) __attribute__((always_inline)) {
        const int use_lds = H->use_lds, fast = H->fast;                  // (both requested before the first branch)
        const float2* fsrc = fr[f].src;
        if (!use_lds || (dbg & 2)) return;
        if (fsrc) prep_src(H, f, fast != 0);
        else if (fast) prep_raw(H, f, std::true_type{});
        else prep_raw(H, f, std::false_type{});
    };
    const int nty = ntiles / ntx;
    // queue position -> tile: the top and bottom rows of tiles (edge items: the slow ones) go first
    auto tile_of = [&](int s) -> int {
        if (s >= ntiles) return s;
        const int r = s / ntx, c = s - r * ntx;
        return (r == 0 ? 0 : r == 1 ? nty - 1 : r - 1) * ntx + c;
    };
    auto next_item = [&](int& tt, int& ff, int& kk) {
        if (++ff == nfr) { ff = 0; ++kk; tt = tring[kk & 3]; }
    };
    auto hdr_word = [&](int tt, int ff) -> int {          // this thread's word of the header of item (tt, ff)
        return tid < FF_HDR_WORDS ? ghdr[((size_t)tt * nfr + ff) * FF_HDR_WORDS + tid] : 0;
    };
    auto hdr_put = [&](int sl, int wd) {
        if (tid < FF_HDR_WORDS) reinterpret_cast<int*>(&HR[sl])[tid] = wd;
    };

    if ((int)blockIdx.x >= ntiles) return;
    int t0 = tile_of(blockIdx.x), f0 = 0, k2 = 0;
    for (int e = tid; e < LZ_FLOATS / 4; e += FD_THREADS)
        reinterpret_cast<float4*>(smem + FF_LDS_HDR)[e] = reinterpret_cast<const float4*>(taptab)[e];
    if (tid == 0) {
        // stacks of one or two frames look two items = up to two tiles ahead
        tring[0] = t0;
        if (nfr <= 2) tring[1] = tile_of(atomicAdd(tilectr, 1));
        if (nfr == 1) tring[2] = tile_of(atomicAdd(tilectr, 1));
    }
    __syncthreads();
    int t1 = t0, f1 = f0;
    next_item(t1, f1, k2);
    int t2 = t1, f2 = f1;
    next_item(t2, f2, k2);
    hdr_put(0, hdr_word(t0, f0));
    if (t1 < ntiles) hdr_put(1, hdr_word(t1, f1));
    __syncthreads();
    dma(&HR[0], f0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    prep(&HR[0], f0, 0);
    __syncthreads();

    // ---- this thread's pixels: column tx, rows 4 wv .. 4 wv + 3 of the 64 x 32 tile (one group)
    const int tx = lane;
    const int cr = wv >> 2;
    const int cell = tx >> 4;
    const float fx = (float)(tx & 15) * (1.f / LSTEP);
    const float fyb = (float)((4 * wv) & 15) * (1.f / LSTEP);
    float S1[NPX], S0[NPX], SW[NPX];
    int32_t MK[NPX];
#pragma unroll
    for (int q = 0; q < NPX; ++q) { S1[q] = 0.f; S0[q] = 0.f; SW[q] = 0.f; MK[q] = -1; }

    // STACK: the samples of item (pt, pfr), held in S1 / S0, to plane pfr of the stack
    int pt = -1, pfr = 0;
    auto flush = [&]() {
        if (pt < 0) return;
        const int ptyi = pt / ntx, ptxi = pt - ptyi * ntx;
        const int pox = ptxi * TW + tx, poy0 = ptyi * RTH + wv * NPX;
        float2* plane = stack + (size_t)pfr * (size_t)fstride;
#pragma unroll
        for (int q = 0; q < NPX; ++q) {
            const int oy = poy0 + q;
            if (pox < onx && oy < ony)
                __builtin_nontemporal_store((zm_v2f){S1[q], S0[q]}, reinterpret_cast<zm_v2f*>(plane + (size_t)oy * onx + pox));
        }
    };
    long long ptk[5] = {0, 0, 0, 0, 0}, tc = 0;      // developer (ZM_FF_PROF=1): shader-clock sums per phase of this wave
#define FD_TICK(k) do { if (DEV && prof) { const long long t_ = __builtin_amdgcn_s_memtime(); ptk[k] += t_ - tc; tc = t_; } } while (0)
    if (DEV && prof) tc = __builtin_amdgcn_s_memtime();
    const int budget = dbg >> 8;
    int ngrab = 0;
    int slot = 0, buf = 0;
    for (;;) {
        const ff_hdr* H = &HR[slot];
        const int nslot = slot == 2 ? 0 : slot + 1;
        const int nnslot = nslot == 2 ? 0 : nslot + 1;
        const zm_ff* F = fr + f0;
        const bool use_lds = H->use_lds, touches = H->touches, fast = H->fast;
        const tile_hdr3* SH = &H->sub[0];
        const int bw = H->bw;
        const int sbx0 = SH->bx0, sby0 = SH->by0;
        const int mx0 = sbx0 & ~7, bwm = (((sbx0 + bw - mx0) + 7) >> 3) << 3;      // box-OR tile: origin, pitch
        const uint16_t* mtile = MSK0 + (size_t)buf * mcap;
        const int tyi = t0 / ntx, txi = t0 - tyi * ntx;
        const int ox0 = txi * TW, oy0 = tyi * RTH + wv * NPX;
        const int ox = ox0 + tx;
        if (STACK) {
            flush();
#pragma unroll
            for (int q = 0; q < NPX; ++q) { S1[q] = 0.f; S0[q] = 0.f; }
            pt = t0;
            pfr = f0;
        }
        int hw2 = 0;
        if (t2 < ntiles) hw2 = hdr_word(t2, f2);
        const bool grab = f2 == nfr - 1;
        int gnext = 0;
        if (grab) {
            // a tile budget (yield mode: the workgroup retires after `budget` tiles and leaves its CU slot to
            // whatever else is queued on the GPU; later workgroups of the launch carry on)
            const bool allowed = budget == 0 || ngrab + 1 < budget;
            if (tid == 0) gnext = allowed ? atomicAdd(tilectr, 1) : ntiles;
            ++ngrab;
        }
        const bool more = t1 < ntiles;
        // the raw planes of the next item: DMA into the raw tiles (free since the last barrier), its
        // box-OR tile into the other mask buffer; its x weights
        if (more) dma(&HR[nslot], f1, buf ^ 1);
        FD_TICK(0);

        const bool do_px = touches && !(dbg & 1);
        {
            // x part of the bilinear lattice interpolation, once per item (k_resample's operations)
            const float x0a = SH->nrel[cr][cell][0], x1a = SH->nrel[cr][cell + 1][0];
            const float y0a = SH->nrel[cr][cell][1], y1a = SH->nrel[cr][cell + 1][1];
            const float x0b = SH->nrel[cr + 1][cell][0], x1b = SH->nrel[cr + 1][cell + 1][0];
            const float y0b = SH->nrel[cr + 1][cell][1], y1b = SH->nrel[cr + 1][cell + 1][1];
            const float xa = __builtin_fmaf(fx, x1a - x0a, x0a), ya = __builtin_fmaf(fx, y1a - y0a, y0a);
            const float xb = __builtin_fmaf(fx, x1b - x0b, x0b), yb = __builtin_fmaf(fx, y1b - y0b, y0b);
            const float xd = xb - xa, yd = yb - ya;
            const bool with_mask = MOP && F->mask != nullptr;
            unsigned slow = !do_px ? 0u : use_lds ? 0u : 0xfu;
            // (does the frame's box-OR plane hold entries that defer to the raw mask - bits above 15?  A flag of
            // the box pre-pass, carried by the header: science masks never do, the pixel loop then has no vote)
            const bool any_raw = MOP && with_mask && use_lds && H->frame_raw != 0;
            const float fscale = F->fscale, fscale2 = F->fscale2;
            const float2* tbase = PREP + (OFF * bw + OFF);
            const uint16_t* mbase = mtile + (OFF * bwm + OFF + (sbx0 - mx0));
            const int enx = F->nx, eny = F->ny;
            // the four vertically adjacent pixels of this thread out of one 9 x 6 window
            // (one instantiation: two - fast / edge - end in a join where every accumulator is copied)
            const bool EDGE = !fast;
            auto group = [&]() __attribute__((always_inline)) {
                float fxf0 = 0.f, fyf0 = 0.f, dxs[4], dys[4];
                bool shape = true;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float fy = fyb;
                    asm volatile("" : "+v"(fy));
                    fy += (float)j * (1.f / LSTEP);
                    const float px = __builtin_fmaf(fy, xd, xa), py = __builtin_fmaf(fy, yd, ya);
                    const float fxf = floorf(px), fyf = floorf(py);
                    const float dx = px - fxf, dy = py - fyf;
                    dxs[j] = dx;
                    dys[j] = dy;
                    if (j == 0) { fxf0 = fxf; fyf0 = fyf; }
                    const float edge = fminf(fminf(dx, 1.f - dx), fminf(dy, 1.f - dy));
                    shape = shape && !(edge < ZM_SNAP) && fxf == fxf0 && fyf == fyf0 + (float)j;
                }
                if (!__all(shape)) {
                    slow |= 0xfu;
                    return;
                }
                const int ix0 = (int)fxf0, iy0 = (int)fyf0;
                const int lo = __mul24(iy0, bw) + ix0;
                const float2* p = tbase + lo;
                unsigned inbm = 0xfu;
                if (EDGE && MOP) {
                    const int ix = sbx0 + OFF + ix0, iy = sby0 + OFF + iy0;
                    const bool xin = ix >= 0 && ix + NT <= enx && ox < onx;
                    inbm = 0u;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        inbm |= (xin && iy + j >= 0 && iy + j + NT <= eny && oy0 + j < ony) ? (1u << j) : 0u;
                }
                int32_t mterm[4] = {-1, -1, -1, -1};
                if (MOP) {
                    const int lom = __mul24(iy0, bwm) + ix0;
                    uint32_t m16[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) m16[j] = mbase[lom + j * bwm];
                    if (any_raw) {
                        bool defer = false;
#pragma unroll
                        for (int j = 0; j < 4; ++j) defer |= m16[j] == ZM_BOX_RAW && ((inbm >> j) & 1u);
                        if (__any(defer)) {
                            slow |= 0xfu;
                            return;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int32_t t = ff_mask_term<MOP>((int32_t)m16[j]);
                        mterm[j] = (with_mask && ((inbm >> j) & 1u)) ? t : -1;
                    }
                }
                zm_v2f txp[4][3], typ[4][3];
                {
                    // (one tap-table node in flight: four waves per SIMD cover the round trip, and a second
                    // node buffer would not fit the 128 registers)
                    lz3_node nd;
                    float dl;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        lz3_issue(ltab, (i & 1) ? dys[i >> 1] : dxs[i >> 1], nd, dl);
                        lz3_wait<0>(nd);
                        if (i & 1) lz3_eval(nd, dl, typ[i >> 1]);
                        else lz3_eval(nd, dl, txp[i >> 1]);
                    }
                }
                zm_v2f av[4];
                lds_row6 ra, rb;
                const unsigned pa = (unsigned)(size_t)p, bw8 = (unsigned)bw * 8u;    // 32-bit LDS address, row pitch in bytes
                asm volatile("; ZM_LGKM_BEGIN" ::: "memory");   // (tests/test_isa_lint.py: no compiler-made lgkm operation up to ZM_LGKM_END)
                lds_issue6(pa, ra);
#pragma unroll
                for (int rho = 0; rho < NT + 3; ++rho) {
                    lds_row6& cur = (rho & 1) ? rb : ra;
                    lds_row6& nxt = (rho & 1) ? ra : rb;
                    if (rho + 1 < NT + 3) {
                        lds_issue6(pa + (unsigned)(rho + 1) * bw8, nxt);
                        lds_wait_n<6>(cur);
                    } else {
                        lds_wait_n<0>(cur);
                    }
                    const unsigned long long rr[NT] = {cur.r0, cur.r1, cur.r2, cur.r3, cur.r4, cur.r5};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = rho - j;
                        if (r < 0 || r >= NT) continue;
                        zm_v2f rv2 = (zm_v2f){0.f, 0.f};
#pragma unroll
                        for (int c = 0; c < NT; ++c) {
                            const float tc = (c & 1) ? txp[j][c >> 1].y : txp[j][c >> 1].x;
                            rv2 = __builtin_elementwise_fma((zm_v2f){tc, tc}, lds_pair(rr[c]), rv2);
                        }
                        const float tr = (r & 1) ? typ[j][r >> 1].y : typ[j][r >> 1].x;
                        av[j] = __builtin_elementwise_fma((zm_v2f){tr, tr}, rv2, r == 0 ? (zm_v2f){0.f, 0.f} : av[j]);
                    }
                }
                asm volatile("; ZM_LGKM_END" ::: "memory");
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float acc = av[j].x, vacc = av[j].y;
                    const bool ok = vacc > 0.f && vacc < ZM_BADVAR_TEST;
                    const float v = ok ? acc * fscale : 0.f;
                    const float w = ok ? __builtin_amdgcn_rcpf(vacc * fscale2) : 0.f;
                    const float ww = AVG ? (w > 0.f ? 1.f : 0.f) : w;
                    if (STACK) {
                        S1[j] = v;
                        S0[j] = w;
                    } else {
                        S1[j] = fmaf(ww, v, S1[j]);
                        S0[j] += ww;
                    }
                    if (AVG) SW[j] += w;
                    if (MOP) MK[j] &= mterm[j];
                }
            };
            if (do_px && use_lds) group();
            // the generic code, once: delta kernels, windows of another shape, footprints beyond the LDS tile
#pragma unroll 1
            while (slow) {
                const int q = __builtin_ctz(slow);
                slow &= slow - 1;
                const int oy = oy0 + q;
                if (ox >= onx || oy >= ony) continue;
                const float fy = fyb + (float)q * (1.f / LSTEP);
                const float px = __builtin_fmaf(fy, xd, xa), py = __builtin_fmaf(fy, yd, ya);
                const ff_px r = ff_generic_pixel<MOP>(F, PREP, ltab, use_lds, touches, sbx0, sby0, bw, px, py);
                const float ww = AVG ? (r.w > 0.f ? 1.f : 0.f) : r.w;
#pragma unroll
                for (int k = 0; k < NPX; ++k) {
                    const bool me = (k == q);
                    S1[k] = me ? (STACK ? r.v : fmaf(ww, r.v, S1[k])) : S1[k];
                    S0[k] = me ? (STACK ? r.w : S0[k] + ww) : S0[k];
                    if (AVG) SW[k] = me ? SW[k] + r.w : SW[k];
                    if (MOP) MK[k] = (me && with_mask && r.inb) ? ff_mask_fold<MOP>(MK[k], r.m) : MK[k];
                }
            }
        }

        if (f0 == nfr - 1) {
            // the tile is complete: coadd (or partial sums) and mask coadd, once
#pragma unroll
            for (int q = 0; q < NPX; ++q) {
                const int oy = oy0 + q;
                if (ox < onx && oy < ony) {
                    const size_t o = (size_t)oy * onx + ox;
                    const float s1 = S1[q], s0 = S0[q];
                    if (STACK) {
                    } else if (partial) {
                        out_img[o] = s1;
                        out_wgt[o] = s0;
                    } else {
                        out_img[o] = s0 > 0.f ? s1 / s0 : 0.f;
                        out_wgt[o] = AVG ? SW[q] : s0;
                    }
                    if (MOP) {
                        const int32_t a = ff_mask_result<MOP>(MK[q]);
                        if (partial) {
                            out_mask[o] = a;
                        } else {
                            out_mask[o] = a == -1 ? 0 : a;
                            if (out_cov) out_cov[o] = a == -1 ? 0.f : 1.f;
                        }
                    }
                }
                if (!STACK) { S1[q] = 0.f; S0[q] = 0.f; }
                SW[q] = 0.f; MK[q] = -1;
            }
        }
        FD_TICK(1);
        // every wave is through with the prepped tile, and every DMA of the next item has landed
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        FD_TICK(2);
        __syncthreads();
        FD_TICK(3);
        if (more) prep(&HR[nslot], f1, buf ^ 1);
        if (t2 < ntiles) hdr_put(nnslot, hw2);
        if (grab && tid == 0) tring[(k2 + 1) & 3] = tile_of(gnext);
        FD_TICK(4);
        __syncthreads();
        if (DEV && prof) { const long long t_ = __builtin_amdgcn_s_memtime(); ptk[3] += t_ - tc; tc = t_; }
        t0 = t1; f0 = f1;
        t1 = t2; f1 = f2;
        next_item(t2, f2, k2);
        slot = nslot;
        buf ^= 1;
        if (t0 >= ntiles) break;
    }
    if (STACK) flush();
    if (DEV && prof && lane == 0)
        for (int k = 0; k < 5; ++k) prof[((size_t)blockIdx.x * NW + wv) * 5 + k] = ptk[k];
#undef FD_TICK
}

// ===========================================================================
// Round 5: the owner-staged form of the same fused coadd (k_coadd_fused_own).
//
// k_coadd_fused_dma closes every item with TWO workgroup barriers: the raw planes of item i + 1 land in one
// buffer (RAW) and are prepped into another (PREP) that the pixels of item i are still reading, so "everybody is
// through with PREP" and "everybody has prepped" are two separate rendezvous, and between them the waves that
// hold one chunk of the box wait for those that hold two.  Here the two buffers are two SLOTS that take turns:
// the raw quads of item i + 1 are DMA'd into the slot the pixels of item i do NOT read and are prepped IN
// PLACE by the wave that issued their DMA - its own `s_waitcnt vmcnt(0)` is all the ordering the DMA -> prep
// hand-over needs (MI355X_MICROARCH.md: nothing orders a ds_read behind a pending LDS-DMA except the issuing
// wave's vmcnt) - so an item ends with ONE barrier: "slot of item i + 1 prepped, slot of item i free".
// What the prep of a chunk needs from OTHER waves' DMA - the x weights and the y table of the background - is
// fetched TWO items ahead into buffers that take turns as well, i.e. it is covered by the barrier of the item
// before (headers are therefore fetched three items ahead, a ring of four).
//
// In place: a chunk is 3 box rows x 20 quads = 60 lanes; the DMA writes its image quads to bytes [0, 960) and
// its weight quads to [960, 1920) of the chunk (lane-linear, 16 B per lane: what the engine can do); the
// prepped chunk is the same 1920 bytes as 60 x {v0, var0, v1, var1, v2, var2, v3, var3} = three rows of the
// {value, variance} pair plane at a FIXED pitch of 80 pixels (640 B).  A wave reads both raw quads of its
// lanes, then writes the pairs: every read of the chunk precedes every write (one wave, LDS in order).
// The fixed pitch and the fixed lane -> (row, quad column) map take the divisions, multiplications and row-pitch
// additions out of all three phases ("what would shrink it"): the DMA address of a piece is
// a clamp and a multiply-add on per-lane constants, the prep knows its row and column without arithmetic, the
// nine window rows of a pixel group are immediate offsets of ONE address register.
// Items whose box is wider than 80 or taller than 42 pixels take the generic per-pixel code (their header says
// so: use_lds = 0); the launcher picks this kernel only for stacks whose planned footprints fit (near-unit
// scale, rotations below about a degree) - everything else runs k_coadd_fused_dma as before.
// Results: bit-identical to k_coadd_fused_dma and to the k_resample path (the same prep_pixel / bk_* / tap
// and filter code in the same order; only LDS addresses differ).
#define FO_PQ 20                                 // quads per staged row
#define FO_P (4 * FO_PQ)                         // the fixed pitch: 80 pixels
#define FO_RPC 3                                 // box rows per raw chunk (60 of 64 lanes)
#define FO_NCH 14                                // raw chunks per slot
#define FO_ROWS (FO_RPC * FO_NCH)                // 42 box rows
#define FO_CHB (FO_RPC * FO_PQ * 32)             // 1920 B: a chunk, raw (image 960 | weight 960) or prepped
#define FO_SLOT (FO_NCH * FO_CHB)                // 26880 B
#define FO_MPC 11                                // 16-byte pieces (8 entries) per row of the box-OR tile
#define FO_MP (8 * FO_MPC)                       // its pitch: 88 entries
#define FO_MRPC 5                                // rows per mask chunk (55 of 64 lanes)
#define FO_NMCH 9                                // mask chunks per tile (45 rows)
#define FO_MCHB (FO_MRPC * FO_MPC * 16)          // 880 B
#define FO_MSLOT (FO_NMCH * FO_MCHB)             // 7920 B
#define FO_YROWS 44                              // rows of a y-table column
#define FO_XWB (4 * FO_PQ * 16)                  // x weights of an item: [weight][quad column] float4, 1280 B
#define FO_YTB (2 * FO_YROWS * 16)               // y table of an item: [mesh column][box row] float4, 1408 B
#define FO_OFF_TAB 896                           // [4 headers 768][tile ring 16] ... tap table
#define FO_OFF_XW (FO_OFF_TAB + FF_LDS_TAB)
#define FO_OFF_YT (FO_OFF_XW + 2 * FO_XWB)
#define FO_OFF_SLOT (FO_OFF_YT + 2 * FO_YTB)
#define FO_OFF_MSK (FO_OFF_SLOT + 2 * FO_SLOT)
#define FO_LDS (FO_OFF_MSK + 2 * FO_MSLOT)
static_assert(4 * sizeof(ff_hdr) + 4 * 4 <= FO_OFF_TAB, "owner-staged kernel: header ring");
static_assert(FO_OFF_XW % 16 == 0 && FO_OFF_YT % 16 == 0 && FO_OFF_SLOT % 16 == 0 && FO_OFF_MSK % 16 == 0, "16-byte LDS pieces");
static_assert(FO_LDS <= 80 * 1024, "owner-staged kernel: LDS budget of half a CU");
static_assert(FF_NSUB == 1, "owner-staged kernel: one k_resample tile per item");

// the six pairs of window row ROW (pitch FO_P pairs) as immediate offsets of one address register
template <int ROW>
__device__ inline void lds_issue6_row(unsigned a, lds_row6& o) {
    asm volatile("ds_read_b64 %0, %6 offset:%7\n\t"
                 "ds_read_b64 %1, %6 offset:%8\n\t"
                 "ds_read_b64 %2, %6 offset:%9\n\t"
                 "ds_read_b64 %3, %6 offset:%10\n\t"
                 "ds_read_b64 %4, %6 offset:%11\n\t"
                 "ds_read_b64 %5, %6 offset:%12"
                 : "=&v"(o.r0), "=&v"(o.r1), "=&v"(o.r2), "=&v"(o.r3), "=&v"(o.r4), "=&v"(o.r5)
                 : "v"(a), "n"(ROW * FO_P * 8), "n"(ROW * FO_P * 8 + 8), "n"(ROW * FO_P * 8 + 16),
                   "n"(ROW * FO_P * 8 + 24), "n"(ROW * FO_P * 8 + 32), "n"(ROW * FO_P * 8 + 40)
                 : "memory");
}

// (launch bounds: the second argument is waves per SIMD - four, i.e. two workgroups per CU, at most 128 vector
// registers.  With "2" the compiler is free to take 256 and did, on an unrelated edit: 208 registers, ONE workgroup
// per CU, 1.75 -> 2.57 ms.)
// A kernel argument fetched where it is used, from the kernarg segment, behind an opaque offset (the load cannot be
// hoisted out of the item loop): the products' pointers are needed once per 32 items, at a tile's completion; held
// in scalar registers for the whole loop they were a third of the kernel's scalar spills.
struct ff_own_args {                 // the argument list of k_coadd_fused_own as the kernarg segment holds it
    const zm_ff* fr; int nfr, onx, ony, lds_cap, ntx, ntiles; const int* ghdr; float* out_img; float* out_wgt;
    int32_t* out_mask; float* out_cov; int partial; const float* taptab; int* tilectr; float2* stack; long long fstride;
    int dbg_arg; long long* prof_arg;
};
template <typename T>
__device__ __forceinline__ T ff_karg(int byte_off) {
    asm volatile("" : "+s"(byte_off));
    typedef const char __attribute__((address_space(4))) kchar;
    kchar* k = (kchar*)__builtin_amdgcn_kernarg_segment_ptr();
    return *(const T __attribute__((address_space(4)))*)(k + byte_off);
}
#define FF_KARG(field) ff_karg<decltype(ff_own_args::field)>((int)offsetof(ff_own_args, field))

// Wave priority behind a wave-uniform condition, as ONE opaque statement: a C++ `if` around s_setprio inside the
// pixel group splits its straight-line block, and the register allocator answered with 208 registers (or, capped
// at 128, 100 spills).  sel: a scalar register; the priority becomes PRIO when sel == WHEN.
template <int WHEN, int PRIO>
__device__ __forceinline__ void ff_setprio_when(int sel) {
    asm volatile("s_cmp_lg_u32 %0, %1\n\ts_cbranch_scc1 1f\n\ts_setprio %2\n1:" : : "s"(sel), "n"(WHEN), "n"(PRIO) : "scc");
}
template <int MOP, bool AVG, bool STACK, bool DEV = false>
__global__ __launch_bounds__(FD_THREADS, 4) void k_coadd_fused_own(
    const zm_ff* __restrict__ fr, int nfr, int onx, int ony, int lds_cap, int ntx, int ntiles,
    const int* __restrict__ ghdr, float* __restrict__ out_img_, float* __restrict__ out_wgt_,
    int32_t* __restrict__ out_mask_, float* __restrict__ out_cov_, int partial_,
    const float* __restrict__ taptab, int* __restrict__ tilectr, float2* __restrict__ stack_, long long fstride_,
    int dbg_arg, long long* __restrict__ prof_arg) {
    // (out_img_ ... fstride_: read through FF_KARG where they are used)
    long long* const prof = DEV ? prof_arg : nullptr;
    const int dbg = (DEV ? dbg_arg : (dbg_arg & ~255)) & 0x00ffffff;   // (bits 8 .. 23: the tile budget of the yield mode)
    // developer switches (ZM_FF_PRIO, ZM_FF_DEAL): s_setprio 1 for 1 = the DMA issue, 2 = the prep, 4 = waves 4 - 7 in
    // their pixel phase, 8 .. 12 = waves 4 - 7 in its first part (below); deal: who stages what (below)
    // (the production instances carry the measured choice as constants: deal 1, switch point 2 - the runtime
    // switches cost scalar registers in a kernel that spills them)
    const int prio = DEV ? ((dbg_arg >> 24) & 15) : 9, deal = DEV ? ((dbg_arg >> 28) & 3) : 1;
    extern __shared__ float4 smem4[];
    char* smem = reinterpret_cast<char*>(smem4);
    ff_hdr* HR = reinterpret_cast<ff_hdr*>(smem);                  // ring of 4 headers
    int* tring = reinterpret_cast<int*>(smem + 4 * sizeof(ff_hdr));   // tiles held, by ordinal & 3
    const float* ltab = reinterpret_cast<const float*>(smem + FO_OFF_TAB);
    constexpr int NT = 6, OFF = -2, NW = FD_THREADS / 64, NPX = 4;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // (prio >= 8: the younger half - waves 4 - 7, which the SIMDs serve after the older waves - runs the FIRST part
    // of its pixel phase at priority 1 and drops back at switch point prio - 7: 1 = behind the tap lookups,
    // 2 .. 5 = behind window row 1, 3, 5, 7)
    int ysw = __builtin_amdgcn_readfirstlane((prio >= 8 && wv >= 4) ? prio - 7 : 0);
    asm volatile("" : "+s"(ysw));
    // Who stages what.  The SIMDs serve their OLDER waves first (MI355X_MICROARCH.md, "Two waves per SIMD"): the
    // phase clocks of the even deal (chunk k to wave k mod 8) showed waves 0 - 3 through their pixels in 1.48 M
    // cycles per launch and waves 4 - 7 in 1.9 M, the former waiting 0.85 M at the barrier.  So the older half gets
    // the staging: raw chunks 0 - 9 go to waves 0 - 3 (three, three, two, two), chunks 10 - 13 one each to waves
    // 4 - 7; the box-OR chunks (no prep) and the tables fill up the lighter waves.
    // deal 0: chunk k to wave k mod 8 (two, two, ..., one, one), box-OR chunks and tables to waves 6, 7;
    // deal 1: the older half heavy, as above; deal 2: three raw chunks each to waves 0 - 3, one each to waves 4, 5,
    // waves 6, 7 stage no raw chunk (and skip the prep): the box-OR chunks and the tables only
    const int own0 = deal == 0 ? wv : deal == 1 ? (wv < 4 ? wv : wv + 6) : (wv < 4 ? wv : wv + 8);   // first raw chunk
    const int owns = deal == 0 ? 8 : 4;                                                                // ... stride
    const int ownn = deal == 0 ? 2 : deal == 1 ? (wv < 2 ? 3 : wv < 4 ? 2 : 1) : (wv < 4 ? 3 : wv < 6 ? 1 : 0);
    // box-OR chunks m0, m0 + ms, ... (nm of them at most)
    const int m0 = deal == 0 ? wv - 6 : deal == 1 ? (wv < 4 ? wv - 2 : wv) : (wv < 6 ? wv - 4 : wv - 4);
    const int ms = deal == 0 ? 2 : deal == 1 ? (wv < 4 ? 2 : 1) : 4;
    const int nm = deal == 0 ? (wv >= 6 ? 5 : 0) : deal == 1 ? (wv < 2 ? 0 : wv < 4 ? 2 : wv == 7 ? 2 : 1)
                                                              : (wv < 4 ? 0 : wv == 4 ? 3 : 2);
    const int ytw = deal == 0 ? 6 : deal == 1 ? 4 : 6, xww = deal == 0 ? 7 : deal == 1 ? 5 : 7;

    // ---- the scalars of an item's staging, fetched in ONE batch (round 5, late).  The header fields and the frame
    // descriptor's fields used to be read where the code needed them, behind the conditions that decide whether it
    // does: in the ISA five to eight scalar loads one after the other, each waited for with lgkmcnt(0) before the
    // branch that guards the next - 2 900 cycles per wave and item of "DMA issue", a fifth of a wave's time, for a
    // handful of address computations.  Now every scalar of the DMA issue (item i + 1: box, frame size, plane
    // pointers; item i + 2: the tables) is requested before the first is used and one asm statement that names them
    // all keeps the compiler from sinking a load back behind a branch: one round trip through the scalar cache.
    auto hdr_g = [&](int tt, int ff) {
        // (the item's number on the scalar unit, 32 bits: with the 64-bit product the compiler moved the address to the
        // vector unit and the header fields became vector loads + v_readfirstlane behind a vmcnt(0))
        const unsigned it = __builtin_amdgcn_readfirstlane((unsigned)tt * (unsigned)nfr + (unsigned)ff);
        return reinterpret_cast<const ff_hdr ZM_GLOBAL*>(zm_gptr(ghdr) + (size_t)it * FF_HDR_WORDS);
    };
    struct dma_sc {                  // item i + 1
        int use_lds, bx0, by0, bw, bh, nx, ny, spitch, mpitch;
        const float2* src; const float* img; const float* wgt; const void* mask; const uint16_t* mbox;
    };
    struct prep_sc {                 // item i + 1, at its prep
        int use_lds, fast, bx0, by0, bh, xb, nx, ny, spitch;
        float vscale, wthresh;
        const float2* src; const float* wgt; const float4* ytab;
    };
    struct tab_sc {                  // item i + 2 (the waves that fetch tables)
        int use_lds, bx0, by0, bh, ia, xb, nx, ny, ytp;
        const float4* ytab; const float4* xtab; const float2* src;
    };
    auto load_dma_sc = [&](int tt, int ff, dma_sc& D) __attribute__((always_inline)) {
        const auto H = hdr_g(tt, ff);
        const zm_ff* F = fr + ff;
        D.use_lds = H->use_lds; D.bx0 = H->bx0; D.by0 = H->by0; D.bw = H->bw; D.bh = H->bh;
        D.nx = F->nx; D.ny = F->ny; D.spitch = F->spitch; D.mpitch = F->mpitch;
        D.src = F->src; D.img = F->img; D.wgt = F->wgt; D.mask = F->mask; D.mbox = F->mbox;
    };
    auto load_tab_sc = [&](int tt, int ff, tab_sc& T) __attribute__((always_inline)) {
        const auto H = hdr_g(tt, ff);
        const zm_ff* F = fr + ff;
        T.use_lds = H->use_lds; T.bx0 = H->bx0; T.by0 = H->by0; T.bh = H->bh; T.ia = H->ia; T.xb = H->xb;
        T.nx = F->nx; T.ny = F->ny; T.ytp = F->ytp;
        T.ytab = F->ytab; T.xtab = F->xtab; T.src = F->src;
    };
#define FO_PIN_DMA(D) asm volatile("; staging scalars (item + 1)" : : "s"(D.use_lds), "s"(D.bx0), "s"(D.by0), "s"(D.bw), "s"(D.bh), \
        "s"(D.nx), "s"(D.ny), "s"(D.spitch), "s"(D.mpitch), "s"(D.src), "s"(D.img), "s"(D.wgt), "s"(D.mask), "s"(D.mbox))
#define FO_PIN_TAB(T) asm volatile("; staging scalars (item + 2)" : : "s"(T.use_lds), "s"(T.bx0), "s"(T.by0), "s"(T.bh), "s"(T.ia), \
        "s"(T.xb), "s"(T.nx), "s"(T.ny), "s"(T.ytp), "s"(T.ytab), "s"(T.xtab), "s"(T.src))
    const bool tabs_wave = wv == ytw || wv == xww;
    // ---- staging, part 1: the DMA of an item's raw planes and of its box-OR tile.  A raw chunk (box rows
    // 3 k .. 3 k + 2) is prepped by the wave that issued its DMA; the box-OR chunks are five rows each.
    auto dma_item = [&](const dma_sc& D, int sl) __attribute__((always_inline)) {
        const int bx0 = D.bx0, by0 = D.by0, bw = D.bw, bh = D.bh;
        if (!D.use_lds || (dbg & 4)) return;
        int ln = lane;
        asm volatile("" : "+v"(ln));             // (the lane map is recomputed per item: held across the pixel
                                                 // phase its four values would cost registers the group needs)
        const int lrow = (ln * 205) >> 12, lcol = ln - lrow * FO_PQ;      // ln / 20 for ln < 80
        const int nx = D.nx, ny = D.ny;
        char* SL = smem + FO_OFF_SLOT + sl * FO_SLOT;
        const float2* fsrc = D.src;
        const int gx = bx0 + 4 * lcol;
        const bool lok = lrow < FO_RPC && lcol < (bw >> 2);
        if (fsrc) {
            const int sp = D.spitch;
            const float ZM_GLOBAL* gS = (const float ZM_GLOBAL*)zm_gptr(fsrc);
            const unsigned xa = (unsigned)min(max(gx, 0), sp - 2), xb = (unsigned)min(max(gx + 2, 0), sp - 2);
#pragma unroll 1
            for (int j = 0; j < ownn; ++j) {
                const int k = own0 + owns * j, r = FO_RPC * k + lrow;
                if (lok && r < bh) {
                    const unsigned gy = (unsigned)min(max(by0 + r, 0), ny - 1);
                    ff_glds16(gS + (gy * (unsigned)sp + xa) * 2u, SL + k * FO_CHB);
                    ff_glds16(gS + (gy * (unsigned)sp + xb) * 2u, SL + k * FO_CHB + FO_CHB / 2);
                }
            }
        } else {
            const float* fw = D.wgt;
            const float ZM_GLOBAL* gI = zm_gptr(D.img);
            const float ZM_GLOBAL* gW = fw ? zm_gptr(fw) : gI;
            const unsigned xo = (unsigned)min(max(gx, 0), nx - 4);
#pragma unroll 1
            for (int j = 0; j < ownn; ++j) {
                const int k = own0 + owns * j, r = FO_RPC * k + lrow;
                if (lok && r < bh) {
                    const unsigned o = (unsigned)min(max(by0 + r, 0), ny - 1) * (unsigned)nx + xo;
                    ff_glds16(gI + o, SL + k * FO_CHB);
                    ff_glds16(gW + o, SL + k * FO_CHB + FO_CHB / 2);
                }
            }
        }
        if (MOP && D.mask && nm > 0) {
            const uint16_t* fmb = D.mbox;
            const int mpitch = D.mpitch;
            const int mrow = (ln * 187) >> 11, mcol = ln - mrow * FO_MPC;   // ln / 11 for ln < 64
            const int mx0 = bx0 & ~7, bwm8 = ((bx0 + bw - mx0) + 7) >> 3;
            char* M = smem + FO_OFF_MSK + sl * FO_MSLOT;
            const uint16_t ZM_GLOBAL* gM = zm_gptr(fmb);
            const bool mok = mrow < FO_MRPC && mcol < bwm8;
            const unsigned gxm = (unsigned)min(max(mx0 + 8 * mcol, 0), mpitch - 8);
#pragma unroll 1
            for (int j = 0; j < nm; ++j) {
                const int m = m0 + ms * j, r = FO_MRPC * m + mrow;
                if (m >= FO_NMCH) break;
                if (mok && r < bh)
                    ff_glds16(gM + ((unsigned)min(max(by0 + r, 0), ny - 1) * (unsigned)mpitch + gxm), M + m * FO_MCHB);
            }
        }
    };
    // ... and of the tables its prep reads (two items ahead): the y part of the background for the box rows,
    // one column per mesh column under the box (wave ytw), the x weights of the box columns as
    // [weight][quad column] (wave xww)
    auto dma_tabs = [&](const tab_sc& T, int tb) __attribute__((always_inline)) {
        const int bx0 = T.bx0, by0 = T.by0, bh = T.bh, hia = T.ia, hxb = T.xb;
        const float4* fyt = T.ytab;
        if (!T.use_lds || (dbg & 4) || !fyt || T.src || !tabs_wave) return;
        if (wv == ytw) {
            const int ny = T.ny, ytp = T.ytp;
            char* YT = smem + FO_OFF_YT + tb * FO_YTB;
            const unsigned gy = (unsigned)min(max(by0 + lane, 0), ny - 1);
            if (lane < bh) {
                ff_glds16(zm_gptr(fyt) + (gy * (unsigned)ytp + (unsigned)min(hia, ytp - 1)), YT);
                if (hxb != 0x7fffffff)
                    ff_glds16(zm_gptr(fyt) + (gy * (unsigned)ytp + (unsigned)min(hia + 1, ytp - 1)), YT + FO_YROWS * 16);
            }
        } else {
            const int nq4 = T.nx >> 2;
            char* XW = smem + FO_OFF_XW + tb * FO_XWB;
            const float4 ZM_GLOBAL* gX = zm_gptr(T.xtab);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int slot = j * 64 + lane;
                if (slot < 4 * FO_PQ) {
                    const int k = (slot * 205) >> 12, c = slot - k * FO_PQ;         // slot / 20 for slot < 80
                    const int gq = min(max((bx0 >> 2) + c, 0), nq4 - 1);
                    ff_glds16(gX + (k * nq4 + gq), XW + j * 1024);
                }
            }
        }
    };
    // ---- staging, part 2: the wave's own chunks, raw quads -> pairs, in place (background off, variance, bad
    // pixels, fill).  Straight-line per chunk: the LDS reads of both chunks first, then the arithmetic; the
    // conditions are item-uniform branches, never per pixel.
    auto prep_raw = [&](const prep_sc& P, int sl, int tb, auto fast_tag) __attribute__((always_inline)) {
        constexpr bool FAST = decltype(fast_tag)::value;
        const int bx0 = P.bx0, by0 = P.by0, bh = P.bh, hxb = P.xb;
        const float vs = P.vscale;
        const float* fw = P.wgt;
        const float4* fyt = P.ytab;
        const float fwth = P.wthresh;
        const int nx = P.nx, ny = P.ny;
        const bool has_w = fw != nullptr, has_y = fyt != nullptr;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int lrow = (ln * 205) >> 12, lcol = ln - lrow * FO_PQ;
        const int gx = bx0 + 4 * lcol;
        char* SL = smem + FO_OFF_SLOT + sl * FO_SLOT;
        const float4* XW = reinterpret_cast<const float4*>(smem + FO_OFF_XW + tb * FO_XWB);
        const float4* YT = reinterpret_cast<const float4*>(smem + FO_OFF_YT + tb * FO_YTB) + ((gx >= hxb) ? FO_YROWS : 0);
        float4 xw[4];
        if (has_y) {
#pragma unroll
            for (int e = 0; e < 4; ++e) xw[e] = XW[e * FO_PQ + lcol];      // weight e of the quad's four pixels
        }
        // one chunk at a time, the next chunk's raw quads requested before the arithmetic of this one.  The (at most
        // three) chunks of a wave are three copies of the code, not a loop: rolled, the look-ahead's twelve registers
        // were copied from "next" to "current" every iteration - six v_mov_b64 and four zeroing moves per chunk, an
        // eighth of the prep's vector instructions.
        float4 ra[3], rb[3], ry[3];
        auto fetch = [&](int j, int k) __attribute__((always_inline)) {
            const char* C = SL + k * FO_CHB;
            ra[j] = reinterpret_cast<const float4*>(C)[ln];
            rb[j] = reinterpret_cast<const float4*>(C + FO_CHB / 2)[ln];
            ry[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (has_y) ry[j] = YT[min(FO_RPC * k + lrow, FO_YROWS - 1)];
        };
        fetch(0, own0);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int k = own0 + owns * j;
            if (j >= ownn || FO_RPC * k >= bh) break;
            const float v[4] = {ra[j].x, ra[j].y, ra[j].z, ra[j].w};
            const float w[4] = {rb[j].x, rb[j].y, rb[j].z, rb[j].w};
            const float4 Y = ry[j];
            if (j + 1 < 3 && j + 1 < ownn) fetch(j + 1, k + owns);
            float bg[4] = {0.f, 0.f, 0.f, 0.f};
            if (has_y) {
                // bk_xpart of the four pixels, two per packed instruction (k_coadd_fused_dma's sequence)
                zm_v2f lo = (zm_v2f){xw[0].x, xw[0].y} * (zm_v2f){Y.x, Y.x};
                zm_v2f hi = (zm_v2f){xw[0].z, xw[0].w} * (zm_v2f){Y.x, Y.x};
                lo = __builtin_elementwise_fma((zm_v2f){xw[1].x, xw[1].y}, (zm_v2f){Y.y, Y.y}, lo);
                hi = __builtin_elementwise_fma((zm_v2f){xw[1].z, xw[1].w}, (zm_v2f){Y.y, Y.y}, hi);
                lo = __builtin_elementwise_fma((zm_v2f){xw[2].x, xw[2].y}, (zm_v2f){Y.z, Y.z}, lo);
                hi = __builtin_elementwise_fma((zm_v2f){xw[2].z, xw[2].w}, (zm_v2f){Y.z, Y.z}, hi);
                lo = __builtin_elementwise_fma((zm_v2f){xw[3].x, xw[3].y}, (zm_v2f){Y.w, Y.w}, lo);
                hi = __builtin_elementwise_fma((zm_v2f){xw[3].z, xw[3].w}, (zm_v2f){Y.w, Y.w}, hi);
                bg[0] = lo.x; bg[1] = lo.y; bg[2] = hi.x; bg[3] = hi.y;
            }
            float2 p[4];
            if (has_w) {
#pragma unroll
                for (int e = 0; e < 4; ++e) p[e] = prep_pixel(v[e], w[e], true, bg[e], vs, fwth);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) p[e] = prep_pixel(v[e], 1.f, false, bg[e], vs, fwth);
            }
            if (!FAST) {
                const int r = FO_RPC * k + lrow;
                const bool ok = (unsigned)(by0 + r) < (unsigned)ny && gx >= 0 && gx + 4 <= nx;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    p[e].x = ok ? p[e].x : 0.f;
                    p[e].y = ok ? p[e].y : ZM_BIGVAR;
                }
            }
            // (every raw quad of chunk k was read before this point - one wave, LDS in order, the values are in
            // v / w - so its pairs may land on the raw bytes of other lanes; the next chunk is another 1920 bytes)
            asm volatile("" ::: "memory");
            if (lrow < FO_RPC) {
                float4* d = reinterpret_cast<float4*>(SL + k * FO_CHB) + 2 * ln;
                d[0] = make_float4(p[0].x, p[0].y, p[1].x, p[1].y);
                d[1] = make_float4(p[2].x, p[2].y, p[3].x, p[3].y);
            }
        }
    };
    // frames that could not be staged raw arrive prepped (zm_ff.src): pairs as they are, fill at the frame edge
    auto prep_src = [&](const prep_sc& P, int sl, bool fast) __attribute__((always_inline)) {
        const int bx0 = P.bx0, by0 = P.by0, bh = P.bh;
        const int ny = P.ny, sp = P.spitch;
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int lrow = (ln * 205) >> 12, lcol = ln - lrow * FO_PQ;
        const int gx = bx0 + 4 * lcol;
        char* SL = smem + FO_OFF_SLOT + sl * FO_SLOT;
#pragma unroll 1
        for (int j = 0; j < ownn; ++j) {
            const int k = own0 + owns * j;
            if (FO_RPC * k >= bh) break;
            const float4 a = reinterpret_cast<const float4*>(SL + k * FO_CHB)[ln];
            const float4 b = reinterpret_cast<const float4*>(SL + k * FO_CHB + FO_CHB / 2)[ln];
            const int r = FO_RPC * k + lrow;
            const bool rowok = fast || (unsigned)(by0 + r) < (unsigned)ny;
            const bool cpa = fast || (gx >= 0 && gx <= sp - 2), cpb = fast || (gx + 2 >= 0 && gx + 2 <= sp - 2);
            const bool oka = rowok && cpa, okb = rowok && cpb;
            asm volatile("" ::: "memory");
            if (lrow < FO_RPC) {
                float4* d = reinterpret_cast<float4*>(SL + k * FO_CHB) + 2 * ln;
                d[0] = make_float4(oka ? a.x : 0.f, oka ? a.y : ZM_BIGVAR, oka ? a.z : 0.f, oka ? a.w : ZM_BIGVAR);
                d[1] = make_float4(okb ? b.x : 0.f, okb ? b.y : ZM_BIGVAR, okb ? b.z : 0.f, okb ? b.w : ZM_BIGVAR);
            }
        }
    };
    // (the scalars of the prep in one batch, like those of the DMA issue - requested BEFORE the wave waits for its DMA)
    auto prep_load = [&](int tt, int ff, prep_sc& P) __attribute__((always_inline)) {
        const auto H = hdr_g(tt, ff);
        const zm_ff* F = fr + ff;
        P.use_lds = H->use_lds; P.fast = H->fast; P.bx0 = H->bx0; P.by0 = H->by0; P.bh = H->bh; P.xb = H->xb;
        P.vscale = H->vscale;
        P.nx = F->nx; P.ny = F->ny; P.spitch = F->spitch; P.wthresh = F->wthresh;
        P.src = F->src; P.wgt = F->wgt; P.ytab = F->ytab;
    };
#define FO_PIN_PREP(P) asm volatile("; prep scalars" : : "s"(P.use_lds), "s"(P.fast), "s"(P.bx0), "s"(P.by0), "s"(P.bh), "s"(P.xb), \
        "s"(P.vscale), "s"(P.nx), "s"(P.ny), "s"(P.spitch), "s"(P.wthresh), "s"(P.src), "s"(P.wgt), "s"(P.ytab))
    auto prep = [&](const prep_sc& P, int sl, int tb) __attribute__((always_inline)) {
        if (!P.use_lds || (dbg & 2) || ownn == 0) return;
        if (P.src) prep_src(P, sl, P.fast != 0);
        else if (P.fast) prep_raw(P, sl, tb, std::true_type{});
        else prep_raw(P, sl, tb, std::false_type{});
    };
    const int nty = ntiles / ntx;
    // queue position -> tile: the top and bottom rows of tiles (edge items: the slow ones) go first
    auto tile_of = [&](int s) -> int {
        if (s >= ntiles) return s;
        const int r = s / ntx, c = s - r * ntx;
        return (r == 0 ? 0 : r == 1 ? nty - 1 : r - 1) * ntx + c;
    };
    auto next_item = [&](int& tt, int& ff, int& kk) {
        if (++ff == nfr) { ff = 0; ++kk; tt = tring[kk & 3]; }
    };
    auto hdr_word = [&](int tt, int ff) -> int {          // this thread's word of the header of item (tt, ff)
        return tid < FF_HDR_WORDS ? ghdr[((size_t)tt * nfr + ff) * FF_HDR_WORDS + tid] : 0;
    };
    auto hdr_put = [&](int sl, int wd) {
        if (tid < FF_HDR_WORDS) reinterpret_cast<int*>(&HR[sl])[tid] = wd;
    };
    // The staging reads the scalar fields of an item's header (box, flags, variance scale, mesh columns) straight
    // from the header array in global memory: wave-uniform addresses, i.e. scalar loads through the constant cache -
    // the words were fetched into the L2 by hdr_word iterations ago.  From the LDS copy every field is a ds_read
    // into a vector register and a v_readfirstlane back: ~20 vector-pipe instructions per wave and item for
    // values the scalar unit can fetch by itself.  (The LDS copy stays for what lanes index: the lattice nodes.)

    if ((int)blockIdx.x >= ntiles) return;
    int t0 = tile_of(blockIdx.x), f0 = 0, k3 = 0;
    for (int e = tid; e < LZ_FLOATS / 4; e += FD_THREADS)
        reinterpret_cast<float4*>(smem + FO_OFF_TAB)[e] = reinterpret_cast<const float4*>(taptab)[e];
    if (tid == 0) {
        // the look-ahead of three items spans 3 / nfr further tiles at the start
        tring[0] = t0;
        for (int o = 1; o <= 3 / nfr; ++o) tring[o] = tile_of(atomicAdd(tilectr, 1));
    }
    __syncthreads();
    int t1 = t0, f1 = f0;
    next_item(t1, f1, k3);
    int t2 = t1, f2 = f1;
    next_item(t2, f2, k3);
    int t3 = t2, f3 = f2;
    next_item(t3, f3, k3);
    hdr_put(0, hdr_word(t0, f0));
    if (t1 < ntiles) hdr_put(1, hdr_word(t1, f1));
    if (t2 < ntiles) hdr_put(2, hdr_word(t2, f2));
    __syncthreads();
    {
        dma_sc D0;
        tab_sc T0, T1;
        load_dma_sc(t0, f0, D0);
        load_tab_sc(t0, f0, T0);
        load_tab_sc(t1 < ntiles ? t1 : t0, t1 < ntiles ? f1 : f0, T1);
        FO_PIN_DMA(D0);
        dma_tabs(T0, 0);
        if (t1 < ntiles) dma_tabs(T1, 1);
        dma_item(D0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {
        prep_sc P0;
        prep_load(t0, f0, P0);
        FO_PIN_PREP(P0);
        prep(P0, 0, 0);
    }
    __syncthreads();

    // ---- this thread's pixels: column tx, rows 4 wv .. 4 wv + 3 of the 64 x 32 tile (one group)
    const int tx = lane;
    const int cr = wv >> 2;
    const int cell = tx >> 4;
    const float fx = (float)(tx & 15) * (1.f / LSTEP);
    const float fyb = (float)((4 * wv) & 15) * (1.f / LSTEP);
    float S1[NPX], S0[NPX], SW[NPX];
    int32_t MK[NPX];
#pragma unroll
    for (int q = 0; q < NPX; ++q) { S1[q] = 0.f; S0[q] = 0.f; SW[q] = 0.f; MK[q] = -1; }

    // STACK: the samples of item (pt, pfr), held in S1 / S0, to plane pfr of the stack
    int pt = -1, pfr = 0;
    auto flush = [&]() {
        if (pt < 0) return;
        const int ptyi = pt / ntx, ptxi = pt - ptyi * ntx;
        const int pox = ptxi * TW + tx, poy0 = ptyi * RTH + wv * NPX;
        float2* plane = FF_KARG(stack) + (size_t)pfr * (size_t)FF_KARG(fstride);
#pragma unroll
        for (int q = 0; q < NPX; ++q) {
            const int oy = poy0 + q;
            if (pox < onx && oy < ony)
                __builtin_nontemporal_store((zm_v2f){S1[q], S0[q]}, reinterpret_cast<zm_v2f*>(plane + (size_t)oy * onx + pox));
        }
    };
    long long ptk[5] = {0, 0, 0, 0, 0}, tc = 0;      // developer (ZM_FF_PROF=1): shader-clock sums per phase of this wave
#define FO_TICK(k) do { if (DEV && prof) { const long long t_ = __builtin_amdgcn_s_memtime(); ptk[k] += t_ - tc; tc = t_; } } while (0)
    if (DEV && prof) tc = __builtin_amdgcn_s_memtime();
    const int budget = dbg >> 8;
    int ngrab = 0;
    int hs = 0, sl = 0;
    for (;;) {
        const ff_hdr* H = &HR[hs];
        const int h1 = (hs + 1) & 3, h2 = (hs + 2) & 3, h3 = (hs + 3) & 3;
        const zm_ff* F = fr + f0;
        const bool use_lds = H->use_lds, touches = H->touches, fast = H->fast;
        const tile_hdr3* SH = &H->sub[0];
        const int sbx0 = SH->bx0, sby0 = SH->by0;
        const int mx0 = sbx0 & ~7;                                    // origin of the box-OR tile
        const float2* tile = reinterpret_cast<const float2*>(smem + FO_OFF_SLOT + sl * FO_SLOT);
        const uint16_t* mtile = reinterpret_cast<const uint16_t*>(smem + FO_OFF_MSK + sl * FO_MSLOT);
        const int tyi = t0 / ntx, txi = t0 - tyi * ntx;
        const int ox0 = txi * TW, oy0 = tyi * RTH + wv * NPX;
        const int ox = ox0 + tx;
        if (STACK) {
            flush();
#pragma unroll
            for (int q = 0; q < NPX; ++q) { S1[q] = 0.f; S0[q] = 0.f; }
            pt = t0;
            pfr = f0;
        }
        int hw3 = 0;
        if (t3 < ntiles) hw3 = hdr_word(t3, f3);
        const bool grab = f3 == nfr - 1;
        int gnext = 0;
        if (grab) {
            // a tile budget (yield mode: the workgroup retires after `budget` tiles and leaves its CU slot to
            // whatever else is queued on the GPU; later workgroups of the launch carry on)
            const bool allowed = budget == 0 || ngrab + 1 < budget;
            if (tid == 0) gnext = allowed ? atomicAdd(tilectr, 1) : ntiles;
            ++ngrab;
        }
        const bool more = t1 < ntiles;
        // the raw planes and the box-OR tile of the next item into the other slot (free since the last barrier);
        // the tables of the item after it into the table buffer the prep of THIS item used
        if (prio == 1 || prio == 3) __builtin_amdgcn_s_setprio(1);
        {
            // (an item that does not exist: the scalars of the current one, valid and unused)
            dma_sc D1;
            tab_sc T2;
            const bool more2 = t2 < ntiles;
            load_dma_sc(more ? t1 : t0, more ? f1 : f0, D1);
            if (tabs_wave) load_tab_sc(more2 ? t2 : t0, more2 ? f2 : f0, T2);
            else T2 = tab_sc{0, 0, 0, 0, 0, 0, 0, 0, 0, nullptr, nullptr, nullptr};
            FO_PIN_DMA(D1);
            FO_PIN_TAB(T2);
            if (more) dma_item(D1, sl ^ 1);
            if (more2) dma_tabs(T2, sl);
        }
        if (prio == 1 || prio == 3) __builtin_amdgcn_s_setprio(0);
        FO_TICK(0);

        const bool do_px = touches && !(dbg & 1);
        if (prio == 4 && wv >= 4) __builtin_amdgcn_s_setprio(1);
        {
            // x part of the bilinear lattice interpolation, once per item (k_resample's operations)
            const float x0a = SH->nrel[cr][cell][0], x1a = SH->nrel[cr][cell + 1][0];
            const float y0a = SH->nrel[cr][cell][1], y1a = SH->nrel[cr][cell + 1][1];
            const float x0b = SH->nrel[cr + 1][cell][0], x1b = SH->nrel[cr + 1][cell + 1][0];
            const float y0b = SH->nrel[cr + 1][cell][1], y1b = SH->nrel[cr + 1][cell + 1][1];
            const float xa = __builtin_fmaf(fx, x1a - x0a, x0a), ya = __builtin_fmaf(fx, y1a - y0a, y0a);
            const float xb = __builtin_fmaf(fx, x1b - x0b, x0b), yb = __builtin_fmaf(fx, y1b - y0b, y0b);
            const float xd = xb - xa, yd = yb - ya;
            const bool with_mask = MOP && F->mask != nullptr;
            unsigned slow = !do_px ? 0u : use_lds ? 0u : 0xfu;
            const bool any_raw = MOP && with_mask && use_lds && H->frame_raw != 0;
            const float fscale = F->fscale, fscale2 = F->fscale2;
            // (32-bit LDS addresses of the window origin (0, 0) and of its box-OR entry)
            const unsigned tbase = (unsigned)(size_t)tile + 8u * (unsigned)(OFF * FO_P + OFF);
            const uint16_t* mbase = mtile + (OFF * FO_MP + OFF + (sbx0 - mx0));
            const int enx = F->nx, eny = F->ny;
            const bool EDGE = !fast;
            auto group = [&]() __attribute__((always_inline)) {
                asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_setprio 1\n1:" : : "s"(ysw) : "scc");
                // Shape of the group (round 5, late: 45 -> 25 vector instructions).  The four pixels share the fast
                // path when they sit in one source column, in four consecutive source rows, none within ZM_SNAP of a
                // sample (delta taps: the generic code).  Positions are rounded FMAs of one linear function of the row
                // fraction - monotone - so equal column floors of pixels 0 and 3 hold for 1 and 2; for the rows,
                // floor(py_3) = floor(py_0) + 3 together with every fraction inside [SNAP, 1 - 2 SNAP] pins the two
                // in between (the exact values are collinear and a rounded one differs from the exact one by less than
                // SNAP: box coordinates stay below 128).  The fractions of pixels 1 and 2 are v_fract (= x - floor(x),
                // exact for these positive values); the test is conservative - who fails it takes the generic code,
                // which gives the same bits.
                float dxs[4], dys[4], pxs[4], pys[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float fy = fyb;
                    if (j == 1) asm volatile("v_add_f32 %0, 0x3d800000, %1" : "=v"(fy) : "v"(fyb));       // + 1 / 16
                    if (j == 2) asm volatile("v_add_f32 %0, 0x3e000000, %1" : "=v"(fy) : "v"(fyb));       // + 2 / 16
                    if (j == 3) asm volatile("v_add_f32 %0, 0x3e400000, %1" : "=v"(fy) : "v"(fyb));       // + 3 / 16
                    pxs[j] = __builtin_fmaf(fy, xd, xa);
                    pys[j] = __builtin_fmaf(fy, yd, ya);
                }
                static_assert(LSTEP == 16, "row fractions of the group: sixteenths");
                const float fxf0 = floorf(pxs[0]), fyf0 = floorf(pys[0]), fxf3 = floorf(pxs[3]), fyf3 = floorf(pys[3]);
                dxs[0] = pxs[0] - fxf0; dys[0] = pys[0] - fyf0;
                dxs[3] = pxs[3] - fxf3; dys[3] = pys[3] - fyf3;
                dxs[1] = __builtin_amdgcn_fractf(pxs[1]); dys[1] = __builtin_amdgcn_fractf(pys[1]);
                dxs[2] = __builtin_amdgcn_fractf(pxs[2]); dys[2] = __builtin_amdgcn_fractf(pys[2]);
                const float dlo = fminf(__builtin_fminf(__builtin_fminf(dxs[0], dys[0]), __builtin_fminf(dxs[1], dys[1])),
                                        __builtin_fminf(__builtin_fminf(dxs[2], dys[2]), __builtin_fminf(dxs[3], dys[3])));
                const float dhi = fmaxf(__builtin_fmaxf(__builtin_fmaxf(dxs[0], dys[0]), __builtin_fmaxf(dxs[1], dys[1])),
                                        __builtin_fmaxf(__builtin_fmaxf(dxs[2], dys[2]), __builtin_fmaxf(dxs[3], dys[3])));
                const bool shape = dlo >= ZM_SNAP && dhi <= 1.f - 2.f * ZM_SNAP && fxf3 == fxf0 && fyf3 == fyf0 + 3.f;
                if (!__all(shape)) {
                    slow |= 0xfu;
                    return;
                }
                const int ix0 = (int)fxf0, iy0 = (int)fyf0;
                // outm: pixels of the group whose box-OR entry does not count (no mask, or the footprint leaves the frame)
                unsigned outm = with_mask ? 0u : 0xfu;
                if (EDGE && MOP) {
                    const int ix = sbx0 + OFF + ix0, iy = sby0 + OFF + iy0;
                    const bool xin = ix >= 0 && ix + NT <= enx && ox < onx;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        outm |= (xin && iy + j >= 0 && iy + j + NT <= eny && oy0 + j < ony) ? 0u : (1u << j);
                }
                int32_t mterm[4] = {-1, -1, -1, -1};
                if (MOP) {
                    const int lom = __mul24(iy0, FO_MP) + ix0;
                    uint32_t m16[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) m16[j] = mbase[lom + j * FO_MP];
                    if (any_raw) {
                        bool defer = false;
#pragma unroll
                        for (int j = 0; j < 4; ++j) defer |= m16[j] == ZM_BOX_RAW && !((outm >> j) & 1u);
                        if (__any(defer)) {
                            slow |= 0xfu;
                            return;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)      // (the term, or -1 = "no vote" where the entry does not count)
                        mterm[j] = ff_mask_term<MOP>((int32_t)m16[j]) | __builtin_amdgcn_sbfe(outm, j, 1);
                }
                zm_v2f txp[4][3], typ[4][3];
                {
                    lz3_node nd;
                    float dl;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        lz3_issue(ltab, (i & 1) ? dys[i >> 1] : dxs[i >> 1], nd, dl);
                        lz3_wait<0>(nd);
                        if (i & 1) lz3_eval(nd, dl, typ[i >> 1]);
                        else lz3_eval(nd, dl, txp[i >> 1]);
                    }
                }
                // (NOTHING conditional in C++ may sit inside the two hand-counted loops: a scalar load the compiler
                // sinks into them - a kernel argument behind a condition - counts in lgkmcnt, returns out of order, and
                // lets lds_wait_n<6> pass early: wrong window rows for the group's last pixel, found the hard way.
                // ff_setprio_when is one opaque statement on a pinned scalar register.)
                ff_setprio_when<1, 0>(ysw);
                zm_v2f av[4];
                lds_row6 ra, rb;
                const unsigned pa = tbase + 8u * (unsigned)(__mul24(iy0, FO_P) + ix0);
                asm volatile("; ZM_LGKM_BEGIN" ::: "memory");   // (tests/test_isa_lint.py: no compiler-made lgkm operation up to ZM_LGKM_END)
                lds_issue6_row<0>(pa, ra);
                zm_static_for<0, NT + 3>([&](auto rho_c) __attribute__((always_inline)) {
                    constexpr int rho = decltype(rho_c)::value;
                    if constexpr (rho == 2 || rho == 4 || rho == 6 || rho == 8) ff_setprio_when<rho / 2 + 1, 0>(ysw);
                    lds_row6& cur = (rho & 1) ? rb : ra;
                    lds_row6& nxt = (rho & 1) ? ra : rb;
                    if constexpr (rho + 1 < NT + 3) {
                        lds_issue6_row<rho + 1>(pa, nxt);
                        lds_wait_n<6>(cur);
                    } else {
                        lds_wait_n<0>(cur);
                    }
                    const unsigned long long rr[NT] = {cur.r0, cur.r1, cur.r2, cur.r3, cur.r4, cur.r5};
                    // (two pixels' row sums side by side: back to back, a packed FMA that takes the high half of its
                    // first operand for both lanes is followed by a compiler-made s_nop before the FMA that reads its
                    // result - 31 of them per group; per pixel the operations and their order are unchanged)
#pragma unroll
                    for (int jp = 0; jp < 4; jp += 2) {
                        const int ra_ = rho - jp, rb_ = rho - jp - 1;
                        const bool oa = ra_ >= 0 && ra_ < NT, ob = rb_ >= 0 && rb_ < NT;
                        if (!oa && !ob) continue;
                        zm_v2f rva = (zm_v2f){0.f, 0.f}, rvb = (zm_v2f){0.f, 0.f};
#pragma unroll
                        for (int c = 0; c < NT; ++c) {
                            if (oa) {
                                const float tc = (c & 1) ? txp[jp][c >> 1].y : txp[jp][c >> 1].x;
                                rva = __builtin_elementwise_fma((zm_v2f){tc, tc}, lds_pair(rr[c]), rva);
                            }
                            if (ob) {
                                const float tc = (c & 1) ? txp[jp + 1][c >> 1].y : txp[jp + 1][c >> 1].x;
                                rvb = __builtin_elementwise_fma((zm_v2f){tc, tc}, lds_pair(rr[c]), rvb);
                            }
                        }
                        if (oa) {
                            const float tr = (ra_ & 1) ? typ[jp][ra_ >> 1].y : typ[jp][ra_ >> 1].x;
                            av[jp] = __builtin_elementwise_fma((zm_v2f){tr, tr}, rva, ra_ == 0 ? (zm_v2f){0.f, 0.f} : av[jp]);
                        }
                        if (ob) {
                            const float tr = (rb_ & 1) ? typ[jp + 1][rb_ >> 1].y : typ[jp + 1][rb_ >> 1].x;
                            av[jp + 1] = __builtin_elementwise_fma((zm_v2f){tr, tr}, rvb, rb_ == 0 ? (zm_v2f){0.f, 0.f} : av[jp + 1]);
                        }
                    }
                });
                asm volatile("; ZM_LGKM_END" ::: "memory");
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float vacc = av[j].y;
                    const bool ok = vacc > 0.f && vacc < ZM_BADVAR_TEST;
                    const zm_v2f sc = av[j] * (zm_v2f){fscale, fscale2};        // (one packed multiplication: the same two products)
                    const float v = ok ? sc.x : 0.f;
                    const float w = ok ? __builtin_amdgcn_rcpf(sc.y) : 0.f;
                    const float ww = AVG ? (w > 0.f ? 1.f : 0.f) : w;
                    if (STACK) {
                        S1[j] = v;
                        S0[j] = w;
                    } else {
                        S1[j] = fmaf(ww, v, S1[j]);
                        S0[j] += ww;
                    }
                    if (AVG) SW[j] += w;
                    if (MOP) MK[j] &= mterm[j];
                }
            };
            if (do_px && use_lds) group();
            if (DEV && (dbg & 24)) {
                // developer (ZM_FF_DBG bits 8 / 16): 64 / 128 extra independent FMAs per wave and item - does the
                // launch grow by their issue time (the vector pipe is the bound) or not (latency is)?
                float e0 = fx, e1 = fyb, e2 = fx + 1.f, e3 = fyb + 1.f;
                const int nrep = (dbg & 16) ? 32 : 16;
#pragma unroll 1
                for (int r = 0; r < nrep; ++r)
                    asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3"
                                 : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3));
                if (e0 + e1 + e2 + e3 == 12345.678f) S1[0] += 1.f;
            }
            // the generic code, once: delta kernels, windows of another shape, footprints beyond the LDS tile
#pragma unroll 1
            while (slow) {
                const int q = __builtin_ctz(slow);
                slow &= slow - 1;
                const int oy = oy0 + q;
                if (ox >= onx || oy >= ony) continue;
                const float fy = fyb + (float)q * (1.f / LSTEP);
                const float px = __builtin_fmaf(fy, xd, xa), py = __builtin_fmaf(fy, yd, ya);
                const ff_px r = ff_generic_pixel<MOP>(F, tile, ltab, use_lds, touches, sbx0, sby0, FO_P, px, py);
                const float ww = AVG ? (r.w > 0.f ? 1.f : 0.f) : r.w;
#pragma unroll
                for (int k = 0; k < NPX; ++k) {
                    const bool me = (k == q);
                    S1[k] = me ? (STACK ? r.v : fmaf(ww, r.v, S1[k])) : S1[k];
                    S0[k] = me ? (STACK ? r.w : S0[k] + ww) : S0[k];
                    if (AVG) SW[k] = me ? SW[k] + r.w : SW[k];
                    if (MOP) MK[k] = (me && with_mask && r.inb) ? ff_mask_fold<MOP>(MK[k], r.m) : MK[k];
                }
            }
        }

        if (f0 == nfr - 1) {
            // the tile is complete: coadd (or partial sums) and mask coadd, once
            float* const out_img = FF_KARG(out_img);
            float* const out_wgt = FF_KARG(out_wgt);
            int32_t* const out_mask = FF_KARG(out_mask);
            float* const out_cov = FF_KARG(out_cov);
            const int partial = FF_KARG(partial);
#pragma unroll
            for (int q = 0; q < NPX; ++q) {
                const int oy = oy0 + q;
                if (ox < onx && oy < ony) {
                    const size_t o = (size_t)oy * onx + ox;
                    const float s1 = S1[q], s0 = S0[q];
                    if (STACK) {
                    } else if (partial) {
                        out_img[o] = s1;
                        out_wgt[o] = s0;
                    } else {
                        out_img[o] = s0 > 0.f ? s1 / s0 : 0.f;
                        out_wgt[o] = AVG ? SW[q] : s0;
                    }
                    if (MOP) {
                        const int32_t a = ff_mask_result<MOP>(MK[q]);
                        if (partial) {
                            out_mask[o] = a;
                        } else {
                            out_mask[o] = a == -1 ? 0 : a;
                            if (out_cov) out_cov[o] = a == -1 ? 0.f : 1.f;
                        }
                    }
                }
                if (!STACK) { S1[q] = 0.f; S0[q] = 0.f; }
                SW[q] = 0.f; MK[q] = -1;
            }
        }
        if (prio == 4 && wv >= 4) __builtin_amdgcn_s_setprio(0);
        FO_TICK(1);
        prep_sc P1;
        prep_load(more ? t1 : t0, more ? f1 : f0, P1);
        FO_PIN_PREP(P1);
        // this wave's DMA has landed: its chunks of the next item are prepped where they lie
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        FO_TICK(2);
        if (prio == 2 || prio == 3) __builtin_amdgcn_s_setprio(1);
        if (more) prep(P1, sl ^ 1, sl ^ 1);
        if (t3 < ntiles) hdr_put(h3, hw3);
        if (grab && tid == 0) tring[(k3 + 1) & 3] = tile_of(gnext);
        if (prio == 2 || prio == 3) __builtin_amdgcn_s_setprio(0);
        FO_TICK(4);
        // the one rendezvous of an item: the next item's slot, box-OR tile and tables are complete, this item's
        // slot is free
        __syncthreads();
        FO_TICK(3);
        t0 = t1; f0 = f1;
        t1 = t2; f1 = f2;
        t2 = t3; f2 = f3;
        next_item(t3, f3, k3);
        hs = h1;
        sl ^= 1;
        if (t0 >= ntiles) break;
    }
    if (STACK) flush();
    if (DEV && prof && lane == 0)
        for (int k = 0; k < 5; ++k) prof[((size_t)blockIdx.x * NW + wv) * 5 + k] = ptk[k];
#undef FO_TICK
#undef FO_PIN_DMA
#undef FO_PIN_TAB
#undef FO_PIN_PREP
}

// ZM_FF_DMA=0: the register-staged kernel (developer: A / B); default: the DMA-staged one
static bool ff_use_dma() {
    const char* e = getenv("ZM_FF_DMA");
    return !(e && e[0] == '0') && FF_TALL == 0;
}
void zm_fused_geometry(int* tile_h, int* lds_cap) {
    *tile_h = FT_H;
    *lds_cap = ff_use_dma() ? FD_LDS_CAP : FF_LDS_CAP;
}

// frames: nfr descriptors on the host (device pointers inside); out_mask may be NULL (no mask coadd)
// geometry of a fused launch: LDS tile, grid, yield budget (shared by the early header pass and the launch)
struct ff_geom {
    bool use_dma, own;
    int lds_elems, ntx, ntiles, G, budget;
    size_t shmem;
};
// ZM_FF_FORM=dma: k_coadd_fused_dma also where the owner-staged kernel would run (developer: A / B)
static bool ff_use_own() {
    const char* e = getenv("ZM_FF_FORM");
    return ff_use_dma() && !(e && !strcmp(e, "dma"));
}
// fits_own: every frame's planned footprint fits the fixed slot of k_coadd_fused_own (fused_prepare's verdict)
static int ff_geometry(zm_ctx* ctx, int onx, int ony, int lds_elems, bool fits_own, ff_geom* g) {
    g->ntx = zm_div_up(onx, TW);
    g->ntiles = g->ntx * zm_div_up(ony, FT_H);
    g->use_dma = ff_use_dma();
    g->own = fits_own && ff_use_own();
    lds_elems = std::min(std::max(lds_elems, 64), g->use_dma ? FD_LDS_CAP : FF_LDS_CAP);
    g->lds_elems = (lds_elems + 7) & ~7;
    g->shmem = g->own ? (size_t)FO_LDS
               : g->use_dma ? (size_t)FD_OFF_RAW + 20 * (size_t)g->lds_elems + 4 * 8 * FD_YROWS
                          : (size_t)FF_LDS_HDR + FF_LDS_TAB + 2 * (size_t)g->lds_elems * (sizeof(float2) + sizeof(uint16_t));
    ZM_CHECK(g->shmem <= 160 * 1024 / FF_WG_PER_CU, "zm_launch_coadd_fused: LDS tile of %zu bytes", g->shmem);
    // persistent grid: FF_WG_PER_CU workgroups per CU (what their LDS tiles leave room for), each starting
    // on the tile of its index and taking further tiles from a queue (a counter behind the item
    // headers, set to G by k_ff_headers)
    int ncu = 256;
    ZM_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device));
    g->G = std::min(g->ntiles, std::max(ncu, 1) * FF_WG_PER_CU);
    // yield mode (a context that shares the GPU: zm_ctx_set_share >= 2, or ZM_FF_YIELD = tiles per workgroup):
    // more workgroups than fit, each retiring after a few tiles, so that the kernels of other streams get CU
    // slots while this launch runs (the persistent form holds every slot for its whole 2.4 ms)
    g->budget = 0;
    if (g->use_dma) {
        const char* ye = getenv("ZM_FF_YIELD");
        g->budget = ye ? atoi(ye) : (ctx->share >= 2 ? 2 : 0);
        if (g->budget > 0 && (g->ntiles + g->budget - 1) / g->budget > g->G) g->G = (g->ntiles + g->budget - 1) / g->budget;
        else g->budget = 0;
    }
    return 0;
}

// descriptors to the device + the item headers, on stream `s`; skip_vscale: see k_ff_vscale
static int ff_upload_and_headers(zm_ctx* ctx, const zm_ff* frames_host, int nfr, int lnx, int lny, int onx, int ony,
                                 const ff_geom& g, hipStream_t s, int skip_vscale, zm_ff** dev_out, int** ghdr_out) {
    // descriptors: pinned staging guarded by an event (a later call must not overwrite a copy in flight)
    zm_ff *pin = nullptr, *dev = nullptr;
    int* ghdr = nullptr;
    hipEvent_t* ev = nullptr;
    ZM_TRY(zm_get_sync_events(ctx, 10, &ev));
    ZM_HIP(hipEventSynchronize(ev[5]));
    ZM_TRY(ctx->get_pinned("ff_frames_h", sizeof(zm_ff) * (size_t)nfr, (void**)&pin));
    ZM_TRY(ctx->get("ff_frames", sizeof(zm_ff) * (size_t)nfr, (void**)&dev));
    ZM_TRY(ctx->get("ff_headers", sizeof(int) * (FF_HDR_WORDS * (size_t)g.ntiles * nfr + 16), (void**)&ghdr));
    int* tilectr = ghdr + FF_HDR_WORDS * (size_t)g.ntiles * nfr;
    memcpy(pin, frames_host, sizeof(zm_ff) * (size_t)nfr);
    ZM_HIP(hipMemcpyAsync(dev, pin, sizeof(zm_ff) * (size_t)nfr, hipMemcpyHostToDevice, s));
    ZM_HIP(hipEventRecord(ev[5], s));
    {
        // (its own scope: `coadd_fused` times the roofline kernel alone, as the kernel trace does)
        zm_scope_timer th(ctx, "ff_headers");
        const long long items = (long long)g.ntiles * nfr;
        hipLaunchKernelGGL(k_ff_headers, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, s, dev, nfr,
                           lnx, lny, onx, ony, g.lds_elems, g.own ? 2 : g.use_dma ? 1 : 0, g.ntx, g.ntiles, ghdr, tilectr, g.G, skip_vscale);
    }
    ZM_HIP(hipGetLastError());
    *dev_out = dev;
    *ghdr_out = ghdr;
    return 0;
}

// Round 4: the item headers of a fused coadd built EARLY, on the context's second stream, beside the mesh
// statistics (they need the lattices and the box-OR flags - both made on that stream just before - but nothing of
// the background chain except the variance scales, which k_ff_vscale drops in later): 80 us off the main stream.
// Call when the descriptors are final; zm_launch_coadd_fused then finds the headers made (ctx->ff_pre_*).
int zm_launch_fused_headers_early(zm_ctx* ctx, const zm_ff* frames_host, int nfr, int lnx, int lny, int onx, int ony,
                                  int lds_elems, bool fits_own) {
    ctx->ff_pre_valid = false;
    static const bool fork_off = getenv("ZM_FF_FORK") && getenv("ZM_FF_FORK")[0] == '0';
    if (!ctx->aux || ctx->timing || fork_off) return 0;      // (scope timers keep a timed kernel on the main stream)
    ff_geom g;
    ZM_TRY(ff_geometry(ctx, onx, ony, lds_elems, fits_own, &g));
    zm_ff* dev = nullptr;
    int* ghdr = nullptr;
    ZM_TRY(ff_upload_and_headers(ctx, frames_host, nfr, lnx, lny, onx, ony, g, ctx->aux, 1, &dev, &ghdr));
    hipEvent_t* ev = nullptr;
    ZM_TRY(zm_get_sync_events(ctx, 10, &ev));
    ZM_HIP(hipEventRecord(ev[9], ctx->aux));
    ctx->ff_pre_valid = true;
    ctx->ff_pre_nfr = nfr;
    ctx->ff_pre_onx = onx;
    ctx->ff_pre_ony = ony;
    ctx->ff_pre_lds = lds_elems;
    ctx->ff_pre_own = fits_own;
    return 0;
}

int zm_launch_coadd_fused(zm_ctx* ctx, const zm_ff* frames_host, int nfr, int lnx, int lny, int onx, int ony,
                          int lds_elems, int combine, int mask_kind, float* out_img, float* out_wgt,
                          int32_t* out_mask, float* out_cov, int partial, int32_t* unmasked_out,
                          float2* stack, int64_t fstride, bool fits_own) {
    if (unmasked_out) {
        // a mask coadd was asked for but no frame carries a mask: "nothing covered" everywhere
        const size_t opix = (size_t)onx * ony;
        ZM_HIP(hipMemsetAsync(unmasked_out, partial ? 0xFF : 0, sizeof(int32_t) * opix, ctx->stream));
        if (!partial && out_cov) ZM_HIP(hipMemsetAsync(out_cov, 0, sizeof(float) * opix, ctx->stream));
    }
    const int lds_in = lds_elems;
    ff_geom g;
    ZM_TRY(ff_geometry(ctx, onx, ony, lds_elems, fits_own, &g));
    const bool use_dma = g.use_dma, own = g.own;
    lds_elems = g.lds_elems;
    const size_t shmem = g.shmem;
    const int ntx = g.ntx, ntiles = g.ntiles, G = g.G, budget = g.budget;
    const float* taptab = nullptr;
    ZM_TRY(zm_get_lanczos_table(ctx, &taptab));
    zm_ff* dev = nullptr;
    int* ghdr = nullptr;
    const bool pre = ctx->ff_pre_valid && ctx->ff_pre_nfr == nfr && ctx->ff_pre_onx == onx && ctx->ff_pre_ony == ony &&
                     ctx->ff_pre_lds == lds_in && ctx->ff_pre_own == fits_own;
    ctx->ff_pre_valid = false;
    if (pre) {
        // the headers were made on the second stream (zm_launch_fused_headers_early): wait for them, fill in the
        // variance scales
        hipEvent_t* ev = nullptr;
        ZM_TRY(zm_get_sync_events(ctx, 10, &ev));
        ZM_HIP(hipStreamWaitEvent(ctx->stream, ev[9], 0));
        ZM_TRY(ctx->get("ff_frames", sizeof(zm_ff) * (size_t)nfr, (void**)&dev));
        ZM_TRY(ctx->get("ff_headers", sizeof(int) * (FF_HDR_WORDS * (size_t)ntiles * nfr + 16), (void**)&ghdr));
        const long long items = (long long)ntiles * nfr;
        hipLaunchKernelGGL(k_ff_vscale, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, ctx->stream, dev, nfr, items, ghdr);
    } else {
        ZM_TRY(ff_upload_and_headers(ctx, frames_host, nfr, lnx, lny, onx, ony, g, ctx->stream, 0, &dev, &ghdr));
    }
    int* tilectr = ghdr + FF_HDR_WORDS * (size_t)ntiles * nfr;
    const bool avg = combine == ZM_COMBINE_AVERAGE;
    const int mop = out_mask ? (mask_kind == ZM_MASK_AND ? 1 : 2) : 0;
    // ZM_FF_DBG (developer, tools/ff_probe.py): 1 no pixel work, 2 no prep / LDS store, 4 no staging loads
    const int dbg = (getenv("ZM_FF_DBG") ? (atoi(getenv("ZM_FF_DBG")) & 255) : 0) | ((budget & 0xffff) << 8) |
                    (own && getenv("ZM_FF_PRIO") ? ((atoi(getenv("ZM_FF_PRIO")) & 15) << 24) : 0) |
                    (own ? (((getenv("ZM_FF_DEAL") ? atoi(getenv("ZM_FF_DEAL")) : 1) & 3) << 28) : 0);
    long long* prof = nullptr;
    const bool want_prof = getenv("ZM_FF_PROF") && atoi(getenv("ZM_FF_PROF")) != 0;
    const int nwv = (use_dma ? FD_THREADS : FF_THREADS) / 64;
    if (want_prof) ZM_TRY(ctx->get("ff_prof", sizeof(long long) * 5 * nwv * (size_t)G, (void**)&prof));
    zm_scope_timer t(ctx, "coadd_fused");
#define ZM_FF_LAUNCH(MOPV, AVGV, STACKV)                                                                      \
    do {                                                                                                       \
        if (own) {                                                                                             \
            const bool devk = want_prof || (dbg & 255);                                                        \
            auto ko = devk ? k_coadd_fused_own<MOPV, AVGV, STACKV, true> : k_coadd_fused_own<MOPV, AVGV, STACKV, false>; \
            static bool oattr[2][64] = {};                                                                     \
            if (!oattr[devk][ctx->device & 63]) {                                                              \
                ZM_HIP(hipFuncSetAttribute((const void*)ko, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024)); \
                oattr[devk][ctx->device & 63] = true;                                                          \
            }                                                                                                  \
            hipLaunchKernelGGL(ko, dim3(G), dim3(FD_THREADS), shmem, ctx->stream, dev, nfr, onx, ony, lds_elems, \
                               ntx, ntiles, ghdr, out_img, out_wgt, out_mask, out_cov, partial, taptab, tilectr, \
                               stack, (long long)fstride, dbg, prof);                                          \
            break;                                                                                             \
        }                                                                                                      \
        if (use_dma) {                                                                                         \
            const bool devk = want_prof || (dbg & 255);                                                        \
            auto kd = devk ? k_coadd_fused_dma<MOPV, AVGV, STACKV, true> : k_coadd_fused_dma<MOPV, AVGV, STACKV, false>; \
            static bool dattr[2][64] = {};                                                                     \
            if (!dattr[devk][ctx->device & 63]) {                                                              \
                ZM_HIP(hipFuncSetAttribute((const void*)kd, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024)); \
                dattr[devk][ctx->device & 63] = true;                                                          \
            }                                                                                                  \
            hipLaunchKernelGGL(kd, dim3(G), dim3(FD_THREADS), shmem, ctx->stream, dev, nfr, onx, ony, lds_elems, \
                               ntx, ntiles, ghdr, out_img, out_wgt, out_mask, out_cov, partial, taptab, tilectr, \
                               stack, (long long)fstride, dbg, prof);                                          \
            break;                                                                                             \
        }                                                                                                      \
        auto kfn = k_coadd_fused<MOPV, AVGV, STACKV>;                                                          \
        static bool attr_set[64] = {};                                                                         \
        if (!attr_set[ctx->device & 63]) {                                                                     \
            ZM_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            attr_set[ctx->device & 63] = true;                                                                 \
        }                                                                                                      \
        hipLaunchKernelGGL(kfn, dim3(G), dim3(FF_THREADS), shmem, ctx->stream, dev, nfr, onx, ony, lds_elems,  \
                           ntx, ntiles, ghdr, out_img, out_wgt, out_mask, out_cov, partial, taptab, tilectr,   \
                           stack, (long long)fstride, dbg, prof);                                                         \
    } while (0)
    if (stack) {
        if (mop == 0) ZM_FF_LAUNCH(0, false, true);
        else if (mop == 1) ZM_FF_LAUNCH(1, false, true);
        else ZM_FF_LAUNCH(2, false, true);
    }
    else if (mop == 0) { if (avg) ZM_FF_LAUNCH(0, true, false); else ZM_FF_LAUNCH(0, false, false); }
    else if (mop == 1) { if (avg) ZM_FF_LAUNCH(1, true, false); else ZM_FF_LAUNCH(1, false, false); }
    else { if (avg) ZM_FF_LAUNCH(2, true, false); else ZM_FF_LAUNCH(2, false, false); }
#undef ZM_FF_LAUNCH
    ZM_HIP(hipGetLastError());
    if (want_prof) {
        std::vector<long long> h((size_t)5 * nwv * G);
        ZM_HIP(hipMemcpyAsync(h.data(), prof, sizeof(long long) * h.size(), hipMemcpyDeviceToHost, ctx->stream));
        ZM_HIP(hipStreamSynchronize(ctx->stream));
        static const char* nm0[5] = {"issue", "pixels", "loadwait", "store", "barrier"};
        static const char* nm1[5] = {"dma issue", "pixels", "dma wait", "barriers", "prep"};
        const char** nm = use_dma ? nm1 : nm0;
        double sum[5] = {0, 0, 0, 0, 0};
        for (size_t w = 0; w < (size_t)nwv * G; ++w)
            for (int k = 0; k < 5; ++k) sum[k] += (double)h[w * 5 + k];
        fprintf(stderr, "k_coadd_fused phases, mean per wave (kilo-cycles of the shader clock):");
        for (int k = 0; k < 5; ++k) fprintf(stderr, " %s %.1f", nm[k], sum[k] / ((double)nwv * G) * 1e-3);
        fprintf(stderr, "\n");
        if (atoi(getenv("ZM_FF_PROF")) >= 2) {
            // by wave index of the workgroup: which waves the barrier waits for
            for (int w = 0; w < nwv; ++w) {
                double sw[5] = {0, 0, 0, 0, 0};
                for (int b = 0; b < G; ++b)
                    for (int k = 0; k < 5; ++k) sw[k] += (double)h[((size_t)b * nwv + w) * 5 + k];
                fprintf(stderr, "  wave %d:", w);
                for (int k = 0; k < 5; ++k) fprintf(stderr, " %s %.1f", nm[k], sw[k] / G * 1e-3);
                fprintf(stderr, "\n");
            }
        }
    }
    return 0;
}
