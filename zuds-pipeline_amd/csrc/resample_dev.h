// Device-side pieces shared by the resampling translation units (resample.hip: lattice, prep, k_resample;
// maskbox.hip: box-OR pre-pass; fused_dma.hip / fused_own.hip: the fused resample -> coadd kernels): tile geometry,
// the pinned background-spline arithmetic, the Lanczos-3 taps and their table, positions / tile headers, the
// box-OR entry convention.  Everything here is inline or a template: each translation unit gets its own copy and
// the results are the same bits in all of them (tests/test_fused_coadd_gpu.py, tests/test_configs_gpu.py).
// Operator: zuds/astromatic/makecoadd/default.swarp:42-88; conventions: oracle/resample.py, oracle/background.py.
#pragma once
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
