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
#include "zm_internal.h"
#include "wcs_math.h"

#define TW 64
#define TH 16
#define LSTEP ZM_LATTICE_STEP
#define HDR_FLOATS 64   // LDS header: 20 node floats + bbox ints, 256 B keeps 16-B alignment

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
__device__ inline float bk_eval(const float* __restrict__ bk, int nbx, int nby, float invmesh,
                                int x, int y) {
    size_t pl = (size_t)nbx * nby;
    float ty = (y + 0.5f) * invmesh - 0.5f;
    float tx = (x + 0.5f) * invmesh - 0.5f;
    int j0 = 0, i0 = 0;
    float dy = 0.f, dx = 0.f;
    if (nby > 1) {
        j0 = min(max((int)floorf(ty), 0), nby - 2);
        dy = ty - j0;
    }
    if (nbx > 1) {
        i0 = min(max((int)floorf(tx), 0), nbx - 2);
        dx = tx - i0;
    }
    int j1 = nby > 1 ? j0 + 1 : j0, i1 = nbx > 1 ? i0 + 1 : i0;
    float dy1 = 1.f - dy, dx1 = 1.f - dx;
    float cdy = dy * dy * dy - dy, cdy1 = dy1 * dy1 * dy1 - dy1;
    float cdx = dx * dx * dx - dx, cdx1 = dx1 * dx1 * dx1 - dx1;
    const float* V = bk;
    const float* DY = bk + pl;
    const float* A = bk + 2 * pl;
    const float* B = bk + 3 * pl;
    int a00 = j0 * nbx + i0, a01 = j0 * nbx + i1, a10 = j1 * nbx + i0, a11 = j1 * nbx + i1;
    float r0 = dy1 * V[a00] + dy * V[a10] + cdy1 * DY[a00] + cdy * DY[a10];
    float r1 = dy1 * V[a01] + dy * V[a11] + cdy1 * DY[a01] + cdy * DY[a11];
    float e0 = dy1 * A[a00] + dy * A[a10] + cdy1 * B[a00] + cdy * B[a10];
    float e1 = dy1 * A[a01] + dy * A[a11] + cdy1 * B[a01] + cdy * B[a11];
    return dx1 * r0 + dx * r1 + cdx1 * e0 + cdx * e1;
}

__global__ __launch_bounds__(256) void k_prep(const float* __restrict__ img,
                                              const float* __restrict__ wgt, int nx, int ny,
                                              const float* __restrict__ bk, int nbx, int nby,
                                              float invmesh,
                                              const float* __restrict__ var_scale_dev,
                                              float wthresh, float2* __restrict__ dst,
                                              int spitch) {
    int xp = blockIdx.x * blockDim.x + threadIdx.x;   // pixel pair index
    int y = blockIdx.y;
    int x = xp * 2;
    if (x >= spitch) return;
    const float var_scale = var_scale_dev ? var_scale_dev[0] : 1.0f;
    float2 o[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        int xx = x + k;
        if (xx < nx) {
            size_t idx = (size_t)y * nx + xx;
            float v = img[idx];
            if (bk) v -= bk_eval(bk, nbx, nby, invmesh, xx, y);
            float var = var_scale;
            if (wgt) {
                float w = wgt[idx];
                var = (w > wthresh) ? var_scale / w : ZM_BIGVAR;
            }
            if (!(v == v)) { v = 0.f; var = ZM_BIGVAR; }   // NaN pixels are bad
            o[k] = make_float2(v, var);
        } else {
            o[k] = make_float2(0.f, ZM_BIGVAR);
        }
    }
    float4* d4 = reinterpret_cast<float4*>(dst + (size_t)y * spitch + x);
    *d4 = make_float4(o[0].x, o[0].y, o[1].x, o[1].y);
}

int zm_launch_prep(zm_ctx* ctx, const float* img, const float* wgt, int nx, int ny,
                   const float* bknodes, int nbx, int nby, int mesh, const float* var_scale_dev,
                   float wthresh, float2* dst, int spitch) {
    dim3 blk(256, 1, 1), grd(zm_div_up(spitch / 2, 256), ny, 1);
    zm_scope_timer t(ctx, "prep");
    hipLaunchKernelGGL(k_prep, grd, blk, 0, ctx->stream, img, wgt, nx, ny, bknodes, nbx, nby,
                       mesh > 0 ? 1.0f / mesh : 0.f, var_scale_dev, wthresh, dst, spitch);
    ZM_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// Unit-sum Lanczos-3 taps for d in [SNAP, 1 - SNAP]; offsets k = -2..3.
// t_k ~ n_k / (d - k)^2 with n = {n1, n2, n3, n1, n2, n3}, the sin recurrence of
// SWarp's make_kernel (sin(a +- 2 pi / 3) expanded), sin/cos by polynomial on
// a = pi d / 3 in [0, pi/3].
__host__ __device__ inline void zm_lanczos3(float d, float t[6]) {
    const float a = d * 1.0471975511965976f;
    const float a2 = a * a;
    float s = a * (1.f + a2 * (-1.6666667e-1f + a2 * (8.3333333e-3f + a2 * (-1.98412698e-4f
              + a2 * (2.7557319e-6f + a2 * -2.5052108e-8f)))));
    float c = 1.f + a2 * (-0.5f + a2 * (4.1666667e-2f + a2 * (-1.3888889e-3f
              + a2 * (2.4801587e-5f + a2 * (-2.7557319e-7f + a2 * 2.0876757e-9f)))));
    float n1 = 0.5f * s - 0.8660254037844386f * c;
    float n2 = 0.5f * s + 0.8660254037844386f * c;
    float n3 = -s;
    float x0 = d + 2.f, x1 = d + 1.f, x2 = d, x3 = d - 1.f, x4 = d - 2.f, x5 = d - 3.f;
#ifdef __HIP_DEVICE_COMPILE__
    float r0 = __builtin_amdgcn_rcpf(x0 * x0), r1 = __builtin_amdgcn_rcpf(x1 * x1),
          r2 = __builtin_amdgcn_rcpf(x2 * x2), r3 = __builtin_amdgcn_rcpf(x3 * x3),
          r4 = __builtin_amdgcn_rcpf(x4 * x4), r5 = __builtin_amdgcn_rcpf(x5 * x5);
#else
    float r0 = 1.f / (x0 * x0), r1 = 1.f / (x1 * x1), r2 = 1.f / (x2 * x2),
          r3 = 1.f / (x3 * x3), r4 = 1.f / (x4 * x4), r5 = 1.f / (x5 * x5);
#endif
    t[0] = n1 * r0; t[1] = n2 * r1; t[2] = n3 * r2;
    t[3] = n1 * r3; t[4] = n2 * r4; t[5] = n3 * r5;
    float sum = ((t[0] + t[1]) + (t[2] + t[3])) + (t[4] + t[5]);
#ifdef __HIP_DEVICE_COMPILE__
    float inv = __builtin_amdgcn_rcpf(sum);
#else
    float inv = 1.f / sum;
#endif
#pragma unroll
    for (int k = 0; k < 6; ++k) t[k] *= inv;
}

extern "C" void zm_debug_lanczos3(float d, float* out6) { zm_lanczos3(d, out6); }

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

template <int KIND>
__device__ inline void make_taps(float d, bool delta, float* t) {
    if (KIND == ZM_RESAMPLE_LANCZOS3) {
        zm_lanczos3(delta ? 0.5f : d, t);
        if (delta) { t[0] = 0.f; t[1] = 0.f; t[2] = 1.f; t[3] = 0.f; t[4] = 0.f; t[5] = 0.f; }
    } else {
        t[0] = 1.f - d;
        t[1] = d;
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

template <int KIND>
__global__ __launch_bounds__(256) void k_resample(const float2* __restrict__ src, int nx, int ny,
                                                  int spitch, const double2* __restrict__ lat,
                                                  int lnx, int lny, float fscale,
                                                  float2* __restrict__ dst, int onx, int ony,
                                                  int lds_cap) {
    extern __shared__ float4 smem4[];
    tile_hdr* hdr = reinterpret_cast<tile_hdr*>(smem4);
    float2* tile = reinterpret_cast<float2*>(smem4) + HDR_FLOATS / 2;
    constexpr int NT = taps_traits<KIND>::N;
    constexpr int OFF = taps_traits<KIND>::OFF;

    const int tid = threadIdx.x;
    const int ox0 = blockIdx.x * TW, oy0 = blockIdx.y * TH;
    if (tid < 64) build_tile_header(lat, lnx, lny, ox0 / LSTEP, oy0 / LSTEP, OFF, OFF + NT - 1, hdr);
    __syncthreads();
    const int bx0 = hdr->bx0, by0 = hdr->by0, bw = hdr->bw, bh = hdr->bh;
    // the footprint may miss the frame entirely: nothing to stage then
    const bool touches = (bx0 < nx) && (bx0 + bw > 0) && (by0 < ny) && (by0 + bh > 0);
    const bool use_lds = touches && ((long long)bw * bh <= (long long)lds_cap);

    if (use_lds) {
        const int wave = tid >> 6, lane = tid & 63;
        const int bw2 = bw >> 1;
        const float4 fill = make_float4(0.f, ZM_BIGVAR, 0.f, ZM_BIGVAR);
        for (int r = wave; r < bh; r += 4) {
            int gy = by0 + r;
            bool rowok = (gy >= 0) && (gy < ny);
            const float2* srow = src + (size_t)(rowok ? gy : 0) * spitch;
            for (int c2 = lane; c2 < bw2; c2 += 64) {
                int gx = bx0 + 2 * c2;
                float4 v = fill;
                if (rowok && gx >= 0 && gx < spitch)
                    v = *reinterpret_cast<const float4*>(srow + gx);
                *reinterpret_cast<float4*>(tile + r * bw + 2 * c2) = v;
            }
        }
    }
    __syncthreads();

    const int tx = tid & 63, tyb = tid >> 6;
    const int ox = ox0 + tx;
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
        const int ty = tyb + 4 * q;
        const int oy = oy0 + ty;
        if (ox >= onx || oy >= ony) continue;
        float px, py;
        tile_position(hdr, tx, ty, &px, &py);
        int ixr, iyr;
        float dx, dy;
        bool ddx, ddy;
        split_pos(px, &ixr, &dx, &ddx);
        split_pos(py, &iyr, &dy, &ddy);
        const int ix = bx0 + ixr + OFF, iy = by0 + iyr + OFF;   // first tap, absolute
        const bool inb = touches && (ix >= 0) && (ix + NT <= nx) && (iy >= 0) && (iy + NT <= ny);
        float2 res = make_float2(0.f, 0.f);
        if (inb) {
            float txw[NT], tyw[NT];
            make_taps<KIND>(dx, ddx, txw);
            make_taps<KIND>(dy, ddy, tyw);
            float acc = 0.f, vacc = 0.f;
            if (use_lds) {
                const float2* p = tile + (iyr + OFF) * bw + (ixr + OFF);
#pragma unroll
                for (int r = 0; r < NT; ++r) {
                    float ra = 0.f, rv = 0.f;
#pragma unroll
                    for (int c = 0; c < NT; ++c) {
                        float2 s = p[c];
                        ra = fmaf(txw[c], s.x, ra);
                        rv = fmaf(txw[c], s.y, rv);
                    }
                    acc = fmaf(tyw[r], ra, acc);
                    vacc = fmaf(tyw[r], rv, vacc);
                    p += bw;
                }
            } else {
                const float2* p = src + (size_t)iy * spitch + ix;
#pragma unroll
                for (int r = 0; r < NT; ++r) {
                    float ra = 0.f, rv = 0.f;
#pragma unroll
                    for (int c = 0; c < NT; ++c) {
                        float2 s = p[c];
                        ra = fmaf(txw[c], s.x, ra);
                        rv = fmaf(txw[c], s.y, rv);
                    }
                    acc = fmaf(tyw[r], ra, acc);
                    vacc = fmaf(tyw[r], rv, vacc);
                    p += spitch;
                }
            }
            if (vacc > 0.f && vacc < ZM_BADVAR_TEST) {
                res.x = acc * fscale;
                res.y = 1.f / (vacc * fscale * fscale);
            }
        }
        dst[(size_t)oy * onx + ox] = res;
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

int zm_launch_resample(zm_ctx* ctx, const float2* src, int nx, int ny, int spitch,
                       const double2* lat, int lnx, int lny, int kernel, float fscale,
                       float2* dst, int onx, int ony, int lds_elems) {
    dim3 blk(256, 1, 1), grd(zm_div_up(onx, TW), zm_div_up(ony, TH), 1);
    size_t shmem = (size_t)HDR_FLOATS * 4 + (size_t)lds_elems * sizeof(float2);
    zm_scope_timer t(ctx, "resample");
    if (kernel == ZM_RESAMPLE_LANCZOS3) {
        hipLaunchKernelGGL(k_resample<ZM_RESAMPLE_LANCZOS3>, grd, blk, shmem, ctx->stream, src, nx,
                           ny, spitch, lat, lnx, lny, fscale, dst, onx, ony, lds_elems);
    } else if (kernel == ZM_RESAMPLE_BILINEAR) {
        hipLaunchKernelGGL(k_resample<ZM_RESAMPLE_BILINEAR>, grd, blk, shmem, ctx->stream, src, nx,
                           ny, spitch, lat, lnx, lny, fscale, dst, onx, ony, lds_elems);
    } else if (kernel == ZM_RESAMPLE_NEAREST) {
        hipLaunchKernelGGL(k_resample_nearest, grd, blk, 0, ctx->stream, src, nx, ny, spitch, lat,
                           lnx, lny, fscale, dst, onx, ony);
    } else {
        zm_set_error("zm_launch_resample: unknown kernel %d", kernel);
        return 2;
    }
    ZM_HIP(hipGetLastError());
    return 0;
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
        if (ix >= 0 && ix + NT <= nx && iy >= 0 && iy + NT <= ny) {
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
