// SWarp's own edge and mask conventions as OPTIONS of the resampler (round 6; zm_ctx_set_conventions):
//
//   ZM_EDGE_TRUNCATE          an output pixel whose position lies on the input frame is computed from the taps that
//                             are on the frame - the kernel truncated at the edge, nothing renormalised - instead
//                             of getting value 0 / weight 0 when a non-zero tap leaves the frame (the default,
//                             ZM_EDGE_ZERO: DESIGN.md section 2, oracle/resample.py).
//   ZM_MASKRES_LANCZOS_ROUND  an integer mask is interpolated like the image SWarp takes it for and rounded to the
//                             nearest integer (mask.swarp keeps RESAMPLING_TYPE LANCZOS3,
//                             zuds/astromatic/makecoadd/mask.swarp:25, zuds/swarp.py:141-152) instead of the OR of
//                             the mask words under the non-zero taps (the default, ZM_MASKRES_OR).
//
// Both are meant for pixel-for-pixel comparisons with real SWarp products, not for speed: with
// either set a stack takes the materialised path (k_resample frame by frame) and the kernels below run beside it -
// k_resample_rim recomputes the pixels whose footprint comes within a pixel of the frame edge with per-tap bounds
// (everything else is k_resample's: its interior arithmetic and register budget stay untouched), and
// k_resample_mask_opts resamples a mask plane under either mask rule and either edge rule.
// Arithmetic: oracle/resample.py (edge=, mask_resample=); parity: tests/test_conventions_gpu.py.
#include "resample_dev.h"

// the nearest input pixel of a position (relative to the box origin b0) lies on an axis of n pixels
__device__ inline bool pos_on_axis(float p, int b0, int n) {
    const int j = b0 + (int)floorf(p + 0.5f);
    return j >= 0 && j < n;
}

template <int KIND>
__device__ inline void taps_of(const float* __restrict__ ltab, float dx, float dy, bool ddx, bool ddy, float* tx, float* ty) {
    constexpr int NT = taps_traits<KIND>::N;
    if (KIND == ZM_RESAMPLE_LANCZOS3) {
        // k_resample's taps: the tabulated quadratics (here read from the table in global memory)
        zm_v2f txp[3], typ[3];
        zm_lz3_lookup(ltab, ddx ? 0.5f : dx, txp);
        zm_lz3_lookup(ltab, ddy ? 0.5f : dy, typ);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const float dl = (k == 2) ? 1.f : 0.f;
            tx[k] = ddx ? dl : ((k & 1) ? txp[k >> 1].y : txp[k >> 1].x);
            ty[k] = ddy ? dl : ((k & 1) ? typ[k >> 1].y : typ[k >> 1].x);
        }
    } else {
        zm_v2f t[NT];
        make_taps2<KIND>(dx, dy, ddx, ddy, t);
#pragma unroll
        for (int k = 0; k < NT; ++k) { tx[k] = t[k].x; ty[k] = t[k].y; }
    }
}

// mop 0: image only; 1: store the resampled mask into macc; 2: fold it into macc with mkind (-1 = nothing yet)
template <int KIND>
__global__ __launch_bounds__(256) void k_resample_rim(const float2* __restrict__ src, int nx, int ny, int spitch,
                                                      const double2* __restrict__ lat, int lnx, int lny, float fscale,
                                                      const float* __restrict__ ltab, float2* __restrict__ dst,
                                                      float* __restrict__ plane_a, float* __restrict__ plane_b,
                                                      int onx, int ony, const int32_t* __restrict__ mask,
                                                      int32_t* __restrict__ macc, int mop, int mkind) {
    __shared__ tile_hdr hdr;
    constexpr int NT = taps_traits<KIND>::N;
    constexpr int OFF = taps_traits<KIND>::OFF;
    const int tid = threadIdx.x;
    const int ox0 = blockIdx.x * TW, oy0 = blockIdx.y * TH;
    if (tid < 64) build_tile_header(lat, lnx, lny, ox0 / LSTEP, oy0 / LSTEP, OFF, OFF + NT - 1, &hdr);
    __syncthreads();
    // a tile whose whole box lies two pixels inside the frame has no rim pixel
    if (hdr.bx0 >= 2 && hdr.by0 >= 2 && hdr.bx0 + hdr.bw <= nx - 2 && hdr.by0 + hdr.bh <= ny - 2) return;
    const int tx_ = tid & 63, tyb = tid >> 6;
    const int ox = ox0 + tx_;
    const float fscale2 = fscale * fscale;
    for (int q = 0; q < 4; ++q) {
        const int ty_ = tyb + 4 * q, oy = oy0 + ty_;
        if (ox >= onx || oy >= ony) continue;
        float px, py;
        tile_position(&hdr, tx_, ty_, &px, &py);
        int ixr, iyr;
        float dx, dy;
        bool ddx, ddy;
        split_pos(px, &ixr, &dx, &ddx);
        split_pos(py, &iyr, &dy, &ddy);
        const int ix = hdr.bx0 + ixr + OFF, iy = hdr.by0 + iyr + OFF;          // first tap, absolute
        // k_resample owns every pixel whose footprint keeps a pixel's distance from the edge (it may class a
        // footprint that touches the edge either way: its positions are relative to another box origin)
        if (ix >= 1 && iy >= 1 && ix + NT <= nx - 1 && iy + NT <= ny - 1) continue;
        if (!(pos_on_axis(px, hdr.bx0, nx) && pos_on_axis(py, hdr.by0, ny))) continue;   // (stays value 0 / weight 0)
        float tx[NT], ty[NT];
        taps_of<KIND>(ltab, dx, dy, ddx, ddy, tx, ty);
        float acc = 0.f, vacc = 0.f;
        int32_t mres = 0;
        for (int r = 0; r < NT; ++r) {
            const int y = iy + r;
            if (y < 0 || y >= ny || ty[r] == 0.f) continue;
            float ra = 0.f, rv = 0.f;
            for (int c = 0; c < NT; ++c) {
                const int x = ix + c;
                if (x < 0 || x >= nx || tx[c] == 0.f) continue;
                const float2 s = src[(size_t)y * spitch + x];
                ra = fmaf(tx[c], s.x, ra);
                rv = fmaf(tx[c], s.y, rv);
                if (mop) mres |= mask[(size_t)y * nx + x];
            }
            acc = fmaf(ty[r], ra, acc);
            vacc = fmaf(ty[r], rv, vacc);
        }
        float2 res = make_float2(0.f, 0.f);
        if (vacc > 0.f && vacc < ZM_BADVAR_TEST) {
            res.x = acc * fscale;
            res.y = __builtin_amdgcn_rcpf(vacc * fscale2);
        }
        const size_t oidx = (size_t)oy * onx + ox;
        if (plane_a) {
            plane_a[oidx] = res.x;
            plane_b[oidx] = res.y;
        } else {
            dst[oidx] = res;
        }
        if (mop == 1) {
            macc[oidx] = mres;
        } else if (mop == 2) {
            const int32_t a = macc[oidx];
            macc[oidx] = (a == -1) ? mres : ((mkind == ZM_MASK_AND) ? (a & mres) : (a | mres));
        }
    }
}

// A mask plane under the conventions: lanczos = interpolate and round (else OR under the non-zero taps), trunc = edge
// rule.  Uncovered pixels get `fill`.
template <int KIND>
__global__ __launch_bounds__(256) void k_resample_mask_opts(const int32_t* __restrict__ mask, int nx, int ny,
                                                            const double2* __restrict__ lat, int lnx, int lny,
                                                            const float* __restrict__ ltab, int32_t* __restrict__ dst,
                                                            int onx, int ony, int32_t fill, int lanczos, int trunc) {
    __shared__ tile_hdr hdr;
    constexpr int NT = taps_traits<KIND>::N;
    constexpr int OFF = taps_traits<KIND>::OFF;
    const int tid = threadIdx.x;
    const int ox0 = blockIdx.x * TW, oy0 = blockIdx.y * TH;
    if (tid < 64) build_tile_header(lat, lnx, lny, ox0 / LSTEP, oy0 / LSTEP, OFF, OFF + NT - 1, &hdr);
    __syncthreads();
    const int tx_ = tid & 63, tyb = tid >> 6;
    const int ox = ox0 + tx_;
    for (int q = 0; q < 4; ++q) {
        const int ty_ = tyb + 4 * q, oy = oy0 + ty_;
        if (ox >= onx || oy >= ony) continue;
        float px, py;
        tile_position(&hdr, tx_, ty_, &px, &py);
        int ixr, iyr;
        float dx, dy;
        bool ddx, ddy;
        split_pos(px, &ixr, &dx, &ddx);
        split_pos(py, &iyr, &dy, &ddy);
        const int ix = hdr.bx0 + ixr + OFF, iy = hdr.by0 + iyr + OFF;
        bool covered;
        if (trunc) {
            covered = pos_on_axis(px, hdr.bx0, nx) && pos_on_axis(py, hdr.by0, ny);
        } else {
            const bool inbx = ddx ? (ix - OFF >= 0 && ix - OFF < nx) : (ix >= 0 && ix + NT <= nx);
            const bool inby = ddy ? (iy - OFF >= 0 && iy - OFF < ny) : (iy >= 0 && iy + NT <= ny);
            covered = inbx && inby;
        }
        int32_t m = fill;
        if (covered) {
            float tx[NT], ty[NT];
            taps_of<KIND>(ltab, dx, dy, ddx, ddy, tx, ty);
            float v = 0.f;
            int32_t o = 0;
            for (int r = 0; r < NT; ++r) {
                const int y = iy + r;
                if (y < 0 || y >= ny || ty[r] == 0.f) continue;
                float rs = 0.f;
                for (int c = 0; c < NT; ++c) {
                    const int x = ix + c;
                    if (x < 0 || x >= nx || tx[c] == 0.f) continue;
                    const int32_t w = mask[(size_t)y * nx + x];
                    rs = fmaf(tx[c], (float)w, rs);
                    o |= w;
                }
                v = fmaf(ty[r], rs, v);
            }
            m = lanczos ? (int32_t)rintf(v) : o;
        }
        dst[(size_t)oy * onx + ox] = m;
    }
}

int zm_launch_resample_rim(zm_ctx* ctx, const float2* src, int nx, int ny, int spitch, const double2* lat, int lnx,
                           int lny, int kernel, float fscale, float2* dst, float* plane_a, float* plane_b, int onx,
                           int ony, const int32_t* mask, int32_t* macc, int mop, int mkind) {
    if (kernel == ZM_RESAMPLE_NEAREST) return 0;              // (no footprint: nothing to truncate)
    const float* ltab = nullptr;
    ZM_TRY(zm_get_lanczos_table(ctx, &ltab));
    if (!mask || !macc) mop = 0;
    dim3 blk(256, 1, 1), grd(zm_div_up(onx, TW), zm_div_up(ony, TH), 1);
    zm_scope_timer t(ctx, "resample_rim");
    if (kernel == ZM_RESAMPLE_LANCZOS3)
        hipLaunchKernelGGL(k_resample_rim<ZM_RESAMPLE_LANCZOS3>, grd, blk, 0, ctx->stream, src, nx, ny, spitch, lat, lnx, lny,
                           fscale, ltab, dst, plane_a, plane_b, onx, ony, mask, macc, mop, mkind);
    else
        hipLaunchKernelGGL(k_resample_rim<ZM_RESAMPLE_BILINEAR>, grd, blk, 0, ctx->stream, src, nx, ny, spitch, lat, lnx, lny,
                           fscale, ltab, dst, plane_a, plane_b, onx, ony, mask, macc, mop, mkind);
    ZM_HIP(hipGetLastError());
    return 0;
}

int zm_launch_resample_mask_opts(zm_ctx* ctx, const int32_t* mask, int nx, int ny, const double2* lat, int lnx, int lny,
                                 int kernel, int32_t* dst, int onx, int ony, int32_t fill, int lanczos, int trunc) {
    const float* ltab = nullptr;
    ZM_TRY(zm_get_lanczos_table(ctx, &ltab));
    dim3 blk(256, 1, 1), grd(zm_div_up(onx, TW), zm_div_up(ony, TH), 1);
    zm_scope_timer t(ctx, "resample_mask");
    if (kernel == ZM_RESAMPLE_LANCZOS3)
        hipLaunchKernelGGL(k_resample_mask_opts<ZM_RESAMPLE_LANCZOS3>, grd, blk, 0, ctx->stream, mask, nx, ny, lat, lnx, lny,
                           ltab, dst, onx, ony, fill, lanczos, trunc);
    else if (kernel == ZM_RESAMPLE_BILINEAR)
        hipLaunchKernelGGL(k_resample_mask_opts<ZM_RESAMPLE_BILINEAR>, grd, blk, 0, ctx->stream, mask, nx, ny, lat, lnx, lny,
                           ltab, dst, onx, ony, fill, lanczos, trunc);
    else
        ZM_CHECK(false, "zm_launch_resample_mask_opts: kernel %d has no footprint", kernel);
    ZM_HIP(hipGetLastError());
    return 0;
}
