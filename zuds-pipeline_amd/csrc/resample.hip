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
#include "resample_dev.h"

// ---------------------------------------------------------------------------
// The map travels as a by-value kernel argument (1.5 KB): no host staging buffer
// whose lifetime would have to outlive the enqueue.
__global__ __launch_bounds__(256) void k_lattice(const zm_map_params mp, int lnx, int lny, double2* __restrict__ lat) {
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
    const bool side = after != nullptr && zm_ctx_aux(ctx) != nullptr && !timed;
    hipStream_t s = side ? zm_ctx_aux(ctx) : ctx->stream;
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

// two images that share a geometry as ONE pair plane {a, b} (pad column {0, 0}): zm_align_pair_dev
__global__ __launch_bounds__(256) void k_prep_pair(const float* __restrict__ a, const float* __restrict__ b, int nx, int ny,
                                                   float2* __restrict__ dst, int spitch) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= spitch) return;
    const size_t k = (size_t)y * nx + x;
    dst[(size_t)y * spitch + x] = x < nx ? make_float2(a[k], b[k]) : make_float2(0.f, 0.f);
}
int zm_launch_prep_pair(zm_ctx* ctx, const float* a, const float* b, int nx, int ny, float2* dst, int spitch) {
    zm_scope_timer t(ctx, "prep");
    hipLaunchKernelGGL(k_prep_pair, dim3(zm_div_up(spitch, 256), ny), dim3(256), 0, ctx->stream, a, b, nx, ny, dst, spitch);
    ZM_HIP(hipGetLastError());
    return 0;
}

template <int KIND, int MASKOP>
__global__ __launch_bounds__(256, 4) void k_resample(
    const float2* __restrict__ src, int nx, int ny, int spitch, const double2* __restrict__ lat,
    int lnx, int lny, float fscale, float2* __restrict__ dst, int onx, int ony, int lds_cap,
    const int32_t* __restrict__ mask, const uint16_t* __restrict__ mbox, int32_t* __restrict__ macc,
    int mkind, int mfirst, int ntx, int ntiles, const float* __restrict__ taptab,
    float* __restrict__ plane_a, float* __restrict__ plane_b, float pair_scale) {
    // (plane_a / plane_b: value and weight as two planes - what an alignment hands back - instead of the pair
    // plane `dst` that a stack keeps: no pass to split them afterwards)
    // pair_scale != 0 (round 6, zm_align_pair_dev): the plane holds TWO images {a, b} that share the geometry - the
    // reference and its rms map on their way to a science grid (zuds/subtraction.py:109, zuds/hotpants.py:51: two
    // SWarp runs there, two launches here until now) - and both channels are values: b leaves scaled by pair_scale,
    // nothing is a variance, validity is the footprint test alone.  Per channel the operations of a single alignment.
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
                if (pair_scale != 0.f) {
                    res.x = acc * fscale;
                    res.y = vacc * pair_scale;
                } else if (vacc > 0.f && vacc < ZM_BADVAR_TEST) {
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
                                int32_t* macc, int mop, int mkind, int mfirst, float* plane_a, float* plane_b,
                                float pair_scale) {
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
                           lnx, lny, fscale, dst, onx, ony, lds_elems, mask, mbox, macc, mkind, mfirst, ntx, ntiles, taptab, plane_a, plane_b, pair_scale);
    else if (mop == 1)
        hipLaunchKernelGGL((k_resample<KIND, 1>), pgrd, blk, shmem, ctx->stream, src, nx, ny, spitch, lat,
                           lnx, lny, fscale, dst, onx, ony, lds_elems, mask, mbox, macc, mkind, mfirst, ntx, ntiles, taptab, plane_a, plane_b, pair_scale);
    else
        hipLaunchKernelGGL((k_resample<KIND, 2>), pgrd, blk, shmem, ctx->stream, src, nx, ny, spitch, lat,
                           lnx, lny, fscale, dst, onx, ony, lds_elems, mask, mbox, macc, mkind, mfirst, ntx, ntiles, taptab, plane_a, plane_b, pair_scale);
    ZM_HIP(hipGetLastError());
    return 0;
}

// mop 0: image only; 1: also store the resampled mask into macc; 2: accumulate it
// into macc with mkind (ZM_MASK_AND / ZM_MASK_OR), mfirst = first frame of the stack.
int zm_launch_resample(zm_ctx* ctx, const float2* src, int nx, int ny, int spitch,
                       const double2* lat, int lnx, int lny, int kernel, float fscale,
                       float2* dst, int onx, int ony, int lds_elems, const int32_t* mask,
                       int32_t* macc, int mop, int mkind, int mfirst, float* plane_a, float* plane_b, float pair_scale) {
    dim3 blk(256, 1, 1), grd(zm_div_up(onx, TW), zm_div_up(ony, TH), 1);
    ZM_CHECK(pair_scale == 0.f || (kernel != ZM_RESAMPLE_NEAREST && plane_a && plane_b && ctx->edge == ZM_EDGE_ZERO),
             "zm_launch_resample: a pair of images takes LANCZOS3 / BILINEAR, two output planes and the default edge rule");
    dim3 rgrd(zm_div_up(onx, TW), zm_div_up(ony, RTH), 1);     // k_resample: 64 x 32 tiles
    if (!mask || !macc) mop = 0;
    if (lds_elems > RS_PFCAP) lds_elems = RS_PFCAP;      // what the prefetch registers can stage
    size_t shmem = (size_t)HDR_FLOATS * 4 + (size_t)lds_elems * sizeof(float2);
    if (kernel == ZM_RESAMPLE_LANCZOS3 || kernel == ZM_RESAMPLE_BILINEAR) {
        // the conventions of the context (zm_ctx_set_conventions; defaults: none of this runs).  LANCZOS_ROUND: the
        // image goes alone and the mask through the interpolating kernel; TRUNCATE: the rim is recomputed with per-tap
        // bounds behind k_resample (csrc/resample_opts.hip)
        const bool lz = mop && ctx->mask_resample == ZM_MASKRES_LANCZOS_ROUND, trunc = ctx->edge == ZM_EDGE_TRUNCATE;
        const int mop_img = lz ? 0 : mop;
        if (kernel == ZM_RESAMPLE_LANCZOS3)
            ZM_TRY(launch_resample_kind<ZM_RESAMPLE_LANCZOS3>(ctx, rgrd, shmem, src, nx, ny, spitch, lat, lnx,
                                                              lny, fscale, dst, onx, ony, lds_elems, mask,
                                                              macc, mop_img, mkind, mfirst, plane_a, plane_b, pair_scale));
        else
            ZM_TRY(launch_resample_kind<ZM_RESAMPLE_BILINEAR>(ctx, rgrd, shmem, src, nx, ny, spitch, lat, lnx,
                                                              lny, fscale, dst, onx, ony, lds_elems, mask,
                                                              macc, mop_img, mkind, mfirst, plane_a, plane_b, pair_scale));
        if (trunc)
            ZM_TRY(zm_launch_resample_rim(ctx, src, nx, ny, spitch, lat, lnx, lny, kernel, fscale, dst, plane_a, plane_b, onx,
                                          ony, mask, macc, mop_img, mkind));
        if (lz) {
            const int64_t opix = (int64_t)onx * ony;
            if (mop == 1) return zm_launch_resample_mask_opts(ctx, mask, nx, ny, lat, lnx, lny, kernel, macc, onx, ony, 0, 1, trunc);
            int32_t* tmp = nullptr;
            ZM_TRY(ctx->get("mask_tmp", sizeof(int32_t) * opix, (void**)&tmp));
            ZM_TRY(zm_launch_resample_mask_opts(ctx, mask, nx, ny, lat, lnx, lny, kernel, tmp, onx, ony, -1, 1, trunc));
            return zm_launch_mask_accum(ctx, macc, tmp, opix, mkind, mfirst);
        }
        return 0;
    }
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
    if (kernel != ZM_RESAMPLE_NEAREST && (ctx->edge == ZM_EDGE_TRUNCATE || ctx->mask_resample == ZM_MASKRES_LANCZOS_ROUND))
        return zm_launch_resample_mask_opts(ctx, mask, nx, ny, lat, lnx, lny, kernel, dst, onx, ony, fill,
                                            ctx->mask_resample == ZM_MASKRES_LANCZOS_ROUND, ctx->edge == ZM_EDGE_TRUNCATE);
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

