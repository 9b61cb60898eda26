// Host side of the fused resample -> coadd: item headers, background row / column tables, geometry and the launch.
#include "fused_dev.h"


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
    // (measured and not kept, round 6: the whole header read and written back here, so that the fused kernel finds
    // the headers made 0.7 ms earlier in the caches again - no change, tools/fork_ab.sh)
    reinterpret_cast<float*>(out)[item * FF_HDR_WORDS + offsetof(ff_hdr, vscale) / 4] = vs ? *vs : 1.f;
    out[item * FF_HDR_WORDS + offsetof(ff_hdr, frame_raw) / 4] = fr[f].mboxflag ? *fr[f].mboxflag : 1;
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


void zm_fused_geometry(int* tile_h, int* lds_cap) {
    *tile_h = FT_H;
    *lds_cap = FD_LDS_CAP;
}

// frames: nfr descriptors on the host (device pointers inside); out_mask may be NULL (no mask coadd)
// geometry of a fused launch: LDS tile, grid, yield budget (shared by the early header pass and the launch)
struct ff_geom {
    bool own;
    int lds_elems, ntx, ntiles, G, budget;
    size_t shmem;
};
// ZM_FF_FORM=dma: k_coadd_fused_dma also where the owner-staged kernel would run (developer: A / B)
static bool ff_use_own() {
    const char* e = getenv("ZM_FF_FORM");
    return !(e && !strcmp(e, "dma"));
}
// fits_own: every frame's planned footprint fits the fixed slot of k_coadd_fused_own (fused_prepare's verdict)
static int ff_geometry(zm_ctx* ctx, int onx, int ony, int lds_elems, bool fits_own, ff_geom* g) {
    g->ntx = zm_div_up(onx, TW);
    g->ntiles = g->ntx * zm_div_up(ony, FT_H);
    g->own = fits_own && ff_use_own();
    lds_elems = std::min(std::max(lds_elems, 64), FD_LDS_CAP);
    g->lds_elems = (lds_elems + 7) & ~7;
    g->shmem = g->own ? (size_t)FO_LDS : (size_t)FD_OFF_RAW + 20 * (size_t)g->lds_elems + 4 * 8 * FD_YROWS;
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
    const char* ye = ZM_DEVENV("ZM_FF_YIELD");
    g->budget = ye ? atoi(ye) : (ctx->share >= 2 ? 2 : 0);
    if (g->budget > 0 && (g->ntiles + g->budget - 1) / g->budget > g->G) g->G = (g->ntiles + g->budget - 1) / g->budget;
    else g->budget = 0;
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
                           lnx, lny, onx, ony, g.lds_elems, g.own ? 2 : 1, g.ntx, g.ntiles, ghdr, tilectr, g.G, skip_vscale);
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
    static const bool fork_off = ZM_DEVENV("ZM_FF_FORK") && ZM_DEVENV("ZM_FF_FORK")[0] == '0';
    // (scope timers keep a TIMED kernel on the main stream - this one's scope is "ff_headers".  Until late in round 6
    // the test was `ctx->timing` alone: the bench's timed steps, which time the fused kernel only, ran the headers
    // on the main stream in front of it - 84 us per step that no untimed caller paid)
    const bool timed = ctx->timing && (ctx->timing_only.empty() || ctx->timing_only == "ff_headers");
    if (!zm_ctx_aux(ctx) || timed || fork_off) return 0;
    ff_geom g;
    ZM_TRY(ff_geometry(ctx, onx, ony, lds_elems, fits_own, &g));
    zm_ff* dev = nullptr;
    int* ghdr = nullptr;
    ZM_TRY(ff_upload_and_headers(ctx, frames_host, nfr, lnx, lny, onx, ony, g, zm_ctx_aux(ctx), 1, &dev, &ghdr));
    hipEvent_t* ev = nullptr;
    ZM_TRY(zm_get_sync_events(ctx, 10, &ev));
    ZM_HIP(hipEventRecord(ev[9], zm_ctx_aux(ctx)));
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
    const bool own = g.own;
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
    // developer build (-DZM_DEV) only - ZM_FF_DBG (tools/ff_probe.py): 1 no pixel work, 2 no prep / LDS store, 4 no
    // staging loads; ZM_FF_PRIO / ZM_FF_DEAL: who stages what at which priority; ZM_FF_PROF: per-wave phase clocks
    const char *e_dbg = ZM_DEVENV("ZM_FF_DBG"), *e_prio = ZM_DEVENV("ZM_FF_PRIO"), *e_deal = ZM_DEVENV("ZM_FF_DEAL"),
               *e_prof = ZM_DEVENV("ZM_FF_PROF");
    const int dbg = (e_dbg ? (atoi(e_dbg) & 255) : 0) | ((budget & 0xffff) << 8) |
                    (own && e_prio ? ((atoi(e_prio) & 15) << 24) : 0) | (own ? (((e_deal ? atoi(e_deal) : 1) & 3) << 28) : 0);
    long long* prof = nullptr;
    const bool want_prof = e_prof && atoi(e_prof) != 0;
    const int nwv = FD_THREADS / 64;
    if (want_prof) ZM_TRY(ctx->get("ff_prof", sizeof(long long) * 5 * nwv * (size_t)G, (void**)&prof));
    {
        zm_scope_timer t(ctx, "coadd_fused");
        const ff_launch_args a = {dev, nfr, onx, ony, lds_elems, ntx, ntiles, ghdr, out_img, out_wgt, out_mask, out_cov,
                                  partial, taptab, tilectr, stack, (long long)fstride, dbg, prof};
        const bool devk = want_prof || (dbg & 255);
        ZM_TRY((own ? zm_ff_launch_own : zm_ff_launch_dma)(ctx, mop, avg, stack != nullptr, devk, G, shmem, a));
        ctx->ff_last_form = own ? 2 : 1;
    }
    ZM_HIP(hipGetLastError());
    if (want_prof) {
        std::vector<long long> h((size_t)5 * nwv * G);
        ZM_HIP(hipMemcpyAsync(h.data(), prof, sizeof(long long) * h.size(), hipMemcpyDeviceToHost, ctx->stream));
        ZM_HIP(hipStreamSynchronize(ctx->stream));
        static const char* nm[5] = {"dma issue", "pixels", "dma wait", "barriers", "prep"};
        double sum[5] = {0, 0, 0, 0, 0};
        for (size_t w = 0; w < (size_t)nwv * G; ++w)
            for (int k = 0; k < 5; ++k) sum[k] += (double)h[w * 5 + k];
        fprintf(stderr, "k_coadd_fused phases, mean per wave (kilo-cycles of the shader clock):");
        for (int k = 0; k < 5; ++k) fprintf(stderr, " %s %.1f", nm[k], sum[k] / ((double)nwv * G) * 1e-3);
        fprintf(stderr, "\n");
        if (atoi(e_prof) >= 2) {
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

