// C-ABI entry points for resampling and co-addition (host- and device-pointer
// flavours).  The host flavour copies borrowed numpy buffers to the device,
// runs the device flavour on the ctx stream and copies the products back.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "zm_internal.h"
#include "wcs_math.h"

extern "C" void zm_coadd_params_default(zm_coadd_params* p) {
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->combine = ZM_COMBINE_CLIPPED;
    p->mask_combine = ZM_MASK_AND;
    p->resample = ZM_RESAMPLE_LANCZOS3;
    p->subtract_back = 1;
    p->back_size = 128;
    p->back_filtersize = 3;
    p->rescale_weights = 1;
    p->clip_sigma = 4.0;
    p->clip_ampfrac = 0.3;
    p->weight_thresh = 1e-30;
}

static int check_wcs(const zm_wcs* w, const char* what) {
    ZM_CHECK(w != nullptr, "%s: WCS is NULL", what);
    ZM_CHECK(w->naxis[0] > 0 && w->naxis[1] > 0, "%s: NAXIS must be positive (got %d x %d)", what,
             w->naxis[0], w->naxis[1]);
    ZM_CHECK(w->naxis[0] <= 65536 && w->naxis[1] <= 65536, "%s: NAXIS too large", what);
    double det = w->cd[0] * w->cd[3] - w->cd[1] * w->cd[2];
    ZM_CHECK(det != 0.0 && std::isfinite(det), "%s: singular CD matrix", what);
    return 0;
}

// LDS elements (float2) a tw x th output tile needs: bound the footprint from the map's Jacobian
// sampled over the output grid.  k_resample: 64 x 32 tiles, at most 8000 elements (64 KB without
// the opt-in); the fused coadd: 64 x 64 tiles made of two k_resample boxes (their union: one more
// alignment step), `cap` elements - a larger plan means the frame's footprints exceed the LDS tile.
static int plan_lds(const zm_map_params* mp, int onx, int ony, int ntaps, int tw = 64, int th = 32, int cap = 8000,
                    bool exact = false, int* box_w = nullptr, int* box_h = nullptr) {
    double wmax = 0, hmax = 0;
    for (int sy = 0; sy < 3; ++sy)
        for (int sx = 0; sx < 3; ++sx) {
            double x = 1.0 + sx * 0.5 * (onx - 1), y = 1.0 + sy * 0.5 * (ony - 1);
            double x0, y0, x1, y1, x2, y2;
            zm_map_point(mp, x, y, &x0, &y0);
            zm_map_point(mp, x + tw, y, &x1, &y1);
            zm_map_point(mp, x, y + th, &x2, &y2);
            if (!std::isfinite(x0 + y0 + x1 + y1 + x2 + y2)) continue;
            wmax = std::max(wmax, fabs(x1 - x0) + fabs(x2 - x0));
            hmax = std::max(hmax, fabs(y1 - y0) + fabs(y2 - y0));
        }
    const int extra = th > 32 ? 4 : 0;
    double w = ceil(wmax * 1.02) + ntaps + 9 + extra, h = ceil(hmax * 1.02) + ntaps + 5 + extra / 2;   // + box alignment (4 px)
    if (exact) {
        // the bound build_tile_header3 obeys: bw = (bx1 - bx0 + 4) & ~3 <= extent + 16, bh <= extent + 10.
        // (No safety factor: an item that exceeds the plan is found by its header and takes the generic code.)
        w = (double)(((long long)(ceil(wmax) + ntaps + 10)) & ~3LL);
        h = ceil(hmax) + ntaps + 4;
    }
    // what build_tile_header3 makes of the largest sampled extents (the fixed slot of k_coadd_fused_own is sized
    // by these, not by the area): bw = (floor(mxx) + 5 - ((floor(mnx) - 3) & ~3) + 4) & ~3, bh = floor(mxy) - floor(mny) + 9
    if (box_w) *box_w = (int)std::min(1e6, (double)((((long long)floor(wmax)) + 16) & ~3LL));
    if (box_h) *box_h = (int)std::min(1e6, floor(hmax) + 10);
    double e = w * h;
    if (!(e > 0) || e > cap) return cap + 1;
    return (int)e;
}

static int ntaps_of(int kernel) {
    return kernel == ZM_RESAMPLE_LANCZOS3 ? 6 : kernel == ZM_RESAMPLE_BILINEAR ? 2 : 1;
}

struct bk_plan {
    float* vs_all = nullptr;      // per frame: 4 floats, [0] = variance scale (RESCALE_WEIGHTS)
    int nslot = 1;                // meshes of the largest frame: the scratch slot of every frame
};

static bool frame_has_bk(const zm_coadd_params* P, const zm_dframe* fr, int i) {
    return P->subtract_back || (P->rescale_weights && fr[i].wgt);
}
static int frame_nmode(const zm_coadd_params* P, const zm_dframe* fr, int i) {
    return (P->rescale_weights && fr[i].wgt) ? 2 : 1;
}

// Phase 1 of a stack: mesh statistics, filter and variance rescale of every frame, batched over
// runs of equally sized frames (one launch each per run instead of one per frame: the
// statistics grid of a single 3072^2 frame is 1152 workgroups against 1024 resident, and
// every per-frame launch costs a ~10 us gap).
// (Tried on MI355X and dropped: the filter / lattices on a second stream pipelined
// against the neighbouring frames - foreign workgroups on its CUs stretch the statically
// partitioned persistent resample kernel by a quarter.)
static int frames_background(zm_ctx* ctx, int n, const zm_dframe* fr, const zm_coadd_params* P, bk_plan* bp) {
    const float wthresh = (float)P->weight_thresh;
    ZM_TRY(ctx->get("var_scale", sizeof(float) * 4 * (size_t)n, (void**)&bp->vs_all));
    int nslot = 1;
    for (int i = 0; i < n; ++i)
        nslot = std::max(nslot, ((fr[i].wcs.naxis[0] - 1) / std::max(P->back_size, 1) + 1) *
                                    ((fr[i].wcs.naxis[1] - 1) / std::max(P->back_size, 1) + 1));
    bp->nslot = nslot;
    for (int i0 = 0; i0 < n;) {
        if (!frame_has_bk(P, fr, i0)) { ++i0; continue; }
        int i1 = i0 + 1;
        while (i1 < n && frame_has_bk(P, fr, i1) && frame_nmode(P, fr, i1) == frame_nmode(P, fr, i0) &&
               fr[i1].wcs.naxis[0] == fr[i0].wcs.naxis[0] && fr[i1].wcs.naxis[1] == fr[i0].wcs.naxis[1])
            ++i1;
        const int nf = i1 - i0, nx = fr[i0].wcs.naxis[0], ny = fr[i0].wcs.naxis[1], nmode = frame_nmode(P, fr, i0);
        std::vector<const float*> imgs(nf), wgts(nf);
        for (int f = 0; f < nf; ++f) { imgs[f] = fr[i0 + f].img; wgts[f] = fr[i0 + f].wgt; }
        ZM_TRY(zm_batch_stats(ctx, nf, imgs.data(), wgts.data(), nx, ny, P->back_size, wthresh, 0, nmode,
                              "cbk", i0, n, nslot));
        ZM_TRY(zm_batch_filter(ctx, nf, nx, ny, P->back_size, P->back_filtersize, nmode, "cbk", i0, n, nslot));
        if (nmode == 2) {
            float *nodes = nullptr, *bstats = nullptr;
            int nbx = 0, nby = 0;
            ZM_TRY(zm_frame_products(ctx, nx, ny, P->back_size, "cbk", i0, n, nslot, &nodes, &bstats, &nbx, &nby));
            ZM_TRY(zm_batch_var_scale(ctx, nf, bstats, bp->vs_all + 4 * (size_t)i0));
        }
        i0 = i1;
    }
    return 0;
}

// The int32 view of a frame's mask for the kernels that read int32 only (k_prep_box, k_resample and its
// raw-mask fallback): an int16 plane (ZM_MASKTYPE_I16) is widened on the device into the scratch slot
// `slot` (stream-ordered: one slot serves the frames of a loop that consumes each before the next is made).
static int frame_mask_i32(zm_ctx* ctx, const zm_dframe* f, const char* slot, const int32_t** out) {
    *out = nullptr;
    if (!f->mask) return 0;
    ZM_CHECK(f->mask_type == ZM_MASKTYPE_I32 || f->mask_type == ZM_MASKTYPE_I16, "frame: unknown mask_type %d", f->mask_type);
    if (f->mask_type == ZM_MASKTYPE_I32) {
        *out = (const int32_t*)f->mask;
        return 0;
    }
    const int64_t n = (int64_t)f->wcs.naxis[0] * f->wcs.naxis[1];
    int32_t* w = nullptr;
    ZM_TRY(ctx->get(slot, sizeof(int32_t) * (size_t)n, (void**)&w));
    ZM_TRY(zm_launch_mask_widen(ctx, (const int16_t*)f->mask, n, w));
    *out = w;
    return 0;
}

// Resample nframes device frames onto `wout` into `stack` (float2 [n][ony*onx]).
// Also accumulates the mask coadd when acc_mask != NULL.
static int resample_frames_fused(zm_ctx* ctx, int n, const zm_dframe* fr, const zm_wcs* wout,
                                 const zm_coadd_params* P, float2* stack, int32_t* acc_mask);
// (a context with SWarp's own edge / mask conventions switched on takes the materialised path: csrc/resample_opts.hip)
static bool default_conventions(const zm_ctx* ctx) { return ctx->edge == ZM_EDGE_ZERO && ctx->mask_resample == ZM_MASKRES_OR; }
static bool fused_stack_ok(const zm_ctx* ctx, const zm_coadd_params* P);
static int resample_frames(zm_ctx* ctx, int n, const zm_dframe* fr, const zm_wcs* wout,
                           const zm_coadd_params* P, float2* stack, int32_t* acc_mask,
                           int32_t mask_fill, int mask_kind) {
    if (fused_stack_ok(ctx, P) && mask_fill == -1 && mask_kind == P->mask_combine)
        return resample_frames_fused(ctx, n, fr, wout, P, stack, acc_mask);
    const int onx = wout->naxis[0], ony = wout->naxis[1];
    const int64_t opix = (int64_t)onx * ony;
    const int lnx = (onx - 1) / ZM_LATTICE_STEP + 2, lny = (ony - 1) / ZM_LATTICE_STEP + 2;

    for (int i = 0; i < n; ++i) ZM_TRY(check_wcs(&fr[i].wcs, "frame"));

    // Phase 1: background statistics of every frame (batched).  Phase 2: prep + resample, frame by frame.
    // everything older on the main stream (the resamples of a previous call read the lattice
    // buffer): the lattices of this call may start once that has drained
    hipEvent_t* evs = nullptr;
    ZM_TRY(zm_get_sync_events(ctx, 6, &evs));
    ZM_HIP(hipEventRecord(evs[3], ctx->stream));
    const float wthresh = (float)P->weight_thresh;
    bk_plan bp;
    ZM_TRY(frames_background(ctx, n, fr, P, &bp));
    float* vs_all = bp.vs_all;
    const int nslot = bp.nslot;
    auto has_bk = [&](int i) { return frame_has_bk(P, fr, i); };
    auto nmode_of = [&](int i) { return frame_nmode(P, fr, i); };
    // The maps, flux scales and LDS plans are host work (fp64 TPV inversions): done here, while
    // the GPU runs phase 1, not before the first launch.
    std::vector<zm_map_params> mp_host(n);
    std::vector<double> fscale(n);
    std::vector<int> lds(n);
    for (int i = 0; i < n; ++i) {
        zm_make_map(wout, &fr[i].wcs, &mp_host[i]);
        ZM_TRY(zm_flux_scale(&fr[i].wcs, wout, fr[i].flxscale, &fscale[i]));
        lds[i] = std::min(plan_lds(&mp_host[i], onx, ony, ntaps_of(P->resample)), 8000);
    }
    double2* lat = nullptr;
    ZM_TRY(ctx->get("lattice", sizeof(double2) * (size_t)lnx * lny * n, (void**)&lat));
    ZM_TRY(zm_launch_lattice_batch(ctx, mp_host.data(), n, lnx, lny, lat, evs[3]));

    bool first_mask = true;
    for (int i = 0; i < n; ++i) {
        const int nx = fr[i].wcs.naxis[0], ny = fr[i].wcs.naxis[1];
        const int spitch = (nx + 1) & ~1;
        float *bknodes = nullptr, *vscale = nullptr;
        int nbx = 0, nby = 0;
        if (has_bk(i)) {
            float *nodes = nullptr, *bstats = nullptr;
            ZM_TRY(zm_frame_products(ctx, nx, ny, P->back_size, "cbk", i, n, nslot, &nodes, &bstats, &nbx, &nby));
            if (nmode_of(i) == 2) vscale = vs_all + 4 * (size_t)i;
            if (P->subtract_back) bknodes = nodes;
        }
        float2* src = nullptr;
        ZM_TRY(ctx->get("prep", sizeof(float2) * (size_t)spitch * ny, (void**)&src));
        // the mask rides along with its frame: same tile, same positions
        const bool with_mask = acc_mask && fr[i].mask;
        const int32_t* m32 = nullptr;
        if (with_mask) ZM_TRY(frame_mask_i32(ctx, &fr[i], "mask_widen", &m32));
        ZM_TRY(zm_launch_prep(ctx, fr[i].img, fr[i].wgt, nx, ny, bknodes, nbx, nby, P->back_size,
                              vscale, wthresh, src, spitch, m32, ntaps_of(P->resample)));
        ZM_TRY(zm_launch_resample(ctx, src, nx, ny, spitch, lat + (size_t)i * lnx * lny, lnx, lny,
                                  P->resample, (float)fscale[i], stack + (size_t)i * opix, onx,
                                  ony, lds[i], m32, acc_mask, with_mask ? 2 : 0, mask_kind,
                                  first_mask ? 1 : 0));
        if (with_mask) first_mask = false;
    }
    if (acc_mask && first_mask) ZM_HIP(hipMemsetAsync(acc_mask, 0xFF, sizeof(int32_t) * opix, ctx->stream));
    return 0;
}


// WEIGHTED / AVERAGE stacks with a Lanczos-3 kernel: every frame is prepped into its own plane
// (N x 75 MB at 3072^2: the place the resampled stack took before), then ONE launch loops the
// frames inside each output tile (k_coadd_fused, resample.hip).
static bool fused_ok(const zm_ctx* ctx, const zm_coadd_params* P) {
    // ZM_COADD_FUSED=0 takes the materialised path (resampled stack + k_combine_sum): the
    // reference the fused kernel is tested against bit for bit (tests/test_coadd_gpu.py)
    const char* e = getenv("ZM_COADD_FUSED");
    const bool off = e && e[0] == '0';
    return !off && default_conventions(ctx) && P->resample == ZM_RESAMPLE_LANCZOS3 &&
           (P->combine == ZM_COMBINE_WEIGHTED || P->combine == ZM_COMBINE_AVERAGE);
}

// Everything in front of the fused launch: backgrounds, lattices, the y part of every frame's
// background spline (k_bk_rows), the box-OR planes of the masks, the frame descriptors.  Frames
// are read RAW by the fused kernel; a prepped plane is only written for a frame that cannot be
// staged that way (see zm_ff.src).
struct fused_stage {
    std::vector<zm_ff> ff;
    int lds = 0, lnx = 0, lny = 0;
    bool any_mask = false;
    bool fits_own = true;          // every frame's planned box fits the fixed slot of k_coadd_fused_own (80 x 42)
};
static int fused_prepare(zm_ctx* ctx, int n, const zm_dframe* fr, const zm_wcs* wout, const zm_coadd_params* P,
                         bool want_mask, fused_stage* S) {
    const int onx = wout->naxis[0], ony = wout->naxis[1];
    const int lnx = (onx - 1) / ZM_LATTICE_STEP + 2, lny = (ony - 1) / ZM_LATTICE_STEP + 2;
    S->lnx = lnx;
    S->lny = lny;
    for (int i = 0; i < n; ++i) ZM_TRY(check_wcs(&fr[i].wcs, "frame"));
    hipEvent_t* evs = nullptr;
    ZM_TRY(zm_get_sync_events(ctx, 9, &evs));
    // Round 4: the mesh statistics need neither the maps nor the staging plans of the frames, so they are enqueued
    // FIRST and the host works out the maps (inverse-map parameters, flux scales and LDS plans of every frame:
    // ~0.2 ms for 32 frames, which used to pass with the GPU idle) while they run.  evs[3]: the inputs are ready
    // (what the second stream's work - box-OR planes, lattices - waits for).
    int* boxflags = nullptr;           // per frame: its box-OR plane holds entries that defer to the raw mask
    ZM_TRY(ctx->get("mask_box_flags", sizeof(int) * (size_t)n, (void**)&boxflags));
    ZM_HIP(hipMemsetAsync(boxflags, 0, sizeof(int) * (size_t)n, ctx->stream));
    ZM_HIP(hipEventRecord(evs[3], ctx->stream));
    bk_plan bp;
    ctx->bk_stats_event_valid = false;
    ZM_TRY(frames_background(ctx, n, fr, P, &bp));
    const float wthresh = (float)P->weight_thresh;
    int ff_th = 32, ff_cap = 3700;
    zm_fused_geometry(&ff_th, &ff_cap);
    const char* e = getenv("ZM_FF_RAW");
    // (raw staging: a quad in one mesh column, a box - at most 96 columns - under at most two of them)
    const bool raw_ok = !(e && e[0] == '0') && ((P->back_size % 8 == 0 && P->back_size >= 96) || !P->subtract_back);
    std::vector<zm_map_params> mp_host(n);
    std::vector<zm_ff>& ff = S->ff;
    ff.resize(n);
    int lds = 64;
    size_t prep_bytes = 0, box_bytes = 0, yt_bytes = 0;
    std::vector<size_t> prep_off(n), box_off(n), yt_off(n);
    std::vector<char> need_src(n, 0);
    bool any_mask = false;
    size_t widen_bytes = 0;
    std::vector<size_t> widen_off(n);
    for (int i = 0; i < n; ++i) {
        const int nx = fr[i].wcs.naxis[0], ny = fr[i].wcs.naxis[1], spitch = (nx + 1) & ~1;
        ZM_CHECK(!fr[i].mask || fr[i].mask_type == ZM_MASKTYPE_I32 || fr[i].mask_type == ZM_MASKTYPE_I16,
                 "frame %d: unknown mask_type %d", i, fr[i].mask_type);
        zm_make_map(wout, &fr[i].wcs, &mp_host[i]);
        double fs = 1.0;
        ZM_TRY(zm_flux_scale(&fr[i].wcs, wout, fr[i].flxscale, &fs));
        int pbw = 0, pbh = 0;
        const int plan = plan_lds(&mp_host[i], onx, ony, 6, 64, ff_th, ff_cap, true, &pbw, &pbh);
        if (pbw > 80 || pbh > 42 || ff_th != 32) S->fits_own = false;
        // a footprint beyond the LDS tile is gathered from global memory: from a prepped plane
        // (a frame without 16-byte rows is prepped into a plane, which has them)
        const int vec_ok = (nx % 4 == 0) && (((uintptr_t)fr[i].img & 15) == 0) && (((uintptr_t)fr[i].wgt & 15) == 0);
        need_src[i] = (!raw_ok || !vec_ok || plan > ff_cap) ? 1 : 0;
        lds = std::max(lds, std::min(plan, ff_cap));
        prep_off[i] = prep_bytes;
        if (need_src[i]) prep_bytes += ((sizeof(float2) * (size_t)spitch * ny) + 255) & ~(size_t)255;
        box_off[i] = box_bytes;
        const bool with_mask = want_mask && fr[i].mask;
        const int mpitch = (nx + 7) & ~7;          // 16-byte pieces of the box-OR plane for the LDS-DMA
        if (with_mask) box_bytes += ((sizeof(uint16_t) * (size_t)mpitch * ny) + 255) & ~(size_t)255;
        any_mask |= with_mask;
        memset(&ff[i], 0, sizeof(zm_ff));
        ff[i].nx = nx; ff[i].ny = ny; ff[i].spitch = spitch;
        ff[i].fscale = (float)fs;
        ff[i].fscale2 = (float)fs * (float)fs;
        ff[i].mask = with_mask ? fr[i].mask : nullptr;
        ff[i].mask16 = (with_mask && fr[i].mask_type == ZM_MASKTYPE_I16) ? 1 : 0;
        // (a frame that goes through k_prep_box - it cannot be staged raw - has its int16 mask widened first)
        widen_off[i] = widen_bytes;
        if (ff[i].mask16 && need_src[i]) widen_bytes += ((sizeof(int32_t) * (size_t)nx * ny) + 255) & ~(size_t)255;
        ff[i].img = fr[i].img;
        ff[i].wgt = fr[i].wgt;
        ff[i].wthresh = wthresh;
        ff[i].vec_ok = vec_ok;
        ff[i].mpitch = mpitch;
    }
    double2* lat = nullptr;
    char *prep_all = nullptr, *box_all = nullptr, *yt_all = nullptr;
    ZM_TRY(ctx->get("lattice", sizeof(double2) * (size_t)lnx * lny * n, (void**)&lat));
    if (prep_bytes) ZM_TRY(ctx->get("prep_all", prep_bytes, (void**)&prep_all));
    if (box_bytes) ZM_TRY(ctx->get("mask_box_all", box_bytes, (void**)&box_all));
    if (widen_bytes) {
        char* wall = nullptr;
        ZM_TRY(ctx->get("mask_widen_all", widen_bytes, (void**)&wall));
        for (int i = 0; i < n; ++i)
            if (ff[i].mask16 && need_src[i]) {
                int32_t* w = (int32_t*)(wall + widen_off[i]);
                ZM_TRY(zm_launch_mask_widen(ctx, (const int16_t*)ff[i].mask, (int64_t)ff[i].nx * ff[i].ny, w));
                ff[i].mask = w;
                ff[i].mask16 = 0;
            }
    }
    // the box-OR planes of the frames that are read raw: they need the masks only, so they go out on the
    // second stream, beside the mesh statistics
    std::vector<zm_boxjob> boxes;
    for (int i = 0; i < n; ++i) {
        uint16_t* mbox = ff[i].mask ? (uint16_t*)(box_all + box_off[i]) : nullptr;
        ff[i].mbox = mbox;
        if (mbox && !need_src[i]) {
            boxes.push_back(zm_boxjob{ff[i].mask, mbox, boxflags + i, ff[i].nx, ff[i].ny, ff[i].mpitch, ff[i].mask16});
            ff[i].mboxflag = boxflags + i;
        }
    }
    hipEvent_t boxes_done = nullptr;
    // Round 6: the box-OR pre-pass (1.2 GB, bandwidth-bound) no longer runs beside the mesh statistics - two
    // bandwidth-bound kernels that stretched each other (statistics 0.74 -> 0.93 ms) - but behind them, beside the small
    // latency-bound kernels that follow (k_mesh_guess, filter, variance scales, background rows: ~0.2 ms with the GPU
    // nearly idle).  The second stream therefore takes the lattices and the item headers FIRST (they are light and
    // still run beside the statistics; the headers no longer read the box flags: k_ff_vscale fills them in) and the
    // boxes last, behind the event the statistics record (ZM_BOX_LATE=0, developer build: the old order).
    static const bool box_late = !(ZM_DEVENV("ZM_BOX_LATE") && ZM_DEVENV("ZM_BOX_LATE")[0] == '0');
    if (!box_late) ZM_TRY(zm_launch_mask_boxes(ctx, boxes.data(), (int)boxes.size(), evs[3], &boxes_done));
    ZM_TRY(zm_launch_lattice_batch(ctx, mp_host.data(), n, lnx, lny, lat, evs[3]));
    std::vector<zm_bkrows> rows;
    struct bkinfo { float* nodes; float* vscale; int nbx, nby; };
    std::vector<bkinfo> bki(n, bkinfo{nullptr, nullptr, 0, 0});
    for (int i = 0; i < n; ++i) {
        const int nx = ff[i].nx, ny = ff[i].ny;
        if (frame_has_bk(P, fr, i)) {
            float *nodes = nullptr, *bstats = nullptr;
            int nbx = 0, nby = 0;
            ZM_TRY(zm_frame_products(ctx, nx, ny, P->back_size, "cbk", i, n, bp.nslot, &nodes, &bstats, &nbx, &nby));
            if (frame_nmode(P, fr, i) == 2) bki[i].vscale = bp.vs_all + 4 * (size_t)i;
            if (P->subtract_back) { bki[i].nodes = nodes; bki[i].nbx = nbx; bki[i].nby = nby; }
        }
        yt_off[i] = yt_bytes;
        if (!need_src[i] && bki[i].nodes) {
            ff[i].ytp = std::max(bki[i].nbx - 1, 1);
            yt_bytes += ((sizeof(float4) * ((size_t)ny * ff[i].ytp + (size_t)nx)) + 255) & ~(size_t)255;   // y table, then x weights
        }
    }
    if (yt_bytes) ZM_TRY(ctx->get("bk_rows_all", yt_bytes, (void**)&yt_all));
    for (int i = 0; i < n; ++i) {
        const int nx = ff[i].nx, ny = ff[i].ny;
        ff[i].lat = lat + (size_t)i * lnx * lny;
        ff[i].vscale = bki[i].vscale;
        if (need_src[i]) {
            // the frame prepped into its own plane (k_prep_box also fills the box-OR plane)
            float2* src = (float2*)(prep_all + prep_off[i]);
            ZM_TRY(zm_launch_prep(ctx, fr[i].img, fr[i].wgt, nx, ny, bki[i].nodes, bki[i].nbx, bki[i].nby, P->back_size,
                                  bki[i].vscale, wthresh, src, ff[i].spitch, (const int32_t*)ff[i].mask, 6, (uint16_t*)ff[i].mbox, ff[i].mpitch));
            ff[i].src = src;
            continue;
        }
        if (bki[i].nodes) {
            ff[i].bk = bki[i].nodes;
            ff[i].nbx = bki[i].nbx;
            ff[i].nby = bki[i].nby;
            ff[i].invmesh = 1.0f / (float)P->back_size;
            ff[i].ytab = (const float4*)(yt_all + yt_off[i]);
            ff[i].xtab = ff[i].ytab + (size_t)ny * ff[i].ytp;
            zm_bkrows r;
            memset(&r, 0, sizeof(r));
            r.bk = bki[i].nodes; r.out = (float4*)(yt_all + yt_off[i]);
            r.nbx = bki[i].nbx; r.nby = bki[i].nby; r.ny = ny; r.ytp = ff[i].ytp;
            r.invmesh = ff[i].invmesh;
            r.nx = nx; r.xout = r.out + (size_t)ny * ff[i].ytp;
            rows.push_back(r);
        }
    }
    ZM_TRY(zm_launch_fused_prepass(ctx, rows.data(), (int)rows.size()));
    // the descriptors are final: the item headers go out on the second stream, behind the box-OR planes and the
    // lattices they read, beside the background chain of the main stream
    ZM_TRY(zm_launch_fused_headers_early(ctx, ff.data(), n, lnx, lny, onx, ony, lds, S->fits_own));
    if (box_late)
        ZM_TRY(zm_launch_mask_boxes(ctx, boxes.data(), (int)boxes.size(), evs[3], &boxes_done,
                                    ctx->bk_stats_event_valid ? ctx->bk_stats_event : nullptr));
    if (boxes_done) ZM_HIP(hipStreamWaitEvent(ctx->stream, boxes_done, 0));
    S->lds = lds;
    S->any_mask = any_mask;
    return 0;
}

static bool fused_stack_ok(const zm_ctx* ctx, const zm_coadd_params* P) {
    // the stack of any Lanczos-3 coadd; ZM_COADD_FUSED=0: k_resample frame by frame (the reference
    // the fused machinery is tested against)
    const char* e = getenv("ZM_COADD_FUSED");
    return !(e && e[0] == '0') && default_conventions(ctx) && P->resample == ZM_RESAMPLE_LANCZOS3;
}

static int coadd_fused(zm_ctx* ctx, int n, const zm_dframe* fr, const zm_wcs* wout, const zm_coadd_params* P,
                       int partial, float* out_img, float* out_wgt, int32_t* out_mask, float* out_cov) {
    const int onx = wout->naxis[0], ony = wout->naxis[1];
    fused_stage S;
    ZM_TRY(fused_prepare(ctx, n, fr, wout, P, out_mask != nullptr, &S));
    return zm_launch_coadd_fused(ctx, S.ff.data(), n, S.lnx, S.lny, onx, ony, S.lds, P->combine, P->mask_combine,
                                 out_img, out_wgt, (out_mask && S.any_mask) ? out_mask : nullptr, out_cov, partial,
                                 out_mask && !S.any_mask ? out_mask : nullptr, nullptr, 0, S.fits_own);
}

// The resampled stack of a CLIPPED / MEDIAN coadd through the same kernel (STACK mode: samples stored,
// not summed; the mask coadd with its -1 markers left in, as resample_frames leaves it).
static int resample_frames_fused(zm_ctx* ctx, int n, const zm_dframe* fr, const zm_wcs* wout,
                                 const zm_coadd_params* P, float2* stack, int32_t* acc_mask) {
    const int onx = wout->naxis[0], ony = wout->naxis[1];
    const int64_t opix = (int64_t)onx * ony;
    fused_stage S;
    ZM_TRY(fused_prepare(ctx, n, fr, wout, P, acc_mask != nullptr, &S));
    return zm_launch_coadd_fused(ctx, S.ff.data(), n, S.lnx, S.lny, onx, ony, S.lds, ZM_COMBINE_WEIGHTED,
                                 P->mask_combine, nullptr, nullptr, (acc_mask && S.any_mask) ? acc_mask : nullptr,
                                 nullptr, 1, acc_mask && !S.any_mask ? acc_mask : nullptr, stack, opix, S.fits_own);
}

extern "C" int zm_resample_stack_dev(zm_ctx* ctx, int nframes, const zm_dframe* frames,
                                     const zm_wcs* wout, const zm_coadd_params* params,
                                     float* stack, int32_t* out_mask_partial) {
    ZM_CHECK(ctx && frames && wout && params && stack, "zm_resample_stack_dev: null argument");
    ZM_CHECK(nframes >= 1, "zm_resample_stack_dev: need at least one frame");
    ZM_HIP(hipSetDevice(ctx->device));
    ZM_TRY(check_wcs(wout, "output grid"));
    return resample_frames(ctx, nframes, frames, wout, params, (float2*)stack, out_mask_partial, -1,
                           params->mask_combine);
}

extern "C" int zm_combine_stack_dev(zm_ctx* ctx, int nframes, const float* stack,
                                    int64_t frame_stride_px, int64_t npix,
                                    const zm_coadd_params* params, float* out_img,
                                    float* out_wgt) {
    ZM_CHECK(ctx && stack && params && out_img && out_wgt, "zm_combine_stack_dev: null argument");
    ZM_HIP(hipSetDevice(ctx->device));
    return zm_launch_combine(ctx, nframes, (const float2*)stack, frame_stride_px, npix,
                             params->combine, (float)params->clip_sigma,
                             (float)params->clip_ampfrac, out_img, out_wgt, 0);
}

extern "C" int zm_coadd_dev(zm_ctx* ctx, int nframes, const zm_dframe* frames,
                            const zm_wcs* wout, const zm_coadd_params* params, int partial,
                            float* out_img, float* out_wgt, int32_t* out_mask,
                            float* out_mask_wgt) {
    ZM_CHECK(ctx && frames && wout && params && out_img && out_wgt, "zm_coadd_dev: null argument");
    ZM_CHECK(nframes >= 1, "zm_coadd_dev: need at least one frame");
    ZM_HIP(hipSetDevice(ctx->device));
    ZM_TRY(check_wcs(wout, "output grid"));
    const int64_t opix = (int64_t)wout->naxis[0] * wout->naxis[1];
    if (fused_ok(ctx, params))
        return coadd_fused(ctx, nframes, frames, wout, params, partial, out_img, out_wgt, out_mask, out_mask_wgt);
    float2* stack = nullptr;
    ZM_TRY(ctx->get("stack", sizeof(float2) * (size_t)opix * nframes, (void**)&stack));
    ZM_TRY(resample_frames(ctx, nframes, frames, wout, params, stack, out_mask, -1,
                           params->mask_combine));
    ZM_TRY(zm_launch_combine(ctx, nframes, stack, opix, opix, params->combine,
                             (float)params->clip_sigma, (float)params->clip_ampfrac, out_img,
                             out_wgt, partial));
    // a partial coadd leaves the "nothing covered yet" marker (-1) in place: the caller folds the
    // partial masks of the other ranks in (zm_mask_accum_dev) and finalises (zm_mask_finalize_dev)
    if (out_mask && !partial) ZM_TRY(zm_launch_mask_finalize(ctx, out_mask, out_mask_wgt, opix));
    return 0;
}

extern "C" int zm_mask_accum_dev(zm_ctx* ctx, int32_t* acc, const int32_t* m, int64_t npix, int kind,
                                 int first) {
    ZM_CHECK(ctx && acc && m, "zm_mask_accum_dev: null argument");
    ZM_CHECK(kind == ZM_MASK_AND || kind == ZM_MASK_OR, "zm_mask_accum_dev: unknown combine %d", kind);
    ZM_HIP(hipSetDevice(ctx->device));
    return zm_launch_mask_accum(ctx, acc, m, npix, kind, first);
}

extern "C" int zm_mask_finalize_dev(zm_ctx* ctx, int32_t* acc, float* cov, int64_t npix) {
    ZM_CHECK(ctx && acc, "zm_mask_finalize_dev: null argument");
    ZM_HIP(hipSetDevice(ctx->device));
    return zm_launch_mask_finalize(ctx, acc, cov, npix);
}

extern "C" int zm_resample_dev(zm_ctx* ctx, const float* img, const float* wgt,
                               const int32_t* mask, const zm_wcs* win, const zm_wcs* wout,
                               int kernel, double fscale, float* out_img, float* out_wgt,
                               int32_t* out_mask) {
    ZM_CHECK(ctx && win && wout, "zm_resample_dev: null argument");
    ZM_CHECK(img || mask, "zm_resample_dev: need an image or a mask");
    ZM_CHECK(!img || (out_img && out_wgt), "zm_resample_dev: out_img/out_wgt required");
    ZM_CHECK(!mask || out_mask, "zm_resample_dev: out_mask required");
    ZM_HIP(hipSetDevice(ctx->device));
    ZM_TRY(check_wcs(win, "input frame"));
    ZM_TRY(check_wcs(wout, "output grid"));
    ZM_CHECK(kernel == ZM_RESAMPLE_LANCZOS3 || kernel == ZM_RESAMPLE_BILINEAR ||
             kernel == ZM_RESAMPLE_NEAREST, "zm_resample_dev: unknown RESAMPLING_TYPE %d", kernel);
    const int onx = wout->naxis[0], ony = wout->naxis[1];
    const int nx = win->naxis[0], ny = win->naxis[1];
    const int64_t opix = (int64_t)onx * ony;
    const int lnx = (onx - 1) / ZM_LATTICE_STEP + 2, lny = (ony - 1) / ZM_LATTICE_STEP + 2;
    zm_map_params mp_host;
    zm_make_map(wout, win, &mp_host);
    int lds = std::min(plan_lds(&mp_host, onx, ony, ntaps_of(kernel)), 8000);
    double2* lat = nullptr;
    ZM_TRY(ctx->get("lattice", sizeof(double2) * (size_t)lnx * lny, (void**)&lat));
    ZM_TRY(zm_launch_lattice(ctx, &mp_host, lnx, lny, lat));
    if (img) {
        const int spitch = (nx + 1) & ~1;
        float2 *src = nullptr, *dst = nullptr;
        ZM_TRY(ctx->get("prep", sizeof(float2) * (size_t)spitch * ny, (void**)&src));
        ZM_TRY(ctx->get("stack", sizeof(float2) * (size_t)opix, (void**)&dst));
        ZM_TRY(zm_launch_prep(ctx, img, wgt, nx, ny, nullptr, 0, 0, 0, nullptr, 1e-30f, src, spitch));
        // (value and weight leave the resample kernel as the two planes the caller asked for)
        ZM_TRY(zm_launch_resample(ctx, src, nx, ny, spitch, lat, lnx, lny, kernel, (float)fscale,
                                  dst, onx, ony, lds, mask, out_mask, mask ? 1 : 0, 0, 1, out_img, out_wgt));
    } else if (mask) {
        ZM_TRY(zm_launch_resample_mask(ctx, mask, nx, ny, lat, lnx, lny, kernel, out_mask, onx,
                                       ony, 0));
    }
    return 0;
}

// Two images on one geometry through ONE resampling launch: img_a (with its mask, optionally) and img_b go from `win`
// to `wout` as zm_resample_dev would take each (no weights: WEIGHT_TYPE NONE, zuds/swarp.py:143-152), out_b scaled by
// fscale_b.  The reference aligns its reference image and that image's rms map to the science grid in two SWarp runs
// (zuds/subtraction.py:109 -> zuds/fitsfile.py:290-314, zuds/hotpants.py:51): same positions, same taps.  Per pixel the
// values are those of the two separate calls, bit for bit (tests/test_device_chain_gpu.py); no weight planes come back
// (uncovered pixels are 0 in both outputs).  Default conventions only (zm_ctx_set_conventions).
extern "C" int zm_align_pair_dev(zm_ctx* ctx, const float* img_a, const float* img_b, const int32_t* mask,
                                 const zm_wcs* win, const zm_wcs* wout, int kernel, double fscale_a, double fscale_b,
                                 float* out_a, float* out_b, int32_t* out_mask) {
    ZM_CHECK(ctx && img_a && img_b && win && wout && out_a && out_b, "zm_align_pair_dev: null argument");
    ZM_CHECK(!mask || out_mask, "zm_align_pair_dev: out_mask required");
    ZM_CHECK(kernel == ZM_RESAMPLE_LANCZOS3 || kernel == ZM_RESAMPLE_BILINEAR, "zm_align_pair_dev: LANCZOS3 or BILINEAR (got %d)", kernel);
    ZM_CHECK(fscale_b != 0.0, "zm_align_pair_dev: fscale_b must not be 0");
    ZM_CHECK(ctx->edge == ZM_EDGE_ZERO && ctx->mask_resample == ZM_MASKRES_OR, "zm_align_pair_dev: default conventions only");
    ZM_HIP(hipSetDevice(ctx->device));
    ZM_TRY(check_wcs(win, "input frame"));
    ZM_TRY(check_wcs(wout, "output grid"));
    const int onx = wout->naxis[0], ony = wout->naxis[1];
    const int nx = win->naxis[0], ny = win->naxis[1];
    const int lnx = (onx - 1) / ZM_LATTICE_STEP + 2, lny = (ony - 1) / ZM_LATTICE_STEP + 2;
    zm_map_params mp_host;
    zm_make_map(wout, win, &mp_host);
    const int lds = std::min(plan_lds(&mp_host, onx, ony, ntaps_of(kernel)), 8000);
    double2* lat = nullptr;
    ZM_TRY(ctx->get("lattice", sizeof(double2) * (size_t)lnx * lny, (void**)&lat));
    ZM_TRY(zm_launch_lattice(ctx, &mp_host, lnx, lny, lat));
    const int spitch = (nx + 1) & ~1;
    float2* src = nullptr;
    ZM_TRY(ctx->get("prep", sizeof(float2) * (size_t)spitch * ny, (void**)&src));
    ZM_TRY(zm_launch_prep_pair(ctx, img_a, img_b, nx, ny, src, spitch));
    return zm_launch_resample(ctx, src, nx, ny, spitch, lat, lnx, lny, kernel, (float)fscale_a, nullptr, onx, ony, lds,
                              mask, out_mask, mask ? 1 : 0, 0, 1, out_a, out_b, (float)fscale_b);
}

extern "C" int zm_resample_i16_dev(zm_ctx* ctx, const float* img, const float* wgt, const int16_t* mask,
                                   const zm_wcs* win, const zm_wcs* wout, int kernel, double fscale,
                                   float* out_img, float* out_wgt, int32_t* out_mask) {
    ZM_CHECK(ctx && win && wout, "zm_resample_i16_dev: null argument");
    ZM_HIP(hipSetDevice(ctx->device));
    int32_t* w = nullptr;
    if (mask) {
        ZM_TRY(check_wcs(win, "input frame"));
        const int64_t n = (int64_t)win->naxis[0] * win->naxis[1];
        ZM_TRY(ctx->get("mask_widen", sizeof(int32_t) * (size_t)n, (void**)&w));
        ZM_TRY(zm_launch_mask_widen(ctx, mask, n, w));
    }
    return zm_resample_dev(ctx, img, wgt, w, win, wout, kernel, fscale, out_img, out_wgt, out_mask);
}

// ---- host-pointer flavours -------------------------------------------------------
static int resample_host(zm_ctx* ctx, const float* img, const float* wgt, const void* mask, int mask_type,
                         const zm_wcs* win, const zm_wcs* wout, int kernel, double fscale,
                         float* out_img, float* out_wgt, int32_t* out_mask);
extern "C" int zm_resample(zm_ctx* ctx, const float* img, const float* wgt, const int32_t* mask,
                           const zm_wcs* win, const zm_wcs* wout, int kernel, double fscale,
                           float* out_img, float* out_wgt, int32_t* out_mask) {
    return resample_host(ctx, img, wgt, mask, ZM_MASKTYPE_I32, win, wout, kernel, fscale, out_img, out_wgt, out_mask);
}
extern "C" int zm_resample_i16(zm_ctx* ctx, const float* img, const float* wgt, const int16_t* mask,
                               const zm_wcs* win, const zm_wcs* wout, int kernel, double fscale,
                               float* out_img, float* out_wgt, int32_t* out_mask) {
    return resample_host(ctx, img, wgt, mask, ZM_MASKTYPE_I16, win, wout, kernel, fscale, out_img, out_wgt, out_mask);
}
static int resample_host(zm_ctx* ctx, const float* img, const float* wgt, const void* mask, int mask_type,
                         const zm_wcs* win, const zm_wcs* wout, int kernel, double fscale,
                         float* out_img, float* out_wgt, int32_t* out_mask) {
    ZM_CHECK(ctx && win && wout, "zm_resample: null argument");
    ZM_HIP(hipSetDevice(ctx->device));
    ZM_TRY(check_wcs(win, "input frame"));
    ZM_TRY(check_wcs(wout, "output grid"));
    const size_t ipix = (size_t)win->naxis[0] * win->naxis[1];
    const size_t opix = (size_t)wout->naxis[0] * wout->naxis[1];
    const size_t mb = mask_type == ZM_MASKTYPE_I16 ? 2 : 4;         // bytes per mask pixel over PCIe
    float *d_img = nullptr, *d_wgt = nullptr, *d_oimg = nullptr, *d_owgt = nullptr;
    void* d_mask = nullptr;
    int32_t* d_omask = nullptr;
    if (img) {
        ZM_TRY(ctx->get("h_img", ipix * 4, (void**)&d_img));
        ZM_HIP(hipMemcpyAsync(d_img, img, ipix * 4, hipMemcpyHostToDevice, ctx->stream));
        if (wgt) {
            ZM_TRY(ctx->get("h_wgt", ipix * 4, (void**)&d_wgt));
            ZM_HIP(hipMemcpyAsync(d_wgt, wgt, ipix * 4, hipMemcpyHostToDevice, ctx->stream));
        }
        ZM_TRY(ctx->get("h_oimg", opix * 4, (void**)&d_oimg));
        ZM_TRY(ctx->get("h_owgt", opix * 4, (void**)&d_owgt));
    }
    if (mask) {
        ZM_TRY(ctx->get("h_mask", ipix * mb, &d_mask));
        ZM_HIP(hipMemcpyAsync(d_mask, mask, ipix * mb, hipMemcpyHostToDevice, ctx->stream));
        ZM_TRY(ctx->get("h_omask", opix * 4, (void**)&d_omask));
    }
    if (mask_type == ZM_MASKTYPE_I16)
        ZM_TRY(zm_resample_i16_dev(ctx, d_img, d_wgt, (const int16_t*)d_mask, win, wout, kernel, fscale, d_oimg,
                                   d_owgt, d_omask));
    else
        ZM_TRY(zm_resample_dev(ctx, d_img, d_wgt, (const int32_t*)d_mask, win, wout, kernel, fscale, d_oimg, d_owgt,
                               d_omask));
    if (img) {
        ZM_HIP(hipMemcpyAsync(out_img, d_oimg, opix * 4, hipMemcpyDeviceToHost, ctx->stream));
        ZM_HIP(hipMemcpyAsync(out_wgt, d_owgt, opix * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (mask)
        ZM_HIP(hipMemcpyAsync(out_mask, d_omask, opix * 4, hipMemcpyDeviceToHost, ctx->stream));
    ZM_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int zm_coadd(zm_ctx* ctx, int nframes, const zm_frame* frames, const zm_wcs* wout,
                        const zm_coadd_params* params, float* out_img, float* out_wgt,
                        int32_t* out_mask, float* out_mask_wgt) {
    ZM_CHECK(ctx && frames && wout && params && out_img && out_wgt, "zm_coadd: null argument");
    ZM_CHECK(nframes >= 1, "zm_coadd: need at least one frame");
    ZM_HIP(hipSetDevice(ctx->device));
    ZM_TRY(check_wcs(wout, "output grid"));
    std::vector<zm_dframe> df(nframes);
    size_t total = 0;
    for (int i = 0; i < nframes; ++i) {
        ZM_TRY(check_wcs(&frames[i].wcs, "frame"));
        ZM_CHECK(frames[i].img != nullptr, "zm_coadd: frame %d has no image", i);
        size_t ip = (size_t)frames[i].wcs.naxis[0] * frames[i].wcs.naxis[1];
        ZM_CHECK(!frames[i].mask || frames[i].mask_type == ZM_MASKTYPE_I32 || frames[i].mask_type == ZM_MASKTYPE_I16,
                 "zm_coadd: frame %d: unknown mask_type %d", i, frames[i].mask_type);
        total += ip * 4 * (1 + (frames[i].wgt ? 1 : 0));
        if (frames[i].mask && out_mask) total += ip * (frames[i].mask_type == ZM_MASKTYPE_I16 ? 2 : 4);
        total = (total + 255) & ~(size_t)255;
    }
    char* base = nullptr;
    ZM_TRY(ctx->get("h_frames", total, (void**)&base));
    size_t off = 0;
    for (int i = 0; i < nframes; ++i) {
        size_t ip = (size_t)frames[i].wcs.naxis[0] * frames[i].wcs.naxis[1];
        df[i].wcs = frames[i].wcs;
        df[i].flxscale = frames[i].flxscale;
        df[i].img = (const float*)(base + off);
        ZM_HIP(hipMemcpyAsync(base + off, frames[i].img, ip * 4, hipMemcpyHostToDevice, ctx->stream));
        off += ip * 4;
        df[i].wgt = nullptr;
        if (frames[i].wgt) {
            df[i].wgt = (const float*)(base + off);
            ZM_HIP(hipMemcpyAsync(base + off, frames[i].wgt, ip * 4, hipMemcpyHostToDevice, ctx->stream));
            off += ip * 4;
        }
        df[i].mask = nullptr;
        df[i].mask_type = frames[i].mask_type;
        df[i].pad_ = 0;
        if (frames[i].mask && out_mask) {
            // (an int16 plane - a ZTF mask as it lies on disk - crosses PCIe as it is)
            const size_t mbytes = ip * (frames[i].mask_type == ZM_MASKTYPE_I16 ? 2 : 4);
            df[i].mask = base + off;
            ZM_HIP(hipMemcpyAsync(base + off, frames[i].mask, mbytes, hipMemcpyHostToDevice, ctx->stream));
            off += mbytes;
        }
        off = (off + 255) & ~(size_t)255;
    }
    const size_t opix = (size_t)wout->naxis[0] * wout->naxis[1];
    float *d_oimg = nullptr, *d_owgt = nullptr, *d_omw = nullptr;
    int32_t* d_omask = nullptr;
    ZM_TRY(ctx->get("h_oimg", opix * 4, (void**)&d_oimg));
    ZM_TRY(ctx->get("h_owgt", opix * 4, (void**)&d_owgt));
    if (out_mask) ZM_TRY(ctx->get("h_omask", opix * 4, (void**)&d_omask));
    if (out_mask && out_mask_wgt) ZM_TRY(ctx->get("h_omw", opix * 4, (void**)&d_omw));
    ZM_TRY(zm_coadd_dev(ctx, nframes, df.data(), wout, params, 0, d_oimg, d_owgt, d_omask, d_omw));
    ZM_HIP(hipMemcpyAsync(out_img, d_oimg, opix * 4, hipMemcpyDeviceToHost, ctx->stream));
    ZM_HIP(hipMemcpyAsync(out_wgt, d_owgt, opix * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (out_mask)
        ZM_HIP(hipMemcpyAsync(out_mask, d_omask, opix * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (out_mask && out_mask_wgt)
        ZM_HIP(hipMemcpyAsync(out_mask_wgt, d_omw, opix * 4, hipMemcpyDeviceToHost, ctx->stream));
    ZM_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}
