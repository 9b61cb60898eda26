#include "fused_dev.h"

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


ZM_FF_DEFINE_LAUNCH(zm_ff_launch_dma, k_coadd_fused_dma)
