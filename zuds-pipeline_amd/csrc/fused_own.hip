#include "fused_dev.h"

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


ZM_FF_DEFINE_LAUNCH(zm_ff_launch_own, k_coadd_fused_own)
