// Device-side pieces shared by the two fused resample -> coadd kernels (fused_dma.hip, fused_own.hip) and the
// pre-passes that build their item headers (fused_host.hip).
#pragma once
#include "resample_dev.h"

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

// Tile shape: 64 x 32 output pixels (one k_resample tile), two workgroups of 512 threads per CU.  (Rounds 2 - 3 also
// had a register-staged kernel with 64 x 64 tiles, FF_TALL; it lost to the LDS-DMA staged forms and left in round 6.)
#define FF_TALL 0
#define FF_NSUB (FF_TALL ? 2 : 1)    // k_resample tiles (64 x 32) stacked in a fused tile
#define FT_H (RTH * FF_NSUB)         // output rows of a fused tile
#define FF_WG_PER_CU 2
#define FF_HDR_WORDS (FF_TALL ? 80 : 48)
#define FF_LDS_HDR 1152              // bytes: 3 headers, tile ring, raw-mask flags
#define FF_LDS_TAB ((LZ_FLOATS * 4 + 127) & ~127)
// the DMA-staged kernels (fused_dma.hip, fused_own.hip)
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
        // what the staging holds (dma == 1, k_coadd_fused_dma: rows of the y table, columns of the x-weight table;
        // dma == 2, k_coadd_fused_own: slots of at most 80 x 42 pixels at a fixed pitch)
        const int use_lds = touches && area <= (long long)lds_cap && bw >= 8 && bh >= 1 &&
                            (dma == 2 ? (bw <= 80 && bh <= 42) : (bw <= FD_XCOLS && bh <= FD_YROWS));
        const int inside = bx0 >= 0 && by0 >= 0 && bx1 <= nx && by1 <= ny && (txi + 1) * TW <= onx &&
                           (tyi + 1) * FT_H <= ony;
        H->bx0 = bx0; H->by0 = by0; H->bw = bw; H->bh = bh;
        H->touches = touches;
        H->use_lds = use_lds;
        // fast: the box lies on the frame and the tile on the grid - no bounds test anywhere.  edge (staged,
        // not fast): what lies off the frame becomes {0, BIGVAR} at the store.
        H->fast = use_lds && inside;
        H->vscale = (F->vscale && !skip_vscale) ? *F->vscale : 1.f;     // (skip_vscale: k_ff_vscale fills it in later)
        // (skip_vscale = the early headers: the box-OR planes may not be made yet - k_ff_vscale fills the flag in too)
        H->frame_raw = skip_vscale ? 1 : (F->mboxflag ? *F->mboxflag : 1);
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

#define FD_OFF_XW (FF_LDS_HDR + FF_LDS_TAB)
#define FD_OFF_YT (FD_OFF_XW + FD_XCOLS * 16)
#define FD_OFF_RAW (FD_OFF_YT + FD_YCOLS * FD_YROWS * 16)
static_assert(FD_OFF_RAW + 20 * FD_LDS_CAP + 4 * 8 * FD_YROWS <= 80 * 1024, "DMA-staged kernel: LDS budget of half a CU");

__device__ inline void ff_glds16(const void ZM_GLOBAL* src, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}


// ---- LDS layout of the owner-staged kernel (fused_own.hip; the host sizes its launch from FO_LDS) --------------------
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

// ---- the launch of one fused kernel, across translation units ------------------------------------------------------
// fused_host.hip picks the form and the geometry; the kernel's own translation unit holds its instances and the
// switch over (mask operator, AVERAGE, STACK) that selects one.  `devk`: the developer instance (phase clocks,
// ablations) - only in a -DZM_DEV build.
struct ff_launch_args {
    const zm_ff* fr;
    int nfr, onx, ony, lds_cap, ntx, ntiles;
    const int* ghdr;
    float *out_img, *out_wgt;
    int32_t* out_mask;
    float* out_cov;
    int partial;
    const float* taptab;
    int* tilectr;
    float2* stack;
    long long fstride;
    int dbg;
    long long* prof;
};
int zm_ff_launch_dma(zm_ctx* ctx, int mop, bool avg, bool stack, bool devk, int G, size_t shmem, const ff_launch_args& a);
int zm_ff_launch_own(zm_ctx* ctx, int mop, bool avg, bool stack, bool devk, int G, size_t shmem, const ff_launch_args& a);

#define ZM_FF_DEFINE_LAUNCH(NAME, KERNEL)                                                                              \
    template <int MOPV, bool AVGV, bool STACKV>                                                                        \
    static int NAME##_one(zm_ctx* ctx, bool devk, int G, size_t shmem, const ff_launch_args& a) {                      \
        auto k = KERNEL<MOPV, AVGV, STACKV, false>;                                                                    \
        ZM_FF_DEV_PICK(KERNEL)                                                                                         \
        static bool attr[2][64] = {};                                                                                  \
        if (!attr[devk][ctx->device & 63]) {                                                                           \
            ZM_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));        \
            attr[devk][ctx->device & 63] = true;                                                                       \
        }                                                                                                              \
        hipLaunchKernelGGL(k, dim3(G), dim3(FD_THREADS), shmem, ctx->stream, a.fr, a.nfr, a.onx, a.ony, a.lds_cap,     \
                           a.ntx, a.ntiles, a.ghdr, a.out_img, a.out_wgt, a.out_mask, a.out_cov, a.partial, a.taptab,  \
                           a.tilectr, a.stack, a.fstride, a.dbg, a.prof);                                              \
        return 0;                                                                                                      \
    }                                                                                                                  \
    int NAME(zm_ctx* ctx, int mop, bool avg, bool stack, bool devk, int G, size_t shmem, const ff_launch_args& a) {    \
        if (stack) {                                                                                                   \
            if (mop == 0) return NAME##_one<0, false, true>(ctx, devk, G, shmem, a);                                   \
            if (mop == 1) return NAME##_one<1, false, true>(ctx, devk, G, shmem, a);                                   \
            return NAME##_one<2, false, true>(ctx, devk, G, shmem, a);                                                 \
        }                                                                                                              \
        if (mop == 0) return avg ? NAME##_one<0, true, false>(ctx, devk, G, shmem, a) : NAME##_one<0, false, false>(ctx, devk, G, shmem, a); \
        if (mop == 1) return avg ? NAME##_one<1, true, false>(ctx, devk, G, shmem, a) : NAME##_one<1, false, false>(ctx, devk, G, shmem, a); \
        return avg ? NAME##_one<2, true, false>(ctx, devk, G, shmem, a) : NAME##_one<2, false, false>(ctx, devk, G, shmem, a); \
    }
#ifdef ZM_DEV
#define ZM_FF_DEV_PICK(KERNEL) if (devk) k = KERNEL<MOPV, AVGV, STACKV, true>;
#else
#define ZM_FF_DEV_PICK(KERNEL) (void)devk;
#endif
