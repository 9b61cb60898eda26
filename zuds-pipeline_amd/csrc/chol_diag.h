// The 32 x 32 diagonal factor of the blocked Cholesky (hotpants.hip: k_chol_fused, k_chol_fused2, k_chol_tp all
// call the same function, so the three forms give the same bits) - in a header of its own so that
// tools/potrf_probe.hip times and checks exactly what the library runs.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

#ifndef CH_NB
#define CH_NB 32
#endif

typedef double chd_double4 __attribute__((ext_vector_type(4)));

__device__ inline double readlane_d(double v, int src) {
    // src is wave-uniform (a compile-time constant after unrolling): v_readlane_b32 x 2
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)u, src);
    const unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(u >> 32), src);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

// The rank-8 update of the block's trailing part after panel p (columns c0 .. c1 - 1 are final and in D):
// two v_mfma_f64_16x16x4 per 16 x 16 tile of the lower triangle that still has unfactored columns, through
// LDS.  One wave; LDS operations of a wave execute in order, so the panel written by the lanes is what the
// matrix-core operands read back.
__device__ __forceinline__ void chol_diag_trailing(double (*D)[CH_NB + 1], int c0, int c1, int li, int lk) {
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj <= ti; ++tj) {
            if (16 * (ti + 1) <= c1 || 16 * (tj + 1) <= c1) continue;     // tile above / left of the trailing part
            chd_double4 c4;
#pragma unroll
            for (int q = 0; q < 4; ++q) c4[q] = D[16 * ti + lk + 4 * q][16 * tj + li];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const double av = -D[16 * ti + li][c0 + 4 * kk + lk];
                const double bv = D[16 * tj + li][c0 + 4 * kk + lk];
                c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c4, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = 16 * ti + lk + 4 * q, jc = 16 * tj + li;
                if (i >= c1 && jc >= c1 && jc <= i) D[i][jc] = c4[q];
            }
        }
}

// The same update with its store conditions spelled out per tile: c1 is a multiple of 8, so "row >= c1" is a
// condition on the accumulator register q alone (row = 16 ti + lk + 4 q, lk < 4) and "column >= c1" is either
// always / never true for a tile or the one lane predicate li >= 8.  Entries above the diagonal are written
// too (nobody reads them).  As three-term runtime conditions per (tile, q) they were 24 loop-invariant 64-bit
// masks: hoisted out of the block loop, spilled to VGPR lanes and reloaded (~100 v_readlane per factor).
__device__ __forceinline__ void chol_diag_trailing_static(double (*D)[CH_NB + 1], int c0, int c1, int li, int lk) {
    int liv = li;
    asm volatile("" : "+v"(liv));                          // (made here: one v_cmp instead of a mask kept across the loop)
    const bool right = liv >= 8;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj <= ti; ++tj) {
            if (16 * (ti + 1) <= c1 || 16 * (tj + 1) <= c1) continue;     // tile above / left of the trailing part
            chd_double4 c4;
#pragma unroll
            for (int q = 0; q < 4; ++q) c4[q] = D[16 * ti + lk + 4 * q][16 * tj + li];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const double av = -D[16 * ti + li][c0 + 4 * kk + lk];
                const double bv = D[16 * tj + li][c0 + 4 * kk + lk];
                c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c4, 0, 0, 0);
            }
            const int cl = c1 - 16 * tj;                   // columns li >= cl of this tile trail the panel: <= 0 (all) or 8
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (16 * ti + 4 * q < c1) continue;        // rows 16 ti + 4 q .. + 3 lie above the trailing part
                const int i = 16 * ti + lk + 4 * q, jc = 16 * tj + li;
                if (cl <= 0 || right) D[i][jc] = c4[q];
            }
        }
}

// Rounds 1 - 3: four panels of eight columns.  A column step: the pivot is broadcast from its lane
// (v_readlane), 1 / sqrt by the hardware estimate + two Newton steps (full fp64 accuracy, no divide), the
// column is scaled, broadcast lane by lane for the rank-1 update of the panel's remaining columns (2 readlanes +
// 1 FMA per column); the rest of the block gets the panel's rank-8 update on the matrix cores.  The 32 dependent
// column steps are the critical path of every form of the factorisation: ~150 ns each, 4.7 us per block.
template <int TAG>
__device__ inline void chol_diag_wave_panel_ref(double (*D)[CH_NB + 1], int nb, int* fail) {
    const int lane = threadIdx.x & 63;
    const int row = lane & 31, li = lane & 15, lk = lane >> 4;
    int bad = 0;
    double dinv = 1.0;                                   // 1 / L[row][row]
#pragma unroll
    for (int p = 0; p < CH_NB / 8; ++p) {
        const int c0 = 8 * p, c1 = c0 + 8;
        double a[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) a[c] = D[row][c0 + c];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            double ajj = readlane_d(a[j], c0 + j);
            if (!(ajj > 1e-14)) { ajj = 1e-14; bad = 1; }
            double ri = __builtin_amdgcn_rsq(ajj);
            ri = ri * (1.5 - 0.5 * ajj * ri * ri);
            ri = ri * (1.5 - 0.5 * ajj * ri * ri);
            const double dj = ajj * ri;
            const double lj = (row == c0 + j) ? dj : a[j] * ri;   // L[row][c0 + j] (meaningful for row >= c0 + j)
            if (row == c0 + j) dinv = ri;
            a[j] = lj;
#pragma unroll
            for (int c = j + 1; c < 8; ++c) {
                const double lc = readlane_d(lj, c0 + c);
                a[c] -= lj * lc;
            }
        }
        if (lane < CH_NB) {
#pragma unroll
            for (int c = 0; c < 8; ++c) D[row][c0 + c] = a[c];    // (above the diagonal: finite values nobody reads)
        }
        if (c1 < CH_NB) chol_diag_trailing(D, c0, c1, li, lk);
    }
    if (lane < CH_NB) D[row][CH_NB] = dinv;
    if (bad && lane == 0 && fail) atomicAdd(fail, 1);
}

// Round 4: the same panels with a shorter column step.  What a step waits for, in order: the pivot's broadcast,
// its reciprocal square root, the scaled column, the update of the NEXT pivot.  Three changes, all on that chain:
//  * every lane keeps its own diagonal entry (dg) up to date beside the panel columns - `dg -= l l` needs only
//    the lane's own l, so the next pivot is ready one broadcast (v_readlane -> SGPR -> FMA) earlier; the panel
//    columns themselves still take their rank-1 updates, off the chain;
//  * one third-order correction of the hardware estimate y0 instead of two second-order ones:
//    e = 1 - a y0^2, 1 / sqrt(a) = y0 (1 + e (1/2 + 3/8 e)) to 2.5 eps0^3 (v_rsq_f64: eps0 ~ 2^-23 .. 2^-26,
//    tools/potrf_probe.hip measures it; two Newton steps were needed because ONE second-order step stops at
//    1.5 eps0^2) - four dependent operations instead of six;
//  * the column scale rides in the correction: l = (a_ij y0) + (a_ij y0 e)(1/2 + 3/8 e), ready with the root;
//  * the pivot clamp is a v_max_f64 (the bad-pivot flag is computed beside the chain).
// Same operation count per entry elsewhere, same matrix-core updates; the factor differs from the old one by
// rounding (<= 1 ulp per entry before propagation), all forms of the factorisation share it.
// Measured (tools/potrf_probe.hip, one wave, MI355X): 4.15 -> 3.85 us per block, ~9 200 shader clocks for ~1 750
// instructions: a lone wave issues one vector instruction per ~5 cycles whatever depends on what, so the factor is
// bound by its instruction COUNT on one SIMD (435 v_readlane among them), not by the chain any more.  Tried and
// not kept: the column steps with only lanes 0 - 31 active (the upper half duplicates the rows): the same code
// ran 3.4 us in some launches and 3.9 - 4.1 us in others.
template <int TAG>
__device__ __forceinline__ void chol_diag_wave_panel_fast(double (*D)[CH_NB + 1], int nb, int* fail) {
    const int lane = threadIdx.x & 63;
    const int row = lane & 31, li = lane & 15, lk = lane >> 4;
    double amin = 1.0;                                   // smallest clamped pivot: 1e-14 iff one was clamped
    double dinv = 1.0;                                   // 1 / L[row][row]
#pragma unroll
    for (int p = 0; p < CH_NB / 8; ++p) {
        const int c0 = 8 * p, c1 = c0 + 8;
        double a[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) a[c] = D[row][c0 + c];
        double dg = D[row][row];                         // (a pivot of THIS panel only when c0 <= row < c1)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // (the pivot lane's predicate is made here, per column: hoisted, the 32 of them live in 64 scalar
            // registers that spill to VGPR lanes - 150 v_readlane reloads on the path)
            int rowj = row;
            asm volatile("" : "+v"(rowj));
            const bool pivot_lane = rowj == c0 + j;
            const double piv = readlane_d(dg, c0 + j);
            double ajj;                                             // max(piv, 1e-14), NaN -> 1e-14: ONE v_max_f64
            asm("v_max_f64 %0, %1, %2" : "=v"(ajj) : "v"(1e-14), "s"(piv));   // (fmax() puts a canonicalising max in front)
            amin = __builtin_fmin(amin, ajj);                       // (beside the chain; as predicates the 32 pivots stay in SGPRs and spill)
            const double y0 = __builtin_amdgcn_rsq(ajj);
            const double l0 = a[j] * y0;                            // beside the correction
            const double t = ajj * y0;
            const double e = __builtin_fma(-t, y0, 1.0);
            const double pc = __builtin_fma(0.375, e, 0.5);
            const double ye = y0 * e, le = l0 * e;
            const double ri = __builtin_fma(ye, pc, y0);
            const double lo = __builtin_fma(le, pc, l0);            // a[j] / sqrt(ajj)
            const double lj = pivot_lane ? ajj * ri : lo;           // L[row][c0 + j] (meaningful for row >= c0 + j)
            dg = __builtin_fma(-lo, lo, dg);                        // the diagonal entry of a row below the pivot
            dinv = pivot_lane ? ri : dinv;
            a[j] = lj;
#pragma unroll
            for (int c = j + 1; c < 8; ++c) {
                const double lc = readlane_d(lj, c0 + c);
                a[c] = __builtin_fma(-lj, lc, a[c]);
            }
        }
        if (lane < CH_NB) {
#pragma unroll
            for (int c = 0; c < 8; ++c) D[row][c0 + c] = a[c];    // (above the diagonal: finite values nobody reads)
        }
        if (c1 < CH_NB) chol_diag_trailing_static(D, c0, c1, li, lk);
    }
    if (lane < CH_NB) D[row][CH_NB] = dinv;
    if (!(amin > 1e-14) && lane == 0 && fail) atomicAdd(fail, 1);
}

// Round 5: the same factor with the broadcasts on the data-parallel-primitive path.  A lone wave issues one
// instruction per ~5 cycles whatever it is, so the factor's time is its instruction count - and a broadcast through
// the scalar file is three instructions per rank-1 update (two v_readlane_b32 and the FMA).  gfx950 has DPP forms of
// the fp64 move and the fp64 fused multiply-add (v_mov_b64_dpp, v_fmac_f64_dpp) with `row_newbcast:k` - lane k of
// every ROW OF 16 LANES to the 16 lanes of that row.  So the lanes are dealt anew for every panel: lanes 0 - 7 of
// each of the four rows all hold the panel's eight matrix rows (four identical copies: the values every update
// broadcasts are then present in every row of lanes), lanes 8 - 15 hold eight of the up to 24 matrix rows below
// the panel.  A rank-1 update is ONE instruction, the pivot's broadcast one instead of two.  Per entry the
// operations and their order are those of chol_diag_wave_panel_fast: the same bits (tools/potrf_probe.hip compares).
// Hazard: a DPP operand written by one of the two preceding vector instructions is read stale (no interlock);
// the asm statements carry their own s_nop where the producer can be that close.
template <int SRC, bool NOP>
__device__ __forceinline__ double chd_bcast16(double v) {
    double r;
    if (NOP)
        asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(SRC));
    else
        asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(SRC));
    return r;
}
// acc -= (lane SRC's b) * m      (NOP: b may have been written by the instruction before)
template <int SRC, bool NOP>
__device__ __forceinline__ void chd_fnma_bcast16(double& acc, double b, double m) {
    if (NOP)
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(m), "n"(SRC));
    else
        asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(m), "n"(SRC));
}
template <int I, int N, typename Fn>
__device__ __forceinline__ void chd_static_for(Fn&& fn) {
    if constexpr (I < N) {
        fn(std::integral_constant<int, I>{});
        chd_static_for<I + 1, N>(fn);
    }
}
template <int TAG>
__device__ __forceinline__ void chol_diag_wave_panel_dpp(double (*D)[CH_NB + 1], int nb, int* fail) {
    const int lane = threadIdx.x & 63;
    const int q = lane & 15, r = lane >> 4;              // lane q of lane row r (also the matrix-core roles li, lk)
    double amin = 1.0;                                   // smallest clamped pivot: 1e-14 iff one was clamped
    chd_static_for<0, CH_NB / 8>([&](auto P) __attribute__((always_inline)) {
        constexpr int p = decltype(P)::value;
        constexpr int c0 = 8 * p, c1 = c0 + 8;
        // this panel's deal: its own rows in lanes 0 - 7 of every lane row, the rows below it in lanes 8 - 15
        const int row = q < 8 ? c0 + q : c1 + 8 * r + (q - 8);
        const bool live = row < CH_NB;                   // (idle lanes work on a copy of the last row, unstored)
        const int rr = live ? row : CH_NB - 1;
        double a[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) a[c] = D[rr][c0 + c];
        double dg = D[rr][rr];                           // (a pivot of THIS panel only in lanes 0 - 7)
        double dinv = 1.0;
        chd_static_for<0, 8>([&](auto J) __attribute__((always_inline)) {
            constexpr int j = decltype(J)::value;
            int qj = q;
            asm volatile("" : "+v"(qj));                 // (the pivot lane's predicate made per column: see the fast form)
            const bool pivot_lane = qj == j;
            // (dg was written by the asm statement of the column before and the updates between are asm statements too:
            // at least two of them for j <= 6, one for j = 7; at j = 0 it comes from LDS)
            const double piv = chd_bcast16<j, j == 7>(dg);
            double ajj;                                             // max(piv, 1e-14), NaN -> 1e-14: ONE v_max_f64
            asm("v_max_f64 %0, %1, %2" : "=v"(ajj) : "v"(1e-14), "v"(piv));
            amin = __builtin_fmin(amin, ajj);
            const double y0 = __builtin_amdgcn_rsq(ajj);
            const double l0 = a[j] * y0;                            // beside the correction
            const double t = ajj * y0;
            const double e = __builtin_fma(-t, y0, 1.0);
            const double pc = __builtin_fma(0.375, e, 0.5);
            const double ye = y0 * e, le = l0 * e;
            const double ri = __builtin_fma(ye, pc, y0);
            const double lo = __builtin_fma(le, pc, l0);            // a[j] / sqrt(ajj)
            const double lj = pivot_lane ? ajj * ri : lo;           // L[row][c0 + j] (meaningful for row >= c0 + j)
            // the diagonal entry of a row below the pivot and the reciprocal diagonal, as asm statements BEHIND lj (their
            // fake operand): three vector instructions between lj's last write and its first DPP read, no s_nop
            asm volatile("v_fma_f64 %0, -%1, %1, %0" : "+v"(dg) : "v"(lo), "v"(lj));
            {
                const unsigned long long pm = __builtin_amdgcn_ballot_w64(pivot_lane);
                unsigned dlo = (unsigned)__double_as_longlong(dinv), dhi = (unsigned)(__double_as_longlong(dinv) >> 32);
                const unsigned rlo = (unsigned)__double_as_longlong(ri), rhi = (unsigned)(__double_as_longlong(ri) >> 32);
                asm volatile("v_cndmask_b32 %0, %0, %2, %4\n\tv_cndmask_b32 %1, %1, %3, %4"
                             : "+v"(dlo), "+v"(dhi) : "v"(rlo), "v"(rhi), "s"(pm), "v"(lj));
                dinv = __longlong_as_double(((unsigned long long)dhi << 32) | dlo);
            }
            a[j] = lj;
            // a[c] -= L[row][c0 + j] L[c0 + c][c0 + j]: the second factor sits in lane c of this lane row
            chd_static_for<j + 1, 8>([&](auto C) __attribute__((always_inline)) {
                constexpr int c = decltype(C)::value;
                chd_fnma_bcast16<c, false>(a[c], lj, lj);
            });
        });
        if (live && (q >= 8 || r == 0)) {
#pragma unroll
            for (int c = 0; c < 8; ++c) D[row][c0 + c] = a[c];    // (above the diagonal: finite values nobody reads)
            if (q < 8) D[row][CH_NB] = dinv;
        }
        if (c1 < CH_NB) chol_diag_trailing_static(D, c0, c1, q, r);
    });
    if (!(amin > 1e-14) && lane == 0 && fail) atomicAdd(fail, 1);
}

#ifndef ZM_CHOL_DIAG_FAST
#define ZM_CHOL_DIAG_FAST 2
#endif
template <int TAG>
__device__ inline void chol_diag_wave_panel_t(double (*D)[CH_NB + 1], int nb, int* fail) {
#if ZM_CHOL_DIAG_FAST == 2
    chol_diag_wave_panel_dpp<TAG>(D, nb, fail);
#elif ZM_CHOL_DIAG_FAST
    chol_diag_wave_panel_fast<TAG>(D, nb, fail);
#else
    chol_diag_wave_panel_ref<TAG>(D, nb, fail);
#endif
}

// The same, always expanded in place: k_chol_df calls the factor from the middle of its step loop, where an
// out-of-line call saves and restores ~120 scalar registers through vector lanes on the critical wave (measured:
// 1.54 -> 1.51 ms per seven factorisations); the barrier forms and k_chol_tp are better off with the call.
template <int TAG>
__device__ __forceinline__ void chol_diag_wave_panel_inl(double (*D)[CH_NB + 1], int nb, int* fail) {
#if ZM_CHOL_DIAG_FAST == 2
    chol_diag_wave_panel_dpp<TAG>(D, nb, fail);
#elif ZM_CHOL_DIAG_FAST
    chol_diag_wave_panel_fast<TAG>(D, nb, fail);
#else
    chol_diag_wave_panel_ref<TAG>(D, nb, fail);
#endif
}
