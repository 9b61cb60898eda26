// Alard-Lupton kernel fit + spatially varying convolution + subtraction on gfx950.
//
// Replaces the hotpants process the reference launches from
// zuds/subtraction.py:162 with the flags of zuds/hotpants.py:77-93
// (-c t -n i, -r, -rss, -nsx/-nsy, -nrx/-nry, -ko, -bgo, -tni/-ini, -imi, -oni,
// -fin).  Algorithm and every convention: oracle/hotpants.py.
//
// Data flow (all device resident, fp64 for the fit, fp32 for the convolution)
//   k_hp_valid      bad[P] (u8) from bpm and the -tl/-tu/-il/-iu ranges
//   k_hp_rowany / k_hp_colany   separable (2 hw + 1)^2 dilations of `bad`
//   k_hp_cells      one workgroup per stamp cell: clipped sky/sigma, greedy
//                   brightest-first substamp centres (argmax reductions)
//   k_hp_vectors    5 workgroups per cell: separable basis convolutions of the
//                   template patch held in LDS -> X [nX][npix] fp64
//   k_hp_gram       8 K-slices per cell: G = X X^T on v_mfma_f64_16x16x4_f64 (the
//                   normal-equation GEMM; LDS-staged operands), k_hp_gram_sum adds them
//   k_hp_build      global normal matrix of a region from the per-cell Grams
//   k_chol_fused    blocked Cholesky (NB = 32) of every region in one launch: region
//                   barriers, MFMA trailing update, look-ahead diagonal factor
//   k_chol_back     back substitution (k_chol_back_cols: column per thread, up to 960 unknowns)
//   k_hp_merit / k_hp_reject    stamp figure of merit, sigma clip, next substamp
//   k_hp_apply<HWK> per output block kernel evaluation (fp64) + register-tiled
//                   fp32 convolution of template and template variance
#include "hp_dev.h"
#include <sched.h>
#include "chol_diag.h"

// ---------------------------------------------------------------------------
// lim != nullptr (zm_hp_params.limits_dev): the lower limits come from the background estimates that
// zm_median_mad2_async_dev left on the device - {median, sigma, count} of the science frame, then of the
// template - with the host's arithmetic (double, then rounded to float as a kernel argument would be)
__global__ void k_hp_valid(const float* __restrict__ sci, const float* __restrict__ ref,
                           const uint8_t* __restrict__ bpm, int64_t n, float il, float iu,
                           float tl, float tu, uint8_t* __restrict__ bad,
                           const double* __restrict__ lim, double nsig) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (lim) {
        il = (float)(lim[0] - nsig * lim[1]);
        tl = (float)(lim[3] - nsig * lim[4]);
    }
    if (p >= n) return;
    float s = sci[p], t = ref[p];
    bool ok = (s == s) && (t == t) && fabsf(s) < 3e38f && fabsf(t) < 3e38f;
    ok = ok && s >= il && s <= iu && t >= tl && t <= tu;
    if (bpm) ok = ok && bpm[p] == 0;
    bad[p] = ok ? 0 : 1;
}

// (2 hw + 1)-wide "any" along rows: one workgroup per row, inclusive prefix count of the row
// in LDS, out = count(x - hw .. x + hw) > 0.  Replaces a 2 hw + 1 byte-load loop per pixel.
#define HP_ROWMAX 16384
__global__ __launch_bounds__(256) void k_hp_rowany(const uint8_t* __restrict__ in, int nx, int ny, int hw,
                                                   uint8_t* __restrict__ out) {
    extern __shared__ int ra_pre[];                       // [nx + 1], ra_pre[0] = 0
    __shared__ int wsum[4];
    const int y = blockIdx.x, tid = threadIdx.x;
    const uint8_t* row = in + (size_t)y * nx;
    const int per = (nx + 255) / 256;                     // consecutive pixels per thread
    const int x0 = tid * per, x1 = min(x0 + per, nx);
    int loc = 0;
    for (int x = x0; x < x1; ++x) loc += row[x] ? 1 : 0;
    int inc = loc;
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int base = inc - loc;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    if (tid == 0) ra_pre[0] = 0;
    int run = base;
    for (int x = x0; x < x1; ++x) { run += row[x] ? 1 : 0; ra_pre[x + 1] = run; }
    __syncthreads();
    for (int x = tid; x < nx; x += 256) {
        const int a = max(x - hw, 0), b = min(x + hw, nx - 1);
        out[(size_t)y * nx + x] = (ra_pre[b + 1] - ra_pre[a]) > 0;
    }
}

// k_hp_valid and both row dilations of a subtraction in one pass over the row (round 4): the validity of a pixel
// is worked out where the prefix counts need it, `bad` is written for the stamp search, and the row goes out
// dilated by hw1 (substamp footprint) and by hw2 (kernel footprint) from the one prefix array.  Three launches
// and two more reads of the validity plane less per subtraction; the same bytes in every plane.
__global__ __launch_bounds__(256) void k_hp_valid_rows(const float* __restrict__ sci, const float* __restrict__ ref,
                                                       const uint8_t* __restrict__ bpm, int nx, int ny, float il, float iu,
                                                       float tl, float tu, const double* __restrict__ lim, double nsig,
                                                       int hw1, int hw2, uint8_t* __restrict__ bad,
                                                       uint8_t* __restrict__ out1, uint8_t* __restrict__ out2) {
    extern __shared__ int ra_pre[];                       // [nx + 1], ra_pre[0] = 0
    __shared__ int wsum[HP_ROWMAX / 64], wtot[4];         // bad pixels per (chunk of 256, wave), then their offsets
    if (lim) {
        il = (float)(lim[0] - nsig * lim[1]);
        tl = (float)(lim[3] - nsig * lim[4]);
    }
    const int y = blockIdx.x, tid = threadIdx.x;
    const size_t r0 = (size_t)y * nx;
    const int lane = tid & 63, wave = tid >> 6;
    const int nch = (nx + 255) / 256;
    // pixels x = 256 c + tid (coalesced loads); counts inside a wave by ballot, the (chunk, wave) totals by one
    // block-wide scan: four barriers per row, whatever its length
    for (int c = 0; c < nch; ++c) {
        const int x = 256 * c + tid;
        bool b = false;
        if (x < nx) {
            const float s = sci[r0 + x], t = ref[r0 + x];
            bool ok = (s == s) && (t == t) && fabsf(s) < 3e38f && fabsf(t) < 3e38f;
            ok = ok && s >= il && s <= iu && t >= tl && t <= tu;
            if (bpm) ok = ok && bpm[r0 + x] == 0;
            b = !ok;
            bad[r0 + x] = b ? 1 : 0;
        }
        const unsigned long long m = __ballot(b);
        if (x < nx) ra_pre[x + 1] = __popcll(m & ((2ull << lane) - 1ull));      // inclusive, inside the wave
        if (lane == 0) wsum[4 * c + wave] = __popcll(m);
    }
    if (tid == 0) ra_pre[0] = 0;
    __syncthreads();
    {
        // exclusive scan of the 4 nch totals (<= 256: one per thread)
        const int v = tid < 4 * nch ? wsum[tid] : 0;
        int inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        int off = inc - v;
        for (int w = 0; w < wave; ++w) off += wtot[w];
        if (tid < 4 * nch) wsum[tid] = off;
    }
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
        const int x = 256 * c + tid;
        if (x < nx) ra_pre[x + 1] += wsum[4 * c + wave];
    }
    __syncthreads();
    for (int x = tid; x < nx; x += 256) {
        const int a1 = max(x - hw1, 0), b1 = min(x + hw1, nx - 1);
        const int a2 = max(x - hw2, 0), b2 = min(x + hw2, nx - 1);
        out1[r0 + x] = (ra_pre[b1 + 1] - ra_pre[a1]) > 0;
        out2[r0 + x] = (ra_pre[b2 + 1] - ra_pre[a2]) > 0;
    }
}

// the same along columns: one thread per column walks a strip of rows with a running count
// of the window (coalesced row reads); edge != 0 also flags the hw-wide frame border
#define HP_COLSTRIP 96
__global__ __launch_bounds__(256) void k_hp_colany(const uint8_t* __restrict__ in, int nx, int ny, int hw,
                                                   int edge, uint8_t* __restrict__ out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y0 = blockIdx.y * HP_COLSTRIP, y1 = min(y0 + HP_COLSTRIP, ny);
    if (x >= nx) return;
    int cnt = 0;
    for (int j = max(y0 - hw, 0); j <= min(y0 + hw, ny - 1); ++j) cnt += in[(size_t)j * nx + x] ? 1 : 0;
    for (int y = y0; y < y1; ++y) {
        uint8_t v = cnt > 0;
        if (edge && (x < hw || x >= nx - hw || y < hw || y >= ny - hw)) v = 1;
        out[(size_t)y * nx + x] = v;
        const int add = y + 1 + hw, sub = y - hw;           // window of row y + 1
        if (add < ny) cnt += in[(size_t)add * nx + x] ? 1 : 0;
        if (sub >= 0) cnt -= in[(size_t)sub * nx + x] ? 1 : 0;
    }
}

// The same with four columns per thread (32-bit loads) and the rows taken eight at a time, the
// sixteen loads of a step issued before the running counts move: nx % 4 == 0, 4-byte aligned rows.
#define HP_COLSTRIP4 32
__global__ __launch_bounds__(256) void k_hp_colany4(const uint8_t* __restrict__ in, int nx, int ny, int hw,
                                                    int edge, uint8_t* __restrict__ out) {
    const int x = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int y0 = blockIdx.y * HP_COLSTRIP4, y1 = min(y0 + HP_COLSTRIP4, ny);
    if (x >= nx) return;
    const unsigned* in4 = reinterpret_cast<const unsigned*>(in + x);
    const size_t pitch = (size_t)nx / 4;                  // rows in 32-bit words
    auto nz = [](unsigned w, int k) -> int { return ((w >> (8 * k)) & 0xffu) ? 1 : 0; };
    int cnt[4] = {0, 0, 0, 0};
    {
        const int j0 = max(y0 - hw, 0), j1 = min(y0 + hw, ny - 1);
        for (int jb = j0; jb <= j1; jb += 8) {
            unsigned w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = (jb + u <= j1) ? in4[(size_t)(jb + u) * pitch] : 0u;
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int k = 0; k < 4; ++k) cnt[k] += nz(w[u], k);
        }
    }
    for (int yb = y0; yb < y1; yb += 8) {
        unsigned wa[8], ws[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int add = yb + u + 1 + hw, sub = yb + u - hw;      // window of row yb + u + 1
            wa[u] = (add < ny) ? in4[(size_t)add * pitch] : 0u;
            ws[u] = (sub >= 0) ? in4[(size_t)sub * pitch] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int y = yb + u;
            if (y < y1) {
                unsigned o = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    unsigned v = cnt[k] > 0;
                    if (edge && (x + k < hw || x + k >= nx - hw || y < hw || y >= ny - hw)) v = 1;
                    o |= v << (8 * k);
                }
                *reinterpret_cast<unsigned*>(out + (size_t)y * nx + x) = o;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) cnt[k] += nz(wa[u], k) - nz(ws[u], k);
        }
    }
}

// One workgroup per stamp cell.  centres[cell * nss + k] = (x, y) or (-1, -1).
__global__ __launch_bounds__(256) void k_hp_cells(const hp_plan P, const float* __restrict__ ref,
                                                  const uint8_t* __restrict__ bad,
                                                  const uint8_t* __restrict__ dirty,
                                                  int2* __restrict__ centres) {
    __shared__ double red[4];
    __shared__ float bval[4];
    __shared__ int bidx[4];
    __shared__ int2 chosen[HP_MAXNSS];
    const int cell = blockIdx.x, tid = threadIdx.x;
    const int r = cell / P.ncellr, c = cell - r * P.ncellr;
    const int sy = c / P.nsx, sx = c - sy * P.nsx;
    const int cw = (P.rx1[r] - P.rx0[r]) / P.nsx, ch = (P.ry1[r] - P.ry0[r]) / P.nsy;
    const int cx0 = P.rx0[r] + sx * cw, cy0 = P.ry0[r] + sy * ch;
    const int n = cw * ch;
    // clipped moments of the valid template pixels (oracle: clipped_moments)
    double m = 0.0, s = 0.0;
    for (int pass = 0; pass < 4; ++pass) {
        double s0 = 0, s1 = 0;
        for (int k = tid; k < n; k += 256) {
            int yy = k / cw, xx = k - yy * cw;
            size_t idx = (size_t)(cy0 + yy) * P.nx + cx0 + xx;
            if (bad[idx]) continue;
            double v = ref[idx];
            if (pass == 0 || fabs(v - m) <= 3.0 * s) { s0 += 1.0; s1 += v; }
        }
        s0 = block_sum256(s0, red);
        s1 = block_sum256(s1, red);
        if (s0 < 1.0) break;
        double mn = s1 / s0;
        double s2 = 0;
        for (int k = tid; k < n; k += 256) {
            int yy = k / cw, xx = k - yy * cw;
            size_t idx = (size_t)(cy0 + yy) * P.nx + cx0 + xx;
            if (bad[idx]) continue;
            double v = ref[idx];
            if (pass == 0 || fabs(v - m) <= 3.0 * s) s2 += (v - mn) * (v - mn);
        }
        s2 = block_sum256(s2, red);
        m = mn;
        s = sqrt(s2 / s0);
    }
    const double thr = m + P.ft * s;
    for (int k = 0; k < P.nss; ++k) {
        float best = -__builtin_inff();
        int bi = 0x7fffffff;
        for (int q = tid; q < n; q += 256) {
            int yy = q / cw, xx = q - yy * cw;
            int x = cx0 + xx, y = cy0 + yy;
            size_t idx = (size_t)y * P.nx + x;
            if (dirty[idx]) continue;
            float v = ref[idx];
            if (!((double)v >= thr)) continue;
            bool excl = false;
            for (int e = 0; e < k; ++e)
                excl |= (abs(x - chosen[e].x) <= P.hwss) && (abs(y - chosen[e].y) <= P.hwss);
            if (excl) continue;
            if (v > best || (v == best && q < bi)) { best = v; bi = q; }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            float ov = __shfl_xor(best, o);
            int oi = __shfl_xor(bi, o);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        __syncthreads();
        if ((tid & 63) == 0) { bval[tid >> 6] = best; bidx[tid >> 6] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; ++w)
                if (bval[w] > best || (bval[w] == best && bidx[w] < bi)) { best = bval[w]; bi = bidx[w]; }
            int2 cc = make_int2(-1, -1);
            if (bi != 0x7fffffff) { int yy = bi / cw; cc = make_int2(cx0 + bi - yy * cw, cy0 + yy); }
            chosen[k] = cc;
            centres[cell * P.nss + k] = cc;
        }
        __syncthreads();
        if (chosen[k].x < 0) {   // nothing left: the remaining slots are empty too
            if (tid == 0)
                for (int e = k + 1; e < P.nss; ++e) centres[cell * P.nss + e] = make_int2(-1, -1);
            break;
        }
    }
}

// The same for cells of at most HC_THREADS * HC_PX pixels: a thread keeps its pixels (k = tid + HC_THREADS i)
// and their two flags in registers; the eight clipping sweeps and the greedy picks then read no memory at all.
#define HC_PX 24
// Round 5 (late): 512 threads and 24 pixels per thread (was 256 x 48) - the kernel is a chain of twelve block sums and
// three picks behind loops over a thread's pixels, 900 workgroups two per CU: per-thread loops of half the length,
// the same number of resident workgroups.  (The partial sums group differently from k_hp_cells' now: m, s and the
// threshold move in their last bits, like between any two summation orders; the oracle's own order is numpy's.)
#define HC_THREADS 512
__global__ __launch_bounds__(HC_THREADS, 4) void k_hp_cells_reg(const hp_plan P, const float* __restrict__ ref,
                                                      const uint8_t* __restrict__ bad,
                                                      const uint8_t* __restrict__ dirty,
                                                      int2* __restrict__ centres) {
    __shared__ double red[HC_THREADS / 64];
    __shared__ float bval[HC_THREADS / 64];
    __shared__ int bidx[HC_THREADS / 64];
    __shared__ int2 chosen[HP_MAXNSS];
    const int cell = blockIdx.x, tid = threadIdx.x;
    const int r = cell / P.ncellr, c = cell - r * P.ncellr;
    const int sy = c / P.nsx, sx = c - sy * P.nsx;
    const int cw = (P.rx1[r] - P.rx0[r]) / P.nsx, ch = (P.ry1[r] - P.ry0[r]) / P.nsy;
    const int cx0 = P.rx0[r] + sx * cw, cy0 = P.ry0[r] + sy * ch;
    const int n = cw * ch;
    const int dyy = HC_THREADS / cw, dxx = HC_THREADS - dyy * cw;        // (yy, xx) advance by HC_THREADS pixels
    const int yy0 = tid / cw, xx0 = tid - yy0 * cw;
    float v[HC_PX];
    unsigned long long okm = 0, cleanm = 0;                // bit i: pixel i is not bad / not dirty
    {
        int yy = yy0, xx = xx0;
#pragma unroll
        for (int i = 0; i < HC_PX; ++i) {
            const int k = tid + HC_THREADS * i;
            float val = 0.f;
            if (k < n) {
                const size_t idx = (size_t)(cy0 + yy) * P.nx + cx0 + xx;
                val = ref[idx];
                if (!bad[idx]) okm |= 1ull << i;
                if (!dirty[idx]) cleanm |= 1ull << i;
            }
            v[i] = val;
            xx += dxx; yy += dyy;
            if (xx >= cw) { xx -= cw; ++yy; }
        }
    }
    double m = 0.0, s = 0.0;
    for (int pass = 0; pass < 4; ++pass) {
        double s0 = 0, s1 = 0;
        // (round 5: the conversions are made HERE, per pass - v[] does not change, so the compiler kept all 48
        // doubles beside the 48 floats across the passes, and the picks' 96 pixel coordinates too: 256 registers
        // and 64 spilled ones, 260 B of scratch per lane.  An opaque copy per use keeps only the floats live.)
#pragma unroll
        for (int i = 0; i < HC_PX; ++i) {
            float vf = v[i];
            asm volatile("" : "+v"(vf));
            const double vv = vf;
            if ((okm >> i & 1) && (pass == 0 || fabs(vv - m) <= 3.0 * s)) { s0 += 1.0; s1 += vv; }
        }
        s0 = block_sum_waves<HC_THREADS / 64>(s0, red);
        s1 = block_sum_waves<HC_THREADS / 64>(s1, red);
        if (s0 < 1.0) break;
        const double mn = s1 / s0;
        double s2 = 0;
#pragma unroll
        for (int i = 0; i < HC_PX; ++i) {
            float vf = v[i];
            asm volatile("" : "+v"(vf));
            const double vv = vf;
            if ((okm >> i & 1) && (pass == 0 || fabs(vv - m) <= 3.0 * s)) s2 += (vv - mn) * (vv - mn);
        }
        s2 = block_sum_waves<HC_THREADS / 64>(s2, red);
        m = mn;
        s = sqrt(s2 / s0);
    }
    const double thr = m + P.ft * s;
    // candidates: clean pixels at or above the threshold
    unsigned long long cand = 0;
#pragma unroll
    for (int i = 0; i < HC_PX; ++i)
        if ((cleanm >> i & 1) && ((double)v[i] >= thr)) cand |= 1ull << i;
    int2 last = make_int2(-1000000, -1000000);           // the pick of the previous round (registers)
    for (int k = 0; k < P.nss; ++k) {
        float best = -__builtin_inff();
        int bi = 0x7fffffff;
        {
            int yy = yy0, xx = xx0;
            asm volatile("" : "+v"(yy), "+v"(xx));       // (the coordinates are walked per pick, not kept: see above)
#pragma unroll
            for (int i = 0; i < HC_PX; ++i) {
                if (cand >> i & 1) {
                    // candidates near an earlier pick were struck out when it was made: only the
                    // latest pick is new
                    const int x = cx0 + xx, y = cy0 + yy, q = tid + HC_THREADS * i;
                    const bool excl = (abs(x - last.x) <= P.hwss) && (abs(y - last.y) <= P.hwss);
                    if (excl) cand &= ~(1ull << i);          // stays excluded for the later picks
                    else if (v[i] > best || (v[i] == best && q < bi)) { best = v[i]; bi = q; }
                }
                xx += dxx; yy += dyy;
                if (xx >= cw) { xx -= cw; ++yy; }
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            float ov = __shfl_xor(best, o);
            int oi = __shfl_xor(bi, o);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        __syncthreads();
        if ((tid & 63) == 0) { bval[tid >> 6] = best; bidx[tid >> 6] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < HC_THREADS / 64; ++w)
                if (bval[w] > best || (bval[w] == best && bidx[w] < bi)) { best = bval[w]; bi = bidx[w]; }
            int2 cc = make_int2(-1, -1);
            if (bi != 0x7fffffff) { int yy = bi / cw; cc = make_int2(cx0 + bi - yy * cw, cy0 + yy); }
            chosen[k] = cc;
            centres[cell * P.nss + k] = cc;
        }
        __syncthreads();
        last = chosen[k];
        if (last.x < 0) {   // nothing left: the remaining slots are empty too
            if (tid == 0)
                for (int e = k + 1; e < P.nss; ++e) centres[cell * P.nss + e] = make_int2(-1, -1);
            break;
        }
    }
}

// ---------------------------------------------------------------------------
// Per-cell state: active[cell] = index of the substamp in use (-1: none),
// need[cell] = vectors / Gram must be (re)computed this round.
// X layout: [cell][nX][npixp] fp64, rows 0..nc-1 kernel vectors, nc..nE-1
// background terms, nE the science pixels; columns >= npix are zero.

// ---------------------------------------------------------------------------
// G = X X^T, 64 x 64 fp64, on the f64 matrix cores.
typedef double double4_t __attribute__((ext_vector_type(4)));

#define GR_KT 32
#define GR_PITCH 34   // doubles; (4 i + 2 k) mod 64 banks are distinct for ds_read_b64

// blockIdx.y = slice of the pixel axis (GR_SPLIT slices of whole K tiles): partial Gram
// matrices go to Gp[cell][slice], k_hp_gram_sum adds them in slice order (deterministic).
// As for the vectors: after the first round a cell's latency sets the kernel time.
#define GR_SPLIT 8
static __device__ __forceinline__ void hp_gram_body(const hp_plan& P, const double* __restrict__ X,
                                                 const int* __restrict__ need,
                                                 const int* __restrict__ active,
                                                 double* __restrict__ Gp, const int* __restrict__ guard,
                                                 const int* __restrict__ list) {
    if (guard && *guard == 0) return;                    // the previous round rejected nothing: this round is void
    __shared__ double L[HP_MAXX * GR_PITCH];
    const int tid = threadIdx.x;
    const int ncl = list ? list[0] : (int)gridDim.x;     // (k_hp_vectors: the cells with a new substamp)
#pragma unroll 1
    for (int ci = blockIdx.x; ci < ncl; ci += gridDim.x) {
    const int cell = list ? list[1 + ci] : ci;
    __syncthreads();
    if (!need[cell] || active[cell] < 0) continue;
    const double* Xc = X + (size_t)cell * P.nX * P.npixp;
    const int wave = tid >> 6, lane = tid & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int ntile = P.npixp / GR_KT;                     // npixp is a multiple of GR_KT
    const int per = (ntile + GR_SPLIT - 1) / GR_SPLIT;
    const int kbeg = blockIdx.y * per * GR_KT, kend = min((int)(blockIdx.y + 1) * per, ntile) * GR_KT;
    double4_t acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = (double4_t){0.0, 0.0, 0.0, 0.0};
    // a K tile = 64 rows x 32 columns = 8 doubles per thread: fetched into registers one tile
    // ahead (all eight loads in flight together, and under the matrix cores' work on the
    // current tile), stored to LDS after the barrier
    constexpr int GR_LD = HP_MAXX * GR_KT / 256;
    double nxt[GR_LD];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < GR_LD; ++q) {
            const int e = tid + 256 * q, row = e >> 5, col = e & 31;
            nxt[q] = row < P.nX ? Xc[(size_t)row * P.npixp + k0 + col] : 0.0;
        }
    };
    if (kbeg < kend) fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += GR_KT) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < GR_LD; ++q) {
            const int e = tid + 256 * q, row = e >> 5, col = e & 31;
            L[row * GR_PITCH + col] = nxt[q];
        }
        __syncthreads();
        if (k0 + GR_KT < kend) fetch(k0 + GR_KT);
#pragma unroll
        for (int kk = 0; kk < GR_KT / 4; ++kk) {
            double a = L[(16 * wave + li) * GR_PITCH + 4 * kk + lk];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                double b = L[(16 * c + li) * GR_PITCH + 4 * kk + lk];
                acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
            }
        }
    }
    // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 reg
    double* Gc = Gp + ((size_t)cell * GR_SPLIT + blockIdx.y) * HP_MAXX * HP_MAXX;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
            Gc[(size_t)(16 * wave + lk + 4 * rg) * HP_MAXX + 16 * c + li] = acc[c][rg];
    }   // cells
}

// (round 4: four entries per thread with all their loads in flight - with sixteen entries per thread taken one
// after the other the kernel was sixteen memory latencies long, 11 us for a handful of cells)
#define GS_THREADS 1024
static __device__ __forceinline__ void hp_gram_sum_body(const double* __restrict__ Gp,
                                                     const int* __restrict__ need,
                                                     const int* __restrict__ active,
                                                     double* __restrict__ G, const int* __restrict__ guard,
                                                     double* __restrict__ Gold, const int* __restrict__ list) {
    if (guard && *guard == 0) return;                    // the previous round rejected nothing: this round is void
    const int ncl = list ? list[0] : (int)gridDim.x;
    constexpr int NE = HP_MAXX * HP_MAXX / GS_THREADS;   // 4
#pragma unroll 1
    for (int ci = blockIdx.x; ci < ncl; ci += gridDim.x) {
        const int cell = list ? list[1 + ci] : ci;
        if (!need[cell] || active[cell] < 0) continue;
        const double* src = Gp + (size_t)cell * GR_SPLIT * HP_MAXX * HP_MAXX;
        double* dst = G + (size_t)cell * HP_MAXX * HP_MAXX;
        double* old = Gold ? Gold + (size_t)cell * HP_MAXX * HP_MAXX : nullptr;
        double part[NE][GR_SPLIT], prev[NE];
#pragma unroll
        for (int q = 0; q < NE; ++q) {
            const int e = threadIdx.x + GS_THREADS * q;
#pragma unroll
            for (int sl = 0; sl < GR_SPLIT; ++sl) part[q][sl] = src[(size_t)sl * HP_MAXX * HP_MAXX + e];
            prev[q] = old ? dst[e] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < NE; ++q) {
            const int e = threadIdx.x + GS_THREADS * q;
            double v = 0.0;
#pragma unroll
            for (int sl = 0; sl < GR_SPLIT; ++sl) v += part[q][sl];      // slice order: deterministic
            if (old) old[e] = prev[q];   // (the Gram matrix of the substamp this one replaces: the fused normal-matrix update takes it out)
            dst[e] = v;
        }
    }   // cells
}

// ---------------------------------------------------------------------------
// Global unknown c -> (source vector n, spatial term p or -1)
__device__ inline void col_decode(const hp_plan& P, int c, int* n, int* p) {
    if (c == 0) { *n = 0; *p = -1; return; }
    int k = c - 1;
    if (k < (P.nc - 1) * P.nkp) { *n = 1 + k / P.nkp; *p = k % P.nkp; return; }
    *n = P.nc + (k - (P.nc - 1) * P.nkp);
    *p = -1;
}

// A0: [reg][nunk][nunk] lower triangle of the unscaled normal matrix, rhs0: [reg][nunk]; both
// persist over the rejection rounds.  sign == 0: built from every active cell.  sign == -1 / +1:
// only the contribution of the cells whose substamp the last rejection changed (chg[cell]) is
// taken out (with the old Gram matrix and spatial terms, before they are recomputed) / put
// back in (with the new ones, cells that still have a substamp): a round touches a few cells
// of a hundred.
__global__ __launch_bounds__(256) void k_hp_build(const hp_plan P, const double* __restrict__ G,
                                                  const double* __restrict__ phi,
                                                  const int* __restrict__ active,
                                                  const int* __restrict__ chg, int sign,
                                                  double* __restrict__ A,
                                                  double* __restrict__ rhs, const int* __restrict__ guard,
                                                  unsigned* __restrict__ zero = nullptr, int nzero = 0) {
    if (guard && *guard == 0) return;                    // the previous round rejected nothing: this round is void
    const int reg = blockIdx.z;
    if (zero && blockIdx.x == 0 && blockIdx.y == 0)      // (the hand-over words of this round's k_chol_df)
        for (int k = threadIdx.x; k < nzero; k += 256) zero[(size_t)reg * nzero + k] = 0u;
    const int c1 = blockIdx.y * 16 + (threadIdx.x >> 4), c2 = blockIdx.x * 16 + (threadIdx.x & 15);
    if (blockIdx.x > blockIdx.y) return;
    if (c1 >= P.nunk || c2 >= P.nunk) return;
    int n1, p1, n2, p2;
    col_decode(P, c1, &n1, &p1);
    col_decode(P, c2, &n2, &p2);
    double acc = 0.0, racc = 0.0;
    const bool do_rhs = (c2 == 0);
    bool any = false;
    const int* list = chg + P.ncell + reg * (P.ncellr + 1);
    const int ncand = (sign == 0) ? P.ncellr : list[0];
    // eight candidates at a time: their loads go out together, the sums keep the cell order
    for (int s0 = 0; s0 < ncand; s0 += 8) {
        double g[8], gr[8], w1[8], w2[8];
        bool on[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int s = s0 + u;
            on[u] = false;
            g[u] = gr[u] = 0.0;
            w1[u] = w2[u] = 1.0;
            if (s < ncand) {
                const int cell = (sign == 0) ? reg * P.ncellr + s : list[1 + s];
                on[u] = !(active[cell] < 0 && sign >= 0);
                if (on[u]) {
                    const double* Gc = G + (size_t)cell * HP_MAXX * HP_MAXX;
                    const double* ph = phi + (size_t)cell * P.nkp;
                    if (p1 >= 0) w1[u] = ph[p1];
                    if (p2 >= 0) w2[u] = ph[p2];
                    g[u] = Gc[n1 * HP_MAXX + n2];
                    if (do_rhs) gr[u] = Gc[n1 * HP_MAXX + P.nE];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (on[u]) {
                any = true;
                acc += w1[u] * w2[u] * g[u];
                if (do_rhs) racc += w1[u] * gr[u];
            }
    }
    const size_t ia = (size_t)reg * (size_t)(P.nunk + 1) * P.nunk + (size_t)c1 * P.nunk + c2;
    if (sign == 0) {
        if (c2 <= c1) A[ia] = acc;
        if (do_rhs) rhs[(size_t)reg * P.nunk + c1] = racc;
    } else if (any) {
        if (c2 <= c1) A[ia] += sign * acc;
        if (do_rhs) rhs[(size_t)reg * P.nunk + c1] += sign * racc;
    }
}

#define BB_CH 128
// The same sums by blocks of source-vector pairs: workgroup (n1, n2 <= n1) of a region owns the
// nkp x nkp unknowns (n1, p1) x (n2, p2); per cell it needs ONE Gram entry G[n1][n2] (a uniform,
// scalar load) and the spatial terms of its threads, i.e. 2 loads and 3 flops per unknown pair and
// cell where k_hp_build decodes and gathers per pair (1.2 x 10^8 pair-cell products per round-1
// build).  Same products, same cell order: the results are identical to the last bit.
static __device__ __forceinline__ void hp_build_blk_body(const int reg, const int pair, const hp_plan& P, const double* __restrict__ G,
                                                      const double* __restrict__ phi,
                                                      const int* __restrict__ active,
                                                      const int* __restrict__ chg, int sign,
                                                      double* __restrict__ A,
                                                      double* __restrict__ rhs, const int* __restrict__ guard,
                                                      unsigned* __restrict__ zero, int nzero,
                                                      const double* __restrict__ Gold,
                                                      const double* __restrict__ phiold,
                                                      const int* __restrict__ need) {
    if (guard && *guard == 0) return;
    // (a triangular grid: x = pair index of (n1, n2 <= n1) - the square grid dispatched as many empty workgroups again)
    int n1 = (int)((sqrtf(8.f * (float)pair + 1.f) - 1.f) * 0.5f);
    while (n1 * (n1 + 1) / 2 > pair) --n1;
    while ((n1 + 1) * (n1 + 2) / 2 <= pair) ++n1;
    const int n2 = pair - n1 * (n1 + 1) / 2;
    if (zero && n1 == 0 && n2 == 0)                      // (the hand-over words of this round's k_chol_df)
        for (int k = threadIdx.x; k < nzero; k += 256) zero[(size_t)reg * nzero + k] = 0u;
    const int p1 = threadIdx.x >> 4, p2 = threadIdx.x & 15;
    const bool k1 = n1 >= 1 && n1 < P.nc, k2 = n2 >= 1 && n2 < P.nc;   // kernel terms carry spatial factors
    const int nk = (P.nc - 1) * P.nkp;
    const int c1 = n1 == 0 ? 0 : (k1 ? 1 + (n1 - 1) * P.nkp + p1 : 1 + nk + (n1 - P.nc));
    const int c2 = n2 == 0 ? 0 : (k2 ? 1 + (n2 - 1) * P.nkp + p2 : 1 + nk + (n2 - P.nc));
    const bool live = p1 < (k1 ? P.nkp : 1) && p2 < (k2 ? P.nkp : 1) && c2 <= c1;
    const bool do_rhs = (c2 == 0);
    const int* list = chg + P.ncell + reg * (P.ncellr + 1);
    const int ncand = (sign == 0) ? P.ncellr : list[0];
    // the candidates' Gram entries, right-hand-side entries, spatial terms and flags are staged in
    // LDS, BB_CH cells at a time (one memory latency per chunk); the sums then run out of LDS
    __shared__ double gs[BB_CH], grs[BB_CH], phs[BB_CH * 16];
    __shared__ int ons[BB_CH];
    const int tid = threadIdx.x;
    const size_t ia = (size_t)reg * (size_t)(P.nunk + 1) * P.nunk + (size_t)c1 * P.nunk + c2;
    // sign == 2 (round 4): both updates of a rejection round in one launch - the changed cells leave with the Gram
    // matrix and spatial terms they had (kept by k_hp_gram_sum / k_hp_vectors when they wrote the new ones: Gold,
    // phiold; a cell that was dropped still has them in place), then come back with the new ones; an entry takes
    // the two sums one after the other, as the two launches applied them: the same roundings.
    double av = 0.0, rv = 0.0;
    bool touched = false;
    const int npass = sign == 2 ? 2 : 1;
    for (int pass = 0; pass < npass; ++pass) {
        const int sg = sign == 2 ? (pass == 0 ? -1 : 1) : sign;
        double acc = 0.0, racc = 0.0;
        bool any = false;
        for (int s0 = 0; s0 < ncand; s0 += BB_CH) {
            const int nch = min(BB_CH, ncand - s0);
            __syncthreads();
            if (tid < nch) {
                const int cell = (sg == 0) ? reg * P.ncellr + s0 + tid : list[1 + s0 + tid];
                const int on = !(active[cell] < 0 && sg >= 0);
                const bool was = sign == 2 && pass == 0 && need[cell];       // recomputed since: the old copy
                const double* Gc = (was ? Gold : G) + (size_t)cell * HP_MAXX * HP_MAXX;
                ons[tid] = on;
                gs[tid] = on ? Gc[n1 * HP_MAXX + n2] : 0.0;
                grs[tid] = (on && n2 == 0) ? Gc[n1 * HP_MAXX + P.nE] : 0.0;
            }
            for (int e = tid; e < nch * P.nkp; e += 256) {
                const int k = e / P.nkp, pp = e - k * P.nkp;
                const int cell = (sg == 0) ? reg * P.ncellr + s0 + k : list[1 + s0 + k];
                const bool was = sign == 2 && pass == 0 && need[cell];
                phs[k * 16 + pp] = (was ? phiold : phi)[(size_t)cell * P.nkp + pp];
            }
            __syncthreads();
            if (live) {
                for (int k = 0; k < nch; ++k) {
                    if (!ons[k]) continue;
                    any = true;
                    const double w1 = k1 ? phs[k * 16 + p1] : 1.0, w2 = k2 ? phs[k * 16 + p2] : 1.0;
                    acc += w1 * w2 * gs[k];
                    if (do_rhs) racc += w1 * grs[k];
                }
            }
        }
        if (!live) continue;
        if (sg == 0) {
            A[ia] = acc;
            if (do_rhs) rhs[(size_t)reg * P.nunk + c1] = racc;
        } else if (any) {
            if (!touched) {
                av = A[ia];
                if (do_rhs) rv = rhs[(size_t)reg * P.nunk + c1];
                touched = true;
            }
            av += (double)sg * acc;
            if (do_rhs) rv += (double)sg * racc;
        }
    }
    if (live && touched) {
        A[ia] = av;
        if (do_rhs) rhs[(size_t)reg * P.nunk + c1] = rv;
    }
}

// The first round's normal matrix on the f64 matrix cores (round 4).  The block of a pair of source vectors
// (n1, n2) is a small matrix product over the region's cells, A[p1][p2] = sum_k phi_k[p1] (phi_k[p2] g_k) with
// g_k = G_k[n1][n2]: C = F1 (F2 diag(g))^T, 15 x 15 (x number of cells) - one wave per pair, a
// v_mfma_f64_16x16x4 per four cells, the right-hand side riding as column 15 (sum_k phi_k[p1] G_k[n1][nE]).  As
// scalar sums (k_hp_build_blk, sign 0) every product read LDS three times: 0.7 G reads, 100 us.  The products are
// grouped differently ((phi g) phi instead of (phi phi) g, four cells per instruction): the entries agree with
// k_hp_build_blk's to fp64 rounding, not to the bit; the later rounds' updates (k_hp_build_blk, sign 2) take a
// changed cell's contribution out in their own rounding - a residue of 1e-16 of an entry, far below what the
// conditioning of the fit resolves (the parity tests against the oracle hold at their tolerances).
#define BM_CH 128     // cells per chunk
static __device__ __forceinline__ void hp_build_mfma_body(const hp_plan& P, const double* __restrict__ G,
                                                       const double* __restrict__ phi, const int* __restrict__ active,
                                                       double* __restrict__ A, double* __restrict__ rhs,
                                                       unsigned* __restrict__ zero, int nzero) {
    typedef double bm_double4 __attribute__((ext_vector_type(4)));
    __shared__ double phs[BM_CH][16];        // spatial terms of the chunk's cells (0 beyond nkp, rows of inactive cells 0)
    __shared__ double gw[4][2][BM_CH];       // per wave: g_k and the right-hand-side entry of its pair
    const int reg = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    if (zero && blockIdx.x == 0)             // (the hand-over words of this round's k_chol_df)
        for (int k = tid; k < nzero; k += 256) zero[(size_t)reg * nzero + k] = 0u;
    const int npair = P.nE * (P.nE + 1) / 2;
    const int pair = blockIdx.x * 4 + wave;
    const bool have = pair < npair;          // (wave-uniform; every wave takes part in the staging barriers)
    int n1 = 0, n2 = 0;
    if (have) {
        n1 = (int)((sqrtf(8.f * (float)pair + 1.f) - 1.f) * 0.5f);
        while (n1 * (n1 + 1) / 2 > pair) --n1;
        while ((n1 + 1) * (n1 + 2) / 2 <= pair) ++n1;
        n2 = pair - n1 * (n1 + 1) / 2;
    }
    const bool k1 = n1 >= 1 && n1 < P.nc, k2 = n2 >= 1 && n2 < P.nc;   // kernel terms carry spatial factors
    const int nk = (P.nc - 1) * P.nkp;
    const int b1 = n1 == 0 ? 0 : (k1 ? 1 + (n1 - 1) * P.nkp : 1 + nk + (n1 - P.nc));
    const int b2 = n2 == 0 ? 0 : (k2 ? 1 + (n2 - 1) * P.nkp : 1 + nk + (n2 - P.nc));
    const int np1 = k1 ? P.nkp : 1, np2 = k2 ? P.nkp : 1;
    bm_double4 c4 = {0.0, 0.0, 0.0, 0.0};
    for (int s0 = 0; s0 < P.ncellr; s0 += BM_CH) {
        const int nch = min(BM_CH, P.ncellr - s0);
        __syncthreads();
        for (int e = tid; e < BM_CH * 16; e += 256) {
            const int k = e >> 4, pp = e & 15;
            double v = 0.0;
            if (k < nch && pp < P.nkp) {
                const int cell = reg * P.ncellr + s0 + k;
                if (active[cell] >= 0) v = phi[(size_t)cell * P.nkp + pp];
            }
            phs[k][pp] = v;
        }
        for (int k = lane; k < BM_CH; k += 64) {
            double g = 0.0, gr = 0.0;
            if (have && k < nch) {
                const int cell = reg * P.ncellr + s0 + k;
                if (active[cell] >= 0) {
                    const double* Gc = G + (size_t)cell * HP_MAXX * HP_MAXX;
                    g = Gc[n1 * HP_MAXX + n2];
                    if (n2 == 0) gr = Gc[n1 * HP_MAXX + P.nE];
                }
            }
            gw[wave][0][k] = g;
            gw[wave][1][k] = gr;
        }
        __syncthreads();
        if (have) {
            for (int kk = 0; kk < (nch + 3) / 4; ++kk) {
                const int k = 4 * kk + lk;                    // (cells beyond the chunk: g = 0 and phi = 0)
                const double f1 = k1 ? phs[k][li] : (li == 0 ? 1.0 : 0.0);    // (an inactive cell has g = 0 on the other side)
                const double f2 = k2 ? phs[k][li] : (li == 0 ? 1.0 : 0.0);
                const double bv = (li == 15) ? gw[wave][1][k] : f2 * gw[wave][0][k];
                c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(f1, bv, c4, 0, 0, 0);
            }
        }
    }
    if (!have) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int p1 = lk + 4 * q, p2 = li;
        if (p1 >= np1) continue;
        const int c1 = b1 + p1;
        if (p2 == 15) {
            if (n2 == 0) rhs[(size_t)reg * P.nunk + c1] = c4[q];
        } else if (p2 < np2) {
            const int c2 = b2 + p2;
            if (c2 <= c1) A[(size_t)reg * (size_t)(P.nunk + 1) * P.nunk + (size_t)c1 * P.nunk + c2] = c4[q];
        }
    }
}

// Jacobi scaling: d = sqrt(diag); A <- A / (d d^T); rhs <- rhs / d
static __device__ __forceinline__ void hp_diag_body(int n, const double* __restrict__ A, double* __restrict__ d, const int* __restrict__ guard) {
    if (guard && *guard == 0) return;                    // the previous round rejected nothing: this round is void
    int reg = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    double v = A[(size_t)reg * (size_t)(n + 1) * n + (size_t)c * n + c];
    d[(size_t)reg * n + c] = v > 0.0 ? sqrt(v) : 1.0;
}

static __device__ __forceinline__ void hp_scale_body(const int reg, int n, int lda, const double* __restrict__ A0, const double* __restrict__ rhs0,
                           double* __restrict__ A, const double* __restrict__ d, unsigned* __restrict__ bar, const int* __restrict__ guard,
                           unsigned* __restrict__ dff, int ndff) {
    if (guard && *guard == 0) return;                    // the previous round rejected nothing: this round is void
    int c2 = blockIdx.x * blockDim.x + threadIdx.x, c1 = blockIdx.y;
    if (c1 == 0 && c2 == 0)
        for (int k = 0; k < 4; ++k) bar[reg * CF_BAR_STRIDE + k] = 0;          // arms k_chol_fused's two region barriers
    if (c1 == 0 && dff && c2 < ndff) dff[(size_t)reg * ndff + c2] = 0u;        // ... and k_chol_df's hand-over words
    if (c2 > c1 || c2 >= n) return;
    const double* dd = d + (size_t)reg * n;
    const size_t base0 = (size_t)reg * (size_t)(n + 1) * n;
    const size_t base = (size_t)reg * (size_t)(n + 1) * lda;     // rows of A are padded to whole 128-B lines
    // (round 4: by the reciprocals of d - k_chol_df scales its tiles on their way into LDS with two multiplications
    // per entry where an fp64 division costs ~40 instructions; every form multiplies the same way: same bits)
    const double r1 = 1.0 / dd[c1], r2 = 1.0 / dd[c2];
    double v = __dmul_rn(A0[base0 + (size_t)c1 * n + c2], __dmul_rn(r1, r2));     // (no contraction: k_chol_df does the same)
    if (c1 == c2) v = __dadd_rn(v, HP_RIDGE);   // keeps a rank-deficient basis solvable (oracle: RIDGE)
    A[base + (size_t)c1 * lda + c2] = v;
    if (c2 == 0) A[base + (size_t)n * lda + c1] = __dmul_rn(rhs0[(size_t)reg * n + c1], r1);   // rhs row
}

// ---- blocked Cholesky, lower, in place ------------------------------------------------
// The right-hand side rides along as row n of the (n + 1) x n lower-triangular storage:
// factoring the augmented matrix leaves y = L^-1 b in that row, so the forward substitution
// costs nothing extra.  Block size 32; see k_chol_fused below.

// (the 32 x 32 diagonal factor every form of the factorisation calls: chol_diag.h)

template <int SRC>
__device__ __forceinline__ double row16_bcast_d(double v) {
    // lane SRC of every row of 16 lanes to all lanes of that row (row_newbcast: 0x150 + SRC)
    const unsigned long long u = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u, 0x150 + SRC, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), 0x150 + SRC, 0xf, 0xf, false);
    return __longlong_as_double(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

template <int I, int N, typename Fn>
__device__ __forceinline__ void hp_static_for(Fn&& fn) {
    if constexpr (I < N) {
        fn(std::integral_constant<int, I>{});
        hp_static_for<I + 1, N>(fn);
    }
}

// ---- the whole factorisation in one launch ----------------------------------------
// W workgroups per region walk the 32-column blocks together; a region-wide barrier (a
// monotone counter in global memory, agent-scope release / acquire) separates the panel
// solve from the trailing update and the update from the next panel.  Replaces 2 launches
// per block (45 for 722 unknowns) whose cost was launch latency, not arithmetic.
// Launched cooperatively: every workgroup is resident, and every workgroup reaches every
// barrier (the loop bounds depend on n only).
// All traffic on the shared matrix goes through agent-scope relaxed atomics (sc1 loads and
// write-through stores, served at the memory side of the per-XCD L2s), so the barrier needs
// no cache maintenance: an agent-scope release / acquire pair would write back and
// invalidate the whole L2 twice per step (measured: 3.5 x slower than separate launches).
__device__ inline double ld_sh(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline void st_sh(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ctr[0]: arrivals (monotone); ctr[1]: set when a workgroup gave up waiting (the launch is
// sized to be fully resident, so this only happens when something else holds the GPU).
// The wait is bounded: no hang, the factorisation is reported as failed instead.
#define CF_SPIN_LIMIT (1 << 16)
__device__ inline bool region_barrier(unsigned* ctr, unsigned target, int spin_limit) {
    __shared__ int dead;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's write-through stores are done
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0, d = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > spin_limit || __hip_atomic_load(ctr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                d = 1;
                break;
            }
        }
        dead = d;
    }
    __syncthreads();
    return dead != 0;
}

// arrival without waiting (the look-ahead workgroup at the panel barrier)
__device__ inline void region_arrive(unsigned* ctr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the panel chain of k_chol_fused on a half strip held as xb[column half][row]: unscaled partial sums, the
// reciprocal diagonal at the end
__device__ __forceinline__ void cf_chain(double (&xb)[2][2], const double2 (*Cf)[16], const double* Rd, int li,
                                         double (&x0)[2], double (&x1)[2]) {
    // The coefficients of step m are LDS reads; left to the scheduler they are issued one step ahead of their use
    // and every step of the chain then waits out an LDS round trip (~100 cycles against ~40 of broadcast + FMA).
    // Here they come in batches of eight steps, a batch ahead: the scheduling barrier keeps the reads of batch
    // b + 1 in front of the arithmetic of batch b.
    double2 cf[4][8];
#pragma unroll
    for (int q = 0; q < 8; ++q) cf[0][q] = Cf[q][li];
    hp_static_for<0, 4>([&](auto B) __attribute__((always_inline)) {
        constexpr int b = decltype(B)::value;
        if constexpr (b < 3) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (8 * (b + 1) + q < CH_NB - 1) cf[b + 1][q] = Cf[8 * (b + 1) + q][li];
        }
        __builtin_amdgcn_sched_barrier(0);
        // Round 5: x += c * (lane sl's u) as ONE instruction - v_fmac_f64 with its first operand through the DPP
        // row broadcast - instead of two 32-bit DPP moves and the FMA: 216 -> 108 instructions per chain, on a wave
        // that issues one per ~5 cycles.  The same products and sums.  A DPP operand written by one of the two
        // instructions before is read stale (no interlock): within a step the order below keeps two instructions
        // between every write of a source and its next broadcast (steps >= 15 touch only the second column half, with
        // an s_nop for the distance); lane sl's own coefficient c.x is 0 - its entry is final - so the source of a
        // step is not changed by the step.  The eight steps of a batch are ONE asm statement that opens with its own
        // s_nop: the compiler cannot put a register copy between two of them (it did, in k_chol_fused2, between
        // steps 13 and 14 of the per-step form - tests/test_isa_lint.py found it and now guards every DPP read).
        if constexpr (b == 0)
            asm volatile("s_nop 1\n\t"
                         "v_fmac_f64_dpp %1, %0, %5 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %0, %0, %4 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %3, %2, %5 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %2, %2, %4 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %1, %0, %7 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %0, %0, %6 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %3, %2, %7 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %2, %2, %6 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %1, %0, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %0, %0, %8 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %3, %2, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %2, %2, %8 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %1, %0, %11 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %0, %0, %10 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %3, %2, %11 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %2, %2, %10 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %1, %0, %13 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %0, %0, %12 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %3, %2, %13 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %2, %2, %12 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %1, %0, %15 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %0, %0, %14 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %3, %2, %15 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %2, %2, %14 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %1, %0, %17 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %0, %0, %16 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %3, %2, %17 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %2, %2, %16 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %1, %0, %19 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %0, %0, %18 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %3, %2, %19 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %2, %2, %18 row_newbcast:7 row_mask:0xf bank_mask:0xf"
                         : "+v"(xb[0][0]), "+v"(xb[1][0]), "+v"(xb[0][1]), "+v"(xb[1][1])
                         : "v"(cf[0][0].x), "v"(cf[0][0].y), "v"(cf[0][1].x), "v"(cf[0][1].y), "v"(cf[0][2].x), "v"(cf[0][2].y), "v"(cf[0][3].x), "v"(cf[0][3].y), "v"(cf[0][4].x), "v"(cf[0][4].y), "v"(cf[0][5].x), "v"(cf[0][5].y), "v"(cf[0][6].x), "v"(cf[0][6].y), "v"(cf[0][7].x), "v"(cf[0][7].y));
        if constexpr (b == 1)
            asm volatile("s_nop 1\n\t"
                         "v_fmac_f64_dpp %1, %0, %5 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %0, %0, %4 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %3, %2, %5 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %2, %2, %4 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %1, %0, %7 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %0, %0, %6 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %3, %2, %7 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %2, %2, %6 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %1, %0, %9 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %0, %0, %8 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %3, %2, %9 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %2, %2, %8 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %1, %0, %11 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %0, %0, %10 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %3, %2, %11 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %2, %2, %10 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %1, %0, %13 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %0, %0, %12 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %3, %2, %13 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %2, %2, %12 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %1, %0, %15 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %0, %0, %14 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %3, %2, %15 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %2, %2, %14 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %1, %0, %17 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %0, %0, %16 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %3, %2, %17 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %2, %2, %16 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 0\n\t"
                         "v_fmac_f64_dpp %1, %0, %19 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %3, %2, %19 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 0"
                         : "+v"(xb[0][0]), "+v"(xb[1][0]), "+v"(xb[0][1]), "+v"(xb[1][1])
                         : "v"(cf[1][0].x), "v"(cf[1][0].y), "v"(cf[1][1].x), "v"(cf[1][1].y), "v"(cf[1][2].x), "v"(cf[1][2].y), "v"(cf[1][3].x), "v"(cf[1][3].y), "v"(cf[1][4].x), "v"(cf[1][4].y), "v"(cf[1][5].x), "v"(cf[1][5].y), "v"(cf[1][6].x), "v"(cf[1][6].y), "v"(cf[1][7].x), "v"(cf[1][7].y));
        if constexpr (b == 2)
            asm volatile("s_nop 1\n\t"
                         "v_fmac_f64_dpp %1, %1, %5 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %3, %3, %5 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 0\n\t"
                         "v_fmac_f64_dpp %1, %1, %7 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %3, %3, %7 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 0\n\t"
                         "v_fmac_f64_dpp %1, %1, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %3, %3, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 0\n\t"
                         "v_fmac_f64_dpp %1, %1, %11 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %3, %3, %11 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 0\n\t"
                         "v_fmac_f64_dpp %1, %1, %13 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %3, %3, %13 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 0\n\t"
                         "v_fmac_f64_dpp %1, %1, %15 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %3, %3, %15 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 0\n\t"
                         "v_fmac_f64_dpp %1, %1, %17 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %3, %3, %17 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 0\n\t"
                         "v_fmac_f64_dpp %1, %1, %19 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %3, %3, %19 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 0"
                         : "+v"(xb[0][0]), "+v"(xb[1][0]), "+v"(xb[0][1]), "+v"(xb[1][1])
                         : "v"(cf[2][0].x), "v"(cf[2][0].y), "v"(cf[2][1].x), "v"(cf[2][1].y), "v"(cf[2][2].x), "v"(cf[2][2].y), "v"(cf[2][3].x), "v"(cf[2][3].y), "v"(cf[2][4].x), "v"(cf[2][4].y), "v"(cf[2][5].x), "v"(cf[2][5].y), "v"(cf[2][6].x), "v"(cf[2][6].y), "v"(cf[2][7].x), "v"(cf[2][7].y));
        if constexpr (b == 3)
            asm volatile("s_nop 1\n\t"
                         "v_fmac_f64_dpp %1, %1, %5 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %3, %3, %5 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 0\n\t"
                         "v_fmac_f64_dpp %1, %1, %7 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %3, %3, %7 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 0\n\t"
                         "v_fmac_f64_dpp %1, %1, %9 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %3, %3, %9 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 0\n\t"
                         "v_fmac_f64_dpp %1, %1, %11 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %3, %3, %11 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 0\n\t"
                         "v_fmac_f64_dpp %1, %1, %13 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %3, %3, %13 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 0\n\t"
                         "v_fmac_f64_dpp %1, %1, %15 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %3, %3, %15 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 0\n\t"
                         "v_fmac_f64_dpp %1, %1, %17 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %3, %3, %17 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
                         "s_nop 0"
                         : "+v"(xb[0][0]), "+v"(xb[1][0]), "+v"(xb[0][1]), "+v"(xb[1][1])
                         : "v"(cf[3][0].x), "v"(cf[3][0].y), "v"(cf[3][1].x), "v"(cf[3][1].y), "v"(cf[3][2].x), "v"(cf[3][2].y), "v"(cf[3][3].x), "v"(cf[3][3].y), "v"(cf[3][4].x), "v"(cf[3][4].y), "v"(cf[3][5].x), "v"(cf[3][5].y), "v"(cf[3][6].x), "v"(cf[3][6].y));
    });
    const double r0 = Rd[li], r1 = Rd[16 + li];
#pragma unroll
    for (int h = 0; h < 2; ++h) { x0[h] = xb[0][h] * r0; x1[h] = xb[1][h] * r1; }
}

// A: [reg][(n + 1)][lda], lda = n rounded up to 16 doubles so that every 16-column segment
// of a tile is one 128-B line; Dg: [reg][2][32][33] published diagonal factors (two slots: the factor
// of block k + 1 is written while slower workgroups may still read that of block k).
// Step k, W >= 2 workgroups per region:
//   all   read the published factor of block k, solve their panel rows X L^T = B (register
//         chain, two rows per wave).  Workgroup 0 owns the first 32 panel rows - the rows of
//         the next diagonal block - the others share the rest.                 | arrive b1
//   wg 0  does not wait: next diagonal block = A11' - X X^T from its own rows, factors it (one
//         wave, block in registers) and publishes it, while
//   wg>0  wait b1, then update the trailing 64 x 64 tiles on the f64 matrix cores (tile 0
//         without the corner that workgroup 0 holds)                           | barrier b2
// so the 32 x 32 factorisation - the longest serial piece - overlaps the panel barrier and the
// trailing update instead of following them.
// fail[reg]: pivots that had to be clamped (the matrix is not positive definite: a property of the
// fit); tmo[reg]: a barrier gave up waiting (a property of the launch: not every workgroup was
// resident - the host repeats the fit on k_chol_tp, which has no region barrier and waits for nobody)
__global__ __launch_bounds__(256) void k_chol_fused(int n, int lda, int W, double* Aall, double* Dgall, int* fail,
                                                    int* tmo, int spin_limit, unsigned* bar, long long* prof,
                                                    const int* __restrict__ guard) {
    if (guard && *guard == 0) return;                    // the previous round rejected nothing: this round is void
    __shared__ double D[CH_NB][CH_NB + 1];
    __shared__ double Li[64][CH_NB + 2];        // pitch 34: conflict-free ds_read_b64 of MFMA operands
    __shared__ double Lj[64][CH_NB + 2];
    // coefficients of the panel chains from the current factor: Cf[m][li] = {c[li][m], c[li + 16][m]},
    // c[i][m] = -(L[i][m] / L[m][m]) for i > m, else 0; Rd[i] = 1 / L[i][i] (what is published per block)
    __shared__ double2 Cf[CH_NB][16];
    __shared__ double Rd[CH_NB];
    const int reg = blockIdx.x / W, w = blockIdx.x - reg * W;
    double* A = Aall + (size_t)reg * (size_t)(n + 1) * lda;
    double* Dg2 = Dgall + (size_t)reg * 2 * CH_NB * (CH_NB + 1);
    unsigned* ctr1 = bar + reg * CF_BAR_STRIDE;          // [0] panel barrier, [1] timed-out flag,
    unsigned* ctr2 = ctr1 + 2;                           // [2] update barrier (its flag is [3])
    const int tid = threadIdx.x;
    const int nrows = n + 1;
    const int nblk = (n + CH_NB - 1) / CH_NB;
    unsigned gen1 = 0, gen2 = 0;
    long long pt[6] = {0, 0, 0, 0, 0, 0}, tc = 0;      // phase clocks (100 MHz), prof != NULL only
#define CF_TICK(k) do { if (prof) { long long t_ = wall_clock64(); pt[k] += t_ - tc; tc = t_; } } while (0)
    if (prof) tc = wall_clock64();
    // factor the diagonal block held (unfactored, lower triangle) in D; publish it in `slot`
    auto factor_and_publish = [&](int k0, int nb, int slot) {
        double* Dg = Dg2 + (size_t)slot * CH_NB * (CH_NB + 1);
        __syncthreads();
        if (tid < 64) chol_diag_wave_panel_t<0>(D, nb, &fail[reg]);
        __syncthreads();
        // published: the chain coefficients and reciprocal diagonals (what every workgroup's panel solve
        // reads; this workgroup keeps them in Cf / Rd for its own rows); L itself goes to A
        for (int e = tid; e < CH_NB * CH_NB; e += 256) {
            const int m = e >> 5, i = e & 31;                      // column m of row i
            const double cv = (i > m) ? -(D[i][m] * D[m][CH_NB]) : 0.0;
            reinterpret_cast<double*>(&Cf[m][i & 15])[i >> 4] = cv;
            st_sh(&Dg[2 * (m * 16 + (i & 15)) + (i >> 4)], cv);
            if (i < nb && m <= i) st_sh(&A[(size_t)(k0 + i) * lda + k0 + m], D[i][m]);
        }
        if (tid < CH_NB) {
            Rd[tid] = D[tid][CH_NB];
            st_sh(&Dg[CH_NB * CH_NB + tid], D[tid][CH_NB]);
        }
    };
    if (w == 0) {
        const int nb0 = min(CH_NB, n);
        for (int e = tid; e < CH_NB * CH_NB; e += 256) {
            const int i = e >> 5, j = e & 31;
            D[i][j] = (i < nb0 && j <= i) ? ld_sh(&A[(size_t)i * lda + j]) : (i == j ? 1.0 : 0.0);
        }
        factor_and_publish(0, nb0, 0);
    }
    gen2 += W;
    bool dead = region_barrier(ctr2, gen2, spin_limit);
    const int wave = tid >> 6, lane = tid & 63;
    for (int kb = 0; kb < nblk && !dead; ++kb) {
        const int k0 = kb * CH_NB;
        const int nb = min(CH_NB, n - k0);
        const int k1 = k0 + nb;                                  // first trailing row / column
        const int below = nrows - k1;                            // panel rows (the rhs row included)
        const int nbn = max(0, min(CH_NB, n - k1));              // size of the next diagonal block
        // rows [0, r0n) of the panel belong to workgroup 0, the rest is dealt to the others
        const int r0n = min(CH_NB, below);
        const int per = (W > 1) ? (below - r0n + W - 2) / (W - 1) : 0;
        const int pbeg = (w == 0) ? 0 : r0n + (w - 1) * per;
        // (a lone workgroup, W = 1 - kept for experiments; the host's fallback is k_chol_tp - solves every panel row itself)
        const int pend = (w == 0) ? (W == 1 ? below : r0n) : min(r0n + w * per, below);
        // workgroup 0: the unfactored next diagonal block, fetched ahead of its use
        // (in the accumulator layout of the f64 matrix cores: wave = 16 x 16 quadrant (ti, tj) of
        // the block, column = lane & 15, row = (lane >> 4) + 4 reg)
        double cpre[4] = {0.0, 0.0, 0.0, 0.0};
        if (w == 0 && nb == CH_NB)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // (entries outside the block or above its diagonal are replaced when D is formed: clamped, not masked)
                const int i = 16 * (wave >> 1) + (lane >> 4) + 4 * q, j = 16 * (wave & 1) + (lane & 15);
                cpre[q] = ld_sh(&A[(size_t)min(k1 + i, nrows - 1) * lda + min(k1 + j, n - 1)]);
            }
        // trailing update: 64 x 64 tiles of the lower triangle dealt to workgroups 1 .. W - 1.
        // The loop over a workgroup's tiles is software-pipelined (the next tile is requested
        // before the matrix cores run on the current one), and the accumulators of its first tile
        // - trailing-matrix entries, final since the last barrier - are requested here, ahead of
        // the panel solve, so that only the panels are left to fetch after the panel barrier.
        const int T = (max(below, 0) + 63) / 64;
        const int ntile = T * (T + 1) / 2;
        const int Wu = (W > 1) ? W - 1 : 1, wu = (W > 1) ? w - 1 : 0;
        auto tile_of = [&](int t, int& i0, int& j0) {
            int ti = (int)((sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);
            while (ti * (ti + 1) / 2 > t) --ti;
            while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
            const int tj = t - ti * (ti + 1) / 2;
            i0 = k1 + ti * 64;
            j0 = k1 + tj * 64;
        };
        // f64 matrix cores: wave v owns rows 16 v .. 16 v + 15 of the tile; C layout of
        // v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 reg.
        const int li = lane & 15, lk = lane >> 4;
        double4_t accn[4];
        double pa[8], pb[8];
        auto tile_fetch_acc = [&](int t) {
            int i0, j0;
            tile_of(t, i0, j0);
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    // (entries outside the matrix, above the diagonal or in workgroup 0's corner are never
                    // stored: they are fetched from a clamped address instead of being masked - sixteen 64-bit
                    // predicates held across the loads overflowed the scalar registers into VGPR lanes)
                    const int i = i0 + 16 * wave + lk + 4 * rg, j = j0 + 16 * c + li;
                    accn[c][rg] = ld_sh(&A[(size_t)min(i, nrows - 1) * lda + min(j, n - 1)]);
                }
        };
        auto tile_fetch_panels = [&](int t) {
            int i0, j0;
            tile_of(t, i0, j0);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = tid + 256 * q, r = e >> 5, m = e & 31;
                // (negated when it is written to LDS: arithmetic here would make every one of these
                // loads wait for itself)
                // (rows beyond the matrix feed tile rows / columns that are never stored: clamped, not masked)
                pa[q] = ld_sh(&A[(size_t)min(i0 + r, nrows - 1) * lda + k0 + m]);
                pb[q] = ld_sh(&A[(size_t)min(j0 + r, nrows - 1) * lda + k0 + m]);
            }
        };
        if ((w != 0 || W == 1) && nb == CH_NB && wu < ntile) tile_fetch_acc(wu);
        // (b) panel rows X L^T = B.  A wave takes 16-row strips of its workgroup's slice in the accumulator layout
        // of the matrix cores (row = lk + 4 rg, column = li + 16 c: loads and stores along rows) and runs the
        // chain on them in place: step m broadcasts column m inside each row of 16 lanes (DPP row_newbcast) and
        // every lane subtracts u_m c[col][m] - the unscaled chain of the first form of this kernel (lane = column,
        // two rows per wave, readlanes), operation for operation, at a third of its instructions per row; the
        // form k_chol_tp runs.  The first strip is requested before anything waits: it was final at the last
        // barrier, so its latency overlaps that of the published coefficients.  Rows beyond the slice come from
        // clamped addresses and are not stored.
        const int nslice = max(pend - pbeg, 0);
        // (a wave takes HALF a strip - rows lk + 4 rg for two of the four rg - so that the 27 rows a workgroup has
        // at 26 workgroups per region keep all four waves busy: the chain is as long, its steps half as wide)
        const int ntask = 2 * ((nslice + 15) >> 4);
        double xb[2][2];
        auto panel_fetch = [&](int t) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int prow = k1 + pbeg + 16 * (t >> 1) + lk + 4 * (2 * (t & 1) + h);
                    xb[c][h] = ld_sh(&A[(size_t)min(prow, nrows - 1) * lda + min(k0 + 16 * c + li, n - 1)]);
                }
        };
        if (wave < ntask) panel_fetch(wave);
        // (a) the published coefficients of block kb; workgroup 0 computed them itself and still has them
        __syncthreads();                                         // Cf / Rd of the previous step are consumed
        if (w != 0) {
            // all loads of a thread first, then the LDS stores: one latency
            const double* Dg = Dg2 + (size_t)(kb & 1) * CH_NB * (CH_NB + 1);
            constexpr int ND = (CH_NB * (CH_NB + 1) + 255) / 256;
            double dv[ND];
#pragma unroll
            for (int q = 0; q < ND; ++q) dv[q] = ld_sh(&Dg[min(tid + 256 * q, CH_NB * (CH_NB + 1) - 1)]);
#pragma unroll
            for (int q = 0; q < ND; ++q) {
                const int e = tid + 256 * q;
                if (e < CH_NB * CH_NB) reinterpret_cast<double*>(&Cf[0][0])[e] = dv[q];
                else if (e < CH_NB * (CH_NB + 1)) Rd[e - CH_NB * CH_NB] = dv[q];
            }
        }
        __syncthreads();
        CF_TICK(0);
        for (int t = wave; t < ntask; t += 4) {
            // columns of a partial (last) block beyond nb count as zero, as the masked loads of the first form did
            if (nb < CH_NB) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int h = 0; h < 2; ++h) xb[c][h] = (16 * c + li < nb) ? xb[c][h] : 0.0;
            }
            double x0[2], x1[2];
            cf_chain(xb, Cf, Rd, li, x0, x1);
            const int tcur = t;
            if (t + 4 < ntask) panel_fetch(t + 4);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int pr = 16 * (tcur >> 1) + lk + 4 * (2 * (tcur & 1) + h), p = pbeg + pr;
                if (pr < nslice) {
                    if (li < nb) st_sh(&A[(size_t)(k1 + p) * lda + k0 + li], x0[h]);
                    if (16 + li < nb) st_sh(&A[(size_t)(k1 + p) * lda + k0 + 16 + li], x1[h]);
                }
                if (w == 0 && p < CH_NB) {
                    Li[p][li] = (pr < nslice) ? x0[h] : 0.0;
                    Li[p][16 + li] = (pr < nslice) ? x1[h] : 0.0;
                }
            }
        }
        if (below <= 0 || nb < CH_NB) break;      // nothing trails the last (partial) block
        CF_TICK(2);
        if (w == 0 && W > 1) {
            // panel rows published: arrive, do not wait - nothing below needs the others' rows
            region_arrive(ctr1);
            gen1 += W;
            // next diagonal block (k1 .. k1 + nbn) = prefetched A - X X^T, X = rows 0 .. nbn - 1
            // of this workgroup's panel slice (in Li); then factor and publish it
            for (int p = pend; p < CH_NB; ++p)                   // rows this slice does not have
                if (tid < CH_NB) Li[p][tid] = 0.0;
            __syncthreads();
            {
                // one quadrant per wave on the f64 matrix cores (8 MFMAs, 16 LDS reads per lane
                // instead of 256 for the scalar dot products); the upper quadrant is not needed
                const int ti = wave >> 1, tj = wave & 1;
                double4_t c4 = {cpre[0], cpre[1], cpre[2], cpre[3]};
                if (tj <= ti) {
#pragma unroll
                    for (int kk = 0; kk < CH_NB / 4; ++kk) {
                        const double a = -Li[16 * ti + li][4 * kk + lk];
                        const double b = Li[16 * tj + li][4 * kk + lk];
                        c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c4, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = 16 * ti + lk + 4 * q, j = 16 * tj + li;
                    D[i][j] = (i < nbn && j <= i) ? c4[q] : ((i == j) ? 1.0 : 0.0);
                }
            }
            factor_and_publish(k1, nbn, (kb + 1) & 1);
            CF_TICK(4);
        } else {
            gen1 += W;
            dead = region_barrier(ctr1, gen1, spin_limit);
            CF_TICK(3);
            if (dead) break;
            // (c) trailing update A22 -= L21 L21^T on 64 x 64 tiles of the lower triangle; the
            // corner of tile 0 (the next diagonal block) is workgroup 0's
            int t = wu;
            if (t < ntile) tile_fetch_panels(t);
            while (t < ntile) {
                int i0, j0;
                tile_of(t, i0, j0);
                double4_t acc[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = accn[c];
                __syncthreads();                       // the previous tile's panels are consumed
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int e = tid + 256 * q, r = e >> 5, m = e & 31;
                    Li[r][m] = -pa[q];
                    Lj[r][m] = pb[q];
                }
                __syncthreads();
                CF_TICK(1);
                const int tn = t + Wu;
                if (tn < ntile) { tile_fetch_acc(tn); tile_fetch_panels(tn); }
#pragma unroll
                for (int kk = 0; kk < CH_NB / 4; ++kk) {
                    const double a = Li[16 * wave + li][4 * kk + lk];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const double b = Lj[16 * c + li][4 * kk + lk];
                        acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
                    }
                }
                if (W == 1 && t == 0) {
                    // a lone workgroup factors the next block here, straight from the accumulators
                    __syncthreads();
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) {
                            const int i = 16 * wave + lk + 4 * rg, j = 16 * c + li;
                            if (i < CH_NB) D[i][j] = (i < nbn && j <= i && j < nbn) ? acc[c][rg] : (i == j ? 1.0 : 0.0);
                        }
                }
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const int i = i0 + 16 * wave + lk + 4 * rg, j = j0 + 16 * c + li;
                        const bool corner = (t == 0) && i < k1 + nbn;              // stored factored instead
                        if (i < nrows && j < n && j <= i && !corner) st_sh(&A[(size_t)i * lda + j], acc[c][rg]);
                    }
                if (W == 1 && t == 0) factor_and_publish(k1, nbn, (kb + 1) & 1);
                t = tn;
            }
            CF_TICK(4);
        }
        gen2 += W;
        dead = region_barrier(ctr2, gen2, spin_limit);
        CF_TICK(5);
    }
    if (dead && tid == 0) atomicAdd(&tmo[reg], 1);
    if (prof && tid == 0)
        for (int k = 0; k < 6; ++k) prof[blockIdx.x * 6 + k] = pt[k];
#undef CF_TICK
}

// (Round 4 also built this factorisation with 64-column super-steps, k_chol_fused2: the same bits, 301 against 263 us -
// the look-ahead workgroup became the critical path.  Measured, written down in DESIGN.md section 4, removed in
// round 6.)
// ---- the same factorisation, cheap in CU-time instead of short --------------------------
// ---- the factorisation as a data-flow of resident tiles (round 4) ---------------------------
// k_chol_fused moves every trailing entry through memory once per block step (accumulators in, accumulators
// out, two panels per tile: ~400 MB per launch, the early steps are bandwidth-bound) and closes every step with
// two region-wide barriers.  Here the lower triangle is cut ONCE into 64 x 64 tiles on a fixed grid; every tile
// has one owner for the whole factorisation and lives in that workgroup's LDS (the accumulator layout of the
// matrix cores, lane-major: 32 KB per tile, three tiles per workgroup) from the first step to its last:
//   * a step of 32 columns (block kb = half h of tile column J) touches memory only for what really changes
//     hands: the factored diagonal block's chain coefficients (Dg[kb]), the solved panel rows (written to A,
//     where the back substitution wants them anyway) and the panels a tile's owner has to read for an update;
//   * no barrier: a flag per (block, tile row) says "these panel rows are in A", a flag per block "the chain
//     coefficients are published"; a workgroup waits for exactly what its next operation reads, and workgroups
//     run as far ahead as their inputs allow;
//   * the critical path - factor block kb, solve the 32 rows below it, update and factor block kb + 1 - stays
//     inside ONE workgroup for two blocks (a diagonal tile holds both) and changes hands once per 64 columns:
//     workgroup d owns the diagonal tile (d, d) AND its left neighbour (d, d - 1), so that everything the
//     next diagonal block needs after the hand-over is computed by its owner from its own LDS.
// Ownership (host and device run the same rule, hp_df_owner): workgroup d < NC (tile columns) owns (d, d) and
// (d, d - 1); the other tiles ("far": two or more tile rows below the diagonal), heaviest first (they take part
// in the most steps), go three each to the workgroups NC .. W - 1 and then fill the diagonal owners up to three.
// Order inside a step: a workgroup first solves the panel tiles it owns (others wait for them), then updates,
// nearest tile column first.  Every wait is a bounded spin on a word only its producer writes; producers never
// wait for consumers, a step's solves wait only for that step's factor and its updates only for its solves:
// no cycle, given that all workgroups are resident (the launch is sized like k_chol_fused's).
// The arithmetic per entry is k_chol_fused's, operation for operation (chol_diag.h for the diagonal block, the
// cf_chain for the panel rows, v_mfma_f64_16x16x4 with the negated row operand in ascending chunks of four
// columns for the updates): same bits (tests/test_subtract_gpu.py compares all forms).
#define DF_THREADS 512
#define DF_MAXT 3                        // resident tiles per workgroup
#define DF_LP (CH_NB + 2)                // pitch of the operand panels: conflict-free b64 reads
struct df_lds {
    double T[DF_MAXT][4][4][4][64];      // [slot][column sixteenth c][register rg][strip s][lane]
    double Li[64][DF_LP];                // row operand of an update: -L[tile row rows][block columns]
    double Lj[64][DF_LP];                // column operand: L[tile column rows][block columns]
    double D[CH_NB][CH_NB + 1];
    double2 Cf[CH_NB][16];
    double Rd[CH_NB];
    double dd[DF_MAXT][2][64];           // Jacobi scale factors of a tile's rows / columns (the scaling folded into the tile load)
    int tI[DF_MAXT], tJ[DF_MAXT], nt;
    int dead;
};
// words per region: [0] a wait gave up, [1 + kb] factor of block kb published, [1 + nblk + kb * NT + I] the
// panel rows of tile row I for block kb are in A
__host__ __device__ inline int hp_df_nflags(int n) {
    const int nblk = (n + CH_NB - 1) / CH_NB, NT = (n + 1 + 63) / 64;
    return 1 + nblk + nblk * NT;
}
// the owner of tile (I, J) and the number of tiles workgroup `me` owns (list in tI / tJ, in processing order:
// tile column ascending, then tile row); returns false when the tiles do not fit 3 per workgroup
inline bool hp_df_owner(int n, int W, int me, int* tI, int* tJ, int* nt) {
    const int NT = (n + 1 + 63) / 64, NC = (n + 63) / 64;
    if (W < NC + 1 && !(NC == 1 && W >= 1)) return false;
    int cnt[128];
    if (W > 128) return false;
    for (int w = 0; w < W; ++w) cnt[w] = 0;
    int mine = 0;
    auto give = [&](int w, int I, int J) {
        if (w == me && mine < DF_MAXT) { tI[mine] = I; tJ[mine] = J; ++mine; }
        ++cnt[w];
    };
    for (int d = 0; d < NC; ++d) {
        give(d, d, d);
        if (d >= 1) give(d, d, d - 1);
    }
    const int nfar = W - NC;
    int nextfar = 0;
    bool ok = true;
    for (int J = NC - 1; J >= 0; --J)
        for (int I = J; I < NT; ++I) {
            if (I < NC && I - J <= 1) continue;                       // a diagonal owner's tile
            int w = -1;
            for (int tries = 0; tries < nfar; ++tries) {              // the far workgroups in turn, three each
                const int c = NC + (nextfar + tries) % nfar;
                if (cnt[c] < DF_MAXT) { w = c; nextfar = (nextfar + tries + 1) % nfar; break; }
            }
            if (w < 0)
                for (int d = 0; d < NC; ++d)
                    if (cnt[d] < DF_MAXT) { w = d; break; }
            if (w < 0) { ok = false; continue; }
            give(w, I, J);
        }
    // processing order: tile column, then tile row (insertion sort of at most three)
    for (int a = 1; a < mine; ++a)
        for (int b = a; b > 0 && (tJ[b] < tJ[b - 1] || (tJ[b] == tJ[b - 1] && tI[b] < tI[b - 1])); --b) {
            int t = tI[b]; tI[b] = tI[b - 1]; tI[b - 1] = t;
            t = tJ[b]; tJ[b] = tJ[b - 1]; tJ[b - 1] = t;
        }
    *nt = mine;
    return ok;
}

// tiles: [W][1 + 2 DF_MAXT] ints per workgroup of a region - the number of its tiles, then (I, J) in processing
// order (hp_df_owner on the host, the same table for every region)
//
// The Jacobi scaling rides in the tile load (A0all != nullptr): a workgroup reads its tiles from the UNSCALED normal
// matrix (pitch n; row n = the right-hand side), forms d = sqrt(diag) for the rows and columns of its tiles and
// divides on the way into LDS - k_hp_diag's and k_hp_scale's operations on every entry, the same bits - and the
// owner of a diagonal tile writes d for the back substitution.  Every entry of the factor's lower triangle and of
// the right-hand-side row is written to A by the solves, so A needs no initialisation: two launches and one pass
// over both matrices per rejection round are gone.
// PROF: wall-clock sums per phase of every workgroup (ZM_CHOL_PROF=1): 0 tile load, 1 waiting for a flag,
// 2 coefficient load, 3 diagonal block out + factor, 4 publish, 5 panel solve + its flag, 6 panel loads, 7 updates.
template <int PROF>
__global__ __launch_bounds__(DF_THREADS) void k_chol_df(int n, int lda, int W, double* Aall, double* Dgall, int* fail,
                                                        int* tmo, int spin_limit, unsigned* flags_all,
                                                        const int* __restrict__ tiles, const int* __restrict__ guard,
                                                        const double* __restrict__ A0all, const double* __restrict__ rhs0all,
                                                        double* __restrict__ dscall, long long* __restrict__ prof) {
    if (guard && *guard == 0) return;                    // the previous round rejected nothing: this round is void
    long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tc = 0;
#define DF_TICK(k) do { if (PROF == 1) { const long long t_ = wall_clock64(); pt[k] += t_ - tc; tc = t_; } } while (0)
    // (PROF: wall-clock stamps of the critical chain per block behind the per-workgroup sums: [0] diagonal block
    // taken up, [1] factored, [2] published by its owner; [3] coefficients in LDS, [4] rows solved, [5] second half /
    // own diagonal tile updated at the owner of the tile below it)
#define DF_STAMP(kb_, e_) do { if (PROF == 2 && prof && threadIdx.x == 0) prof[(size_t)gridDim.x * 8 + ((size_t)(blockIdx.x / W) * ((n + CH_NB - 1) / CH_NB) + (kb_)) * 8 + (e_)] = wall_clock64(); } while (0)
    if (PROF == 1) tc = wall_clock64();
    extern __shared__ char df_raw[];
    df_lds& S = *reinterpret_cast<df_lds*>(df_raw);
    // (the region and the workgroup's number in it are wave-uniform, but a division leaves them in vector registers -
    // and with them every pointer derived below, two registers each for the whole kernel: readfirstlane moves the
    // lot to the scalar file; round 5)
    const int reg = __builtin_amdgcn_readfirstlane((int)(blockIdx.x / W)), w = __builtin_amdgcn_readfirstlane((int)(blockIdx.x - reg * W));
    const int nrows = n + 1;
    const int nblk = (n + CH_NB - 1) / CH_NB, NT = (n + 1 + 63) / 64;
    double* A = Aall + (size_t)reg * (size_t)nrows * lda;
    double* Dgr = Dgall + (size_t)reg * nblk * CH_NB * (CH_NB + 1);
    unsigned* F = flags_all + (size_t)reg * hp_df_nflags(n);
    unsigned* Fdiag = F + 1;
    unsigned* Fpan = F + 1 + nblk;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int ws = wave >> 1, wp = wave & 1;             // strip (16 tile rows) and half of this wave
    if (tid == 0) {
        const int* tl = tiles + w * (1 + 2 * DF_MAXT);
        S.nt = tl[0];
        for (int k = 0; k < DF_MAXT; ++k) { S.tI[k] = tl[1 + 2 * k]; S.tJ[k] = tl[2 + 2 * k]; }
        S.dead = 0;
    }
    __syncthreads();
    const int nt = S.nt;
    if (nt == 0) return;                                 // (more workgroups than tiles: small systems)
    // ---- the resident tiles: this wave's planes are (c = 2 wp, 2 wp + 1; rg = 0 .. 3; strip ws)
    const double* A0 = A0all ? A0all + (size_t)reg * (size_t)(n + 1) * n : nullptr;
    const double* r0 = A0all ? rhs0all + (size_t)reg * n : nullptr;
    if (A0all) {
        double* dsc = dscall + (size_t)reg * n;
        // d = sqrt(diag) (1 where the diagonal is not positive: k_hp_diag) for the rows and columns of the tiles;
        // the right-hand-side row (index n) is not scaled by a row factor
        for (int e = tid; e < nt * 128; e += DF_THREADS) {
            const int sl = e >> 7, col = (e >> 6) & 1, k = e & 63;
            const int idx = 64 * (col ? S.tJ[sl] : S.tI[sl]) + k;
            double d = 1.0;
            if (idx < n) {
                const double v = A0[(size_t)idx * n + idx];
                d = v > 0.0 ? sqrt(v) : 1.0;
                if (col && S.tI[sl] == S.tJ[sl]) dsc[idx] = d;
            }
            S.dd[sl][col][k] = 1.0 / d;                  // (k_hp_scale multiplies by the reciprocals)
        }
        __syncthreads();
    }
    // every load of the (up to three) tiles goes out before the first value is used
    {
        double a[DF_MAXT][2][4];
#pragma unroll
        for (int sl = 0; sl < DF_MAXT; ++sl) {
            const int I = S.tI[min(sl, nt - 1)], J = S.tJ[min(sl, nt - 1)];
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int c = 2 * wp + cc;
                    const int i = min(64 * I + 16 * ws + lk + 4 * rg, nrows - 1), j = min(64 * J + 16 * c + li, n - 1);
                    if (A0) a[sl][cc][rg] = (i < n) ? A0[(size_t)i * n + j] : r0[j];
                    else a[sl][cc][rg] = ld_sh(&A[(size_t)i * lda + j]);
                }
        }
#pragma unroll
        for (int sl = 0; sl < DF_MAXT; ++sl) {
            if (sl >= nt) break;
            const int I = S.tI[sl], J = S.tJ[sl];
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int c = 2 * wp + cc;
                    const int i = min(64 * I + 16 * ws + lk + 4 * rg, nrows - 1), j = min(64 * J + 16 * c + li, n - 1);
                    double v = a[sl][cc][rg];
                    if (A0) {
                        v = __dmul_rn(v, __dmul_rn(S.dd[sl][0][i - 64 * I], S.dd[sl][1][j - 64 * J]));
                        if (i == j) v = __dadd_rn(v, HP_RIDGE);          // (k_hp_scale: keeps a rank-deficient basis solvable)
                    }
                    S.T[sl][c][rg][ws][lane] = v;
                }
        }
    }
    DF_TICK(0);
    // this workgroup's stores to A / Dg are complete, then the word is set - together with a panel flag that was
    // left pending: after a panel solve the workgroup carries on with what only needs the solved rows in LDS (its
    // own updates: the critical path) and lets the stores to A land meanwhile; the flag goes out at the next point
    // where the workgroup would wait or publish anyway (producers still never wait for consumers)
    unsigned *pend0 = nullptr, *pend1 = nullptr, *pend2 = nullptr;
    int npend = 0;
    auto set_flag = [&](unsigned* f) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (npend > 0) __hip_atomic_store(pend0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (npend > 1) __hip_atomic_store(pend1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (npend > 2) __hip_atomic_store(pend2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (f) __hip_atomic_store(f, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        npend = 0;
    };
    auto flush = [&]() { if (npend) set_flag(nullptr); };
    // a bounded wait for a word to become non-zero (thread 0 polls; everybody learns the outcome)
    auto wait_flag = [&](const unsigned* f) -> bool {
        if (npend) set_flag(nullptr);
        __syncthreads();
        if (tid == 0) {
            int spins = 0, d = 0;
            while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > spin_limit || __hip_atomic_load(F, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(F, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    d = 1;
                    break;
                }
            }
            if (d) S.dead = 1;
        }
        __syncthreads();
        return S.dead == 0;
    };
    int cf_block = -1;                                   // block whose chain coefficients are in S.Cf / S.Rd
    int li_row = -1, lj_row = -1, l_block = -1;          // tile rows whose panels (block l_block) are in S.Li / S.Lj
    // panel rows [r0, r1) of tile row R for block kb from A into an operand buffer (negated for the row operand)
    auto load_panel = [&](int R, int kb, bool neg, double (*P)[DF_LP], int r0, int r1) {
        const int k0 = kb * CH_NB;
        double v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + DF_THREADS * q, r = e >> 5, m = e & 31;
            v[q] = ld_sh(&A[(size_t)min(64 * R + r, nrows - 1) * lda + min(k0 + m, n - 1)]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + DF_THREADS * q, r = e >> 5, m = e & 31;
            if (r >= r0 && r < r1) P[r][m] = neg ? -v[q] : v[q];
        }
    };
    // acc(slot) -= Li Lj^T on the column sixteenths [cbeg, cend) of the strips [sbeg, 4)
    auto update_tile = [&](int sl, int cbeg, int sbeg) {
        if (ws < sbeg) return;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const int c = 2 * wp + cc;
            if (c < cbeg) continue;
            double4_t acc;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) acc[rg] = S.T[sl][c][rg][ws][lane];
#pragma unroll
            for (int kk = 0; kk < CH_NB / 4; ++kk) {
                const double a = S.Li[16 * ws + li][4 * kk + lk];
                const double b = S.Lj[16 * c + li][4 * kk + lk];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
            }
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) S.T[sl][c][rg][ws][lane] = acc[rg];
        }
    };
    for (int kb = 0; kb < nblk && !S.dead; ++kb) {
        const int J = kb >> 1, h = kb & 1;
        const int k0 = kb * CH_NB, nb = min(CH_NB, n - k0);
        const bool more = kb + 1 < nblk;                 // a block follows: the trailing part takes this one's update
        double* Dg = Dgr + (size_t)kb * CH_NB * (CH_NB + 1);
        for (int sl = 0; sl < nt; ++sl) {
            const int I = S.tI[sl], Jt = S.tJ[sl];
            if (Jt < J) continue;                        // finished
            if (Jt == J) {
                // ---- a tile of the panel column
                const bool diag = I == J;
                if (diag) {
                    // the diagonal block (half h of the tile) out of the planes, factored, published
                    DF_TICK(7);
                    DF_STAMP(kb, 0);
                    // (no barrier in front: a wave copies from its own planes, and S.D was last read before the
                    // barriers of the step before)
                    if (wp == h && (ws >> 1) == h) {
#pragma unroll
                        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                            for (int rg = 0; rg < 4; ++rg) {
                                const int i = 16 * (ws & 1) + lk + 4 * rg, j = 16 * cc + li;
                                const double v = S.T[sl][2 * h + cc][rg][ws][lane];
                                S.D[i][j] = (i < nb && j <= i) ? v : ((i == j) ? 1.0 : 0.0);
                            }
                    }
                    __syncthreads();
                    if (tid < 64) chol_diag_wave_panel_inl<2>(S.D, nb, &fail[reg]);
                    __syncthreads();
                    DF_TICK(3);
                    DF_STAMP(kb, 1);
                    for (int e = tid; e < CH_NB * CH_NB; e += DF_THREADS) {
                        const int m = e >> 5, i = e & 31;                      // column m of row i
                        const double cv = (i > m) ? -(S.D[i][m] * S.D[m][CH_NB]) : 0.0;
                        reinterpret_cast<double*>(&S.Cf[m][i & 15])[i >> 4] = cv;
                        st_sh(&Dg[2 * (m * 16 + (i & 15)) + (i >> 4)], cv);
                        if (i < nb && m <= i) st_sh(&A[(size_t)(k0 + i) * lda + k0 + m], S.D[i][m]);
                    }
                    if (tid < CH_NB) {
                        S.Rd[tid] = S.D[tid][CH_NB];
                        st_sh(&Dg[CH_NB * CH_NB + tid], S.D[tid][CH_NB]);
                    }
                    cf_block = kb;
                    set_flag(&Fdiag[kb]);
                    DF_TICK(4);
                    DF_STAMP(kb, 2);
                } else if (cf_block != kb) {
                    DF_TICK(7);
                    if (!wait_flag(&Fdiag[kb])) break;
                    DF_TICK(1);
                    constexpr int ND = (CH_NB * (CH_NB + 1) + DF_THREADS - 1) / DF_THREADS;
                    double dv[ND];
#pragma unroll
                    for (int q = 0; q < ND; ++q) dv[q] = ld_sh(&Dg[min(tid + DF_THREADS * q, CH_NB * (CH_NB + 1) - 1)]);
#pragma unroll
                    for (int q = 0; q < ND; ++q) {
                        const int e = tid + DF_THREADS * q;
                        if (e < CH_NB * CH_NB) reinterpret_cast<double*>(&S.Cf[0][0])[e] = dv[q];
                        else if (e < CH_NB * (CH_NB + 1)) S.Rd[e - CH_NB * CH_NB] = dv[q];
                    }
                    cf_block = kb;
                    if (PROF == 1) { __syncthreads(); DF_TICK(2); }
                }
                __syncthreads();
                if (I == J + 1) DF_STAMP(kb, 3);
                // the rows of this tile below the diagonal block: X L^T = B, a half strip per wave
                // (rows 16 ws + lk + 4 (2 wp + h2), h2 = 0, 1), in place, stored to A and kept as operands
                const int rfirst = diag ? CH_NB * h + nb : 0;             // first tile row that is a panel row
                const bool solve = 16 * ws + 15 >= rfirst && 64 * I + 16 * ws < nrows;
                if (solve) {
                    double xb[2][2];
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2) {
                            const double v = S.T[sl][2 * h + c][2 * wp + h2][ws][lane];
                            xb[c][h2] = (16 * c + li < nb) ? v : 0.0;
                        }
                    double x0[2], x1[2];
                    cf_chain(xb, S.Cf, S.Rd, li, x0, x1);
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        const int r = 16 * ws + lk + 4 * (2 * wp + h2), gr = 64 * I + r;
                        const bool row_ok = r >= rfirst && gr < nrows;
                        if (row_ok) {
                            if (li < nb) st_sh(&A[(size_t)gr * lda + k0 + li], x0[h2]);
                            if (16 + li < nb) st_sh(&A[(size_t)gr * lda + k0 + 16 + li], x1[h2]);
                        }
                        const double y0 = row_ok ? x0[h2] : 0.0, y1 = row_ok ? x1[h2] : 0.0;
                        S.Li[r][li] = -y0;
                        S.Li[r][16 + li] = -y1;
                        S.Lj[r][li] = y0;
                        S.Lj[r][16 + li] = y1;
                    }
                } else {
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        const int r = 16 * ws + lk + 4 * (2 * wp + h2);
                        S.Li[r][li] = 0.0; S.Li[r][16 + li] = 0.0;
                        S.Lj[r][li] = 0.0; S.Lj[r][16 + li] = 0.0;
                    }
                }
                li_row = I; lj_row = I; l_block = kb;
                __syncthreads();                                         // (the operand writes)
                if (npend == DF_MAXT) flush();
                {
                    unsigned* pf = &Fpan[(size_t)kb * NT + I];           // the stores to A land while the updates run
                    if (npend == 0) pend0 = pf; else if (npend == 1) pend1 = pf; else pend2 = pf;
                    ++npend;
                }
                DF_TICK(5);
                if (I == J + 1) DF_STAMP(kb, 4);
                // the second half of the tile takes this block's update: columns 32 .. 63 -= X L21^T, L21 = the
                // rows 32 .. 63 of tile row J (the diagonal owner's solve of this step)
                if (h == 0 && more) {
                    if (!diag) {
                        if (!wait_flag(&Fpan[(size_t)kb * NT + J])) break;
                        DF_TICK(1);
                        load_panel(J, kb, false, S.Lj, 32, 64);
                        lj_row = -1;                                     // (half a panel: not a cached operand)
                        __syncthreads();
                        DF_TICK(6);
                    }
                    update_tile(sl, 2, diag ? 2 : 0);
                    __syncthreads();
                    DF_TICK(7);
                    if (I == J + 1) DF_STAMP(kb, 5);
                }
                continue;
            }
            // ---- a trailing tile: acc -= L[rows of tile row I][block] L[rows of tile row Jt][block]^T
            if (!more) continue;
            const bool have_i = l_block == kb && li_row == I, have_j = l_block == kb && lj_row == Jt;
            DF_TICK(7);
            if (!have_i) { if (!wait_flag(&Fpan[(size_t)kb * NT + I])) break; }
            if (!have_j && Jt != I) { if (!wait_flag(&Fpan[(size_t)kb * NT + Jt])) break; }
            __syncthreads();                                             // the previous update's operands are consumed
            DF_TICK(1);
            if (!have_i) load_panel(I, kb, true, S.Li, 0, 64);
            if (!have_j) load_panel(Jt, kb, false, S.Lj, 0, 64);
            li_row = I; lj_row = Jt; l_block = kb;
            __syncthreads();
            DF_TICK(6);
            update_tile(sl, 0, 0);
            if (PROF == 2 && I == Jt && I == J + 1 && h == 1) { __syncthreads(); DF_STAMP(kb, 5); }
        }
        flush();
    }
    flush();
    if (S.dead && tid == 0) atomicAdd(&tmo[reg], 1);
    if (PROF == 1 && prof && tid == 0)
        for (int k = 0; k < 8; ++k) prof[(size_t)blockIdx.x * 8 + k] = pt[k];
#undef DF_TICK
#undef DF_STAMP
}

// One workgroup of 512 threads per region and nothing shared between workgroups: no region
// barrier, nothing that has to be resident together, 9 CUs instead of 234.  Built for the case
// where many subtractions are in flight on one GPU (zm_ctx_set_share >= 2) and as the form a
// fit is repeated on after a barrier time-out of k_chol_fused.
// Left-looking: block column kb is   C = A[k0:, k0:k0+32] - L[k0:, 0:k0] L[k0:k0+32, 0:k0]^T
// on the f64 matrix cores, 16-row strips dealt to the waves (three per wave and pass, the
// accumulators stay in registers), the block row L[k0:k0+32, 0:k0] staged through LDS in
// 32-column chunks (double-buffered, one barrier per chunk), the strips' own rows read from
// global memory in the operand layout (a lane's eight loads of a chunk fall into two lines
// that stay in the vector L1).  Then wave 0 factors the diagonal block (chol_diag_wave_panel,
// the code k_chol_fused runs) and every wave solves its strips X L^T = C in place, in the
// accumulator layout: step m broadcasts column m inside each row of 16 lanes (DPP
// row_newbcast) and every lane subtracts u_m c[col][m].
// The arithmetic is that of k_chol_fused, operation for operation: an entry receives the
// products of the earlier block columns through the same v_mfma_f64_16x16x4 (negated row
// operand, chunks of four columns in ascending order - k_chol_fused applies them right-looking,
// one block step at a time, which is the same sequence per entry), the panel solve is the
// same chain (u_m unscaled, c = -(L[i][m] / L[m][m]), 1 / L[i][i] at the end).  Same bits:
// tests/test_subtract_gpu.py compares the two forms.
#define CT_THREADS 512
#define CT_WAVES (CT_THREADS / 64)
#define CT_NQ 3                          // strips per wave and pass
#define CT_NS (CH_NB * CH_NB / CT_THREADS)   // chunk entries staged per thread

// (one out-of-line copy for the four instances of ct_pass; k_chol_fused keeps its inlined one)
__device__ __noinline__ void chol_diag_wave_panel_call(double (*D)[CH_NB + 1], int nb, int* fail) {
    chol_diag_wave_panel_t<1>(D, nb, fail);
}

struct ct_lds {
    double D[CH_NB][CH_NB + 1];
    double Lb[2][CH_NB][CH_NB + 2];           // block-row chunk; pitch 34: conflict-free operand reads
    double2 Cf[CH_NB][16];                    // Cf[m][li] = {c[li][m], c[li + 16][m]}
    double Rd[CH_NB];                         // 1 / L[i][i]
};

// One pass of block column kb for a wave that owns NQ strips in it (NQ is wave-uniform: the
// instances differ in the work between the barriers, not in the barriers).
template <int NQ>
__device__ __forceinline__ void ct_pass(double* __restrict__ A, double* __restrict__ AT, const int n, const int lda,
                                        const int ldt, const int kb, const int pass, ct_lds* __restrict__ S, int* fail,
                                        long long* pt) {
    // phase clocks (100 MHz) when pt != NULL: 0 C = A, 1 products, 2 diagonal block, 3 its factor, 4 panel solve + store
    long long tc = pt ? wall_clock64() : 0;
#define CT_TICK(k) do { if (pt) { long long t_ = wall_clock64(); pt[k] += t_ - tc; tc = t_; } } while (0)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    const int sr = tid >> 5, sm = tid & 31;               // staging role: rows sr + 16 e, column sm of a 32 x 32 chunk
    const int nrows = n + 1;
    const int k0 = kb * CH_NB;
    const int nb = min(CH_NB, n - k0);
    const int k1 = k0 + nb;
    // (offsets into the region's matrix fit 32 bits: one base, unsigned offsets)
    bool bok[CT_NS];
    unsigned brow[CT_NS];
#pragma unroll
    for (int e = 0; e < CT_NS; ++e) {
        const int r = sr + (CH_NB / CT_NS) * e;
        bok[e] = k0 + r < n;
        brow[e] = (unsigned)(min(k0 + r, n - 1) * lda + sm);
    }
    constexpr int NA = NQ > 0 ? NQ : 1;
    int row0[NA];
#pragma unroll
    for (int q = 0; q < NQ; ++q) row0[q] = k0 + 16 * (CT_NQ * CT_WAVES * pass + wave + CT_WAVES * q);
    double4_t acc[NA][2];
    // C = A (entries beyond the matrix: 0); clamped addresses, selected afterwards
    {
        double t[NA][2][4];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int i = row0[q] + lk + 4 * rg, j = k0 + 16 * c + li;
                    t[q][c][rg] = A[(unsigned)(min(i, nrows - 1) * lda + min(j, n - 1))];
                }
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int i = row0[q] + lk + 4 * rg, j = k0 + 16 * c + li;
                    acc[q][c][rg] = (i < nrows && j < n) ? t[q][c][rg] : 0.0;
                }
    }
    CT_TICK(0);
    if (kb > 0) {
        {
            double b0[CT_NS];
#pragma unroll
            for (int e = 0; e < CT_NS; ++e) b0[e] = A[brow[e]];
#pragma unroll
            for (int e = 0; e < CT_NS; ++e) S->Lb[0][sr + (CH_NB / CT_NS) * e][sm] = bok[e] ? b0[e] : 0.0;
        }
        // the strips' own rows come from the column-major copy AT (16 consecutive lanes = 16 consecutive rows
        // = one 128-B line; from the row-major matrix every lane of a load would touch a line of its own)
        unsigned ap[NA];
#pragma unroll
        for (int q = 0; q < NQ; ++q) ap[q] = (unsigned)(lk * ldt + row0[q] + li);
        double av[NA][8];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) av[q][kk] = AT[ap[q] + 4 * kk * ldt];
        __syncthreads();
        for (int kc = 0; kc < kb; ++kc) {
            // the next chunk is requested before the matrix cores run on this one (the last
            // iteration requests the last chunk again: no branch around the loads)
            const int kn = min(kc + 1, kb - 1) * CH_NB;
            double bnext[CT_NS], avn[NA][8];
#pragma unroll
            for (int e = 0; e < CT_NS; ++e) bnext[e] = A[brow[e] + kn];
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) avn[q][kk] = AT[ap[q] + (kn + 4 * kk) * ldt];
            const double (*B)[CH_NB + 2] = S->Lb[kc & 1];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const double b0 = B[li][4 * kk + lk], b1 = B[16 + li][4 * kk + lk];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const double a = -av[q][kk];
                    acc[q][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, acc[q][0], 0, 0, 0);
                    acc[q][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, acc[q][1], 0, 0, 0);
                }
            }
#pragma unroll
            for (int e = 0; e < CT_NS; ++e)
                S->Lb[(kc + 1) & 1][sr + (CH_NB / CT_NS) * e][sm] = bok[e] ? bnext[e] : 0.0;
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) av[q][kk] = avn[q][kk];
            __syncthreads();
        }
    }
    CT_TICK(1);
    if (pass == 0) {
        // the diagonal block (strips 0 and 1: waves 0 and 1) -> D, identity outside nb
        for (int e = tid; e < CH_NB * (CH_NB + 1); e += CT_THREADS) {
            const int i = e / (CH_NB + 1), j = e - i * (CH_NB + 1);
            S->D[i][j] = (i == j) ? 1.0 : 0.0;
        }
        __syncthreads();
        if constexpr (NQ > 0) {
            if (wave < 2) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const int i = 16 * wave + lk + 4 * rg, j = 16 * c + li;
                        if (i < nb && j <= i) S->D[i][j] = acc[0][c][rg];
                    }
            }
        }
        __syncthreads();
        CT_TICK(2);
        if (wave == 0) chol_diag_wave_panel_call(S->D, nb, fail);
        __syncthreads();
        CT_TICK(3);
#pragma unroll
        for (int e = 0; e < CT_NS; ++e) {
            const int m = sr + (CH_NB / CT_NS) * e, i = sm;               // column m of row i
            const double cv = -(S->D[i][m] * S->D[m][CH_NB]);
            reinterpret_cast<double*>(&S->Cf[m][i & 15])[i >> 4] = (i > m) ? cv : 0.0;
            if (i < nb && m <= i) A[(unsigned)((k0 + i) * lda + k0 + m)] = S->D[i][m];
        }
        if (tid < CH_NB) S->Rd[tid] = S->D[tid][CH_NB];
        __syncthreads();
    }
    if constexpr (NQ > 0) {
        // panel rows X L^T = C in the accumulator layout (rows of the diagonal block ride along, unstored)
        hp_static_for<0, CH_NB - 1>([&](auto M) __attribute__((always_inline)) {
            constexpr int m = decltype(M)::value;
            constexpr int tm = m >> 4, sl = m & 15;
            const double2 cf = S->Cf[m][li];
            // (round 5: one v_fmac_f64 with a DPP row broadcast per update, as in cf_chain; the second column half
            // first - it reads the sources this step's first-half updates rewrite - and 4 NQ independent rows
            // between a write and the next broadcast of the same register: no s_nop except for one strip at m >= 16)
            if constexpr (m == 0) asm volatile("s_nop 1");
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
                    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                                 : "+v"(acc[q][1][rg]) : "v"(acc[q][tm][rg]), "v"(cf.y), "n"(sl));
            if constexpr (m < 15) {
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg)
                        asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf"
                                     : "+v"(acc[q][0][rg]) : "v"(cf.x), "n"(sl));
            }
        });
        const double r0 = S->Rd[li], r1 = S->Rd[16 + li];
        // (the chunk buffers are idle here: 16 x 17 doubles of them per wave turn a tile for the column-major copy)
        double (*T)[17] = reinterpret_cast<double (*)[17]>(&S->Lb[0][0][0] + wave * (16 * 17));
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            double x[2][4];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int i = row0[q] + lk + 4 * rg;
                x[0][rg] = acc[q][0][rg] * r0;
                x[1][rg] = acc[q][1][rg] * r1;
                if (i >= k1 && i < nrows) {
                    if (li < nb) A[(unsigned)(i * lda + k0 + li)] = x[0][rg];
                    if (16 + li < nb) A[(unsigned)(i * lda + k0 + 16 + li)] = x[1][rg];
                }
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) T[li][lk + 4 * rg] = x[c][rg];
                double tv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) tv[j] = T[lk + 4 * j][li];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = 16 * c + lk + 4 * j, i = row0[q] + li;
                    if (i >= k1 && i < nrows && col < nb) AT[(unsigned)((k0 + col) * ldt + i)] = tv[j];
                }
            }
        }
    }
    __syncthreads();          // the stored rows are the next column's operands; Lb / D / Cf are reused
    CT_TICK(4);
#undef CT_TICK
}

// AT: [reg][n][ldt] scratch, ldt = n + 1 rounded up to 16: the panels of L once more, column by column
static __device__ __forceinline__ void chol_tp_body(int n, int lda, int ldt, double* Aall, double* ATall, int* fail,
                                                        long long* prof, const int* __restrict__ guard) {
    if (guard && *guard == 0) return;                    // the previous round rejected nothing: this round is void
    __shared__ ct_lds S;
    const int reg = blockIdx.x;
    double* A = Aall + (size_t)reg * (size_t)(n + 1) * lda;
    double* AT = ATall + (size_t)reg * (size_t)n * ldt;
    const int wave = threadIdx.x >> 6;
    const int nblk = (n + CH_NB - 1) / CH_NB;
    // (phase clocks: developer build only - the array's address is taken, it lived in 48 bytes of scratch memory of
    // the shipped k_chol_tp whether or not anybody asked for the clocks)
    long long ptv[5] = {0, 0, 0, 0, 0};
    long long* pt = (ZM_DEV_BUILD && prof) ? ptv : nullptr;
    for (int kb = 0; kb < nblk; ++kb) {
        const int ns = (n + 1 - kb * CH_NB + 15) >> 4;       // 16-row strips from row k0 down
        const int npass = (ns + CT_NQ * CT_WAVES - 1) / (CT_NQ * CT_WAVES);
        for (int pass = 0; pass < npass; ++pass) {
            // strips of this wave: first + wave + CT_WAVES q, q < nq
            const int left = ns - CT_NQ * CT_WAVES * pass - wave;
            const int nq = left <= 0 ? 0 : min(CT_NQ, (left + CT_WAVES - 1) / CT_WAVES);
            switch (nq) {
                case 0: ct_pass<0>(A, AT, n, lda, ldt, kb, pass, &S, &fail[reg], pt); break;
                case 1: ct_pass<1>(A, AT, n, lda, ldt, kb, pass, &S, &fail[reg], pt); break;
                case 2: ct_pass<2>(A, AT, n, lda, ldt, kb, pass, &S, &fail[reg], pt); break;
                default: ct_pass<3>(A, AT, n, lda, ldt, kb, pass, &S, &fail[reg], pt); break;
            }
        }
    }
    if (ZM_DEV_BUILD && prof && threadIdx.x == 0)
        for (int k = 0; k < 5; ++k) prof[blockIdx.x * 5 + k] = ptv[k];
}

// Back substitution L^T x = y (y = row n of the factored storage), one workgroup
// of 1024 threads per region; then x /= d (Jacobi scaling) into xout.
__global__ __launch_bounds__(1024) void k_chol_back(int n, int lda, const double* __restrict__ Aall,
                                                    const double* __restrict__ dall,
                                                    double* __restrict__ xall, const int* __restrict__ guard) {
    if (guard && *guard == 0) return;                    // the previous round rejected nothing: this round is void
    extern __shared__ double cb_smem[];
    double* y = cb_smem;                                   // [n]
    double (*D)[CH_NB + 1] = reinterpret_cast<double (*)[CH_NB + 1]>(cb_smem + ((n + 1) & ~1));
    const double* A = Aall + (size_t)blockIdx.x * (size_t)(n + 1) * lda;
    const double* d = dall + (size_t)blockIdx.x * n;
    double* xo = xall + (size_t)blockIdx.x * n;
    const int tid = threadIdx.x;
    for (int i = tid; i < n; i += 1024) y[i] = A[(size_t)n * lda + i];
    const int nblk = (n + CH_NB - 1) / CH_NB;
    for (int kb = nblk - 1; kb >= 0; --kb) {
        const int k0 = kb * CH_NB, nb = min(CH_NB, n - k0);
        __syncthreads();
        {
            const int i = tid >> 5, j = tid & 31;          // 1024 threads = 32 x 32
            const double v = (i < nb && j <= i) ? A[(size_t)(k0 + i) * lda + k0 + j] : (i == j ? 1.0 : 0.0);
            D[i][j] = v;
            if (i == j) D[i][CH_NB] = 1.0 / v;             // 32 divisions side by side, none in the chain
        }
        __syncthreads();
        if (tid < 64) {
            // lane i keeps column i of the block (L[j][i], j = 0 .. 31) and the reciprocal
            // diagonal in registers, so a step of the chain is a readlane, a multiply and an
            // FMA - no LDS round trip, no ds_bpermute (rows >= nb are identity rows)
            const int li = tid & 31;
            double col[CH_NB];
#pragma unroll
            for (int j = 0; j < CH_NB; ++j) col[j] = D[j][li];
            const double rdl = D[li][CH_NB];                 // lane j holds 1 / L[j][j]
            double bi = (tid < nb) ? y[k0 + tid] : 0.0;
#pragma unroll
            for (int j = CH_NB - 1; j >= 0; --j) {
                const double xj = readlane_d(bi, j) * readlane_d(rdl, j);
                bi = (li == j) ? xj : ((li < j) ? bi - col[j] * xj : bi);   // L^T[i][j] = L[j][i]
            }
            if (tid < nb) y[k0 + tid] = bi;
        }
        __syncthreads();
        for (int c = tid; c < k0; c += 1024) {
            double acc = 0.0;
#pragma unroll 8
            for (int m = 0; m < nb; ++m) acc += A[(size_t)(k0 + m) * lda + c] * y[k0 + m];
            y[c] -= acc;
        }
    }
    __syncthreads();
    for (int i = tid; i < n; i += 1024) xo[i] = y[i] / d[i];
}

// Back substitution for n <= CBC_COLS unknowns, one workgroup per region: thread 64 + c owns
// column c and keeps y[c] in a register, wave 0 only runs the 32-step chain of each diagonal
// block - beside the column sweep of the block solved before it (look-ahead, round 2: 72 -> 56 us).  The 32 rows of L a column thread needs for block kb - 1 are requested as soon as those
// of block kb are consumed, and the diagonal block one step further ahead, so the chain of one
// block covers the memory latency of the next: a step costs the chain plus two barriers
// (k_chol_back above pays a dependent global load per block on top of it).
#define CBC_THREADS 1024
#define CBC_COLS (CBC_THREADS - 64)
static __device__ __forceinline__ void chol_back_cols_body(int n, int lda, const double* __restrict__ Aall,
                                                                const double* __restrict__ dall,
                                                                double* __restrict__ xall, const int* __restrict__ guard) {
    if (guard && *guard == 0) return;                    // the previous round rejected nothing: this round is void
    __shared__ double Dn[CH_NB][CH_NB + 1];      // diagonal block of the coming chain, 1 / diag in column 32
    __shared__ double xs[2][CH_NB];              // the block just solved / the one being solved beside the sweep
    __shared__ double ys[CH_NB];                 // right-hand side of the coming chain
    const double* A = Aall + (size_t)blockIdx.x * (size_t)(n + 1) * lda;
    const int tid = threadIdx.x;
    const int nblk = (n + CH_NB - 1) / CH_NB;
    // The two roles are two code paths with the same sequence of barriers (s_barrier counts waves, whatever their
    // program counter): in one path the chain's 32 coefficients and the column threads' 32 rows of L were live
    // together and the 128 registers of a wave (1024 threads) spilled.
    if (tid < 64) {
        // ---- wave 0: the 32-step chain of the block whose diagonal is in Dn and whose right-hand side is in ys.
        // Round 4: the chain carries the UNSCALED partial sums (the panel chains of the factorisation do the same):
        // u_i = y_i - sum_{k > i} L[k][i] x_k, x_j = u_j / L[j][j].  Lane i keeps c[j] = L[j][i] / L[j][j] for the
        // rows j > i of its column and subtracts c[j] u_j as soon as u_j is final - a step is the broadcast of u_j
        // (v_readlane) and one FMA, the reciprocal diagonal comes in once at the end (before: two broadcasts, a
        // multiply, an FMA and two selects per step).  Dn holds 0 on and above the diagonal and the reciprocal
        // diagonal in column 32: no select anywhere.
        auto run_chain = [&](double* xout) {
            const int li = tid & 31;
            double col[CH_NB];
#pragma unroll
            for (int j = 0; j < CH_NB; ++j) col[j] = Dn[j][li];
            const double rdl = Dn[li][CH_NB];
            double u = ys[li];
#pragma unroll
            for (int j = 1; j < CH_NB; ++j) col[j] *= readlane_d(rdl, j);
#pragma unroll
            for (int j = CH_NB - 1; j >= 1; --j) u = __builtin_fma(-col[j], readlane_d(u, j), u);
            if (tid < CH_NB) xout[tid] = u * rdl;
        };
        __syncthreads();
        run_chain(xs[0]);
        __syncthreads();
        for (int kb = nblk - 1; kb >= 1; --kb) {
            __syncthreads();                                 // (phase A of the column threads)
            run_chain(xs[(nblk - kb) & 1]);                  // block kb - 1, beside their phase B
            __syncthreads();
        }
        return;
    }
    const int c = tid - 64;
    double yc = 0.0, a[CH_NB], dnext[2] = {0.0, 0.0};
    // element e of a diagonal block (row e >> 5, column e & 31); rows >= nb are identity rows
    auto diag_elem = [&](int k0, int nb, int e) -> double {
        const int i = e >> 5, j = e & 31;
        return (i < nb && j <= i) ? A[(size_t)(k0 + i) * lda + k0 + j] : (i == j ? 1.0 : 0.0);
    };
    auto diag_put = [&](int e, double v) {
        const int i = e >> 5, j = e & 31;
        Dn[i][j] = (i == j) ? 0.0 : v;                   // (the chain wants the diagonal as its reciprocal only)
        if (i == j) Dn[i][CH_NB] = 1.0 / v;
    };
    {
        const int kb = nblk - 1, k0 = kb * CH_NB, nb = n - k0;
        yc = (c < n) ? A[(size_t)n * lda + c] : 0.0;
        diag_put(c, diag_elem(k0, nb, c));
        if (c + CBC_COLS < CH_NB * CH_NB) diag_put(c + CBC_COLS, diag_elem(k0, nb, c + CBC_COLS));
        if (c >= k0 && c < k0 + CH_NB) ys[c - k0] = yc;          // 0 beyond column n - 1
#pragma unroll
        for (int m = 0; m < CH_NB; ++m) a[m] = (c < k0 && m < nb) ? A[(size_t)(k0 + m) * lda + c] : 0.0;
        if (kb > 0) {
            dnext[0] = diag_elem(k0 - CH_NB, CH_NB, c);
            if (c + CBC_COLS < CH_NB * CH_NB) dnext[1] = diag_elem(k0 - CH_NB, CH_NB, c + CBC_COLS);
        }
    }
    __syncthreads();
    // Look-ahead: once block kb is solved, the 32 columns of block kb - 1 take its contribution first
    // (phase A), then wave 0 runs the chain of block kb - 1 WHILE the other columns take theirs and
    // request the rows of the next block (phase B) - the chain overlaps the column sweep and its
    // loads instead of standing between them.  Every y[c] still receives the same products in the
    // same order (one block of 32 at a time, m ascending).
    __syncthreads();                                         // (the chain of the last block)
    for (int kb = nblk - 1; kb >= 1; --kb) {
        const int k0 = kb * CH_NB, nb = min(CH_NB, n - k0);
        const int k0n = k0 - CH_NB;
        const double* xc = xs[(nblk - 1 - kb) & 1];          // solution of block kb
        // ---- phase A: this block's columns are final; the next block's columns finish their sums
        if (c >= k0 && c < k0 + nb) yc = xc[c - k0];
        if (c >= k0n && c < k0) {
            double acc = 0.0;
#pragma unroll
            for (int m = 0; m < CH_NB; ++m) acc += a[m] * xc[m];
            yc -= acc;
            ys[c - k0n] = yc;
        }
        diag_put(c, dnext[0]);
        if (c + CBC_COLS < CH_NB * CH_NB) diag_put(c + CBC_COLS, dnext[1]);
        __syncthreads();
        // ---- phase B: the sweep of the columns left of block kb - 1, beside its chain
        if (c < k0n) {
            double acc = 0.0;
#pragma unroll
            for (int m = 0; m < CH_NB; ++m) acc += a[m] * xc[m];
            yc -= acc;
        }
#pragma unroll
        for (int m = 0; m < CH_NB; ++m) a[m] = (c < k0n) ? A[(size_t)(k0n + m) * lda + c] : 0.0;
        if (kb > 1) {
            dnext[0] = diag_elem(k0n - CH_NB, CH_NB, c);
            if (c + CBC_COLS < CH_NB * CH_NB) dnext[1] = diag_elem(k0n - CH_NB, CH_NB, c + CBC_COLS);
        }
        __syncthreads();
    }
    if (c < min(CH_NB, n)) yc = xs[(nblk - 1) & 1][c];       // block 0
    if (c < n) xall[(size_t)blockIdx.x * n + c] = yc / dall[(size_t)blockIdx.x * n + c];
}

// ---------------------------------------------------------------------------
// merit[cell] = (I.I - 2 c.b + c^T Q c) / (npix vbar), c = per-cell coefficients
static __device__ __forceinline__ void hp_merit_body(const hp_plan& P, const double* __restrict__ G,
                                                 const double* __restrict__ phi,
                                                 const double* __restrict__ vbar,
                                                 const int* __restrict__ active,
                                                 const double* __restrict__ xsol,
                                                 double* __restrict__ merit, const int* __restrict__ guard,
                                                 int* __restrict__ needlist) {
    if (guard && *guard == 0) return;                    // the previous round rejected nothing: this round is void
    __shared__ double c[HP_MAXX];
    const int cell = blockIdx.x, lane = threadIdx.x;
    if (needlist && cell == 0 && lane == 0) needlist[0] = 0;        // (this round's rejection starts a new list)
    if (active[cell] < 0) { if (lane == 0) merit[cell] = -1.0; return; }
    const int reg = cell / P.ncellr;
    const double* x = xsol + (size_t)reg * P.nunk;
    const double* ph = phi + (size_t)cell * P.nkp;
    const double* Gc = G + (size_t)cell * HP_MAXX * HP_MAXX;
    if (lane < P.nE) {
        double v;
        if (lane == 0) v = x[0];
        else if (lane < P.nc) {
            v = 0.0;
            for (int p = 0; p < P.nkp; ++p) v += x[1 + (lane - 1) * P.nkp + p] * ph[p];
        } else v = x[1 + (P.nc - 1) * P.nkp + (lane - P.nc)];
        c[lane] = v;
    }
    __syncthreads();
    double part = 0.0;
    if (lane < P.nE) {
        double q = 0.0;
        for (int m = 0; m < P.nE; ++m) q += Gc[lane * HP_MAXX + m] * c[m];
        part = c[lane] * (q - 2.0 * Gc[lane * HP_MAXX + P.nE]);
    }
    part = wave_sum_d(part);
    if (lane == 0) merit[cell] = (Gc[P.nE * HP_MAXX + P.nE] + part) / (P.npix * vbar[cell]);
}

// one workgroup per region: clipped moments of the merits, reject, advance
// bit of a round's flag: a barrier of the many-workgroup factorisation timed out in this attempt (the flag is then
// not 0, later rounds are not void: their work is thrown away with the attempt)
#define HP_RFLAG_TMO (1 << 30)
// Round 6: the host hears of a round's flag from the rejection kernel itself - the last of its workgroups to finish
// writes {generation, flag | HP_RFLAG_DONE} into a word of coherent pinned memory the host spins on - instead of a
// 4-byte copy behind the kernel and an event (a blit kernel of 4 us and a barrier packet of 6 per round: 70 us of a
// subtraction).  `host` NULL (the batch's kernels): nothing.
#define HP_RFLAG_DONE 0x80000000u
struct hp_sig {
    int* done;                     // device counter of the round: workgroups that have finished
    unsigned long long* host;      // the round's word, as the device addresses it
    unsigned seq;                  // generation of the attempt (words are not reset between calls)
};
static __device__ __forceinline__ void hp_round_signal(const hp_sig& sg, int* round_flag, int nblocks) {
    if (!sg.host) return;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(sg.done, 1) == nblocks - 1) {
        const unsigned v = (unsigned)atomicAdd(round_flag, 0);
        __threadfence_system();
        __hip_atomic_store(sg.host, ((unsigned long long)sg.seq << 32) | v | HP_RFLAG_DONE, __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ __launch_bounds__(256) void k_hp_reject(const hp_plan P, const double* __restrict__ merit,
                                                   const int2* __restrict__ centres,
                                                   int* __restrict__ active, int* __restrict__ need,
                                                   int* __restrict__ chg, int* __restrict__ nrej,
                                                   double* __restrict__ stats, const int* __restrict__ guard, int* __restrict__ round_flag,
                                                   int* __restrict__ needlist, const int* __restrict__ tmo = nullptr,
                                                   const hp_sig sig = hp_sig{nullptr, nullptr, 0u}) {
    if (guard && *guard == 0) {                          // the previous round rejected nothing: this round is void
        hp_round_signal(sig, round_flag, gridDim.x);
        return;
    }
    __shared__ double red[4];
    const int reg = blockIdx.x, tid = threadIdx.x;
    double m = 0.0, s = 0.0;
    for (int pass = 0; pass < 4; ++pass) {
        double s0 = 0, s1 = 0;
        for (int k = tid; k < P.ncellr; k += 256) {
            double v = merit[reg * P.ncellr + k];
            if (active[reg * P.ncellr + k] < 0) continue;
            if (pass == 0 || fabs(v - m) <= 3.0 * s) { s0 += 1.0; s1 += v; }
        }
        s0 = block_sum256(s0, red);
        s1 = block_sum256(s1, red);
        if (s0 < 1.0) break;
        double mn = s1 / s0, s2 = 0;
        for (int k = tid; k < P.ncellr; k += 256) {
            double v = merit[reg * P.ncellr + k];
            if (active[reg * P.ncellr + k] < 0) continue;
            if (pass == 0 || fabs(v - m) <= 3.0 * s) s2 += (v - mn) * (v - mn);
        }
        s2 = block_sum256(s2, red);
        m = mn;
        s = sqrt(s2 / s0);
    }
    const double lim = m + P.ks * s;
    double cnt = 0, msum = 0, used = 0;
    for (int k = tid; k < P.ncellr; k += 256) {
        const int cell = reg * P.ncellr + k;
        int a = active[cell];
        need[cell] = 0;
        chg[cell] = 0;
        if (a < 0) continue;
        double v = merit[cell];
        used += 1.0;
        msum += v;
        if (v > lim) {
            cnt += 1.0;
            a += 1;
            if (a >= P.nss || centres[cell * P.nss + a].x < 0) a = -1;
            else {
                need[cell] = 1;
                if (needlist) needlist[1 + atomicAdd(&needlist[0], 1)] = cell;   // (the order does not matter: per-cell work)
            }
            active[cell] = a;
            chg[cell] = 1;              // its contribution leaves the normal matrix (and may come back)
        }
    }
    cnt = block_sum256(cnt, red);
    msum = block_sum256(msum, red);
    used = block_sum256(used, red);
    if (tid == 0) {
        nrej[reg] = (int)cnt;
        if (cnt > 0) atomicAdd(round_flag, (int)cnt);        // any rejection: the next round is live
        if (tmo && tmo[reg]) atomicOr(round_flag, HP_RFLAG_TMO);   // (the host hears of a barrier time-out with the round)
        stats[reg * 2 + 0] = used > 0 ? msum / used : 0.0;   // mean merit of the stamps fitted
        stats[reg * 2 + 1] = used;
    }
    // the changed cells of the region in cell order (a fixed order: the incremental normal
    // matrix sums their contributions in it): chg[cell] flags -> list behind the flags
    __syncthreads();
    if (tid == 0) {
        int* list = chg + P.ncell + reg * (P.ncellr + 1);     // [count, cells ...]
        int k = 0;
        for (int c = 0; c < P.ncellr; ++c)
            if (chg[reg * P.ncellr + c]) list[1 + k++] = reg * P.ncellr + c;
        list[0] = k;
    }
    hp_round_signal(sig, round_flag, gridDim.x);
}

// The same for regions of at most 256 cells, by one wave: lane l keeps cells l, l + 64, l + 128,
// l + 192 in registers and every sum is a wave reduction - no barrier (the 256-thread version
// spends its time in fifteen two-barrier block sums over a hundred values).
static __device__ __forceinline__ void hp_reject_wave_body(const hp_plan& P, const double* __restrict__ merit,
                                                       const int2* __restrict__ centres,
                                                       int* __restrict__ active, int* __restrict__ need,
                                                       int* __restrict__ chg, int* __restrict__ nrej,
                                                       double* __restrict__ stats, const int* __restrict__ guard,
                                                       int* __restrict__ round_flag, int* __restrict__ needlist,
                                                       const int* __restrict__ tmo = nullptr) {
    if (guard && *guard == 0) return;
    const int reg = blockIdx.x, lane = threadIdx.x;
    double mv[4];
    int av[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int k = lane + 64 * u;
        mv[u] = k < P.ncellr ? merit[reg * P.ncellr + k] : 0.0;
        av[u] = k < P.ncellr ? active[reg * P.ncellr + k] : -1;
    }
    double m = 0.0, s = 0.0;
    for (int pass = 0; pass < 4; ++pass) {
        double s0 = 0, s1 = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (av[u] >= 0 && (pass == 0 || fabs(mv[u] - m) <= 3.0 * s)) { s0 += 1.0; s1 += mv[u]; }
        s0 = wave_sum_d(s0);
        s1 = wave_sum_d(s1);
        if (s0 < 1.0) break;
        const double mn = s1 / s0;
        double s2 = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (av[u] >= 0 && (pass == 0 || fabs(mv[u] - m) <= 3.0 * s)) s2 += (mv[u] - mn) * (mv[u] - mn);
        s2 = wave_sum_d(s2);
        m = mn;
        s = sqrt(s2 / s0);
    }
    const double lim = m + P.ks * s;
    double cnt = 0, msum = 0, used = 0;
    unsigned long long chm[4];                                // changed-cell ballots, cell order
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int k = lane + 64 * u;
        bool changed = false;
        if (k < P.ncellr) {
            const int cell = reg * P.ncellr + k;
            int a = av[u];
            int nd = 0;
            if (a >= 0) {
                used += 1.0;
                msum += mv[u];
                if (mv[u] > lim) {
                    cnt += 1.0;
                    a += 1;
                    if (a >= P.nss || centres[cell * P.nss + a].x < 0) a = -1;
                    else nd = 1;
                    active[cell] = a;
                    changed = true;              // its contribution leaves the normal matrix (and may come back)
                }
            }
            need[cell] = nd;
            chg[cell] = changed ? 1 : 0;
            // the cells with a new substamp, all regions in one list: what the next round's vector / Gram
            // kernels run over (the order does not matter: per-cell work into per-cell slots)
            if (nd && needlist) needlist[1 + atomicAdd(&needlist[0], 1)] = cell;
        }
        chm[u] = __ballot(changed);
    }
    cnt = wave_sum_d(cnt);
    msum = wave_sum_d(msum);
    used = wave_sum_d(used);
    if (lane == 0) {
        nrej[reg] = (int)cnt;
        if (cnt > 0) atomicAdd(round_flag, (int)cnt);        // any rejection: the next round is live
        if (tmo && tmo[reg]) atomicOr(round_flag, HP_RFLAG_TMO);   // (the host hears of a barrier time-out with the round)
        stats[reg * 2 + 0] = used > 0 ? msum / used : 0.0;   // mean merit of the stamps fitted
        stats[reg * 2 + 1] = used;
    }
    // the changed cells in cell order behind the flags: [count, cells ...]
    int* list = chg + P.ncell + reg * (P.ncellr + 1);
    int base = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const unsigned long long bm = chm[u];
        if (bm >> lane & 1ull)
            list[1 + base + __popcll(bm & ((1ull << lane) - 1ull))] = reg * P.ncellr + lane + 64 * u;
        base += __popcll(bm);
    }
    if (lane == 0) list[0] = base;
}

__global__ void k_hp_init_active(const hp_plan P, const int2* __restrict__ centres,
                                 int* __restrict__ active, int* __restrict__ need,
                                 int* __restrict__ ntotal) {
    int cell = blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= P.ncell) return;
    bool has = centres[cell * P.nss].x >= 0;
    active[cell] = has ? 0 : -1;
    need[cell] = has ? 1 : 0;
    if (has) atomicAdd(&ntotal[cell / P.ncellr], 1);
}

__global__ __launch_bounds__(256) void k_hp_gram(const hp_plan P, const double* __restrict__ X,
                                                 const int* __restrict__ need, const int* __restrict__ active,
                                                 double* __restrict__ Gp, const int* __restrict__ guard,
                                                 const int* __restrict__ list) {
    hp_gram_body(P, X, need, active, Gp, guard, list);
}
__global__ __launch_bounds__(256) void k_hp_gram_b(const hp_plan P, const hp_job* __restrict__ jobs, int round) {
    const hp_job& J = jobs[blockIdx.z];
    hp_gram_body(P, J.X, J.need, J.active, J.Gp, HPJ_GUARD(J, round), round > 1 ? J.needlist : nullptr);
}

__global__ __launch_bounds__(GS_THREADS) void k_hp_gram_sum(const double* __restrict__ Gp, const int* __restrict__ need,
                                                     const int* __restrict__ active, double* __restrict__ G,
                                                     const int* __restrict__ guard, double* __restrict__ Gold,
                                                     const int* __restrict__ list) {
    hp_gram_sum_body(Gp, need, active, G, guard, Gold, list);
}
__global__ __launch_bounds__(GS_THREADS) void k_hp_gram_sum_b(const hp_job* __restrict__ jobs, int round) {
    const hp_job& J = jobs[blockIdx.z];
    hp_gram_sum_body(J.Gp, J.need, J.active, J.G, HPJ_GUARD(J, round), J.Gold, round > 1 ? J.needlist : nullptr);
}

#define HBB_PAIRS 5
__global__ __launch_bounds__(256) void k_hp_build_blk(const hp_plan P, const double* __restrict__ G,
                                                      const double* __restrict__ phi, const int* __restrict__ active,
                                                      const int* __restrict__ chg, int sign, double* __restrict__ A,
                                                      double* __restrict__ rhs, const int* __restrict__ guard,
                                                      unsigned* __restrict__ zero = nullptr, int nzero = 0,
                                                      const double* __restrict__ Gold = nullptr,
                                                      const double* __restrict__ phiold = nullptr,
                                                      const int* __restrict__ need = nullptr) {
    // (grid: x = HBB_PAIRS pairs of source vectors, z = region: see k_hp_build_blk_b.  A region whose last rejection
    // changed nothing has nothing to update - sign 2 - and its workgroups leave once the hand-over words of this
    // round's k_chol_df are cleared: 11 475 workgroups per round became 2 295, most of them gone at once)
    if (guard && *guard == 0) return;
    const int reg = blockIdx.z;
    if (sign == 2 && chg[P.ncell + reg * (P.ncellr + 1)] == 0) {
        if (zero && blockIdx.x == 0)
            for (int k = threadIdx.x; k < nzero; k += 256) zero[(size_t)reg * nzero + k] = 0u;
        return;
    }
    const int npair = P.nE * (P.nE + 1) / 2;
#pragma unroll 1
    for (int pair = blockIdx.x * HBB_PAIRS; pair < min((int)(blockIdx.x + 1) * HBB_PAIRS, npair); ++pair)
        hp_build_blk_body(reg, pair, P, G, phi, active, chg, sign, A, rhs, guard, zero, nzero, Gold, phiold, need);
}
// (grid: x = HBB_PAIRS pairs of source vectors, y = job, z = region.  One workgroup per pair, job and region is
// 183 600 workgroups for 16 jobs, most of them - every region whose last rejection changed nothing, every job that
// has converged - without work: the launch was bound by their dispatch, 0.3 - 0.45 ms per round.  A workgroup
// takes several pairs one after the other and leaves at once when its region has nothing to update.)
__global__ __launch_bounds__(256) void k_hp_build_blk_b(const hp_plan P, const hp_job* __restrict__ jobs, int round) {
    const hp_job& J = jobs[blockIdx.y];
    const int* guard = HPJ_GUARD(J, round);
    if (guard && *guard == 0) return;
    const int reg = blockIdx.z;
    if (HPJ_REGION_IDLE(J, round, reg, P.ncell, P.ncellr)) return;
    const int npair = P.nE * (P.nE + 1) / 2;
#pragma unroll 1
    for (int pair = blockIdx.x * HBB_PAIRS; pair < min((int)(blockIdx.x + 1) * HBB_PAIRS, npair); ++pair)
        hp_build_blk_body(reg, pair, P, J.G, J.phi, J.active, J.chg, round == 1 ? 0 : 2, J.A0, J.rhs0, guard,
                          nullptr, 0, J.Gold, J.phiold, J.need);
}

__global__ __launch_bounds__(256) void k_hp_build_mfma(const hp_plan P, const double* __restrict__ G,
                                                       const double* __restrict__ phi, const int* __restrict__ active,
                                                       double* __restrict__ A, double* __restrict__ rhs,
                                                       unsigned* __restrict__ zero, int nzero) {
    hp_build_mfma_body(P, G, phi, active, A, rhs, zero, nzero);
}
__global__ __launch_bounds__(256) void k_hp_build_mfma_b(const hp_plan P, const hp_job* __restrict__ jobs) {
    const hp_job& J = jobs[blockIdx.z];
    hp_build_mfma_body(P, J.G, J.phi, J.active, J.A0, J.rhs0, nullptr, 0);
}

__global__ void k_hp_diag(int n, const double* __restrict__ A, double* __restrict__ d, const int* __restrict__ guard) {
    hp_diag_body(n, A, d, guard);
}
__global__ void k_hp_diag_b(int n, const hp_job* __restrict__ jobs, int round) {
    const hp_job& J = jobs[blockIdx.z];
    hp_diag_body(n, J.A0, J.dsc, HPJ_GUARD(J, round));
}

__global__ void k_hp_scale(int n, int lda, const double* __restrict__ A0, const double* __restrict__ rhs0,
                           double* __restrict__ A, const double* __restrict__ d, unsigned* __restrict__ bar,
                           const int* __restrict__ guard, unsigned* __restrict__ dff, int ndff) {
    hp_scale_body(blockIdx.z, n, lda, A0, rhs0, A, d, bar, guard, dff, ndff);
}
// The batch's form: a workgroup takes HSB_ROWS rows of one region of one job (grid: x = row block, y = job * nreg +
// region) - k_hp_scale's grid of one workgroup per row and 256 columns is 312 000 workgroups for 16 jobs and is bound
// by their dispatch (222 us, whatever the number of jobs still fitting).  The arithmetic per entry is k_hp_scale's.
#define HSB_ROWS 8
__global__ __launch_bounds__(256) void k_hp_scale_b(int n, int lda, int nreg, const hp_job* __restrict__ jobs, int round,
                                                    int ncell, int ncellr) {
    const hp_job& J = jobs[blockIdx.y / nreg];
    const int* guard = HPJ_GUARD(J, round);
    if (guard && *guard == 0) return;
    const int reg = blockIdx.y % nreg;
    if (HPJ_REGION_IDLE(J, round, reg, ncell, ncellr)) return;
    const double* __restrict__ dd = J.dsc + (size_t)reg * n;
    const double* __restrict__ A0 = J.A0 + (size_t)reg * (size_t)(n + 1) * n;
    const double* __restrict__ rhs0 = J.rhs0 + (size_t)reg * n;
    double* __restrict__ A = J.A + (size_t)reg * (size_t)(n + 1) * lda;
#pragma unroll 1
    for (int k = 0; k < HSB_ROWS; ++k) {
        const int c1 = blockIdx.x * HSB_ROWS + k;
        if (c1 >= n) break;
        const double r1 = 1.0 / dd[c1];
        for (int c2 = threadIdx.x; c2 <= c1; c2 += 256) {
            const double r2 = 1.0 / dd[c2];
            double v = __dmul_rn(A0[(size_t)c1 * n + c2], __dmul_rn(r1, r2));
            if (c1 == c2) v = __dadd_rn(v, HP_RIDGE);
            A[(size_t)c1 * lda + c2] = v;
        }
        if (threadIdx.x == 0) A[(size_t)n * lda + c1] = __dmul_rn(rhs0[c1], r1);   // rhs row
    }
}

// AT: [reg][n][ldt] scratch, ldt = n + 1 rounded up to 16: the panels of L once more, column by column
__global__ __launch_bounds__(CT_THREADS) void k_chol_tp(int n, int lda, int ldt, double* Aall, double* ATall, int* fail,
                                                        long long* prof, const int* __restrict__ guard) {
    chol_tp_body(n, lda, ldt, Aall, ATall, fail, prof, guard);
}
__global__ __launch_bounds__(CT_THREADS) void k_chol_tp_b(int n, int lda, int ldt, const hp_job* __restrict__ jobs, int round,
                                                          int ncell, int ncellr) {
    const hp_job& J = jobs[blockIdx.z];
    if (HPJ_REGION_IDLE(J, round, (int)blockIdx.x, ncell, ncellr)) return;
    chol_tp_body(n, lda, ldt, J.A, J.AT, HPJ_FAIL(J), nullptr, HPJ_GUARD(J, round));
}

__global__ __launch_bounds__(CBC_THREADS) void k_chol_back_cols(int n, int lda, const double* __restrict__ Aall,
                                                                const double* __restrict__ dall,
                                                                double* __restrict__ xall, const int* __restrict__ guard) {
    chol_back_cols_body(n, lda, Aall, dall, xall, guard);
}
__global__ __launch_bounds__(CBC_THREADS) void k_chol_back_cols_b(int n, int lda, const hp_job* __restrict__ jobs, int round,
                                                                  int ncell, int ncellr) {
    const hp_job& J = jobs[blockIdx.z];
    if (HPJ_REGION_IDLE(J, round, (int)blockIdx.x, ncell, ncellr)) return;
    chol_back_cols_body(n, lda, J.A, J.dsc, J.rhs, HPJ_GUARD(J, round));
}

__global__ __launch_bounds__(64) void k_hp_merit(const hp_plan P, const double* __restrict__ G,
                                                 const double* __restrict__ phi, const double* __restrict__ vbar,
                                                 const int* __restrict__ active, const double* __restrict__ xsol,
                                                 double* __restrict__ merit, const int* __restrict__ guard,
                                                 int* __restrict__ needlist) {
    hp_merit_body(P, G, phi, vbar, active, xsol, merit, guard, needlist);
}
__global__ __launch_bounds__(64) void k_hp_merit_b(const hp_plan P, const hp_job* __restrict__ jobs, int round) {
    const hp_job& J = jobs[blockIdx.z];
    hp_merit_body(P, J.G, J.phi, J.vbar, J.active, J.rhs, J.merit, HPJ_GUARD(J, round), J.needlist);
}

__global__ __launch_bounds__(64) void k_hp_reject_wave(const hp_plan P, const double* __restrict__ merit,
                                                       const int2* __restrict__ centres, int* __restrict__ active,
                                                       int* __restrict__ need, int* __restrict__ chg,
                                                       int* __restrict__ nrej, double* __restrict__ stats,
                                                       const int* __restrict__ guard, int* __restrict__ round_flag,
                                                       int* __restrict__ needlist, const int* __restrict__ tmo,
                                                       const hp_sig sig) {
    hp_reject_wave_body(P, merit, centres, active, need, chg, nrej, stats, guard, round_flag, needlist, tmo);
    hp_round_signal(sig, round_flag, gridDim.x);
}
__global__ __launch_bounds__(64) void k_hp_reject_wave_b(const hp_plan P, const hp_job* __restrict__ jobs, int round) {
    const hp_job& J = jobs[blockIdx.z];
    hp_reject_wave_body(P, J.merit, J.centres, J.active, J.need, J.chg, HPJ_NREJ(J), J.stats, HPJ_GUARD(J, round),
                        J.rflags + round, J.needlist);
}

// zm_hp_params.flag_mask_dev: mask |= bit where the difference image carries hotpants' fill value (k_mask_flag,
// elementwise.hip) - unless a wait of this attempt's factorisation gave up (tmo: the fit is going to be repeated
// and this attempt's fill pattern is not the product's)
__global__ void k_hp_flag(int32_t* __restrict__ mask, const float* __restrict__ img, float value, int32_t bit,
                          int64_t n, const int* __restrict__ tmo, int nreg) {
    int gaveup = 0;
    for (int r = 0; r < nreg; ++r) gaveup |= tmo[r];
    if (gaveup) return;
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    if (img[p] == value) mask[p] |= bit;
}

// ---------------------------------------------------------------------------
extern "C" void zm_hp_params_default(zm_hp_params* p) {
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->tu = 5e3; p->tl = 0.0; p->iu = 5e3; p->il = 0.0;
    p->r = 10.0; p->rss = 15.0;
    p->fin = sqrt(50000.0); p->fi = 1e-30;
    p->nsx = p->nsy = 10; p->nrx = p->nry = 1;
    p->ko = 4; p->bgo = 0; p->nss = 3; p->normalize = 0;
    p->ft = 20.0; p->ks = 2.0;
    p->ngauss = 3;
    p->deg[0] = 6; p->deg[1] = 4; p->deg[2] = 2;
    p->sigma[0] = 0.7; p->sigma[1] = 1.5; p->sigma[2] = 3.0;
    p->limits_dev = nullptr; p->limits_nsigma = 10.0;
}

static int make_plan(const zm_hp_params* hp, int nx, int ny, hp_plan* P, std::vector<double>* filt,
                     std::vector<double>* basis) {
    memset(P, 0, sizeof(*P));
    P->nx = nx; P->ny = ny;
    P->hwk = (int)hp->r; P->hwss = (int)hp->rss;
    ZM_CHECK(P->hwk >= 1 && P->hwk <= HP_MAX_HWK, "zm_subtract: kernel half width int(r) = %d outside [1, %d]", P->hwk, HP_MAX_HWK);
    ZM_CHECK(P->hwss >= 1 && P->hwss <= HP_MAX_HWSS, "zm_subtract: substamp half width int(rss) = %d outside [1, %d]", P->hwss, HP_MAX_HWSS);
    P->hw = P->hwk + P->hwss;
    P->step = 2 * P->hwk + 1; P->sw = 2 * P->hwss + 1; P->pw = 2 * P->hw + 1;
    P->npix = P->sw * P->sw;
    P->npixp = (P->npix + GR_KT - 1) / GR_KT * GR_KT;
    ZM_CHECK(hp->ngauss >= 1 && hp->ngauss <= 4, "zm_subtract: ngauss %d outside [1, 4]", hp->ngauss);
    ZM_CHECK(hp->ko >= 0 && hp->ko <= 6, "zm_subtract: -ko %d outside [0, 6]", hp->ko);
    ZM_CHECK(hp->bgo >= 0 && hp->bgo <= 3, "zm_subtract: -bgo %d outside [0, 3]", hp->bgo);
    ZM_CHECK(hp->nss >= 1 && hp->nss <= HP_MAXNSS, "zm_subtract: nss %d outside [1, %d]", hp->nss, HP_MAXNSS);
    ZM_CHECK(hp->nrx >= 1 && hp->nry >= 1 && hp->nrx * hp->nry <= HP_MAXREG, "zm_subtract: bad -nrx/-nry");
    ZM_CHECK(hp->nsx >= 1 && hp->nsy >= 1 && hp->nsx * hp->nsy <= 4096, "zm_subtract: bad -nsx/-nsy");
    P->ko = hp->ko; P->bgo = hp->bgo; P->nss = hp->nss; P->normalize = hp->normalize;
    P->nrx = hp->nrx; P->nry = hp->nry; P->nsx = hp->nsx; P->nsy = hp->nsy;
    P->nreg = hp->nrx * hp->nry; P->ncellr = hp->nsx * hp->nsy; P->ncell = P->nreg * P->ncellr;
    P->tu = hp->tu; P->tl = hp->tl; P->iu = hp->iu; P->il = hp->il; P->ft = hp->ft; P->ks = hp->ks;
    P->fi = (float)hp->fi; P->fin = (float)hp->fin;
    ZM_CHECK(nx / hp->nrx / hp->nsx >= 1 && ny / hp->nry / hp->nsy >= 1, "zm_subtract: stamps smaller than a pixel");
    for (int ry = 0; ry < hp->nry; ++ry)
        for (int rx = 0; rx < hp->nrx; ++rx) {
            int r = ry * hp->nrx + rx;
            P->rx0[r] = rx * (nx / hp->nrx);
            P->rx1[r] = rx == hp->nrx - 1 ? nx : (rx + 1) * (nx / hp->nrx);
            P->ry0[r] = ry * (ny / hp->nry);
            P->ry1[r] = ry == hp->nry - 1 ? ny : (ry + 1) * (ny / hp->nry);
        }
    // 1-D filters and term tables
    const int step = P->step, hwk = P->hwk;
    std::vector<int> fidx;   // filter index of (g, a)
    int nf1 = 0, nc = 0;
    std::vector<double> fsum;
    filt->clear();
    int base[4] = {0, 0, 0, 0};
    for (int g = 0; g < hp->ngauss; ++g) {
        ZM_CHECK(hp->deg[g] >= 0 && hp->deg[g] <= 8 && hp->sigma[g] > 0, "zm_subtract: bad basis");
        base[g] = nf1;
        P->gdeg[g] = hp->deg[g];
        P->gbase[g] = nf1;
        for (int a = 0; a <= hp->deg[g]; ++a) {
            double s = 0.0;
            for (int u = -hwk; u <= hwk; ++u) {
                double v = exp(-(double)u * u / (2.0 * hp->sigma[g] * hp->sigma[g])) * pow((double)u, (double)a);
                filt->push_back(v);
                s += v;
            }
            fsum.push_back(s);
            ++nf1;
        }
    }
    ZM_CHECK(nf1 <= HP_MAXF1, "zm_subtract: too many 1-D filters");
    P->nf1 = nf1;
    P->ngauss = hp->ngauss;
    for (int g = 0; g < hp->ngauss; ++g)
        for (int a = 0; a <= hp->deg[g]; ++a)
            for (int b = 0; b <= hp->deg[g] - a; ++b) {
                ZM_CHECK(nc < HP_MAXX, "zm_subtract: basis too large");
                if (a == 0 && b == 0) P->gterm0[g] = nc;
                P->tfx[nc] = base[g] + a;
                P->tfy[nc] = base[g] + b;
                bool ee = (a % 2 == 0) && (b % 2 == 0);
                P->tscale[nc] = ee ? 1.0 / (fsum[base[g] + a] * fsum[base[g] + b]) : 1.0;
                P->tsub0[nc] = (ee && nc > 0) ? 1 : 0;
                ++nc;
            }
    P->nc = nc;
    for (int f = 0; f < HP_MAXF1; ++f) { P->tf0[f] = 0; P->tfn[f] = 0; }
    for (int n = 0; n < nc; ++n) {                       // terms are ordered (g, a, b): equal tfx are consecutive
        const int f = P->tfx[n];
        if (P->tfn[f] == 0) P->tf0[f] = n;
        P->tfn[f] += 1;
    }
    int nkp = 0;
    for (int i = 0; i <= P->ko; ++i)
        for (int j = 0; j <= P->ko - i; ++j) { P->kpi[nkp] = i; P->kpj[nkp] = j; ++nkp; }
    P->nkp = nkp;
    int nbg = 0;
    for (int i = 0; i <= P->bgo; ++i)
        for (int j = 0; j <= P->bgo - i; ++j) { P->bpi[nbg] = i; P->bpj[nbg] = j; ++nbg; }
    P->nbg = nbg;
    P->nE = nc + nbg;
    P->nX = P->nE + 1;
    ZM_CHECK(P->nX <= HP_MAXX, "zm_subtract: %d basis + %d background terms exceed the %d-row Gram tile", nc, nbg, HP_MAXX);
    P->nunk = 1 + (nc - 1) * nkp + nbg;
    // 2-D basis (hotpants normalisation), K_n[v][u]
    basis->assign((size_t)nc * step * step, 0.0);
    for (int n = 0; n < nc; ++n) {
        const double* fx = filt->data() + (size_t)P->tfx[n] * step;
        const double* fy = filt->data() + (size_t)P->tfy[n] * step;
        for (int v = 0; v < step; ++v)
            for (int u = 0; u < step; ++u) {
                double k = fy[v] * fx[u] * P->tscale[n];
                if (P->tsub0[n]) k -= (*basis)[(size_t)v * step + u];
                (*basis)[((size_t)n * step + v) * step + u] = k;
            }
    }
    return 0;
}

// Which regions have a usable fit - at least one stamp fitted, no clamped pivot, no wait that gave up, a finite
// solution - as a bit mask in device memory: the convolution is enqueued right behind the last rejection round and
// reads it there, instead of the host reading the fit summary first (round 4: three small copies, a
// synchronisation and ~70 us of idle GPU per subtraction; the host evaluates the same rule on its one copy of
// the summary after the convolution, for zm_hp_info and for the repeat after a time-out).
static __device__ __forceinline__ void hp_solved_body(int nreg, int nunk, const double* __restrict__ stats, const int* __restrict__ fail,
                            const int* __restrict__ tmo, const double* __restrict__ x,
                            unsigned long long* __restrict__ mask, double* __restrict__ x0out = nullptr) {
    unsigned long long m = 0;
    const int reg = threadIdx.x;
    if (reg < nreg) {
        const double x0 = x[(size_t)reg * nunk];
        if (x0out) x0out[reg] = x0;              // (the kernel sum of the region: all the host wants of the solution)
        const bool ok = stats[2 * reg + 1] >= 1.0 && fail[reg] == 0 && tmo[reg] == 0 && isfinite(x0);
        m = ok ? 1ull << reg : 0ull;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m |= __shfl_xor(m, o);
    if (threadIdx.x == 0) *mask = m;
}
__global__ void k_hp_solved(int nreg, int nunk, const double* __restrict__ stats, const int* __restrict__ fail,
                            const int* __restrict__ tmo, const double* __restrict__ x,
                            unsigned long long* __restrict__ mask, double* __restrict__ x0out = nullptr) {
    hp_solved_body(nreg, nunk, stats, fail, tmo, x, mask, x0out);
}


// The validity mask of a subtraction and its two dilations (substamp footprint: `dirty`; kernel footprint: `outbad`)
static int hp_launch_masks(zm_ctx* ctx, const hp_plan& P, const zm_hp_params* hp, const float* sci, const float* ref,
                           const uint8_t* bpm, uint8_t* bad, uint8_t* tmp8, uint8_t* dirty, uint8_t* outbad) {
    const int nx = P.nx, ny = P.ny;
    const int64_t np = (int64_t)nx * ny;
    hipStream_t st = ctx->stream;
    const dim3 b256(256);
    zm_scope_timer t(ctx, "hp_masks");
    ZM_CHECK(nx <= HP_ROWMAX, "zm_subtract: frames wider than %d pixels are not supported", HP_ROWMAX);
    const size_t rsh = sizeof(int) * ((size_t)nx + 1);
    dim3 gc(zm_div_up(nx, 256), zm_div_up(ny, HP_COLSTRIP));
    if (!ctx->hp_rset && rsh > 65536) {
        ZM_HIP(hipFuncSetAttribute((const void*)k_hp_rowany, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)(sizeof(int) * (HP_ROWMAX + 1))));
        ZM_HIP(hipFuncSetAttribute((const void*)k_hp_valid_rows, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)(sizeof(int) * (HP_ROWMAX + 1))));
        ctx->hp_rset = true;
    }
    const bool col4 = (nx % 4 == 0) && (((uintptr_t)tmp8 | (uintptr_t)dirty | (uintptr_t)outbad) & 3) == 0;
    dim3 gc4(zm_div_up(nx / 4, 256), zm_div_up(ny, HP_COLSTRIP4));
    // validity + both row dilations in one pass (k_hp_valid_rows); ZM_HP_MASKS=split: the five launches of rounds 1 - 3
    static const bool split = ZM_DEVENV("ZM_HP_MASKS") && !strcmp(ZM_DEVENV("ZM_HP_MASKS"), "split");
    uint8_t* tmp8b = nullptr;
    if (!split) ZM_TRY(ctx->get("hp_tmp8b", np, (void**)&tmp8b));
    if (!split && (((uintptr_t)tmp8b & 3) == 0)) {
        hipLaunchKernelGGL(k_hp_valid_rows, dim3(ny), b256, rsh, st, sci, ref, bpm, nx, ny, (float)P.il, (float)P.iu,
                           (float)P.tl, (float)P.tu, hp->limits_dev, hp->limits_nsigma, P.hw, P.hwk, bad, tmp8, tmp8b);
        if (col4) {
            hipLaunchKernelGGL(k_hp_colany4, gc4, b256, 0, st, tmp8, nx, ny, P.hw, 1, dirty);
            hipLaunchKernelGGL(k_hp_colany4, gc4, b256, 0, st, tmp8b, nx, ny, P.hwk, 1, outbad);
        } else {
            hipLaunchKernelGGL(k_hp_colany, gc, b256, 0, st, tmp8, nx, ny, P.hw, 1, dirty);
            hipLaunchKernelGGL(k_hp_colany, gc, b256, 0, st, tmp8b, nx, ny, P.hwk, 1, outbad);
        }
    } else {
        hipLaunchKernelGGL(k_hp_valid, dim3((unsigned)((np + 255) / 256)), b256, 0, st, sci, ref, bpm, np,
                           (float)P.il, (float)P.iu, (float)P.tl, (float)P.tu, bad, hp->limits_dev, hp->limits_nsigma);
        hipLaunchKernelGGL(k_hp_rowany, dim3(ny), b256, rsh, st, bad, nx, ny, P.hw, tmp8);
        if (col4) hipLaunchKernelGGL(k_hp_colany4, gc4, b256, 0, st, tmp8, nx, ny, P.hw, 1, dirty);
        else hipLaunchKernelGGL(k_hp_colany, gc, b256, 0, st, tmp8, nx, ny, P.hw, 1, dirty);
        hipLaunchKernelGGL(k_hp_rowany, dim3(ny), b256, rsh, st, bad, nx, ny, P.hwk, tmp8);
        if (col4) hipLaunchKernelGGL(k_hp_colany4, gc4, b256, 0, st, tmp8, nx, ny, P.hwk, 1, outbad);
        else hipLaunchKernelGGL(k_hp_colany, gc, b256, 0, st, tmp8, nx, ny, P.hwk, 1, outbad);
    }
    ZM_HIP(hipGetLastError());
    return 0;
}

// The stamp search of every cell and the first list of active cells
static int hp_launch_cells(zm_ctx* ctx, const hp_plan& P, const float* ref, const uint8_t* bad, const uint8_t* dirty,
                           int2* centres, int* active, int* need, int* ntotal) {
    hipStream_t st = ctx->stream;
    const dim3 b256(256);
    zm_scope_timer t(ctx, "hp_cells");
    int maxcell = 0;
    for (int r = 0; r < P.nreg; ++r)
        maxcell = std::max(maxcell, ((P.rx1[r] - P.rx0[r]) / P.nsx) * ((P.ry1[r] - P.ry0[r]) / P.nsy));
    // ZM_CELLS_FORM=global (tests): every cell through the kernel that keeps its pixels in global memory, whatever its
    // size - ADVICE r5: the two kernels sum a cell's moments in different orders, tests/test_subtract_gpu.py holds
    // their picks against each other on the same cells
    const char* cf = getenv("ZM_CELLS_FORM");
    if (maxcell <= HC_THREADS * HC_PX && !(cf && !strcmp(cf, "global")))
        hipLaunchKernelGGL(k_hp_cells_reg, dim3(P.ncell), dim3(HC_THREADS), 0, st, P, ref, bad, dirty, centres);
    else
        hipLaunchKernelGGL(k_hp_cells, dim3(P.ncell), b256, 0, st, P, ref, bad, dirty, centres);
    hipLaunchKernelGGL(k_hp_init_active, dim3(zm_div_up(P.ncell, 256)), b256, 0, st, P, centres, active, need, ntotal);
    ZM_HIP(hipGetLastError());
    return 0;
}

static std::atomic<int> g_hp_fitting[64];    // per device: contexts inside the kernel fit
// A context inside its kernel fit (zm_subtract_dev and, ADVICE r4, zm_subtract_batch_dev: a lone subtraction that
// starts beside a batch must see it and take the one-workgroup-per-region factorisation, whose launches need
// nothing resident - the batch's kernels hold CUs the many-workgroup form would wait for until its spins give up).
struct hp_fit_guard {
    std::atomic<int>* c;
    int others;
    explicit hp_fit_guard(std::atomic<int>* cc) : c(cc), others(cc->fetch_add(1)) {}
    bool shared() const { return c && (others > 0 || c->load(std::memory_order_relaxed) > 1); }
    void release() { if (c) c->fetch_sub(1); c = nullptr; }
    ~hp_fit_guard() { release(); }
};

// The fit summary as the host reads it (one pinned buffer: counters, stamp statistics, the regions' kernel sums).
#define HP_NIBUF_ (4 * HP_MAXREG + 4)
#define HP_SUMBYTES_ (sizeof(int) * HP_NIBUF_ + sizeof(double) * 3 * HP_MAXREG)
static void hp_fill_info(const char* h_sum, int nreg, int nunk, int rounds, int retries, zm_hp_info* info) {
    const int* h_int = reinterpret_cast<const int*>(h_sum);
    const double* h_stats = reinterpret_cast<const double*>(h_sum + sizeof(int) * HP_NIBUF_);
    const double* h_x0 = h_stats + 2 * HP_MAXREG;
    // a region is solved when it fitted at least one stamp and the factorisation held (k_hp_solved's rule)
    auto reg_solved = [&](int reg) {
        return h_stats[2 * reg + 1] >= 1.0 && h_int[2 * HP_MAXREG + reg] == 0 &&
               h_int[3 * HP_MAXREG + 4 + reg] == 0 && std::isfinite(h_x0[reg]);
    };
    memset(info, 0, sizeof(*info));
    double ks = 0, chi = 0;
    int nsolved = 0;
    for (int r = 0; r < nreg; ++r) {
        info->nstamps_total += h_int[HP_MAXREG + r];
        info->nstamps_used += (int)h_stats[2 * r + 1];
        if (reg_solved(r)) {
            ks += h_x0[r];
            chi += h_stats[2 * r];
            ++nsolved;
        }
    }
    info->niter = rounds;
    info->ncoeff = nunk;
    info->kernel_sum = nsolved ? ks / nsolved : 0.0;
    info->chi2 = nsolved ? chi / nsolved : 0.0;
    info->nmasked = h_int[3 * HP_MAXREG];
    // status bits: ZM_HP_UNSOLVED - a region without a usable fit (no stamps left / normal matrix
    // not positive definite: its pixels carry the fill value).  ZM_HP_TIMEOUT is reserved: barrier
    // time-outs can only happen in the first attempt (k_chol_fused); the repeat runs k_chol_tp, which
    // has no barrier between workgroups and therefore nothing that could time out - `retries` says
    // that a repeat happened, a hung device surfaces as a HIP error of the stream synchronisation.
    info->status = (nsolved == nreg ? 0 : ZM_HP_UNSOLVED);
    info->nunsolved = nreg - nsolved;
    info->retries = retries;
}

extern "C" int zm_subtract_dev(zm_ctx* ctx, const float* sci, const float* sci_rms, const float* ref,
                               const float* ref_rms, const uint8_t* bpm, int nx, int ny,
                               const zm_hp_params* hp, float* out_diff, float* out_rms,
                               zm_hp_info* info) {
    ZM_CHECK(ctx && sci && sci_rms && ref && ref_rms && hp && out_diff && out_rms,
             "zm_subtract_dev: null argument");
    ZM_CHECK(nx > 0 && ny > 0, "zm_subtract_dev: empty image");
    ZM_HIP(hipSetDevice(ctx->device));
    hp_plan P;
    std::vector<double> filt, basis;
    ZM_TRY(make_plan(hp, nx, ny, &P, &filt, &basis));
    ZM_CHECK(nx > 2 * P.hw + 1 && ny > 2 * P.hw + 1, "zm_subtract_dev: image smaller than a substamp");
    const int64_t np = (int64_t)nx * ny;
    hipStream_t st = ctx->stream;

    uint8_t *bad = nullptr, *tmp8 = nullptr, *dirty = nullptr, *outbad = nullptr;
    ZM_TRY(ctx->get("hp_bad", np, (void**)&bad));
    ZM_TRY(ctx->get("hp_tmp8", np, (void**)&tmp8));
    ZM_TRY(ctx->get("hp_dirty", np, (void**)&dirty));
    ZM_TRY(ctx->get("hp_outbad", np, (void**)&outbad));
    int2* centres = nullptr;
    int *active = nullptr, *need = nullptr, *ibuf = nullptr;
    double *X = nullptr, *G = nullptr, *phi = nullptr, *vbar = nullptr, *A = nullptr, *rhs = nullptr,
           *dsc = nullptr, *merit = nullptr, *stats = nullptr, *d_filt = nullptr;
    ZM_TRY(ctx->get("hp_centres", sizeof(int2) * P.ncell * P.nss, (void**)&centres));
    ZM_TRY(ctx->get("hp_active", sizeof(int) * P.ncell, (void**)&active));
    ZM_TRY(ctx->get("hp_need", sizeof(int) * P.ncell, (void**)&need));
    int* needlist = nullptr;             // [count, cells ...]: the cells the last rejection gave a new substamp
    ZM_TRY(ctx->get("hp_needlist", sizeof(int) * ((size_t)P.ncell + 1), (void**)&needlist));
    int* chg = nullptr;                  // cells whose substamp the last rejection changed
    ZM_TRY(ctx->get("hp_chg", sizeof(int) * (2 * (size_t)P.ncell + P.nreg), (void**)&chg));   // flags + per-region lists
    constexpr int HP_NIBUF = 4 * HP_MAXREG + 4;
    // the fit summary - counters, stamp statistics, the regions' kernel sums - is ONE buffer: one copy back when the
    // subtraction is done (round 4, late: three copies with the GPU idle cost a launch latency each)
    constexpr size_t HP_SUMBYTES = sizeof(int) * HP_NIBUF + sizeof(double) * 3 * HP_MAXREG;
    static_assert((sizeof(int) * HP_NIBUF) % 8 == 0, "the statistics behind the counters are doubles");
    char* sumbuf = nullptr;
    ZM_TRY(ctx->get("hp_summary", HP_SUMBYTES, (void**)&sumbuf));
    ibuf = reinterpret_cast<int*>(sumbuf);
    int *nrej = ibuf, *ntotal = ibuf + HP_MAXREG, *fail = ibuf + 2 * HP_MAXREG, *nmasked = ibuf + 3 * HP_MAXREG;
    int* tmo = ibuf + 3 * HP_MAXREG + 4;  // barrier time-outs of k_chol_fused per region
    unsigned* cbar = nullptr;            // region barrier counters of k_chol_fused (zeroed by k_hp_scale)
    ZM_TRY(ctx->get("hp_cbar", sizeof(unsigned) * CF_BAR_STRIDE * HP_MAXREG, (void**)&cbar));
    double* cdg = nullptr;               // published diagonal factors of k_chol_fused
    ZM_TRY(ctx->get("hp_cdg", sizeof(double) * 4 * CH_NB * (CH_NB + 1) * HP_MAXREG, (void**)&cdg));   // (k_chol_fused2: four slots)
    ZM_TRY(ctx->get("hp_X", sizeof(double) * (size_t)P.ncell * P.nX * P.npixp, (void**)&X));
    ZM_TRY(ctx->get("hp_G", sizeof(double) * (size_t)P.ncell * HP_MAXX * HP_MAXX, (void**)&G));
    double* Gp = nullptr;                // per-slice partial Gram matrices
    ZM_TRY(ctx->get("hp_Gp", sizeof(double) * (size_t)P.ncell * GR_SPLIT * HP_MAXX * HP_MAXX, (void**)&Gp));
    ZM_TRY(ctx->get("hp_phi", sizeof(double) * (size_t)P.ncell * P.nkp, (void**)&phi));
    // what a cell's Gram matrix and spatial terms were before its substamp was replaced (the fused update of the
    // normal matrix takes the old contribution out and puts the new one in in one launch)
    double *Gold = nullptr, *phiold = nullptr;
    ZM_TRY(ctx->get("hp_Gold", sizeof(double) * (size_t)P.ncell * HP_MAXX * HP_MAXX, (void**)&Gold));
    ZM_TRY(ctx->get("hp_phiold", sizeof(double) * (size_t)P.ncell * P.nkp, (void**)&phiold));
    ZM_TRY(ctx->get("hp_vbar", sizeof(double) * P.ncell, (void**)&vbar));
    const int lda = (P.nunk + 15) & ~15;       // factored storage: rows padded to whole 128-B lines
    ZM_TRY(ctx->get("hp_A", sizeof(double) * (size_t)P.nreg * (P.nunk + 1) * lda, (void**)&A));
    ZM_TRY(ctx->get("hp_rhs", sizeof(double) * (size_t)P.nreg * P.nunk, (void**)&rhs));
    double *A0 = nullptr, *rhs0 = nullptr;   // unscaled normal matrix / right-hand side, kept over the rounds
    ZM_TRY(ctx->get("hp_A0", sizeof(double) * (size_t)P.nreg * (P.nunk + 1) * P.nunk, (void**)&A0));
    ZM_TRY(ctx->get("hp_rhs0", sizeof(double) * (size_t)P.nreg * P.nunk, (void**)&rhs0));
    ZM_TRY(ctx->get("hp_dsc", sizeof(double) * (size_t)P.nreg * P.nunk, (void**)&dsc));
    ZM_TRY(ctx->get("hp_merit", sizeof(double) * P.ncell, (void**)&merit));
    stats = reinterpret_cast<double*>(sumbuf + sizeof(int) * HP_NIBUF);
    double* x0sum = stats + 2 * HP_MAXREG;
    ZM_TRY(ctx->get("hp_filt", sizeof(double) * filt.size(), (void**)&d_filt));
    // small constant tables: staged through pinned memory owned per call generation.  The table depends on the
    // parameters only (half width, Gaussians): a context that subtracts frame after frame with the same ones keeps
    // the device copy and neither waits for the stream nor copies (round 4: with the data limits taken on the
    // device - limits_dev - nothing between the background estimates and the fit waits for the host any more)
    if (!(ctx->hp_filt_dev == (const void*)d_filt && ctx->hp_filt_host == filt)) {
        double* h_tab = nullptr;
        ZM_TRY(ctx->get_pinned("hp_tab", sizeof(double) * filt.size(), (void**)&h_tab));
        ZM_HIP(hipStreamSynchronize(st));   // the previous call may still be reading the staging area
        memcpy(h_tab, filt.data(), sizeof(double) * filt.size());
        ZM_HIP(hipMemcpyAsync(d_filt, h_tab, sizeof(double) * filt.size(), hipMemcpyHostToDevice, st));
        ctx->hp_filt_host = filt;
        ctx->hp_filt_dev = d_filt;
    }

    const dim3 b256(256);
    ZM_TRY(hp_launch_masks(ctx, P, hp, sci, ref, bpm, bad, tmp8, dirty, outbad));
    // The fit (stamp search ... rejection rounds) is one repeatable attempt: when a barrier of the
    // fused factorisation timed out - its workgroups were not all resident, something else held
    // the GPU - every later round worked on a garbage solution, so the whole fit is run again on
    // the one-workgroup-per-region form of the factorisation (k_chol_tp: it waits for nobody; slower, same bits).
    // ZM_CHOL_SPIN_LIMIT (developer / tests): spins before a barrier gives up in the FIRST attempt.
    char* h_sum = nullptr;
    ZM_TRY(ctx->get_pinned("hp_summary_h", HP_SUMBYTES, (void**)&h_sum));
    const int* h_int = reinterpret_cast<const int*>(h_sum);
    const double* h_stats = reinterpret_cast<const double*>(h_sum + sizeof(int) * HP_NIBUF);
    const double* h_x0 = h_stats + 2 * HP_MAXREG;
    int rounds = 0, retries = 0, ntimeouts = 0;
    const char* spin_env = getenv("ZM_CHOL_SPIN_LIMIT");
    // Contexts of this process that are fitting on this device right now: the many-workgroup form of the
    // factorisation wants the GPU to itself, so a context that finds another one at work takes the
    // one-workgroup-per-region form without having been told (zm_ctx_set_share) - two engines used side by
    // side without a pool are safe by default.  Other processes on the card are not seen: there
    // the barrier time-out and the repeat below bound the damage.
    // (the count is read again whenever a factorisation is enqueued - a context that started alone
    // gives up the many-workgroup form as soon as a second one begins to fit - and it covers the fit only,
    // not the convolution behind it.)
    hp_fit_guard fitting(&g_hp_fitting[ctx->device & 63]);
    for (int attempt = 0; attempt < 2; ++attempt) {
    const bool safe = attempt > 0;
    const int spin_limit = (!safe && spin_env) ? atoi(spin_env) : CF_SPIN_LIMIT;
    ZM_HIP(hipMemsetAsync(ibuf, 0, sizeof(int) * HP_NIBUF, st));
    ZM_TRY(hp_launch_cells(ctx, P, ref, bad, dirty, centres, active, need, ntotal));
    // LDS of k_hp_vectors
    const hv_cfg hvc = hp_vectors_cfg(P.pw, P.sw, P.npix);
    ZM_CHECK(hvc.cw >= 8, "zm_subtract: r = %d, rss = %d: the template patch alone exceeds the LDS", P.hwk, P.hwss);
    size_t vsh = hvc.shmem;
    // (the chunked form: term 0 of every workgroup of the launch in global memory; the grid is capped, cells loop)
    const int hv_gx = hvc.big ? std::min(P.ncell, 128) : P.ncell;
    double* hv_w0g = nullptr;
    if (hvc.w0_global)
        ZM_TRY(ctx->get("hp_w0g", sizeof(double) * (size_t)hv_gx * HV_SPLIT_FEW * P.npix, (void**)&hv_w0g));

    // Rejection rounds without a host round trip on the critical path: round r + 1 is enqueued
    // before the host learns whether round r rejected anything.  k_hp_reject of round r adds its
    // rejections to rflags[r]; every kernel of round r + 1 starts with `if (rflags[r] == 0)
    // return`, so a round enqueued in vain costs a dozen empty launches while the GPU never
    // waits for the host in between.
    int* rflags = nullptr;
    int* h_rflags = nullptr;
    hipEvent_t* evs = nullptr;
    ZM_TRY(ctx->get("hp_rflags", sizeof(int) * 32, (void**)&rflags));        // [16] flags, [16] finished-workgroup counters
    ZM_TRY(ctx->get_pinned("hp_rflags_h", sizeof(int) * 16, (void**)&h_rflags));
    ZM_TRY(zm_get_sync_events(ctx, 3, &evs));
    ZM_HIP(hipMemsetAsync(rflags, 0, sizeof(int) * 32, st));
    // the words the rejection kernels signal through (coherent, mapped; one per round; never reset: a word counts
    // when it carries this attempt's generation)
    if (!ctx->hp_sig_h) {
        ZM_HIP(hipHostMalloc((void**)&ctx->hp_sig_h, sizeof(unsigned long long) * 16, hipHostMallocCoherent | hipHostMallocMapped));
        memset(ctx->hp_sig_h, 0, sizeof(unsigned long long) * 16);
        ZM_HIP(hipHostGetDevicePointer((void**)&ctx->hp_sig_d, ctx->hp_sig_h, 0));
    }
    static const bool sig_off = ZM_DEVENV("ZM_HP_SIGNAL") && ZM_DEVENV("ZM_HP_SIGNAL")[0] == '0';   // (developer build, 0: the copy + event of rounds 1 - 5)
    const unsigned seq = ++ctx->hp_seq;
    auto enqueue_round = [&](const int rounds) -> int {
        const int* guard = rounds > 1 ? rflags + (rounds - 1) : nullptr;
        // later rounds: the vector / Gram kernels run over the list of cells with a new substamp; this many
        // workgroup columns walk it (a handful of cells per round is the rule, more loop)
        const int ncl_grid = std::min(P.ncell, 48);
        if (rounds > 1) {
            // the rejected cells leave the normal matrix with their old Gram matrices / spatial
            // terms, before k_hp_vectors / k_hp_gram overwrite them
            zm_scope_timer t(ctx, "hp_solve");
            int nt = zm_div_up(P.nunk, 16);
            // (k_hp_build_blk takes both updates of the round in one launch, below: sign 2)
            if (P.nkp > 16)
                hipLaunchKernelGGL(k_hp_build, dim3(nt, nt, P.nreg), b256, 0, st, P, G, phi, active, chg, -1, A0, rhs0, guard);
        }
        {
            zm_scope_timer t(ctx, "hp_vectors");
            ZM_TRY(zm_hp_launch_vectors(ctx, st, P, hvc, rounds, ncl_grid, hv_gx, sci, ref, sci_rms, ref_rms, d_filt, centres, active,
                                        need, X, phi, vbar, guard, phiold, needlist, hv_w0g));
            ZM_HIP(hipGetLastError());
        }
        {
            zm_scope_timer t(ctx, "hp_gram");
            hipLaunchKernelGGL(k_hp_gram, dim3(rounds == 1 ? P.ncell : ncl_grid, GR_SPLIT), b256, 0, st, P, X, need, active, Gp, guard,
                               rounds == 1 ? nullptr : needlist);
            hipLaunchKernelGGL(k_hp_gram_sum, dim3(rounds == 1 ? P.ncell : ncl_grid), dim3(GS_THREADS), 0, st, Gp, need, active, G, guard, Gold,
                               rounds == 1 ? nullptr : needlist);
            ZM_HIP(hipGetLastError());
        }
        {
            zm_scope_timer t(ctx, "hp_solve");
            int nt = zm_div_up(P.nunk, 16);
            // Which form of the factorisation: the throughput form (one workgroup per region, nothing shared) when
            // `share` contexts subtract at the same time (zm_ctx_set_share), when another context of this process is
            // fitting, and for the repeat after a time-out; otherwise a latency form - the data-flow form (k_chol_df)
            // where its tiles fit three per workgroup, else the barrier form (k_chol_fused).  Same bits either way.
            // ZM_CHOL_FORM = tp / lat (k_chol_fused) / df overrides (tests, A / B timing).
            const char* form_env = getenv("ZM_CHOL_FORM");
            bool tp = safe || ctx->share >= 2 || fitting.shared();
            if (form_env && !strcmp(form_env, "tp")) tp = true;
            if (form_env && (!strcmp(form_env, "lat") || !strcmp(form_env, "df")) && !safe) tp = false;
            if (!ctx->hp_wg_cap) {
                // One workgroup per CU: a second one on the same CU slows the serial chains of the
                // look-ahead workgroup (measured: 806 us at 26 workgroups per region, 917 at 32).
                // The occupancy API only bounds it (it can be one block per CU high; keep a margin).
                int occ = 0, ncu = 0;
                ZM_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k_chol_fused, 256, 0));
                ZM_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device));
                ctx->hp_wg_cap = std::max(1, std::min(occ, 1) * (ncu - ncu / 16));
                if (ZM_DEVENV("ZM_CHOL_PROF")) fprintf(stderr, "chol: occupancy %d x %d CUs\n", occ, ncu);
            }
            // this context's share of the resident workgroups: each of `share` contexts keeps its launch fully resident
            const int wg_cap = ctx->hp_wg_cap / std::max(ctx->share, 1);
            const int W = std::max(2, std::min(68, wg_cap / P.nreg));
            const int nunk = P.nunk;
            static const bool want_prof = ZM_DEVENV("ZM_CHOL_PROF") && atoi(ZM_DEVENV("ZM_CHOL_PROF")) != 0;
            const bool want_df = !tp && !(form_env && !strcmp(form_env, "lat")) && 2 * P.nreg <= wg_cap && W <= 128;
            const int ndff = hp_df_nflags(P.nunk);
            unsigned* dff = nullptr;
            double* dfdg = nullptr;
            int* dftiles = nullptr;
            bool df = want_df;
            if (df) {
                // the tile table of (unknowns, W): built once, kept on the device
                constexpr int TW = 1 + 2 * DF_MAXT;
                int* htl = nullptr;
                ZM_TRY(ctx->get_pinned("hp_dftiles_h", sizeof(int) * (3 + 128 * TW), (void**)&htl));
                ZM_TRY(ctx->get("hp_dftiles", sizeof(int) * 128 * TW, (void**)&dftiles));
                constexpr int DF_MAGIC = 0x64663031;               // (the buffer is not zeroed: a stamp says its header is ours)
                const bool known = htl[2] == DF_MAGIC && htl[1] == W && (htl[0] == nunk || htl[0] == -nunk - 1);
                if (!known) {
                    ZM_HIP(hipStreamSynchronize(st));              // (an earlier copy out of this buffer may be in flight)
                    bool fits = true;
                    for (int wg = 0; wg < W; ++wg) {
                        int ti[DF_MAXT] = {0, 0, 0}, tj[DF_MAXT] = {0, 0, 0}, ntl = 0;
                        fits = hp_df_owner(nunk, W, wg, ti, tj, &ntl) && fits;
                        htl[3 + wg * TW] = ntl;
                        for (int k = 0; k < DF_MAXT; ++k) { htl[3 + wg * TW + 1 + 2 * k] = ti[k]; htl[3 + wg * TW + 2 + 2 * k] = tj[k]; }
                    }
                    htl[0] = fits ? nunk : -nunk - 1;              // (negative: this size does not fit, remembered too)
                    htl[1] = W;
                    htl[2] = DF_MAGIC;
                    if (fits) ZM_HIP(hipMemcpyAsync(dftiles, htl + 3, sizeof(int) * W * TW, hipMemcpyHostToDevice, st));
                }
                df = htl[0] == nunk;
            }
            if (df) {
                ZM_TRY(ctx->get("hp_dfflags", sizeof(unsigned) * (size_t)ndff * P.nreg, (void**)&dff));
                ZM_TRY(ctx->get("hp_dfdg", sizeof(double) * (size_t)P.nreg * zm_div_up(P.nunk, CH_NB) * CH_NB * (CH_NB + 1),
                                (void**)&dfdg));
            }
            static const bool build_scalar = ZM_DEVENV("ZM_BUILD_FORM") && !strcmp(ZM_DEVENV("ZM_BUILD_FORM"), "scalar");
            if (P.nkp <= 15 && rounds == 1 && !build_scalar)
                hipLaunchKernelGGL(k_hp_build_mfma, dim3(zm_div_up(P.nE * (P.nE + 1) / 2, 4), P.nreg), b256, 0, st, P, G, phi, active,
                                   A0, rhs0, dff, ndff);
            else if (P.nkp <= 16)
                hipLaunchKernelGGL(k_hp_build_blk, dim3(zm_div_up(P.nE * (P.nE + 1) / 2, HBB_PAIRS), 1, P.nreg), b256, 0, st, P, G, phi, active, chg,
                                   rounds == 1 ? 0 : 2, A0, rhs0, guard, dff, ndff, Gold, phiold, need);
            else
                hipLaunchKernelGGL(k_hp_build, dim3(nt, nt, P.nreg), b256, 0, st, P, G, phi, active, chg,
                                   rounds == 1 ? 0 : 1, A0, rhs0, guard, dff, ndff);
            if (!df) {
                // (the data-flow form scales on its way into LDS and writes the scale factors itself)
                hipLaunchKernelGGL(k_hp_diag, dim3(zm_div_up(P.nunk, 256), P.nreg), b256, 0, st, P.nunk, A0, dsc, guard);
                hipLaunchKernelGGL(k_hp_scale, dim3(zm_div_up(P.nunk, 256), P.nunk, P.nreg), b256, 0, st, P.nunk, lda,
                                   A0, rhs0, A, dsc, cbar, guard, nullptr, 0);
            }
            {
                if (tp) {
                    static const bool tp_prof = ZM_DEVENV("ZM_CHOL_PROF") && atoi(ZM_DEVENV("ZM_CHOL_PROF")) != 0;
                    long long* parg = nullptr;
                    if (tp_prof) ZM_TRY(ctx->get("hp_cprof", sizeof(long long) * 5 * P.nreg, (void**)&parg));
                    const int ldt = (P.nunk + 1 + 15) & ~15;
                    double* AT = nullptr;
                    ZM_TRY(ctx->get("hp_chol_at", sizeof(double) * (size_t)P.nreg * P.nunk * ldt, (void**)&AT));
                    {
                        zm_scope_timer tc(ctx, "hp_chol");         // (inside hp_solve: the factorisation alone)
                        hipLaunchKernelGGL(k_chol_tp, dim3(P.nreg), dim3(CT_THREADS), 0, st, P.nunk, lda, ldt, A, AT, fail, parg, guard);
                    }
                    ZM_HIP(hipGetLastError());
                    if (tp_prof) {
                        std::vector<long long> hp((size_t)5 * P.nreg);
                        ZM_HIP(hipMemcpyAsync(hp.data(), parg, sizeof(long long) * hp.size(), hipMemcpyDeviceToHost, st));
                        ZM_HIP(hipStreamSynchronize(st));
                        static const char* nm[5] = {"init", "products", "diag", "factor", "solve"};
                        fprintf(stderr, "chol_tp wg 0:");
                        for (int k = 0; k < 5; ++k) fprintf(stderr, " %s %.1f us", nm[k], hp[k] * 0.01);
                        fprintf(stderr, "\n");
                    }
                } else {
                // A latency form: W workgroups per region, all resident, one per CU.  A plain launch sized to be
                // fully resident (hipLaunchCooperativeKernel does not order against the following launches of the
                // stream on its first use).
                ZM_CHECK(2 * P.nreg <= wg_cap,
                         "zm_subtract: %d regions exceed the %d resident workgroups of this context's share (1 / %d)",
                         P.nreg, wg_cap, ctx->share);
                // ZM_CHOL_PROF=1: per-phase clocks of every workgroup, printed after the launch
                const int nprof = df ? 8 : 6;
                long long* parg = nullptr;
                const int nblkp = zm_div_up(nunk, CH_NB);
                if (want_prof) ZM_TRY(ctx->get("hp_cprof", sizeof(long long) * 8 * ((size_t)P.nreg * W + (size_t)P.nreg * nblkp), (void**)&parg));
                {
                    zm_scope_timer tc(ctx, "hp_chol");             // (inside hp_solve: the factorisation alone)
                    if (df) {
                        static const int prof_mode = want_prof ? std::min(2, atoi(ZM_DEVENV("ZM_CHOL_PROF"))) : 0;
#ifdef ZM_DEV
                        auto kf = prof_mode == 2 ? k_chol_df<2> : prof_mode == 1 ? k_chol_df<1> : k_chol_df<0>;
#else
                        auto kf = k_chol_df<0>;
#endif
                        static bool df_attr[3][64] = {};
                        if (!df_attr[prof_mode][ctx->device & 63]) {
                            ZM_HIP(hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                       (int)sizeof(df_lds)));
                            df_attr[prof_mode][ctx->device & 63] = true;
                        }
                        hipLaunchKernelGGL(kf, dim3(P.nreg * W), dim3(DF_THREADS), sizeof(df_lds), st, nunk, lda, W, A,
                                           dfdg, fail, tmo, spin_limit, dff, dftiles, guard, A0, rhs0, dsc, parg);
                    } else {
                        hipLaunchKernelGGL(k_chol_fused, dim3(P.nreg * W), b256, 0, st, nunk, lda, W, A, cdg, fail, tmo,
                                           spin_limit, cbar, parg, guard);
                    }
                }
                ZM_HIP(hipGetLastError());
                if (want_prof) {
                    std::vector<long long> hp((size_t)8 * ((size_t)P.nreg * W + (size_t)P.nreg * nblkp));
                    ZM_HIP(hipMemcpyAsync(hp.data(), parg, sizeof(long long) * hp.size(), hipMemcpyDeviceToHost, st));
                    ZM_HIP(hipStreamSynchronize(st));
                    static const char* nm6[6] = {"load", "tilewait", "panel", "barrier1", "update", "barrier2"};
                    static const char* nm8[8] = {"load", "flagwait", "coeff", "factor", "publish", "solve", "panels", "update"};
                    // (k_chol_fused2, workgroup 0: "barrier1" = first look-ahead block, "update" = the second)
                    const int NCd = (nunk + 63) / 64;
                    std::vector<int> show = {0, 1, W - 1, W, (P.nreg - 1) * W};
                    if (df) show = {0, 1, NCd / 2, NCd - 1, std::min(NCd, W - 1), W - 1, (P.nreg - 1) * W + 1};
                    for (int wg : show) {
                        fprintf(stderr, "chol wg %3d:", wg);
                        for (int k = 0; k < nprof; ++k)
                            fprintf(stderr, " %s %.1f us", df ? nm8[k] : nm6[k], hp[(size_t)wg * nprof + k] * 0.01);
                        fprintf(stderr, "\n");
                    }
                    if (df && ZM_DEVENV("ZM_CHOL_PROF") && atoi(ZM_DEVENV("ZM_CHOL_PROF")) >= 2) {
                        // region 0: the chain per block, microseconds since the first block was taken up
                        const long long* ts = hp.data() + (size_t)8 * P.nreg * W;
                        for (int kb = 0; kb < nblkp; ++kb) {
                            fprintf(stderr, "chol kb %2d:", kb);
                            for (int e = 0; e < 6; ++e) fprintf(stderr, " %7.2f", (ts[kb * 8 + e] - ts[0]) * 0.01);
                            fprintf(stderr, "\n");
                        }
                    }
                }
                }   // latency form
            }
            {
                const size_t bsh = sizeof(double) * (((size_t)P.nunk + 1) & ~(size_t)1) +
                                   sizeof(double) * CH_NB * (CH_NB + 1);
                if (!ctx->hp_bset && bsh > 65536) {
                    ZM_HIP(hipFuncSetAttribute((const void*)k_chol_back,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
                    ctx->hp_bset = true;
                }
                ZM_CHECK(bsh <= 160 * 1024 - 64, "zm_subtract: %d unknowns exceed the solver's LDS", P.nunk);
                if (P.nunk <= CBC_COLS)
                    hipLaunchKernelGGL(k_chol_back_cols, dim3(P.nreg), dim3(CBC_THREADS), 0, st, P.nunk, lda, A, dsc, rhs, guard);
                else
                    hipLaunchKernelGGL(k_chol_back, dim3(P.nreg), dim3(1024), bsh, st, P.nunk, lda, A, dsc, rhs, guard);
            }
            hipLaunchKernelGGL(k_hp_merit, dim3(P.ncell), dim3(64), 0, st, P, G, phi, vbar, active, rhs, merit, guard, needlist);
            const hp_sig sig = sig_off ? hp_sig{nullptr, nullptr, 0u} : hp_sig{rflags + 16 + rounds, ctx->hp_sig_d + rounds, seq};
            if (P.ncellr <= 256)
                hipLaunchKernelGGL(k_hp_reject_wave, dim3(P.nreg), dim3(64), 0, st, P, merit, centres, active, need,
                                   chg, nrej, stats, guard, rflags + rounds, needlist, (const int*)tmo, sig);
            else
                hipLaunchKernelGGL(k_hp_reject, dim3(P.nreg), b256, 0, st, P, merit, centres, active, need, chg,
                                   nrej, stats, guard, rflags + rounds, needlist, (const int*)tmo, sig);
            ZM_HIP(hipGetLastError());
        }
        if (sig_off) {
            ZM_HIP(hipMemcpyAsync(h_rflags + rounds, rflags + rounds, sizeof(int), hipMemcpyDeviceToHost, st));
            ZM_HIP(hipEventRecord(evs[1 + (rounds & 1)], st));
        }
        return 0;
    };
    // the flag of round r as the host learns it
    auto wait_round = [&](int r, int* flag) -> int {
        if (sig_off) {
            ZM_HIP(hipEventSynchronize(evs[1 + (r & 1)]));
            *flag = h_rflags[r];
            return 0;
        }
        for (unsigned long spins = 1;; ++spins) {
            const unsigned long long w = __atomic_load_n(&ctx->hp_sig_h[r], __ATOMIC_ACQUIRE);
            if ((unsigned)(w >> 32) == seq && ((unsigned)w & HP_RFLAG_DONE)) {
                *flag = (int)((unsigned)w & ~HP_RFLAG_DONE);
                return 0;
            }
            if ((spins & 0x3ffff) == 0) {                // now and then: is the stream still alive?
                const hipError_t e = hipStreamQuery(st);
                if (e == hipSuccess) {                   // everything enqueued has run: the word must be there
                    const unsigned long long w2 = __atomic_load_n(&ctx->hp_sig_h[r], __ATOMIC_ACQUIRE);
                    ZM_CHECK((unsigned)(w2 >> 32) == seq && ((unsigned)w2 & HP_RFLAG_DONE),
                             "zm_subtract: round %d finished without signalling the host", r);
                } else if (e != hipErrorNotReady) {
                    ZM_HIP(e);
                }
            }
            // (a round takes 0.2 - 0.4 ms: a short busy wait, then the core is offered to whoever wants it - a pool
            // of sixteen chains has sixteen threads here, beside the readers and writers of the file ring)
            if (spins < 4096) __builtin_ia32_pause();
            else sched_yield();
        }
    };
    rounds = 0;
    ZM_TRY(enqueue_round(1));
    bool tmo_seen = false;
    for (int r = 1; r <= 8; ++r) {
        if (r < 8) ZM_TRY(enqueue_round(r + 1));          // void if round r rejects nothing
        int flag = 0;
        ZM_TRY(wait_round(r, &flag));
        rounds = r;
        tmo_seen = tmo_seen || (flag & HP_RFLAG_TMO);
        if ((flag & ~HP_RFLAG_TMO) == 0 || tmo_seen) break;
    }
    // the convolution behind the last round, on the device's own view of which regions are solved
    {
        zm_scope_timer t(ctx, "hp_apply");
        unsigned long long* smask = nullptr;
        ZM_TRY(ctx->get("hp_smask", sizeof(unsigned long long), (void**)&smask));
        hipLaunchKernelGGL(k_hp_solved, dim3(1), dim3(64), 0, st, P.nreg, P.nunk, stats, fail, tmo, rhs, smask, x0sum);
        {
            ZM_TRY(zm_hp_launch_apply(ctx, P, smask, sci, ref, sci_rms, ref_rms, outbad, d_filt, rhs, out_diff, out_rms, nmasked));
        }
    }
    // what the reference does behind hotpants - bit 17 where the fill value landed - enqueued here when asked for
    // (zm_hp_params.flag_mask_dev), not by the caller after this call has waited for the GPU
    if (hp->flag_mask_dev) {
        hipLaunchKernelGGL(k_hp_flag, dim3((unsigned)((np + 255) / 256)), b256, 0, st, hp->flag_mask_dev, out_diff, 1e-30f,
                           hp->flag_bit, np, tmo, P.nreg);
        ZM_HIP(hipGetLastError());
    }
    // ... and ONE read of the fit summary (counters incl. the convolution's masked-pixel count, stamp statistics,
    // the regions' kernel sums), one copy, when everything is done
    ZM_HIP(hipMemcpyAsync(h_sum, sumbuf, HP_SUMBYTES, hipMemcpyDeviceToHost, st));
    // zm_hp_params.async_info: the rounds are over (the host has seen the last flag, and no time-out with it), what is
    // left - convolution, bit 17, the summary's copy - is enqueued: return, zm_subtract_info waits for the rest.  The
    // host then enqueues whatever comes next (the next coadd's statistics, the next job's alignment) while the
    // convolution runs, instead of starting on it when the summary has arrived.
    if (hp->async_info && !tmo_seen) {
        if (!ctx->hp_done) ZM_HIP(hipEventCreateWithFlags(&ctx->hp_done, hipEventDisableTiming));
        ZM_HIP(hipEventRecord(ctx->hp_done, st));
        ctx->hp_pending = true;
        ctx->hp_pend_rounds = rounds;
        ctx->hp_pend_retries = retries;
        ctx->hp_pend_nreg = P.nreg;
        ctx->hp_pend_nunk = P.nunk;
        fitting.release();
        if (info) {
            memset(info, 0, sizeof(*info));
            info->status = ZM_HP_PENDING;
            info->niter = rounds;
            info->ncoeff = P.nunk;
            info->retries = retries;
        }
        return 0;
    }
    ZM_HIP(hipStreamSynchronize(st));
    ntimeouts = 0;
    for (int reg = 0; reg < P.nreg; ++reg) ntimeouts += h_int[3 * HP_MAXREG + 4 + reg];
    if (ntimeouts == 0) break;
    if (!safe) {
        ++retries;
        if (getenv("ZM_VERBOSE"))
            fprintf(stderr, "zm_subtract: %d barrier time-out(s) in the fused factorisation; repeating the fit on the "
                            "one-workgroup form\n", ntimeouts);
    }
    }   // attempts
    fitting.release();
    ctx->hp_pending = false;
    if (info) hp_fill_info(h_sum, P.nreg, P.nunk, rounds, retries, info);
    return 0;
}

extern "C" int zm_subtract_info(zm_ctx* ctx, zm_hp_info* info) {
    ZM_CHECK(ctx && info, "zm_subtract_info: null argument");
    ZM_CHECK(ctx->hp_pending, "zm_subtract_info: no subtraction with async_info is pending on this context");
    ZM_HIP(hipSetDevice(ctx->device));
    ZM_HIP(hipEventSynchronize(ctx->hp_done));
    ctx->hp_pending = false;
    char* h_sum = nullptr;
    ZM_TRY(ctx->get_pinned("hp_summary_h", HP_SUMBYTES_, (void**)&h_sum));
    const int* h_int = reinterpret_cast<const int*>(h_sum);
    for (int reg = 0; reg < ctx->hp_pend_nreg; ++reg)
        ZM_CHECK(h_int[3 * HP_MAXREG + 4 + reg] == 0, "zm_subtract_info: a barrier time-out went unnoticed (region %d)", reg);
    hp_fill_info(h_sum, ctx->hp_pend_nreg, ctx->hp_pend_nunk, ctx->hp_pend_rounds, ctx->hp_pend_retries, info);
    return 0;
}

// ---------------------------------------------------------------------------
// Many subtractions in one call (include/zudsmi.h: zm_subtract_batch_dev).  A lone subtraction is bound by the
// latency of its fit (a dozen small launches per rejection round, a 23-step factorisation); a pool of J contexts on J
// streams gets at most six kernels in flight whatever J is.  Here the job is a grid dimension of every launch of the
// fit: nine regions x J jobs of k_chol_tp are 9 J workgroups on 9 J compute units for the time ONE factorisation
// takes.  The throughput kernels (masks, stamp search, convolution: each fills the GPU alone) are enqueued job after
// job around it.  Same kernels' bodies, same order of operations per job: the bits of zm_subtract_dev.
static bool hp_plans_agree(const hp_plan& a, const hp_plan& b) {
    hp_plan x = a, y = b;
    x.tu = x.tl = x.iu = x.il = 0; y.tu = y.tl = y.iu = y.il = 0;
    x.fi = x.fin = 0; y.fi = y.fin = 0;
    return memcmp(&x, &y, sizeof(hp_plan)) == 0;
}

extern "C" int zm_subtract_batch_dev(zm_ctx* ctx, int njobs, const zm_sub_job* jobs, int nx, int ny,
                                     zm_hp_info* infos) {
    ZM_CHECK(ctx && jobs && njobs >= 1 && njobs <= ZM_SUB_BATCH_MAX, "zm_subtract_batch_dev: 1 .. %d jobs", ZM_SUB_BATCH_MAX);
    ZM_CHECK(nx > 0 && ny > 0, "zm_subtract_batch_dev: empty image");
    ZM_HIP(hipSetDevice(ctx->device));
    std::vector<hp_plan> Pj(njobs);
    std::vector<double> filt, basis, filt_j, basis_j;
    for (int j = 0; j < njobs; ++j) {
        const zm_sub_job& jb = jobs[j];
        ZM_CHECK(jb.sci && jb.sci_rms && jb.ref && jb.ref_rms && jb.params && jb.out_diff && jb.out_rms,
                 "zm_subtract_batch_dev: null argument in job %d", j);
        ZM_TRY(make_plan(jb.params, nx, ny, &Pj[j], j == 0 ? &filt : &filt_j, j == 0 ? &basis : &basis_j));
        ZM_CHECK(j == 0 || (hp_plans_agree(Pj[0], Pj[j]) && filt_j == filt),
                 "zm_subtract_batch_dev: job %d asks for a different fit (half widths, basis, orders, regions, stamps or "
                 "thresholds) than job 0 - batch jobs of one configuration", j);
    }
    const hp_plan& P = Pj[0];
    ZM_CHECK(nx > 2 * P.hw + 1 && ny > 2 * P.hw + 1, "zm_subtract_batch_dev: image smaller than a substamp");
    if (P.nkp > 15 || P.nunk > CBC_COLS || P.ncellr > 256 || njobs == 1 || P.hwk > 15 || hp_vectors_cfg(P.pw, P.sw, P.npix).big) {
        // what the batched kernels do not cover (and the batch of one): job by job
        for (int j = 0; j < njobs; ++j)
            ZM_TRY(zm_subtract_dev(ctx, jobs[j].sci, jobs[j].sci_rms, jobs[j].ref, jobs[j].ref_rms, jobs[j].bpm, nx, ny,
                                   jobs[j].params, jobs[j].out_diff, jobs[j].out_rms, infos ? &infos[j] : nullptr));
        return 0;
    }
    const int64_t np = (int64_t)nx * ny;
    hipStream_t st = ctx->stream;
    const int lda = (P.nunk + 15) & ~15;
    const int ldt = (P.nunk + 1 + 15) & ~15;
    constexpr int HP_NIBUF = 4 * HP_MAXREG + 4;

    // one slab of scratch per job, every buffer at the same offset in every slab
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_bad = take(np), o_dirty = take(np), o_outbad = take(np);
    const size_t o_centres = take(sizeof(int2) * P.ncell * P.nss);
    const size_t o_active = take(sizeof(int) * P.ncell), o_need = take(sizeof(int) * P.ncell);
    const size_t o_needlist = take(sizeof(int) * ((size_t)P.ncell + 1));
    const size_t o_chg = take(sizeof(int) * (2 * (size_t)P.ncell + P.nreg));
    const size_t o_X = take(sizeof(double) * (size_t)P.ncell * P.nX * P.npixp);
    const size_t o_G = take(sizeof(double) * (size_t)P.ncell * HP_MAXX * HP_MAXX);
    const size_t o_Gp = take(sizeof(double) * (size_t)P.ncell * GR_SPLIT * HP_MAXX * HP_MAXX);
    const size_t o_Gold = take(sizeof(double) * (size_t)P.ncell * HP_MAXX * HP_MAXX);
    const size_t o_phi = take(sizeof(double) * (size_t)P.ncell * P.nkp);
    const size_t o_phiold = take(sizeof(double) * (size_t)P.ncell * P.nkp);
    const size_t o_vbar = take(sizeof(double) * P.ncell);
    const size_t o_A = take(sizeof(double) * (size_t)P.nreg * (P.nunk + 1) * lda);
    const size_t o_AT = take(sizeof(double) * (size_t)P.nreg * P.nunk * ldt);
    const size_t o_rhs = take(sizeof(double) * (size_t)P.nreg * P.nunk);
    const size_t o_A0 = take(sizeof(double) * (size_t)P.nreg * (P.nunk + 1) * P.nunk);
    const size_t o_rhs0 = take(sizeof(double) * (size_t)P.nreg * P.nunk);
    const size_t o_dsc = take(sizeof(double) * (size_t)P.nreg * P.nunk);
    const size_t o_merit = take(sizeof(double) * P.ncell);
    const size_t o_smask = take(sizeof(unsigned long long));
    const size_t slab = off;
    char* base = nullptr;
    ZM_TRY(ctx->get("hpb_slab", slab * (size_t)njobs, (void**)&base));
    uint8_t* tmp8 = nullptr;
    ZM_TRY(ctx->get("hp_tmp8", np, (void**)&tmp8));
    int *ibuf_all = nullptr, *rflags_all = nullptr, *h_rflags = nullptr;
    ZM_TRY(ctx->get("hpb_ibuf", sizeof(int) * HP_NIBUF * (size_t)njobs, (void**)&ibuf_all));
    ZM_TRY(ctx->get("hpb_rflags", sizeof(int) * 16 * (size_t)njobs, (void**)&rflags_all));
    // stamp statistics and kernel sums of all jobs side by side: with the counters, two copies bring the batch's
    // fit summaries back (the solution vectors stay on the device)
    double* sum_all = nullptr;
    ZM_TRY(ctx->get("hpb_sum", sizeof(double) * 3 * HP_MAXREG * (size_t)njobs, (void**)&sum_all));
    // (two host copies of the flags, by parity of the round: the copy of round r + 1 may land while round r's is read)
    ZM_TRY(ctx->get_pinned("hpb_rflags_h", sizeof(int) * 32 * (size_t)njobs, (void**)&h_rflags));
    // job tables: the full one, and one per later round holding only the jobs that can still be fitting then
    hp_job *h_jobs = nullptr, *d_jobs = nullptr;
    ZM_TRY(ctx->get_pinned("hpb_jobs_h", sizeof(hp_job) * (size_t)njobs * 9, (void**)&h_jobs));
    ZM_TRY(ctx->get("hpb_jobs", sizeof(hp_job) * (size_t)njobs * 9, (void**)&d_jobs));
    double* d_filt = nullptr;
    ZM_TRY(ctx->get("hp_filt", sizeof(double) * filt.size(), (void**)&d_filt));
    hipEvent_t* evs = nullptr;
    ZM_TRY(zm_get_sync_events(ctx, 3, &evs));
    ZM_HIP(hipStreamSynchronize(st));           // (the pinned tables of the call before are no longer being read)
    if (!(ctx->hp_filt_dev == (const void*)d_filt && ctx->hp_filt_host == filt)) {
        double* h_tab = nullptr;
        ZM_TRY(ctx->get_pinned("hp_tab", sizeof(double) * filt.size(), (void**)&h_tab));
        memcpy(h_tab, filt.data(), sizeof(double) * filt.size());
        ZM_HIP(hipMemcpyAsync(d_filt, h_tab, sizeof(double) * filt.size(), hipMemcpyHostToDevice, st));
        ctx->hp_filt_host = filt;
        ctx->hp_filt_dev = d_filt;
    }
    for (int j = 0; j < njobs; ++j) {
        char* sb = base + slab * (size_t)j;
        hp_job& J = h_jobs[j];
        J.sci = jobs[j].sci; J.ref = jobs[j].ref; J.srms = jobs[j].sci_rms; J.trms = jobs[j].ref_rms;
        J.centres = (int2*)(sb + o_centres);
        J.active = (int*)(sb + o_active); J.need = (int*)(sb + o_need); J.needlist = (int*)(sb + o_needlist);
        J.chg = (int*)(sb + o_chg);
        J.ibuf = ibuf_all + (size_t)HP_NIBUF * j; J.rflags = rflags_all + 16 * (size_t)j;
        J.X = (double*)(sb + o_X); J.G = (double*)(sb + o_G); J.Gp = (double*)(sb + o_Gp); J.Gold = (double*)(sb + o_Gold);
        J.phi = (double*)(sb + o_phi); J.phiold = (double*)(sb + o_phiold); J.vbar = (double*)(sb + o_vbar);
        J.A = (double*)(sb + o_A); J.AT = (double*)(sb + o_AT); J.rhs = (double*)(sb + o_rhs);
        J.A0 = (double*)(sb + o_A0); J.rhs0 = (double*)(sb + o_rhs0); J.dsc = (double*)(sb + o_dsc);
        J.merit = (double*)(sb + o_merit); J.stats = sum_all + (size_t)3 * HP_MAXREG * j;
        J.smask = (unsigned long long*)(sb + o_smask);
    }
    ZM_HIP(hipMemcpyAsync(d_jobs, h_jobs, sizeof(hp_job) * (size_t)njobs, hipMemcpyHostToDevice, st));
    ZM_HIP(hipMemsetAsync(ibuf_all, 0, sizeof(int) * HP_NIBUF * (size_t)njobs, st));
    ZM_HIP(hipMemsetAsync(rflags_all, 0, sizeof(int) * 16 * (size_t)njobs, st));

    // per job: validity masks, stamp search
    for (int j = 0; j < njobs; ++j) {
        char* sb = base + slab * (size_t)j;
        const hp_job& J = h_jobs[j];
        ZM_TRY(hp_launch_masks(ctx, Pj[j], jobs[j].params, jobs[j].sci, jobs[j].ref, jobs[j].bpm, (uint8_t*)(sb + o_bad), tmp8,
                               (uint8_t*)(sb + o_dirty), (uint8_t*)(sb + o_outbad)));
        ZM_TRY(hp_launch_cells(ctx, Pj[j], jobs[j].ref, (uint8_t*)(sb + o_bad), (uint8_t*)(sb + o_dirty), J.centres, J.active,
                               J.need, HPJ_NTOTAL(J)));
    }

    // the fit of all jobs: the rejection rounds, one ahead of the host as in zm_subtract_dev; a job is guarded by its
    // own round flags
    const size_t vsh = hp_vectors_cfg(P.pw, P.sw, P.npix).shmem;             // (the resident form: checked above)
    const dim3 b256(256);
    // A job leaves the launches two rounds after it converged: round r + 1 is enqueued when the host has the flags
    // of round r - 1, and takes the jobs that rejected something then (the others' rounds would be void anyway: the
    // table of a later round is the list of jobs that can still be fitting, and the grids shrink with it)
    std::vector<int> live(njobs);
    for (int j = 0; j < njobs; ++j) live[j] = j;
    auto enqueue_round = [&](const int round) -> int {
        zm_scope_timer t(ctx, "hpb_fit");
        const hp_job* d_tab = d_jobs;
        unsigned NJ = (unsigned)njobs;
        if (round >= 3) {
            hp_job* h_tab = h_jobs + (size_t)njobs * (round - 1);
            for (size_t k = 0; k < live.size(); ++k) h_tab[k] = h_jobs[live[k]];
            NJ = (unsigned)live.size();
            if (NJ == 0) return 0;
            d_tab = d_jobs + (size_t)njobs * (round - 1);
            ZM_HIP(hipMemcpyAsync((void*)d_tab, h_tab, sizeof(hp_job) * NJ, hipMemcpyHostToDevice, st));
        }
        // (later rounds: a job has a handful of cells with a new substamp; the workgroup columns walk its list - a
        // dozen per job here, where the lone subtraction takes 48: the batch pays for every empty workgroup J times)
        const int ncl_grid = std::min(P.ncell, 12);
        const unsigned gcells = round == 1 ? P.ncell : ncl_grid;
        ZM_TRY(zm_hp_launch_vectors_b(ctx, st, P, vsh, gcells, NJ, d_tab, d_filt, round));
        hipLaunchKernelGGL(k_hp_gram_b, dim3(gcells, GR_SPLIT, NJ), b256, 0, st, P, d_tab, round);
        hipLaunchKernelGGL(k_hp_gram_sum_b, dim3(gcells, 1, NJ), dim3(GS_THREADS), 0, st, d_tab, round);
        static const bool build_scalar = ZM_DEVENV("ZM_BUILD_FORM") && !strcmp(ZM_DEVENV("ZM_BUILD_FORM"), "scalar");
        if (round == 1 && !build_scalar)
            hipLaunchKernelGGL(k_hp_build_mfma_b, dim3(zm_div_up(P.nE * (P.nE + 1) / 2, 4), P.nreg, NJ), b256, 0, st, P, d_tab);
        else
            hipLaunchKernelGGL(k_hp_build_blk_b, dim3(zm_div_up(P.nE * (P.nE + 1) / 2, HBB_PAIRS), NJ, P.nreg), b256, 0, st, P, d_tab, round);
        hipLaunchKernelGGL(k_hp_diag_b, dim3(zm_div_up(P.nunk, 256), P.nreg, NJ), b256, 0, st, P.nunk, d_tab, round);
        hipLaunchKernelGGL(k_hp_scale_b, dim3(zm_div_up(P.nunk, HSB_ROWS), P.nreg * NJ), b256, 0, st, P.nunk, lda, P.nreg,
                           d_tab, round, P.ncell, P.ncellr);
        hipLaunchKernelGGL(k_chol_tp_b, dim3(P.nreg, 1, NJ), dim3(CT_THREADS), 0, st, P.nunk, lda, ldt, d_tab, round, P.ncell, P.ncellr);
        hipLaunchKernelGGL(k_chol_back_cols_b, dim3(P.nreg, 1, NJ), dim3(CBC_THREADS), 0, st, P.nunk, lda, d_tab, round, P.ncell, P.ncellr);
        hipLaunchKernelGGL(k_hp_merit_b, dim3(P.ncell, 1, NJ), dim3(64), 0, st, P, d_tab, round);
        hipLaunchKernelGGL(k_hp_reject_wave_b, dim3(P.nreg, 1, NJ), dim3(64), 0, st, P, d_tab, round);
        ZM_HIP(hipGetLastError());
        ZM_HIP(hipMemcpyAsync(h_rflags + 16 * (size_t)njobs * (round & 1), rflags_all, sizeof(int) * 16 * (size_t)njobs,
                              hipMemcpyDeviceToHost, st));
        ZM_HIP(hipEventRecord(evs[1 + (round & 1)], st));
        return 0;
    };
    // The convolution of a job is enqueued as soon as the host knows that the job has converged - on the context's
    // second stream, beside the rounds the other jobs still need (the factorisations of a late round leave most of
    // the GPU idle).  The host has seen the event behind the job's last live round by then, and the void rounds
    // that follow write nothing of the job: no further ordering is needed.
    auto enqueue_apply = [&](const int j) -> int {
        zm_scope_timer t(ctx, "hp_apply");
        char* sb = base + slab * (size_t)j;
        const hp_job& J = h_jobs[j];
        hipLaunchKernelGGL(k_hp_solved, dim3(1), dim3(64), 0, ctx->stream, P.nreg, P.nunk, J.stats, HPJ_FAIL(J), HPJ_TMO(J),
                           J.rhs, J.smask, J.stats + 2 * HP_MAXREG);
        ZM_TRY(zm_hp_launch_apply(ctx, Pj[j], J.smask, jobs[j].sci, jobs[j].ref, jobs[j].sci_rms, jobs[j].ref_rms,
                                  (uint8_t*)(sb + o_outbad), d_filt, J.rhs, jobs[j].out_diff, jobs[j].out_rms, HPJ_NMASKED(J)));
        if (jobs[j].params->flag_mask_dev)
            ZM_TRY(zm_mask_flag_dev(ctx, jobs[j].params->flag_mask_dev, jobs[j].out_diff, 1e-30f, jobs[j].params->flag_bit, np));
        return 0;
    };
    static const bool apply_beside = !(ZM_DEVENV("ZM_BATCH_APPLY") && !strcmp(ZM_DEVENV("ZM_BATCH_APPLY"), "after"));
    struct stream_swap {                                  // (launch_apply and the timers enqueue on ctx->stream)
        zm_ctx* c; hipStream_t keep;
        stream_swap(zm_ctx* cc, hipStream_t s) : c(cc), keep(cc->stream) { c->stream = s; }
        ~stream_swap() { c->stream = keep; }
    };
    std::vector<int> rounds(njobs, 0);
    //  the batch counts as a fitting context of this device for as long as its rounds run, and
    // convolutions enqueued on the second stream are joined to the main one on EVERY way out of the call - also
    // when a launch in between fails and the caller goes on to reuse the products' planes
    hp_fit_guard fitting(&g_hp_fitting[ctx->device & 63]);
    struct aux_join {
        zm_ctx* c; hipStream_t st; hipEvent_t ev; bool armed;
        ~aux_join() {
            if (!armed) return;
            if (hipEventRecord(ev, c->aux) == hipSuccess) (void)hipStreamWaitEvent(st, ev, 0);
        }
    } join{ctx, st, evs[0], false};
    ZM_TRY(enqueue_round(1));
    for (int r = 1; r <= 8; ++r) {
        if (r < 8) ZM_TRY(enqueue_round(r + 1));
        ZM_HIP(hipEventSynchronize(evs[1 + (r & 1)]));
        const int* fl = h_rflags + 16 * (size_t)njobs * (r & 1);
        live.clear();
        for (int j = 0; j < njobs; ++j) {
            if (rounds[j]) continue;                     // (converged in an earlier round)
            if (fl[16 * j + r] == 0 || r == 8) {
                rounds[j] = r;
                if (apply_beside) {
                    stream_swap sw(ctx, zm_ctx_aux(ctx));
                    join.armed = true;
                    ZM_TRY(enqueue_apply(j));
                }
            } else {
                live.push_back(j);
            }
        }
        if (live.empty()) break;
    }
    fitting.release();
    if (apply_beside) {
        join.armed = false;
        ZM_HIP(hipEventRecord(evs[0], zm_ctx_aux(ctx)));
        ZM_HIP(hipStreamWaitEvent(st, evs[0], 0));
    } else {
        for (int j = 0; j < njobs; ++j) ZM_TRY(enqueue_apply(j));
    }
    // one read of the fit summaries
    std::vector<int> h_int((size_t)HP_NIBUF * njobs);
    std::vector<double> h_sum((size_t)3 * HP_MAXREG * njobs);
    ZM_HIP(hipMemcpyAsync(h_int.data(), ibuf_all, sizeof(int) * h_int.size(), hipMemcpyDeviceToHost, st));
    ZM_HIP(hipMemcpyAsync(h_sum.data(), sum_all, sizeof(double) * h_sum.size(), hipMemcpyDeviceToHost, st));
    ZM_HIP(hipStreamSynchronize(st));
    for (int j = 0; j < njobs && infos; ++j) {
        const int* hi = h_int.data() + (size_t)HP_NIBUF * j;
        const double* hs = h_sum.data() + (size_t)3 * HP_MAXREG * j;
        const double* hx0 = hs + 2 * HP_MAXREG;
        zm_hp_info* info = &infos[j];
        memset(info, 0, sizeof(*info));
        double ks = 0, chi = 0;
        int nsolved = 0;
        for (int r = 0; r < P.nreg; ++r) {
            info->nstamps_total += hi[HP_MAXREG + r];
            info->nstamps_used += (int)hs[2 * r + 1];
            const bool solved = hs[2 * r + 1] >= 1.0 && hi[2 * HP_MAXREG + r] == 0 && hi[3 * HP_MAXREG + 4 + r] == 0 &&
                                std::isfinite(hx0[r]);
            if (solved) { ks += hx0[r]; chi += hs[2 * r]; ++nsolved; }
        }
        info->niter = rounds[j];
        info->ncoeff = P.nunk;
        info->kernel_sum = nsolved ? ks / nsolved : 0.0;
        info->chi2 = nsolved ? chi / nsolved : 0.0;
        info->nmasked = hi[3 * HP_MAXREG];
        info->status = (nsolved == P.nreg ? 0 : ZM_HP_UNSOLVED);
        info->nunsolved = P.nreg - nsolved;
        info->retries = 0;
    }
    return 0;
}

extern "C" int zm_subtract(zm_ctx* ctx, const float* sci, const float* sci_rms, const float* ref,
                           const float* ref_rms, const uint8_t* bpm, int nx, int ny,
                           const zm_hp_params* hp, float* out_diff, float* out_rms, zm_hp_info* info) {
    ZM_CHECK(ctx && sci && sci_rms && ref && ref_rms && hp && out_diff && out_rms,
             "zm_subtract: null argument");
    ZM_CHECK(nx > 0 && ny > 0, "zm_subtract: empty image");
    ZM_CHECK(!hp->flag_mask_dev && !hp->limits_dev, "zm_subtract: limits_dev / flag_mask_dev belong to the device entry points");
    ZM_HIP(hipSetDevice(ctx->device));
    const size_t np = (size_t)nx * ny;
    float* d[6];
    const float* h[4] = {sci, sci_rms, ref, ref_rms};
    const char* nm[6] = {"hs_sci", "hs_srms", "hs_ref", "hs_rrms", "hs_diff", "hs_noise"};
    for (int i = 0; i < 6; ++i) ZM_TRY(ctx->get(nm[i], np * 4, (void**)&d[i]));
    for (int i = 0; i < 4; ++i) ZM_HIP(hipMemcpyAsync(d[i], h[i], np * 4, hipMemcpyHostToDevice, ctx->stream));
    uint8_t* d_bpm = nullptr;
    if (bpm) {
        ZM_TRY(ctx->get("hs_bpm", np, (void**)&d_bpm));
        ZM_HIP(hipMemcpyAsync(d_bpm, bpm, np, hipMemcpyHostToDevice, ctx->stream));
    }
    zm_hp_params hps = *hp;
    hps.async_info = 0;                          // (host entry point: the planes are copied back below, the summary with them)
    ZM_TRY(zm_subtract_dev(ctx, d[0], d[1], d[2], d[3], d_bpm, nx, ny, &hps, d[4], d[5], info));
    ZM_HIP(hipMemcpyAsync(out_diff, d[4], np * 4, hipMemcpyDeviceToHost, ctx->stream));
    ZM_HIP(hipMemcpyAsync(out_rms, d[5], np * 4, hipMemcpyDeviceToHost, ctx->stream));
    ZM_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

// zm_subtract_batch_dev on host planes: the jobs' planes are staged in one device buffer (seven planes per job),
// the products copied back when the batch is done.
extern "C" int zm_subtract_batch(zm_ctx* ctx, int njobs, const zm_sub_job* jobs, int nx, int ny, zm_hp_info* infos) {
    ZM_CHECK(ctx && jobs && njobs >= 1 && njobs <= ZM_SUB_BATCH_MAX, "zm_subtract_batch: 1 .. %d jobs", ZM_SUB_BATCH_MAX);
    ZM_CHECK(nx > 0 && ny > 0, "zm_subtract_batch: empty image");
    ZM_HIP(hipSetDevice(ctx->device));
    const size_t np = (size_t)nx * ny, npa = (np + 63) & ~(size_t)63;
    char* stage = nullptr;
    ZM_TRY(ctx->get("hsb_planes", (size_t)njobs * (6 * 4 + 1) * npa, (void**)&stage));
    std::vector<zm_sub_job> dj(njobs);
    for (int j = 0; j < njobs; ++j) {
        const zm_sub_job& jb = jobs[j];
        ZM_CHECK(jb.sci && jb.sci_rms && jb.ref && jb.ref_rms && jb.params && jb.out_diff && jb.out_rms,
                 "zm_subtract_batch: null argument in job %d", j);
        ZM_CHECK(!jb.params->limits_dev && !jb.params->flag_mask_dev,
                 "zm_subtract_batch: limits_dev / flag_mask_dev belong to the device entry points");
        char* b = stage + (size_t)j * 25 * npa;
        float* d[6];
        for (int i = 0; i < 6; ++i) d[i] = (float*)(b + (size_t)i * 4 * npa);
        uint8_t* d_bpm = (uint8_t*)(b + (size_t)24 * npa);
        const float* h[4] = {jb.sci, jb.sci_rms, jb.ref, jb.ref_rms};
        for (int i = 0; i < 4; ++i) ZM_HIP(hipMemcpyAsync(d[i], h[i], np * 4, hipMemcpyHostToDevice, ctx->stream));
        if (jb.bpm) ZM_HIP(hipMemcpyAsync(d_bpm, jb.bpm, np, hipMemcpyHostToDevice, ctx->stream));
        dj[j] = jb;
        dj[j].sci = d[0]; dj[j].sci_rms = d[1]; dj[j].ref = d[2]; dj[j].ref_rms = d[3];
        dj[j].bpm = jb.bpm ? d_bpm : nullptr;
        dj[j].out_diff = d[4]; dj[j].out_rms = d[5];
    }
    ZM_TRY(zm_subtract_batch_dev(ctx, njobs, dj.data(), nx, ny, infos));
    for (int j = 0; j < njobs; ++j) {
        ZM_HIP(hipMemcpyAsync(jobs[j].out_diff, dj[j].out_diff, np * 4, hipMemcpyDeviceToHost, ctx->stream));
        ZM_HIP(hipMemcpyAsync(jobs[j].out_rms, dj[j].out_rms, np * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    ZM_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

