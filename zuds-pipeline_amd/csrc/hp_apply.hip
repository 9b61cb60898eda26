// The spatially varying convolution of the subtraction (hotpants -c t -n i: template and template variance convolved
// with the kernel of each (2 r + 1)^2 block, zuds/hotpants.py:77-84), one instance per kernel half width 1 .. 20.
#include "hp_dev.h"

// background coefficient t of a region's solution vector
__device__ inline double xs_bg(const double* __restrict__ xsol, int reg, const hp_plan& P, int t) {
    return xsol[(size_t)reg * P.nunk + 1 + (size_t)(P.nc - 1) * P.nkp + t];
}

// ---------------------------------------------------------------------------
// Apply.  One workgroup = NB consecutive kernel blocks of one block row; each lane
// owns R consecutive output pixels of one row of one block and slides a register
// window over the LDS tile; kernel taps come from LDS (few distinct addresses per
// wave: broadcast).  Template and template-variance planes are convolved together.
// worst number of lanes of a half wave that meet in one of the 64 LDS banks when lane l reads the
// float2 at (l / lpr) * pitch + (l % lpr) * r: the window loads of the convolution
constexpr int apply_bank_passes(int pitch, int lpr, int r, int step) {
    int worst = 0;
    for (int half = 0; half < 2; ++half) {
        int cnt[64] = {};
        for (int l = 32 * half; l < 32 * half + 32; ++l) {
            const int row = l / lpr, strip = l % lpr;
            if (row >= step) continue;
            const int e = row * pitch + strip * r;
            ++cnt[(2 * e) % 64];
            ++cnt[(2 * e + 1) % 64];
        }
        for (int b = 0; b < 64; ++b) worst = cnt[b] > worst ? cnt[b] : worst;
    }
    return worst;
}
// smallest row pitch >= width (at most 16 more) with the fewest bank conflicts
constexpr int apply_pitch(int width, int lpr, int r, int step) {
    int best = width, bw = apply_bank_passes(width, lpr, r, step);
    for (int p = width + 1; p <= width + 16; ++p) {
        const int w = apply_bank_passes(p, lpr, r, step);
        if (w < bw) { bw = w; best = p; }
    }
    return best;
}

constexpr int apply_pitch_n(int width, int lpr, int r, int step, int range) {
    int best = width, bw = apply_bank_passes(width, lpr, r, step);
    for (int p = width + 1; p <= width + range; ++p) {
        const int w = apply_bank_passes(p, lpr, r, step);
        if (w < bw) { bw = w; best = p; }
    }
    return best;
}

template <int HWK> struct apply_cfg {
    enum { STEP = 2 * HWK + 1,
           LPR = (STEP <= 11) ? 1 : (STEP <= 22 ? 2 : 3),   // lanes per block row
           R = (STEP + LPR - 1) / LPR,
           LPB = STEP * LPR,                 // lanes per block
           NB = 256 / LPB > 0 ? 256 / LPB : 1,
           TW = NB * STEP + 2 * HWK,         // tile width
           TP = apply_pitch(NB * STEP + 2 * HWK, LPR, (STEP + LPR - 1) / LPR, STEP) };   // tile row pitch
};

template <int HWK>
__global__ __launch_bounds__(256) void k_hp_apply(const hp_plan P, const unsigned long long* __restrict__ solved_mask,
                                                  const float* __restrict__ sci,
                                                  const float* __restrict__ ref,
                                                  const float* __restrict__ srms,
                                                  const float* __restrict__ trms,
                                                  const uint8_t* __restrict__ outbad,
                                                  const double* __restrict__ filt,    // [nf1][STEP] 1-D filters
                                                  const double* __restrict__ xsol,
                                                  float* __restrict__ diff,
                                                  float* __restrict__ noise,
                                                  int* __restrict__ nmasked) {
    // every region in one launch: blockIdx.z = region (the grid covers the largest one)
    const int reg = blockIdx.z;
    const int solved = (int)((*solved_mask >> reg) & 1ull);      // (k_hp_solved: the fit's outcome, read on the device)
    typedef apply_cfg<HWK> C;
    constexpr int STEP = C::STEP, R = C::R, LPR = C::LPR, LPB = C::LPB, NB = C::NB;
    constexpr int TW = C::TW;                 // tile width
    constexpr int TP = C::TP;                 // row pitch of the tile in LDS (bank-conflict padding)
    constexpr int TH = STEP + 2 * HWK;
    // LDS: template and template variance interleaved ({T, V} pairs), the per-block kernel as
    // {k, k^2} pairs: one v_pk_fma_f32 per tap and pixel feeds both planes, one ds_read_b64 per
    // operand
    typedef float ap_v2f __attribute__((ext_vector_type(2)));
    extern __shared__ float ap_smem[];
    ap_v2f* tTV = reinterpret_cast<ap_v2f*>(ap_smem);              // [TH][TP]
    // the tile's space first serves the kernel evaluation (solution vector, term scales, 1-D
    // filters, g_f, s0: nunk + nc + (NB + 1) nf1 STEP + NB doubles) and last the output staging
    const int tvn = max(TH * TP, P.nunk + 2 * P.nc + (NB + 1) * P.nf1 * STEP + NB + 2 * NB * P.nkp);
    ap_v2f* kc = tTV + tvn;                                        // [NB][STEP*STEP]
    double* cf = reinterpret_cast<double*>(kc + NB * STEP * STEP);  // [NB][nc]
    __shared__ int wmask[4];
    double* xs = reinterpret_cast<double*>(ap_smem);                // [nunk] this region's solution
    double* ts = xs + P.nunk;                                       // [nc] term scales
    double* sb = ts + P.nc;                                         // [nc] 1.0 where term 0 is subtracted, else 0.0
    double* fl = sb + P.nc;                                         // [nf1][STEP]
    double* gf = fl + P.nf1 * STEP;                                 // [NB][nf1][STEP]
    double* s0v = gf + NB * P.nf1 * STEP;                           // [NB] sum of the c_n with sub0_n
    double* pxy = s0v + NB;                                         // [NB][nkp][2] x^i, y^j at the block centres
    const int tid = threadIdx.x;
    const int x0r = P.rx0[reg], x1r = P.rx1[reg], y0r = P.ry0[reg], y1r = P.ry1[reg];
    const int gx0 = x0r + blockIdx.x * NB * STEP;      // first block of this workgroup
    const int gy0 = y0r + blockIdx.y * STEP;
    if (gx0 >= x1r || gy0 >= y1r) return;              // beyond this (smaller) region
    const double xc = x0r + 0.5 * (x1r - x0r), hx = 0.5 * (x1r - x0r);
    const double yc = y0r + 0.5 * (y1r - y0r), hy = 0.5 * (y1r - y0r);
    const double* x = xsol + (size_t)reg * P.nunk;
    // tables into LDS, every load of a thread issued before its first store (a copy loop pays a
    // memory latency per iteration; the exponent tables sit in the kernel-argument segment, and
    // indexing them inside the polynomial loop below would cost two latencies per term)
    for (int e0 = tid; e0 < P.nunk; e0 += 256 * 4) {
        double t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = (e0 + 256 * u < P.nunk) ? x[e0 + 256 * u] : 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (e0 + 256 * u < P.nunk) xs[e0 + 256 * u] = t[u];
    }
    for (int e0 = tid; e0 < P.nf1 * STEP; e0 += 256 * 4) {
        double t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = (e0 + 256 * u < P.nf1 * STEP) ? filt[e0 + 256 * u] : 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (e0 + 256 * u < P.nf1 * STEP) fl[e0 + 256 * u] = t[u];
    }
    for (int e = tid; e < P.nc; e += 256) { ts[e] = P.tscale[e]; sb[e] = P.tsub0[e] ? 1.0 : 0.0; }
    for (int e = tid; e < NB * P.nkp; e += 256) {
        const int b = e / P.nkp, pp = e - b * P.nkp;
        const double fx = (gx0 + b * STEP + HWK - xc) / hx, fy = (gy0 + HWK - yc) / hy;
        pxy[2 * e] = ipowd(fx, P.kpi[pp]);
        pxy[2 * e + 1] = ipowd(fy, P.kpj[pp]);
    }
    __syncthreads();
    // per-block basis coefficients at the nominal block centre (fp64)
    for (int e = tid; e < NB * P.nc; e += 256) {
        int b = e / P.nc, n = e - b * P.nc;
        double v;
        if (n == 0) v = xs[0];
        else {
            v = 0.0;
            const double* xb = xs + 1 + (n - 1) * P.nkp;
            const double* pb = pxy + 2 * b * P.nkp;
            for (int p = 0; p < P.nkp; ++p) v += xb[p] * pb[2 * p] * pb[2 * p + 1];
        }
        cf[e] = v;
    }
    __syncthreads();
    if (tid < NB) {
        // (flags staged in LDS with the scales: a scalar table load per term would sit in this
        // loop's critical path while the rest of the workgroup waits at the barrier)
        double t = 0.0;
        for (int n = 0; n < P.nc; ++n) t += cf[tid * P.nc + n] * sb[n];
        s0v[tid] = t;
    }
    __syncthreads();
    // the block kernels from the separable form of the basis, all operands in LDS:
    //   K[v][u] = sum_n c_n (s_n fy_n[v] fx_n[u] - [sub0_n] s_0 fy_0[v] fx_0[u]) = sum_f fx_f[u] g_f[v],
    //   g_f[v] = sum_{n: fx_n = f} c_n s_n fy_n[v]  -  [f = fx_0] (sum_{n: sub0_n} c_n) s_0 fy_0[v]
    // 15 terms per tap instead of 49 fp64 rows of the 2-D basis fetched from L2.  The terms of
    // Gaussian g are ordered (a, b): those sharing the x filter base_g + a are consecutive.
    for (int e = tid; e < NB * STEP * P.ngauss; e += 256) {
        const int g = e / (NB * STEP), r = e - g * NB * STEP;
        const int b = r / STEP, v = r - b * STEP;
        const double* cb = cf + b * P.nc;
        const int deg = P.gdeg[g], fb = P.gbase[g];
        int n = P.gterm0[g];
        for (int a = 0; a <= deg; ++a) {
            double acc = 0.0;
            for (int bb = 0; bb <= deg - a; ++bb, ++n) acc += cb[n] * ts[n] * fl[(fb + bb) * STEP + v];
            if (fb + a == P.tfx[0])                                 // the x filter of term 0
                acc -= s0v[b] * ts[0] * fl[P.tfy[0] * STEP + v];
            gf[((size_t)b * P.nf1 + fb + a) * STEP + v] = acc;
        }
    }
    __syncthreads();
    for (int e = tid; e < NB * STEP * STEP; e += 256) {
        const int b = e / (STEP * STEP), tap = e - b * STEP * STEP;
        const int v = tap / STEP, u = tap - v * STEP;
        const double* gb = gf + (size_t)b * P.nf1 * STEP + v;
        double acc = 0.0;
#pragma unroll 5
        for (int f = 0; f < P.nf1; ++f) acc += fl[f * STEP + u] * gb[f * STEP];
        const float k = (float)acc;
        kc[e] = (ap_v2f){k, k * k};
    }
    const double bg0 = xs[1 + (size_t)(P.nc - 1) * P.nkp];       // constant background term
    const float norm = P.normalize ? (float)(1.0 / xs[0]) : 1.f;
    __syncthreads();                                   // the evaluation scratch is free
    // tile of template and template variance (zeros outside the frame / non-finite)
    // (every load of a thread goes out before the first LDS store: one memory latency per
    // workgroup instead of one per loop iteration)
    {
        constexpr int NIT = (TH * TW + 255) / 256;
        float tt[NIT], tr[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = tid + 256 * it;
            const int yy = e / TW, xx = e - yy * TW;
            const int gx = gx0 - HWK + xx, gy = gy0 - HWK + yy;
            tt[it] = 0.f;
            tr[it] = 0.f;
            if (e < TH * TW && gx >= 0 && gx < P.nx && gy >= 0 && gy < P.ny) {
                const size_t idx = (size_t)gy * P.nx + gx;
                tt[it] = ref[idx];
                tr[it] = trms[idx];
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = tid + 256 * it;
            float t = tt[it], v = tr[it] * tr[it];
            if (!(fabsf(t) < 3e38f)) t = 0.f;
            if (!(fabsf(v) < 3e38f)) v = 0.f;
            if (e < TH * TW) tTV[(e / TW) * TP + e % TW] = (ap_v2f){t, v};
        }
    }
    __syncthreads();
    const int b = tid / LPB;
    const int l = tid - b * LPB;
    const int row = l / LPR, strip = l - row * LPR;
    const int ox0 = b * STEP + strip * R;              // tile-relative (without halo) x of first px
    const bool live = b < NB && gy0 + row < y1r && gy0 + row < P.ny;
    ap_v2f acc2[R];                                    // {sum k T, sum k^2 V}
#pragma unroll
    for (int q = 0; q < R; ++q) acc2[q] = (ap_v2f){0.f, 0.f};
    if (live) {
        const ap_v2f* kb = kc + b * STEP * STEP;
        // true convolution: out(x, y) = sum_{u,v} K[v][u] T(x - u, y - v); K index (v + HWK, u + HWK)
        for (int v = -HWK; v <= HWK; ++v) {
            const ap_v2f* rt = tTV + (row + HWK - v) * TP + ox0;    // T(x - u): column ox0 + q + HWK - u
            ap_v2f w2[R + 2 * HWK];
#pragma unroll
            for (int q = 0; q < R + 2 * HWK; ++q) w2[q] = rt[q];
            const ap_v2f* kr = kb + (v + HWK) * STEP;
#pragma unroll
            for (int u = -HWK; u <= HWK; ++u) {
                const ap_v2f k2 = kr[u + HWK];
#pragma unroll
                for (int q = 0; q < R; ++q) acc2[q] = __builtin_elementwise_fma(k2, w2[q + HWK - u], acc2[q]);
            }
        }
    }
    // the sums go through LDS (the tile's space) so that the science / noise planes are read and
    // the outputs written along rows: NB STEP consecutive pixels per row instead of R per thread
    constexpr int OW = NB * STEP;
    __syncthreads();
    if (live) {
#pragma unroll
        for (int q = 0; q < R; ++q)
            if (strip * R + q < STEP) tTV[row * OW + ox0 + q] = acc2[q];
    }
    __syncthreads();
    int masked = 0;
    {
        constexpr int NE = (STEP * OW + 255) / 256;
        float es[NE], er[NE];
        bool eb[NE], ein[NE];
#pragma unroll
        for (int it = 0; it < NE; ++it) {
            const int e = tid + 256 * it;
            const int orow = e / OW, ocol = e - orow * OW;
            const int gx = gx0 + ocol, gy = gy0 + orow;
            ein[it] = e < STEP * OW && gx < x1r && gx < P.nx && gy < y1r && gy < P.ny;
            es[it] = er[it] = 0.f;
            eb[it] = true;
            if (ein[it]) {
                const size_t idx = (size_t)gy * P.nx + gx;
                eb[it] = outbad[idx] != 0;
                es[it] = sci[idx];
                er[it] = srms[idx];
            }
        }
#pragma unroll
        for (int it = 0; it < NE; ++it) {
            if (!ein[it]) continue;
            const int e = tid + 256 * it;
            const int orow = e / OW, ocol = e - orow * OW;
            const int gx = gx0 + ocol, gy = gy0 + orow;
            const size_t idx = (size_t)gy * P.nx + gx;
            float d = P.fi, nz = P.fin;
            if (solved && !eb[it]) {
                double bg = bg0;
                if (P.nbg > 1) {
                    const double xf = (gx - xc) / hx, yf = (gy - yc) / hy;
                    bg = 0.0;
                    for (int t = 0; t < P.nbg; ++t)
                        bg += xs_bg(xsol, reg, P, t) * ipowd(xf, P.bpi[t]) * ipowd(yf, P.bpj[t]);
                }
                const ap_v2f a = tTV[e];
                d = (es[it] - a.x - (float)bg) * norm;
                nz = sqrtf(fmaxf(er[it] * er[it] + a.y, 0.f)) * fabsf(norm);
            } else {
                masked += 1;
            }
            diff[idx] = d;
            noise[idx] = nz;
        }
    }
    // one atomic per workgroup
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) masked += __shfl_xor(masked, o);
    if ((tid & 63) == 0) wmask[tid >> 6] = masked;
    __syncthreads();
    if (tid == 0) {
        const int tot = wmask[0] + wmask[1] + wmask[2] + wmask[3];
        if (tot) atomicAdd(nmasked, tot);
    }
}

// ---------------------------------------------------------------------------
// Apply, one wave per kernel block (round 4).  k_hp_apply above spends more time around its convolution than in
// it (41 % of the vector issue slots at 3 waves per SIMD): every workgroup first evaluates its blocks' kernels in
// fp64 behind five barriers, the {k, k^2} taps take a third of the LDS reads of the inner loop (a block is 42
// lanes: a wave straddles two blocks, the taps are not uniform), and 70 KB of LDS per workgroup leave two
// workgroups per CU.  Here:
//   * the block kernels are evaluated once by a kernel of their own (k_hp_kernels: the arithmetic of k_hp_apply's
//     prologue, the same bits) into a {k, k^2} table in global memory, 3.5 KB per block;
//   * a wave owns ONE block (STEP rows x LPR strips of R columns <= 64 lanes), so its taps are wave-uniform: they
//     arrive through the scalar cache (s_load_dwordx16, a tap row ahead) and enter v_pk_fma_f32 as a scalar
//     operand - no LDS read, no vector register;
//   * LDS holds the {T, V} tile only (34 KB for four blocks of 21): four workgroups per CU.
// The sums run in k_hp_apply's order (v outer, u inner, one packed FMA per tap and pixel): the same bits.
template <int HWK> struct applyw_cfg {
    enum { STEP = 2 * HWK + 1,
           LPR = 64 / STEP > 0 ? 64 / STEP : 1,          // lanes per block row (strips)
           R = (STEP + LPR - 1) / LPR,                   // output pixels per lane
           NBW = 4,                                      // blocks (= waves) per workgroup
           TW = NBW * STEP + 2 * HWK,
           TH = STEP + 2 * HWK,
           // tile row pitch in float2 units: the window loads of a wave (lane = (row, strip)) free of bank
           // conflicts where a pitch within 32 of the width allows it (HWK 10: 117 - with 105 every load took
           // two passes and the kernel was bound by the LDS pipe: 308 us)
           TP = apply_pitch_n(NBW * STEP + 2 * HWK, 64 / STEP > 0 ? 64 / STEP : 1,
                              (STEP + (64 / STEP > 0 ? 64 / STEP : 1) - 1) / (64 / STEP > 0 ? 64 / STEP : 1), STEP, 32) };
    // lanes doing useful work x columns doing useful work, in percent
    enum { EFF = (100 * STEP * LPR / 64) * STEP / (LPR * R) };
};

// The table of block kernels.  k_hp_apply evaluates a block's kernel as K = sum_n c_n B_n with the coefficients
// c_n = sum_p x[n, p] X^i_p Y^j_p taken at the block centre first - per block a chain of small fp64 stages behind
// five barriers (as a kernel of its own: 115 us per frame, all latency).  The same sum with the spatial terms
// outside, K = x_0 B_0 + sum_p (X^i_p Y^j_p) M_p, M_p = sum_n x[n, p] B_n, has per-REGION matrices M_p
// (k_hp_kbasis, a few hundred thousand products per subtraction) and leaves nkp fused multiply-adds per tap and
// block (k_hp_ktable: a thread keeps the M_p of its taps in registers and walks along a row of blocks).  Equal to
// k_hp_apply's kernels up to fp64 rounding of the reordered sums, i.e. to the last bit of the fp32 taps in all but
// ~1e-8 of them.
#define HPK_MAXP 15   // spatial terms the register path holds (ko <= 4); more: k_hp_apply
#define HPK_NBK 16    // blocks per workgroup of k_hp_ktable
template <int HWK>
__global__ __launch_bounds__(256) void k_hp_kbasis(const hp_plan P, const double* __restrict__ filt,
                                                   const double* __restrict__ xsol, double* __restrict__ Mt) {
    constexpr int STEP = 2 * HWK + 1, NT = STEP * STEP;
    const int reg = blockIdx.y, p = blockIdx.x;          // p == HPK_MAXP: the constant part x_0 B_0
    const double* x = xsol + (size_t)reg * P.nunk;
    // the term tables, this spatial term's coefficients and the 1-D filters in LDS first (read through the
    // kernel-argument segment and global memory inside the sum, every term paid two dependent latencies: 17 us)
    __shared__ double xq[HP_MAXX], sn[HP_MAXX], flt[HP_MAXF1 * STEP];
    __shared__ int txn[HP_MAXX], tyn[HP_MAXX], sbn[HP_MAXX];
    const int tid = threadIdx.x;
    for (int n = tid; n < P.nc; n += 256) {
        sn[n] = P.tscale[n];
        txn[n] = P.tfx[n] * STEP;
        tyn[n] = P.tfy[n] * STEP;
        sbn[n] = P.tsub0[n];
        xq[n] = (n >= 1 && p < P.nkp) ? x[1 + (size_t)(n - 1) * P.nkp + p] : 0.0;
    }
    for (int e = tid; e < P.nf1 * STEP; e += 256) flt[e] = filt[e];
    __syncthreads();
    for (int tap = tid; tap < NT; tap += 256) {
        const int v = tap / STEP, u = tap - v * STEP;
        const double b0 = sn[0] * flt[tyn[0] + v] * flt[txn[0] + u];
        double acc = 0.0;
        if (p == HPK_MAXP) {
            acc = x[0] * (b0 - (sbn[0] ? b0 : 0.0));
        } else if (p < P.nkp) {
#pragma unroll 8
            for (int n = 1; n < P.nc; ++n) {
                const double bn = sn[n] * flt[tyn[n] + v] * flt[txn[n] + u];
                acc += xq[n] * (bn - (sbn[n] ? b0 : 0.0));
            }
        }
        Mt[((size_t)reg * (HPK_MAXP + 1) + p) * NT + tap] = acc;
    }
}

template <int HWK>
__global__ __launch_bounds__(512) void k_hp_ktable(const hp_plan P, const double* __restrict__ Mt, int maxbx, int maxby,
                                                   float2* __restrict__ kcg) {
    constexpr int STEP = 2 * HWK + 1, NT = STEP * STEP, TPT = (NT + 511) / 512;
    __shared__ double W[HPK_NBK][HPK_MAXP + 1];
    const int reg = blockIdx.z, tid = threadIdx.x;
    const int x0r = P.rx0[reg], x1r = P.rx1[reg], y0r = P.ry0[reg], y1r = P.ry1[reg];
    const int bx0 = blockIdx.x * HPK_NBK;
    const int gx0 = x0r + bx0 * STEP, gy0 = y0r + blockIdx.y * STEP;
    if (gx0 >= x1r || gy0 >= y1r) return;
    const double xc = x0r + 0.5 * (x1r - x0r), hx = 0.5 * (x1r - x0r);
    const double yc = y0r + 0.5 * (y1r - y0r), hy = 0.5 * (y1r - y0r);
    // the spatial terms at the nominal block centres (k_hp_apply's coordinates)
    for (int e = tid; e < HPK_NBK * (HPK_MAXP + 1); e += 512) {
        const int b = e / (HPK_MAXP + 1), pp = e - b * (HPK_MAXP + 1);
        double w = 0.0;
        if (pp < P.nkp) {
            const double fx = (gx0 + b * STEP + HWK - xc) / hx, fy = (gy0 + HWK - yc) / hy;
            w = ipowd(fx, P.kpi[pp]) * ipowd(fy, P.kpj[pp]);
        }
        W[b][pp] = w;
    }
    double m[TPT][HPK_MAXP + 1];
#pragma unroll
    for (int t = 0; t < TPT; ++t)
#pragma unroll
        for (int pp = 0; pp <= HPK_MAXP; ++pp) {
            const int tap = min(tid + 512 * t, NT - 1);
            m[t][pp] = Mt[((size_t)reg * (HPK_MAXP + 1) + pp) * NT + tap];
        }
    __syncthreads();
    for (int b = 0; b < HPK_NBK; ++b) {
        const int bx = bx0 + b;
        if (bx >= maxbx || gx0 + b * STEP >= x1r) break;
        float2* out = kcg + (((size_t)reg * maxby + blockIdx.y) * maxbx + bx) * NT;
        double w[HPK_MAXP];
#pragma unroll
        for (int pp = 0; pp < HPK_MAXP; ++pp) w[pp] = W[b][pp];
#pragma unroll
        for (int t = 0; t < TPT; ++t) {
            double acc = m[t][HPK_MAXP];
#pragma unroll
            for (int pp = 0; pp < HPK_MAXP; ++pp) acc = fma(w[pp], m[t][pp], acc);
            const float k = (float)acc;
            const int tap = tid + 512 * t;
            if (tap < NT) out[tap] = make_float2(k, k * k);
        }
    }
}

template <int HWK>
__global__ __launch_bounds__(256) void k_hp_apply_w(const hp_plan P, const unsigned long long* __restrict__ solved_mask,
                                                    const float* __restrict__ sci,
                                                    const float* __restrict__ ref,
                                                    const float* __restrict__ srms,
                                                    const float* __restrict__ trms,
                                                    const uint8_t* __restrict__ outbad,
                                                    const double* __restrict__ xsol,
                                                    const float2* __restrict__ kcg, int maxbx, int maxby,
                                                    float* __restrict__ diff,
                                                    float* __restrict__ noise,
                                                    int* __restrict__ nmasked) {
    const int reg = blockIdx.z;
    const int solved = (int)((*solved_mask >> reg) & 1ull);      // (k_hp_solved: the fit's outcome, read on the device)
    typedef applyw_cfg<HWK> C;
    constexpr int STEP = C::STEP, R = C::R, LPR = C::LPR, NB = C::NBW, TW = C::TW, TP = C::TP, TH = C::TH;
    typedef float ap_v2f __attribute__((ext_vector_type(2)));
    extern __shared__ float apw_smem[];
    ap_v2f* tTV = reinterpret_cast<ap_v2f*>(apw_smem);             // [TH][TP]
    __shared__ int wmask[4];
    const int tid = threadIdx.x;
    const int x0r = P.rx0[reg], x1r = P.rx1[reg], y0r = P.ry0[reg], y1r = P.ry1[reg];
    const int gx0 = x0r + blockIdx.x * NB * STEP;      // first block of this workgroup
    const int gy0 = y0r + blockIdx.y * STEP;
    if (gx0 >= x1r || gy0 >= y1r) return;              // beyond this (smaller) region
    const double xc = x0r + 0.5 * (x1r - x0r), hx = 0.5 * (x1r - x0r);
    const double yc = y0r + 0.5 * (y1r - y0r), hy = 0.5 * (y1r - y0r);
    const double bg0 = xsol[(size_t)reg * P.nunk + 1 + (size_t)(P.nc - 1) * P.nkp];     // constant background term
    const float norm = P.normalize ? (float)(1.0 / xsol[(size_t)reg * P.nunk]) : 1.f;
    // tile of template and template variance (zeros outside the frame / non-finite), every load of a thread
    // ahead of its first LDS store
    {
        constexpr int NIT = (TH * TW + 255) / 256;
        float tt[NIT], tr[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = tid + 256 * it;
            const int yy = e / TW, xx = e - yy * TW;
            const int gx = gx0 - HWK + xx, gy = gy0 - HWK + yy;
            tt[it] = 0.f;
            tr[it] = 0.f;
            if (e < TH * TW && gx >= 0 && gx < P.nx && gy >= 0 && gy < P.ny) {
                const size_t idx = (size_t)gy * P.nx + gx;
                tt[it] = ref[idx];
                tr[it] = trms[idx];
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = tid + 256 * it;
            float t = tt[it], v = tr[it] * tr[it];
            if (!(fabsf(t) < 3e38f)) t = 0.f;
            if (!(fabsf(v) < 3e38f)) v = 0.f;
            if (e < TH * TW) tTV[(e / TW) * TP + e % TW] = (ap_v2f){t, v};
        }
    }
    __syncthreads();
    const int b = __builtin_amdgcn_readfirstlane(tid >> 6);        // this wave's block
    const int l = tid & 63;
    const int row = l / LPR, strip = l - row * LPR;
    const int ox0 = b * STEP + strip * R;              // tile-relative (without halo) x of first px
    const bool live = row < STEP && gx0 + b * STEP < x1r && gy0 + row < y1r && gy0 + row < P.ny;
    ap_v2f acc2[R];                                    // {sum k T, sum k^2 V}
#pragma unroll
    for (int q = 0; q < R; ++q) acc2[q] = (ap_v2f){0.f, 0.f};
    {
        // (every lane runs the loop - rows beyond the block read a clamped tile row - so that the taps stay
        // wave-uniform scalar loads; dead lanes do not store)
        const int bxi = min((int)blockIdx.x * NB + b, maxbx - 1);
        const ap_v2f* kb = reinterpret_cast<const ap_v2f*>(kcg) +
                           (((size_t)reg * maxby + blockIdx.y) * maxbx + bxi) * (STEP * STEP);
        const int rowc = min(row, STEP - 1);
        // true convolution: out(x, y) = sum_{u,v} K[v][u] T(x - u, y - v); K index (v + HWK, u + HWK)
#pragma unroll 1
        for (int v = -HWK; v <= HWK; ++v) {
            const ap_v2f* rt = tTV + (rowc + HWK - v) * TP + ox0;    // T(x - u): column ox0 + q + HWK - u
            ap_v2f w2[R + 2 * HWK];
            // (strips that reach beyond the block's last column - LPR R > STEP - stay inside the tile row)
#pragma unroll
            for (int q = 0; q < R + 2 * HWK; ++q) w2[q] = rt[(LPR * R == STEP) ? q : min(q, TW - 1 - ox0)];
            const ap_v2f* kr = kb + (v + HWK) * STEP;
#pragma unroll
            for (int u = -HWK; u <= HWK; ++u) {
                const ap_v2f k2 = kr[u + HWK];
#pragma unroll
                for (int q = 0; q < R; ++q) acc2[q] = __builtin_elementwise_fma(k2, w2[q + HWK - u], acc2[q]);
            }
        }
    }
    // the sums go through LDS (the tile's space) so that the science / noise planes are read and
    // the outputs written along rows: NB STEP consecutive pixels per row instead of R per thread
    constexpr int OW = NB * STEP;
    __syncthreads();
    if (live) {
#pragma unroll
        for (int q = 0; q < R; ++q)
            if (strip * R + q < STEP) tTV[row * OW + ox0 + q] = acc2[q];
    }
    __syncthreads();
    int masked = 0;
    {
        constexpr int NE = (STEP * OW + 255) / 256;
        float es[NE], er[NE];
        bool eb[NE], ein[NE];
#pragma unroll
        for (int it = 0; it < NE; ++it) {
            const int e = tid + 256 * it;
            const int orow = e / OW, ocol = e - orow * OW;
            const int gx = gx0 + ocol, gy = gy0 + orow;
            ein[it] = e < STEP * OW && gx < x1r && gx < P.nx && gy < y1r && gy < P.ny;
            es[it] = er[it] = 0.f;
            eb[it] = true;
            if (ein[it]) {
                const size_t idx = (size_t)gy * P.nx + gx;
                eb[it] = outbad[idx] != 0;
                es[it] = sci[idx];
                er[it] = srms[idx];
            }
        }
#pragma unroll
        for (int it = 0; it < NE; ++it) {
            if (!ein[it]) continue;
            const int e = tid + 256 * it;
            const int orow = e / OW, ocol = e - orow * OW;
            const int gx = gx0 + ocol, gy = gy0 + orow;
            const size_t idx = (size_t)gy * P.nx + gx;
            float d = P.fi, nz = P.fin;
            if (solved && !eb[it]) {
                double bg = bg0;
                if (P.nbg > 1) {
                    const double xf = (gx - xc) / hx, yf = (gy - yc) / hy;
                    bg = 0.0;
                    for (int t = 0; t < P.nbg; ++t)
                        bg += xs_bg(xsol, reg, P, t) * ipowd(xf, P.bpi[t]) * ipowd(yf, P.bpj[t]);
                }
                const ap_v2f a = tTV[e];
                d = (es[it] - a.x - (float)bg) * norm;
                nz = sqrtf(fmaxf(er[it] * er[it] + a.y, 0.f)) * fabsf(norm);
            } else {
                masked += 1;
            }
            diff[idx] = d;
            noise[idx] = nz;
        }
    }
    // one atomic per workgroup
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) masked += __shfl_xor(masked, o);
    if ((tid & 63) == 0) wmask[tid >> 6] = masked;
    __syncthreads();
    if (tid == 0) {
        const int tot = wmask[0] + wmask[1] + wmask[2] + wmask[3];
        if (tot) atomicAdd(nmasked, tot);
    }
}

// ---------------------------------------------------------------------------
template <int HWK>
static int launch_apply(zm_ctx* ctx, const hp_plan& P, const unsigned long long* solved_mask, const float* sci,
                        const float* ref, const float* srms, const float* trms, const uint8_t* outbad,
                        const double* filt, const double* xsol, float* diff, float* noise,
                        int* nmasked) {
    typedef apply_cfg<HWK> C;
    constexpr int STEP = C::STEP, NB = C::NB;
    constexpr int TP = C::TP, TH = STEP + 2 * HWK;
    const size_t tvn = std::max((size_t)TH * TP, (size_t)P.nunk + 2 * (size_t)P.nc + (size_t)(NB + 1) * P.nf1 * STEP + NB + (size_t)2 * NB * P.nkp);
    size_t fl = 2 * tvn + (size_t)2 * NB * STEP * STEP;      // {T, V} tile (or the evaluation scratch) + {k, k^2} kernels
    size_t shmem = fl * sizeof(float) + (size_t)NB * P.nc * sizeof(double);
    // (remembered per context and kernel instance, not per process: contexts may sit on
    // different devices)
    size_t& set_max = ctx->hp_set_max[HWK];
    if (set_max < 65536) set_max = 65536;
    if (shmem > set_max) {
        ZM_HIP(hipFuncSetAttribute((const void*)k_hp_apply<HWK>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        set_max = shmem;
    }
    int W = 0, H = 0;
    for (int reg = 0; reg < P.nreg; ++reg) {
        W = std::max(W, P.rx1[reg] - P.rx0[reg]);
        H = std::max(H, P.ry1[reg] - P.ry0[reg]);
    }
    // Round 4: one wave per block with the taps as scalar operands (k_hp_kernels + k_hp_apply_w) where a block
    // fills most of a wave; ZM_APPLY_FORM=tile runs the workgroup-per-six-blocks kernel (A / B, tests).
    typedef applyw_cfg<HWK> CW;
    const char* form = getenv("ZM_APPLY_FORM");
    const bool wave_form = (form ? !strcmp(form, "wave") : (CW::EFF >= 60 && HWK >= 4)) && P.nkp <= HPK_MAXP;
    if (wave_form && !(form && !strcmp(form, "tile"))) {
        const int maxbx = zm_div_up(W, STEP), maxby = zm_div_up(H, STEP);
        float2* kcg = nullptr;
        ZM_TRY(ctx->get("hp_kcg", sizeof(float2) * (size_t)P.nreg * maxby * maxbx * STEP * STEP, (void**)&kcg));
        double* Mt = nullptr;
        ZM_TRY(ctx->get("hp_kmt", sizeof(double) * (size_t)P.nreg * (HPK_MAXP + 1) * STEP * STEP, (void**)&Mt));
        hipLaunchKernelGGL(k_hp_kbasis<HWK>, dim3(HPK_MAXP + 1, P.nreg), dim3(256), 0, ctx->stream, P, filt, xsol, Mt);
        hipLaunchKernelGGL(k_hp_ktable<HWK>, dim3(zm_div_up(maxbx, HPK_NBK), maxby, P.nreg), dim3(512), 0, ctx->stream,
                           P, Mt, maxbx, maxby, kcg);
        const size_t wsh = sizeof(float2) * (size_t)CW::TH * CW::TP;
        static bool wset[HP_MAX_HWK + 1][64] = {};
        if (wsh > 65536 && !wset[HWK][ctx->device & 63]) {
            ZM_HIP(hipFuncSetAttribute((const void*)k_hp_apply_w<HWK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wsh));
            wset[HWK][ctx->device & 63] = true;
        }
        hipLaunchKernelGGL(k_hp_apply_w<HWK>, dim3(zm_div_up(maxbx, CW::NBW), maxby, P.nreg), dim3(256), wsh, ctx->stream,
                           P, solved_mask, sci, ref, srms, trms, outbad, xsol, kcg, maxbx, maxby, diff, noise, nmasked);
        ZM_HIP(hipGetLastError());
        return 0;
    }
    dim3 grd(zm_div_up(zm_div_up(W, STEP), NB), zm_div_up(H, STEP), P.nreg);
    hipLaunchKernelGGL(k_hp_apply<HWK>, grd, dim3(256), shmem, ctx->stream, P, solved_mask, sci, ref, srms,
                       trms, outbad, filt, xsol, diff, noise, nmasked);
    ZM_HIP(hipGetLastError());
    return 0;
}


int zm_hp_launch_apply(zm_ctx* ctx, const hp_plan& P, const unsigned long long* solved_mask, const float* sci,
                       const float* ref, const float* srms, const float* trms, const uint8_t* outbad,
                       const double* filt, const double* xsol, float* diff, float* noise, int* nmasked) {
#define C(H) case H: return launch_apply<H>(ctx, P, solved_mask, sci, ref, srms, trms, outbad, filt, xsol, diff, noise, nmasked);
    switch (P.hwk) {
        C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16) C(17) C(18) C(19) C(20)
        default: zm_set_error("zm_subtract: unsupported kernel half width %d", P.hwk); return 2;
    }
#undef C
}
