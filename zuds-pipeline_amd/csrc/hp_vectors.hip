// k_hp_vectors: the separable basis convolutions of the template patch of every substamp (oracle/hotpants.py,
// hotpants -r / -rss of zuds/hotpants.py:42-44), one instance per kernel half width 1 .. 20.
#include "hp_dev.h"

// BIG (round 5: half widths up to 20, substamps up to 60 - hotpants takes -r 2.5 SEEING, -rss 6 SEEING unclamped,
// zuds/hotpants.py:42-44): where the x-filtered patch, term 0 and the template patch do not fit 160 KB of LDS
// together, the x-filtered patch is built and consumed in chunks of `cwarg` substamp columns (a column's y pass
// needs that column only) and term 0 lives in global memory (`w0g`: one vector per workgroup of the launch, written
// and read by that workgroup alone, a barrier between).  Per entry the sums are those of the resident form.
template <int HWK, bool BIG = false>
static __device__ __forceinline__ void hp_vectors_body(const hp_plan& P, const float* __restrict__ sci,
                                                    const float* __restrict__ ref,
                                                    const float* __restrict__ srms,
                                                    const float* __restrict__ trms,
                                                    const double* __restrict__ filt,   // [nf1][step]
                                                    const int2* __restrict__ centres,
                                                    const int* __restrict__ active,
                                                    const int* __restrict__ need,
                                                    double* __restrict__ X,
                                                    double* __restrict__ phi,         // [cell][nkp]
                                                    double* __restrict__ vbar, const int* __restrict__ guard,
                                                    double* __restrict__ phiold, const int* __restrict__ list,
                                                    int special, int cwarg = 0, double* __restrict__ w0g = nullptr) {
    if (guard && *guard == 0) return;                    // the previous round rejected nothing: this round is void
    extern __shared__ double hp_smem[];
    constexpr int STEP = 2 * HWK + 1;
    constexpr int WIN = HV_R + 2 * HWK;
    const int tid = threadIdx.x;
    const int part = blockIdx.y, nparts = gridDim.y;
    // cells: every cell of the grid (first round: list == nullptr), or the cells the last rejection gave a new
    // substamp (`list`: [count, cells ...], written by k_hp_reject*) - a grid over all 900 cells of which a
    // handful have work spent a third of the launch dispatching workgroups that return at once
    const int ncl = list ? list[0] : (BIG ? P.ncell : (int)gridDim.x);   // (BIG: a capped grid whose workgroups loop)
#pragma unroll 1
    for (int ci = blockIdx.x; ci < ncl; ci += gridDim.x) {
    const int cell = list ? list[1 + ci] : ci;
    __syncthreads();                                     // (the LDS of the cell before is consumed)
    if (!need[cell]) continue;
    const int act = active[cell];
    if (act < 0) continue;
    const int2 cc = centres[cell * P.nss + act];
    const int r = cell / P.ncellr;
    const int pw = P.pw, sw = P.sw, hwss = P.hwss;
    // `special` (the later rounds, where a handful of cells is all there is and the slowest workgroup of a cell is
    // the kernel time): part 0 does nothing but the science row, the background rows, the variance mean and the
    // spatial terms - ~11 us of loads and a serial loop of one thread that used to sit on top of its share of the
    // filters - and the filters are dealt to the other parts
    const bool only_special = special && part == 0;
    const int fpart = special ? part - 1 : part, fparts = special ? nparts - 1 : nparts;
    // xp has HV_R zero rows below, patch HV_R zero columns to the right of the data: the register
    // windows of the two passes run over the edge unconditionally (a conditional LDS read is
    // waited for one by one)
    const int pp = pw + HV_R;                           // patch row pitch
    const int cw = BIG ? cwarg : P.sw;                  // columns of the x-filtered patch held at a time (a multiple of HV_R, or sw)
    double* xp = hp_smem;                               // [pw + HV_R][cw]
    double* w0 = (BIG && w0g) ? w0g + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (size_t)P.npix
                              : xp + (size_t)(pw + HV_R) * cw;         // [npix]
    double* red = (BIG && w0g) ? xp + (size_t)(pw + HV_R) * cw : w0 + P.npix;   // [8]
    float* patch = reinterpret_cast<float*>(red + 8);   // [pw][pp]
    // (loads in batches, stores after: a loop of load -> store pays one memory latency per
    // iteration, and after the first round a cell's latency is the kernel time)
    for (int k0 = tid; !only_special && k0 < pw * pw; k0 += HV_THREADS * 10) {
        float t[10];
#pragma unroll
        for (int u = 0; u < 10; ++u) {
            const int k = k0 + HV_THREADS * u;
            const int yy = k / pw, xx = k - yy * pw;
            t[u] = (k < pw * pw) ? ref[(size_t)(cc.y - P.hw + yy) * P.nx + (cc.x - P.hw + xx)] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 10; ++u) {
            const int k = k0 + HV_THREADS * u;
            const int yy = k / pw, xx = k - yy * pw;
            if (k < pw * pw) patch[yy * pp + xx] = t[u];
        }
    }
    // (two short zeroing loops: interleaved eight-fold by the compiler, their hoisted addresses and trip counts were
    // what k_hp_vectors_b<10, 11> spilled to scratch memory)
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
    for (int k = tid; k < pw * HV_R; k += HV_THREADS) patch[(k / HV_R) * pp + pw + k % HV_R] = 0.f;
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
    for (int k = tid; k < HV_R * cw; k += HV_THREADS) xp[(size_t)pw * cw + k] = 0.0;
    const double xc = P.rx0[r] + 0.5 * (P.rx1[r] - P.rx0[r]), hx = 0.5 * (P.rx1[r] - P.rx0[r]);
    const double yc = P.ry0[r] + 0.5 * (P.ry1[r] - P.ry0[r]), hy = 0.5 * (P.ry1[r] - P.ry0[r]);
    double* Xc = X + (size_t)cell * P.nX * P.npixp;
    // science row, background rows, variance mean, zero padding: part 0
    double vs = 0.0;
    // (the first 256 threads, as before: the partial sums of the variance mean keep their grouping)
    for (int k0 = tid; part == 0 && tid < 256 && k0 < P.npixp; k0 += 256 * 5) {
        float ts[5], ta[5], tb[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int k = k0 + 256 * u;
            ts[u] = ta[u] = tb[u] = 0.f;
            if (k < P.npix) {
                const int i = k / sw, j = k - i * sw;
                const size_t idx = (size_t)(cc.y - hwss + i) * P.nx + (cc.x - hwss + j);
                ts[u] = sci[idx];
                ta[u] = srms[idx];
                tb[u] = trms[idx];
            }
        }
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            int k = k0 + 256 * u;
            // (opaque to the compiler: it otherwise keeps five 64-bit store addresses of this batch alive across the
            // whole cell loop - and spills two of them in k_hp_vectors_b<11>)
            asm volatile("" : "+v"(k));
            if (k < P.npix) {
                const int i = k / sw, j = k - i * sw;
                const int x = cc.x - hwss + j, y = cc.y - hwss + i;
                Xc[(size_t)P.nE * P.npixp + k] = (double)ts[u];
                const double a = ta[u], b = tb[u];
                vs += a * a + b * b;
                const double xf = (x - xc) / hx, yf = (y - yc) / hy;
                for (int q = 0; q < P.nbg; ++q)
                    Xc[(size_t)(P.nc + q) * P.npixp + k] = ipowd(xf, P.bpi[q]) * ipowd(yf, P.bpj[q]);
            } else if (k < P.npixp) {
                for (int q = 0; q < P.nX; ++q) Xc[(size_t)q * P.npixp + k] = 0.0;
            }
        }
    }
    vs = block_sum_waves<HV_THREADS / 64>(vs, red);
    if (tid == 0 && part == 0) {
        vbar[cell] = vs / P.npix;
        double fx = (cc.x - xc) / hx, fy = (cc.y - yc) / hy;
        // (the spatial terms of the substamp this one replaces stay available to the fused normal-matrix update)
        // (one thread: the exponent tables sit in the kernel-argument segment - indexed by a lane they are copied
        // to scratch by every thread of the launch, and the first round took 365 us instead of 220)
        // (plain loops of one thread: vectorised and interleaved by the compiler they cost the batched kernel its
        // last registers - k_hp_vectors_b<10, 11> spilled 3 and 6 of them to scratch memory)
        if (phiold) {
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
            for (int p = 0; p < P.nkp; ++p) phiold[(size_t)cell * P.nkp + p] = phi[(size_t)cell * P.nkp + p];
        }
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
        for (int p = 0; p < P.nkp; ++p)
            phi[(size_t)cell * P.nkp + p] = ipowd(fx, P.kpi[p]) * ipowd(fy, P.kpj[p]);
    }
    __syncthreads();
    if (only_special) continue;
    const int nstrip = (sw + HV_R - 1) / HV_R;
    // basis vectors: for each x filter, one x pass, then a y pass per term using it.
    // Both passes slide a register window: HV_R outputs share HV_R + 2 HWK loads.
    // term 0 (subtracted from the later terms, P.tsub0) is built by every part for itself;
    // only the owner of its x filter stores it
    int fidx = 0;
    for (int f = 0; f < P.nf1; ++f) {
        // (the per-filter term ranges come from the plan: scanning the term table here costs a
        // scalar load and its latency per entry, 15 x 49 of them per workgroup)
        const int tn0 = P.tf0[f], tn1 = tn0 + P.tfn[f];
        if (tn1 == tn0) continue;
        const bool mine = (fidx % fparts) == fpart;
        ++fidx;
        const bool for_w0 = (P.tfx[0] == f);
        if (!mine && !for_w0) continue;
        const double* fxv = filt + f * STEP;            // uniform address: scalar loads, no LDS traffic
        // (one chunk - a loop of constant trip count 1 over constants the compiler folds: the resident form keeps its
        // 124 registers and two workgroups per CU - unless BIG.  Not a lambda: a closure takes the plan's address,
        // and the batched kernel then keeps a copy of the plan in scratch - 3.5 KB per lane, the pool 2 x slower)
        const int nchunks = BIG ? (sw + cw - 1) / cw : 1;
        for (int ch = 0; ch < nchunks; ++ch) {
        const int c0 = BIG ? ch * cw : 0;
        const int cwe = BIG ? min(cw, sw - c0) : sw;
        const int nstripc = BIG ? (cwe + HV_R - 1) / HV_R : nstrip;
        // x pass: xp[yy][j] = sum_m fx[2 HWK - m] patch[yy][j + m]
        for (int e = tid; e < pw * nstripc; e += HV_THREADS) {
            const int yy = e / nstripc, jl0 = (e - yy * nstripc) * HV_R, j0 = c0 + jl0;
            const float* pr = patch + yy * pp + j0;
            double wv[WIN];
#pragma unroll
            for (int k = 0; k < WIN; ++k) wv[k] = (double)pr[k];      // j0 + k < pw + HV_R: zeros beyond pw
            double acc[HV_R];
#pragma unroll
            for (int q = 0; q < HV_R; ++q) acc[q] = 0.0;
#pragma unroll
            for (int m = 0; m < STEP; ++m) {
                const double cf = fxv[2 * HWK - m];
#pragma unroll
                for (int q = 0; q < HV_R; ++q) acc[q] += cf * wv[q + m];
            }
#pragma unroll
            for (int q = 0; q < HV_R; ++q)
                if (j0 + q < sw) xp[yy * cw + jl0 + q] = acc[q];
        }
        __syncthreads();
        for (int n = tn0; n < tn1; ++n) {
            if (!mine && n != 0) continue;
            const double* fyv = filt + P.tfy[n] * STEP;
            const double sc = P.tscale[n];
            const bool sub0 = n != 0 && P.tsub0[n] != 0;
            // y pass: W[i][j] = sum_m fy[2 HWK - m] xp[i + m][j]
            for (int e = tid; e < cwe * nstrip; e += HV_THREADS) {
                const int s = e / cwe, jl = e - s * cwe, j = c0 + jl;      // consecutive lanes = consecutive columns
                const int i0 = s * HV_R;
                const double* col = xp + i0 * cw + jl;
                double wv[WIN];
#pragma unroll
                for (int k = 0; k < WIN; ++k) wv[k] = col[k * cw];        // rows >= pw are zero
                double acc[HV_R];
#pragma unroll
                for (int q = 0; q < HV_R; ++q) acc[q] = 0.0;
#pragma unroll
                for (int m = 0; m < STEP; ++m) {
                    const double cf = fyv[2 * HWK - m];
#pragma unroll
                    for (int q = 0; q < HV_R; ++q) acc[q] += cf * wv[q + m];
                }
                // term 0's vector is read for all eight outputs at once, outside any per-pixel
                // condition (rows beyond the stamp read a clamped, unused entry)
                double wsub[HV_R];
#pragma unroll
                for (int q = 0; q < HV_R; ++q) wsub[q] = 0.0;
                if (sub0) {
#pragma unroll
                    for (int q = 0; q < HV_R; ++q) wsub[q] = w0[min(i0 + q, sw - 1) * sw + j];
                }
#pragma unroll
                for (int q = 0; q < HV_R; ++q) {
                    const int i = i0 + q;
                    const int k = i * sw + j;
                    const double v = acc[q] * sc - wsub[q];
                    if (i < sw) {
                        if (n == 0) w0[k] = v;
                        if (mine) Xc[(size_t)n * P.npixp + k] = v;
                    }
                }
            }
            if (n == 0) __syncthreads();
        }
        __syncthreads();
        }   // column chunks
    }
    }   // cells
}

template <int HWK>
__global__ __launch_bounds__(HV_THREADS) void k_hp_vectors(const hp_plan P, const float* __restrict__ sci,
                                                    const float* __restrict__ ref, const float* __restrict__ srms,
                                                    const float* __restrict__ trms, const double* __restrict__ filt,
                                                    const int2* __restrict__ centres, const int* __restrict__ active,
                                                    const int* __restrict__ need, double* __restrict__ X,
                                                    double* __restrict__ phi, double* __restrict__ vbar,
                                                    const int* __restrict__ guard, double* __restrict__ phiold,
                                                    const int* __restrict__ list, int special) {
    hp_vectors_body<HWK>(P, sci, ref, srms, trms, filt, centres, active, need, X, phi, vbar, guard, phiold, list, special);
}
template <int HWK>
__global__ __launch_bounds__(HV_THREADS) void k_hp_vectors_big(const hp_plan P, const float* __restrict__ sci,
                                                    const float* __restrict__ ref, const float* __restrict__ srms,
                                                    const float* __restrict__ trms, const double* __restrict__ filt,
                                                    const int2* __restrict__ centres, const int* __restrict__ active,
                                                    const int* __restrict__ need, double* __restrict__ X,
                                                    double* __restrict__ phi, double* __restrict__ vbar,
                                                    const int* __restrict__ guard, double* __restrict__ phiold,
                                                    const int* __restrict__ list, int special, int cw, double* __restrict__ w0g) {
    hp_vectors_body<HWK, true>(P, sci, ref, srms, trms, filt, centres, active, need, X, phi, vbar, guard, phiold, list, special,
                               cw, w0g);
}
// (four waves per SIMD - two workgroups per CU - like the one-job kernel: without the bound the job table's pointers
// push this instance to 131 registers, one workgroup per CU, and the first round of a batch ran 40 % slower per job;
// half widths above 11 need more than 128 registers in the one-job kernel too)
template <int HWK>
__global__ __launch_bounds__(HV_THREADS, (HWK <= 11 ? 4 : 2)) void k_hp_vectors_b(const hp_plan P, const hp_job* __restrict__ jobs,
                                                      const double* __restrict__ filt, int round) {
    const hp_job& J = jobs[blockIdx.z];
    hp_vectors_body<HWK>(P, J.sci, J.ref, J.srms, J.trms, filt, J.centres, J.active, J.need, J.X, J.phi, J.vbar,
                         HPJ_GUARD(J, round), J.phiold, round > 1 ? J.needlist : nullptr, round > 1 ? 1 : 0);
}


int zm_hp_launch_vectors(zm_ctx* ctx, hipStream_t st, const hp_plan& P, const hv_cfg& hvc, int rounds, int ncl_grid, int hv_gx,
                         const float* sci, const float* ref, const float* sci_rms, const float* ref_rms, const double* d_filt,
                         const int2* centres, const int* active, const int* need, double* X, double* phi, double* vbar,
                         const int* guard, double* phiold, const int* needlist, double* hv_w0g) {
    const size_t vsh = hvc.shmem;
    const dim3 grid_s(rounds == 1 ? P.ncell : ncl_grid, rounds == 1 ? HV_SPLIT_ALL : HV_SPLIT_FEW);
    const dim3 grid_b(rounds == 1 ? hv_gx : std::min(ncl_grid, hv_gx), rounds == 1 ? HV_SPLIT_ALL : HV_SPLIT_FEW);
    const int* list = rounds == 1 ? nullptr : needlist;
    const int special = rounds == 1 ? 0 : 1;
    if (hvc.big) {
#define C(H) case H: \
    ZM_HIP(hipFuncSetAttribute((const void*)k_hp_vectors_big<H>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)vsh)); \
    hipLaunchKernelGGL(k_hp_vectors_big<H>, grid_b, dim3(HV_THREADS), vsh, st, P, sci, ref, sci_rms, ref_rms, d_filt, \
                       centres, active, need, X, phi, vbar, guard, phiold, list, special, hvc.cw, hv_w0g); break;
        switch (P.hwk) {
            C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16) C(17) C(18) C(19) C(20)
            default: zm_set_error("zm_subtract: unsupported kernel half width %d", P.hwk); return 2;
        }
#undef C
    } else {
#define C(H) case H: \
    ZM_HIP(hipFuncSetAttribute((const void*)k_hp_vectors<H>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)vsh)); \
    hipLaunchKernelGGL(k_hp_vectors<H>, grid_s, dim3(HV_THREADS), vsh, st, P, sci, ref, sci_rms, ref_rms, d_filt, \
                       centres, active, need, X, phi, vbar, guard, phiold, list, special); break;
        switch (P.hwk) {
            C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16) C(17) C(18) C(19) C(20)
            default: zm_set_error("zm_subtract: unsupported kernel half width %d", P.hwk); return 2;
        }
#undef C
    }
    ZM_HIP(hipGetLastError());
    return 0;
}

// the first fit kernel of a batch (zm_subtract_batch_dev): half widths 1 .. 15 (a batch of wider kernels is taken apart
// by the pool and run job by job)
int zm_hp_launch_vectors_b(zm_ctx* ctx, hipStream_t st, const hp_plan& P, size_t vsh, unsigned gcells, unsigned NJ,
                           const hp_job* d_tab, const double* d_filt, int round) {
#define C(H) case H: \
    ZM_HIP(hipFuncSetAttribute((const void*)k_hp_vectors_b<H>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)vsh)); \
    hipLaunchKernelGGL(k_hp_vectors_b<H>, dim3(gcells, round == 1 ? HV_SPLIT_ALL : HV_SPLIT_FEW, NJ), dim3(HV_THREADS), vsh, st, \
                       P, d_tab, d_filt, round); break;
    switch (P.hwk) {
        C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15)
        default: zm_set_error("zm_subtract: unsupported kernel half width %d", P.hwk); return 2;
    }
#undef C
    return 0;
}
