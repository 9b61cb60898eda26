// Shared by the translation units of the subtraction (hotpants.hip: masks, stamp search, Gram / normal matrix,
// Cholesky forms, rejection, host; hp_vectors.hip: the basis convolutions of the substamps; hp_apply.hip: the
// spatially varying convolution): the plan of a fit, block sums, the job table of a batched fit and the launchers
// that cross translation units.  Operator: zuds/hotpants.py:77-93; algorithm: oracle/hotpants.py.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>

#include "zm_internal.h"

#define HP_MAXX 64        // rows of the Gram tile (nc + nbg + 1 <= 64)
#define HP_MAXPOLY 28     // (ko + 1)(ko + 2) / 2 for ko <= 6
#define HP_MAXNSS 8
#define HP_MAXREG 64
#define CF_BAR_STRIDE 32                // one k_chol_fused barrier counter per region, 128 B apart
#define HP_MAXF1 32       // distinct 1-D filters
#define HP_RIDGE 1e-10

struct hp_plan {
    int nx, ny, hwk, hwss, hw, step, sw, npix, npixp, pw;   // sw = 2 hwss + 1, pw = 2 hw + 1
    int nc, nbg, nE, nX, nkp, nunk, ko, bgo;
    int nrx, nry, nsx, nsy, nss, nreg, ncellr, ncell, nf1;
    int normalize;
    double tu, tl, iu, il, ft, ks;
    float fi, fin;
    int rx0[HP_MAXREG], rx1[HP_MAXREG], ry0[HP_MAXREG], ry1[HP_MAXREG];
    // basis term tables
    int tfx[HP_MAXX], tfy[HP_MAXX], tsub0[HP_MAXX];
    double tscale[HP_MAXX];
    int kpi[HP_MAXPOLY], kpj[HP_MAXPOLY];   // kernel spatial terms x^i y^j
    int bpi[16], bpj[16];                   // background terms
    int ngauss, gdeg[4], gbase[4], gterm0[4];   // per Gaussian: degree, first 1-D filter, first term
    int tf0[HP_MAXF1], tfn[HP_MAXF1];           // terms whose x filter is f: tf0[f] .. tf0[f] + tfn[f] - 1 (consecutive)
};


// ---------------------------------------------------------------------------
__device__ inline double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// the same sum for NW waves (a power of two): pairwise - waves that contribute 0 leave the bits alone
template <int NW>
__device__ inline double block_sum_waves(double v, double* red) {
    v = wave_sum_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    // (pairwise, block_sum256's grouping for the first four waves)
    double t[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) t[w] = red[w];
#pragma unroll
    for (int step = 1; step < NW; step *= 2)
#pragma unroll
        for (int w = 0; w + step < NW; w += 2 * step) t[w] += t[w + step];
    return t[0];
}

__device__ inline double block_sum256(double v, double* red) {
    v = wave_sum_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// x^n for the small non-negative integer exponents of the spatial polynomials: a handful of
// multiplications instead of the ~200 instructions of a general fp64 pow()
__device__ inline double ipowd(double x, int n) {
    double r = 1.0;
    for (int k = 0; k < n; ++k) r *= x;
    return r;
}

#define HV_R 8     // outputs per thread along the filter direction (register sliding window)
// Workgroups per cell (blockIdx.y): the x filters in use are dealt round-robin, each
// workgroup runs its x passes and the y passes of the terms built on them.  After the first
// round only the few cells with a replaced substamp are recomputed, so a cell's latency,
// not the throughput, sets the kernel time: the first round runs HV_SPLIT_ALL parts per cell
// (less of term 0 rebuilt), the later ones HV_SPLIT_FEW.
#define HV_SPLIT_ALL 5
#define HV_SPLIT_FEW 16   // later rounds: part 0 = the science / background rows and the spatial terms, 15 parts of filters

// Round 4: 512 threads per workgroup.  A pass has 343 (y) or 483 (x) work items of eight outputs: with 256 threads
// it ran as two rounds, the second a third full, and after the first rejection round the latency of one cell's
// passes is the kernel time (38 us); the LDS footprint (70 KB) allows two workgroups per CU either way, so 512
// threads also double the waves that cover each other's LDS reads in the first round.
#define HV_THREADS 512
#define HP_MAX_HWK 20                    // kernel half widths 1 .. 20 (41 x 41 taps), substamp half widths 1 .. 60: SEEING up to
#define HP_MAX_HWSS 60                   // 8 px at hotpants' -r 2.5 SEEING -rss 6 SEEING (zuds/hotpants.py:42-44)
// LDS plan of k_hp_vectors for a substamp geometry: resident form where everything fits, else term 0 in global memory
// and, if still too large, the x-filtered patch in column chunks
struct hv_cfg { bool big; bool w0_global; int cw; size_t shmem; };
static inline hv_cfg hp_vectors_cfg(int pw, int sw, int npix) {
    const size_t lim = 160 * 1024, patch = sizeof(float) * (size_t)pw * (pw + 8) + 16;   // (HV_R = 8)
    const size_t fast = sizeof(double) * ((size_t)(pw + 8) * sw + npix + 8) + patch;
    if (fast <= lim) return {false, false, sw, fast};
    const size_t full = sizeof(double) * ((size_t)(pw + 8) * sw + 8) + patch;
    if (full <= lim) return {true, true, sw, full};
    if (patch + sizeof(double) * 8 + sizeof(double) * (size_t)(pw + 8) * 8 > lim) return {true, true, 0, 0};   // (does not fit at all)
    int cw = (int)((lim - patch - sizeof(double) * 8) / (sizeof(double) * (size_t)(pw + 8)));
    cw &= ~7;
    return {true, true, cw, sizeof(double) * ((size_t)(pw + 8) * cw + 8) + patch};
}

// The kernels of the fit, twice: for ONE subtraction (the arguments are that job's buffers), and for a BATCH of
// subtractions in one launch (`*_b`: one more grid dimension picks the job, whose buffers come from a table in device
// memory, read through the scalar cache; its guard is the job's own round flag, so a job that has converged costs
// empty workgroups while the others go on).  Both forms inline the same body: the same bits per job.
struct hp_job {                      // one job of a batched fit: its planes and its slice of the batch's scratch
    const float *sci, *ref, *srms, *trms;
    int2* centres;
    int *active, *need, *needlist, *chg, *ibuf, *rflags;
    double *X, *G, *Gp, *Gold, *phi, *phiold, *vbar, *A, *AT, *rhs, *A0, *rhs0, *dsc, *merit, *stats;
    unsigned long long* smask;
};
#define HPJ_NREJ(J) ((J).ibuf)
#define HPJ_NTOTAL(J) ((J).ibuf + HP_MAXREG)
#define HPJ_FAIL(J) ((J).ibuf + 2 * HP_MAXREG)
#define HPJ_NMASKED(J) ((J).ibuf + 3 * HP_MAXREG)
#define HPJ_TMO(J) ((J).ibuf + 3 * HP_MAXREG + 4)
#define HPJ_GUARD(J, round) ((round) > 1 ? (J).rflags + ((round) - 1) : nullptr)
// A region whose last rejection changed nothing keeps its normal matrix, hence its factor and its solution: from the
// second round on the scaling, the factorisation and the back substitution of such a region are skipped (the list of
// changed cells of a region, k_hp_reject*: chg[ncell + reg (ncellr + 1)] is its length) - the same bits, fewer
// workgroups holding a CU each while other jobs' kernels wait
#define HPJ_REGION_IDLE(J, round, reg, ncell, ncellr) ((round) > 1 && (J).chg[(ncell) + (reg) * ((ncellr) + 1)] == 0)

// ---- launchers across translation units (each switches over the kernel half width its instances are compiled for) ---
// hp_vectors.hip
int zm_hp_launch_vectors(zm_ctx* ctx, hipStream_t st, const hp_plan& P, const hv_cfg& hvc, int rounds, int ncl_grid, int hv_gx,
                         const float* sci, const float* ref, const float* sci_rms, const float* ref_rms, const double* filt,
                         const int2* centres, const int* active, const int* need, double* X, double* phi, double* vbar,
                         const int* guard, double* phiold, const int* needlist, double* hv_w0g);
int zm_hp_launch_vectors_b(zm_ctx* ctx, hipStream_t st, const hp_plan& P, size_t vsh, unsigned gcells, unsigned njobs,
                           const hp_job* d_tab, const double* filt, int round);
// hp_apply.hip
int zm_hp_launch_apply(zm_ctx* ctx, const hp_plan& P, const unsigned long long* solved_mask, const float* sci,
                       const float* ref, const float* srms, const float* trms, const uint8_t* outbad,
                       const double* filt, const double* xsol, float* diff, float* noise, int* nmasked);
