// Mesh background and background-RMS maps on gfx950.
//
// Replaces SWarp's SUBTRACT_BACK stage (zuds/astromatic/makecoadd/default.swarp:
// 77-88, -BACK_SIZE 128 zuds/swarp.py:69) and the SExtractor runs that produce
// the -BACKGROUND / BACKGROUND_RMS / BACKGROUND check-images
// (zuds/sextractor.py:21-26,74,110-150; callers zuds/hotpants.py:28,
// zuds/image.py:206).  Algorithm and conventions: oracle/background.py.
//
// k_mesh_stats   one workgroup per mesh: 2-sigma pre-clip (fp64 wave reductions),
//                quantised histogram in LDS (integer atomics), integer prefix
//                sums, then the iterated +-3 sigma clip and the two-pointer
//                median walk of `backguess` evaluated exactly from the prefix
//                arrays (merge-path search), all in LDS.
// k_mesh_filter  one workgroup: bad-mesh fill, 3x3 median filter, global medians,
//                natural-spline second derivatives (4 node planes per map).
// k_bk_expand    per pixel bicubic-spline evaluation (also fused into k_prep).
#include <algorithm>

#include "zm_internal.h"

#define BK_BIG 1e30f
#define BK_NLEVELS 4096
#define BK_THREADS 256

__device__ inline double block_sum(double v, double* red) {
    // 256 threads = 4 waves; deterministic tree
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

#define BK_PER (BK_NLEVELS / BK_THREADS)   // 16 histogram bins per thread

// 21 KB per workgroup, so several meshes share a CU: only the count prefix is
// kept per bin; the first and second moment prefixes are kept per 16-bin block
// and completed on demand from the counts.
struct mesh_lds {
    int histo[BK_NLEVELS];          // counts, then inclusive prefix P0
    long long b1[BK_THREADS];       // sum of h * i over the bins before block t
    long long b2[BK_THREADS];       // sum of h * i * i over the bins before block t
    double red[4];
    long long wsum[3][4];
};

// Histogram quantisation exactly as SExtractor's backstat / backhisto: the mesh
// mean and sigma are rounded to float, qscale / qzero / cste are floats, the bin
// is (int)(pix / qscale + cste) in float arithmetic (truncation toward zero).
struct bk_quant {
    float qscale, qzero, cste;
    int nlevels;
};

__device__ inline bk_quant make_quant(double mean, double sig, double npix) {
    bk_quant q;
    const float mean32 = (float)mean, sig32 = (float)sig;
    int nl = (int)(0.7978845608028654 * 5.0 / 4.0 * npix + 1.0);
    q.nlevels = nl > BK_NLEVELS ? BK_NLEVELS : nl;
    q.qscale = sig32 > 0.f ? (float)(2.0 * 5.0 * (double)sig32 / q.nlevels) : 1.0f;
    q.qzero = (float)((double)mean32 - 5.0 * (double)sig32);
    q.cste = (float)(0.499999 - (double)(q.qzero / q.qscale));
    return q;
}

// Iterated +-3 sigma clip of `backguess` on the prefix arrays (exact integers).
// Run by one whole wave: the two-pointer median walk is replaced by its
// merge-path equivalent, searched 64 candidates at a time.
template <typename PT>
__device__ inline void backguess_wave(const PT* __restrict__ P0, const long long* __restrict__ B1,
                                      const long long* __restrict__ B2, const bk_quant q,
                                      double mean0, float* ob, float* os) {
    const int lane = threadIdx.x & 63;
    auto p0 = [&](int i) -> long long { return i < 0 ? 0 : (long long)P0[i]; };
    const int nlm1 = q.nlevels - 1;
    if (p0(nlm1) == 0) {
        if (lane == 0) { *ob = -BK_BIG; *os = -BK_BIG; }
        return;
    }
    int lcut = 0, hcut = nlm1;
    double sg = 10.0 * nlm1, sg1 = 1.0, mea = mean0, med = mean0;
    for (int n = 100; n-- && sg >= 0.1 && fabs(sg / sg1 - 1.0) > 1e-4;) {
        sg1 = sg;
        // moments over [lcut, hcut]: lanes 0-15 complete the prefix at hcut, lanes
        // 16-31 the prefix at lcut - 1, from the per-block sums and the counts
        long long m1 = 0, m2 = 0;
        {
            const int grp = lane >> 4, l16 = lane & 15;
            const int x = grp == 0 ? hcut : lcut - 1;
            long long c1 = 0, c2 = 0;
            if (grp < 2 && x >= 0) {
                const int t = x >> 4, i = 16 * t + l16;
                if (i <= x) {
                    long long hh = p0(i) - p0(i - 1);
                    c1 = hh * i;
                    c2 = c1 * i;
                }
                if (l16 == 0) { c1 += B1[t]; c2 += B2[t]; }
            }
#pragma unroll
            for (int o = 8; o >= 1; o >>= 1) { c1 += __shfl_xor(c1, o); c2 += __shfl_xor(c2, o); }
            m1 = __shfl(c1, 0) - __shfl(c1, 16);
            m2 = __shfl(c2, 0) - __shfl(c2, 16);
        }
        const long long sum = p0(hcut) - p0(lcut - 1);
        mea = (double)m1;
        sg = (double)m2;
        // largest a in [0, T] with a == 0 or L(a - 1) < H(T - a)
        const int T = hcut - lcut + 1;
        const long long base_lo = p0(lcut - 1), top = p0(hcut);
        int lo = 0, hi = T;
        while (hi > lo) {
            const int span = hi - lo;
            const int step = (span + 63) >> 6;
            const int a = lo + (lane + 1) * step;
            bool ok = false;
            if (a <= hi) {
                long long La = p0(lcut + a - 2) - base_lo;
                long long Hb = top - p0(hcut - (T - a));
                ok = La < Hb;
            }
            const int k = __popcll(__ballot(ok));      // predicate is monotone: k leading trues
            const int nlo = lo + k * step;
            const int nhi = lo + (k + 1) * step - 1;
            lo = nlo < hi ? nlo : hi;
            hi = nhi < hi ? nhi : hi;
            if (lo > hi) hi = lo;
        }
        const int a = lo, b = T - a;
        const long long lowsum = p0(lcut + a - 1) - base_lo;
        const long long highsum = top - p0(hcut - b);
        const int ihigh = hcut - b, ilow = lcut + a;
        if (ihigh >= 0) {
            long long ha = p0(ilow) - p0(ilow - 1), hb = p0(ihigh) - p0(ihigh - 1);
            double den = 2.0 * (double)(ha > hb ? ha : hb);
            med = ihigh + 0.5 + (den > 0 ? (double)(highsum - lowsum) / den : 0.0);
        } else {
            med = 0.0;
        }
        if (sum) {
            mea /= (double)sum;
            sg = sg / (double)sum - mea * mea;
        }
        sg = sg > 0.0 ? sqrt(sg) : 0.0;
        double ft = med - 3.0 * sg;
        lcut = ft > 0.0 ? (int)(ft + 0.5) : 0;
        ft = med + 3.0 * sg;
        hcut = ft < nlm1 ? (ft > 0.0 ? (int)(ft + 0.5) : (int)(ft - 0.5)) : nlm1;
    }
    const double qz = q.qzero, qs = q.qscale;
    double modev;
    if (sg > 0.0)
        modev = fabs((mea - med) / sg) < 0.3 ? qz + (2.5 * med - 1.5 * mea) * qs : qz + med * qs;
    else
        modev = qz + mea * qs;
    if (lane == 0) { *ob = (float)modev; *os = (float)(sg * qs); }
}

// In-place inclusive prefix of the counts, plus the exclusive block sums of
// h * i and h * i * i (one 16-bin block per thread).
__device__ inline void histo_prefix(mesh_lds* S) {
    const int tid = threadIdx.x;
    long long a0 = 0, a1 = 0, a2 = 0;
    int hloc[BK_PER];
#pragma unroll
    for (int k = 0; k < BK_PER; ++k) {
        const int i = tid * BK_PER + k;
        const int hh = S->histo[i];
        hloc[k] = hh;
        a0 += hh;
        a1 += (long long)hh * i;
        a2 += (long long)hh * i * i;
    }
    long long e0 = a0, e1 = a1, e2 = a2;
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        long long t0 = __shfl_up(e0, o), t1 = __shfl_up(e1, o), t2 = __shfl_up(e2, o);
        if (lane >= o) { e0 += t0; e1 += t1; e2 += t2; }
    }
    if (lane == 63) { S->wsum[0][wave] = e0; S->wsum[1][wave] = e1; S->wsum[2][wave] = e2; }
    __syncthreads();
    long long o0 = 0, o1 = 0, o2 = 0;
    for (int w = 0; w < wave; ++w) { o0 += S->wsum[0][w]; o1 += S->wsum[1][w]; o2 += S->wsum[2][w]; }
    S->b1[tid] = o1 + e1 - a1;
    S->b2[tid] = o2 + e2 - a2;
    int run = (int)(o0 + e0 - a0);
#pragma unroll
    for (int k = 0; k < BK_PER; ++k) {
        run += hloc[k];
        S->histo[tid * BK_PER + k] = run;
    }
    __syncthreads();
}

// Fast path: mesh area <= 16384 px.  1024 threads per mesh, 16 px per thread in
// registers (four 16-byte loads per plane), so HBM is read once and a mesh's
// latency chain is short.  blockIdx.z selects the statistic: mode0 + z, mode 0 =
// image, mode 1 = 1 / weight (variance level).
#ifndef BKF_THREADS
#define BKF_THREADS 512
#endif
#define BKF_WAVES (BKF_THREADS / 64)
#define BKF_PX (16384 / BKF_THREADS)   // pixels of a 128 x 128 mesh per thread
#define BKF_ROWS (BKF_THREADS / 32)   // mesh rows covered per load pass

struct meshf_lds {
    int histo[BK_NLEVELS];
    double red[BKF_WAVES];
    double red3[3][BKF_WAVES];
    long long wsum[3][4];
    // block sums with ONE barrier each (round 4): three slot sets used in rotation - a set is written again
    // only after two more barriers, when every wave has long read it
    double rot[3][3][BKF_WAVES];
};

// per-mesh hand-off from k_mesh_stats_fast to k_mesh_guess (global memory): the histogram itself; its
// prefix arrays are built by k_mesh_guess (one wave per mesh, every mesh of the batch in flight), not at
// the tail of the statistics workgroup, where half its waves idled and the CU slot was held meanwhile
struct mesh_dump {
    unsigned short h[BK_NLEVELS];    // counts (a fast-path mesh has at most 16384 pixels)
    bk_quant q;
    double mean0;
    int valid;
    int pad_;
};

__device__ inline double blockf_sum(double v, double* red) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < BKF_WAVES; ++w) t += red[w];     // fixed order: deterministic
    return t;
}

// three sums, one barrier: the waves' partial sums go into slot set `slot` (see meshf_lds::rot); fixed order:
// deterministic
__device__ inline void blockf_sum3_1b(double& a, double& b, double& c, double (*red)[BKF_WAVES]) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        a += __shfl_xor(a, o);
        b += __shfl_xor(b, o);
        c += __shfl_xor(c, o);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = a;
        red[1][threadIdx.x >> 6] = b;
        red[2][threadIdx.x >> 6] = c;
    }
    __syncthreads();
    a = b = c = 0.0;
#pragma unroll
    for (int w = 0; w < BKF_WAVES; ++w) { a += red[0][w]; b += red[1][w]; c += red[2][w]; }
}

// three sums in one pass of barriers; fixed order: deterministic
__device__ inline void blockf_sum3(double& a, double& b, double& c, double (*red)[BKF_WAVES]) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        a += __shfl_xor(a, o);
        b += __shfl_xor(b, o);
        c += __shfl_xor(c, o);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = a;
        red[1][threadIdx.x >> 6] = b;
        red[2][threadIdx.x >> 6] = c;
    }
    __syncthreads();
    a = b = c = 0.0;
#pragma unroll
    for (int w = 0; w < BKF_WAVES; ++w) { a += red[0][w]; b += red[1][w]; c += red[2][w]; }
}

// up to BK_BATCH equally sized frames per launch: blockIdx.z = frame * nmode + statistic.
// __launch_bounds__(512, 4): HIP's second argument is waves per SIMD - 4 = two workgroups per
// CU at <= 128 VGPRs (left alone the kernel takes 206 and one workgroup per CU: 2 waves per
// SIMD on average, 52 % of the wave cycles waiting).
#define BK_BATCH 64
struct bk_batch {
    const float* img[BK_BATCH];
    const float* wgt[BK_BATCH];
};

// The statistic of one mesh from its 32 px per thread (NaN = not a sample): pivot, two moment passes,
// histogram, prefix arrays -> *D.  Every branch is uniform over the workgroup.
// Round 4: half the barriers.  `khint`: the value of the mesh's first pixel when that pixel is a sample (every
// thread loaded it: a uniform pivot without a search - NaN otherwise); the block sums take one barrier each
// (rotating slots `rot0`, `rot0 + 1`), the histogram is zeroed before the first of them.
__device__ __forceinline__ void mesh_general(float (&v)[BKF_PX], const int area, meshf_lds* S,
                                             mesh_dump* __restrict__ D, const int dbg, const float khint,
                                             const int rot0) {
    const int tid = threadIdx.x;
    const float qnan = __builtin_nanf("");
    // ---- pivot: some valid pixel of the mesh.  Moments are accumulated about it, in fp64:
    // (x - K) is exact, a constant mesh gives exactly zero variance (as numpy's two-pass var
    // in the oracle does) and a nearly flat map loses nothing to cancellation, in one sweep
    // per clipping pass instead of two.
    float kf = khint;
    if (!(khint == khint)) {
#pragma unroll
    for (int k = BKF_PX - 1; k >= 0; --k) kf = (v[k] == v[k]) ? v[k] : kf;
    {
        const unsigned long long has = __ballot(kf == kf);
        const int src = has ? __ffsll((long long)has) - 1 : 0;
        kf = __shfl(kf, src);                                // first valid value of the wave (or NaN)
        __syncthreads();
        if ((tid & 63) == 0) S->red[tid >> 6] = (double)kf;
        __syncthreads();
        double kd = __builtin_nan("");
#pragma unroll
        for (int w8 = BKF_WAVES - 1; w8 >= 0; --w8) { const double c = S->red[w8]; kd = (c == c) ? c : kd; }
        kf = (float)kd;
    }
    }
    if (!(kf == kf)) {                                       // no valid pixel at all
        if (tid == 0) D->valid = 0;
        return;
    }
    const double K = (double)kf;
    // (the histogram is cleared here, ahead of the two barriers of the moment passes: nobody touches it before
    // the third.  Cleared between the passes it cost 25 registers and spills at this kernel's budget of 128.)
    for (int k = tid; k < BK_NLEVELS; k += BKF_THREADS) S->histo[k] = 0;
    // ---- pass 1: all valid pixels
    // branch-free: an excluded pixel is replaced by the pivot (d = 0 adds nothing, exactly) and
    // counted with integers, so a pixel costs one conversion and three fp64 operations
    double s0 = 0, s1 = 0, s2 = 0;
    {
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < BKF_PX; ++k) {
            const float x = v[k];
            const bool in = (x == x);
            const double d = (double)(in ? x : kf) - K;
            cnt += in ? 1 : 0;
            s1 += d;
            s2 = fma(d, d, s2);
        }
        s0 = (double)cnt;
    }
    blockf_sum3_1b(s0, s1, s2, S->rot[rot0 % 3]);
    if (s0 < area * 0.5 || s0 < 1.0) {   // BACK_MINGOODFRAC
        if (tid == 0) D->valid = 0;
        return;
    }
    double dm = s1 / s0;                                     // mean - K
    double var = s2 / s0 - dm * dm;
    double sig = var > 0 ? sqrt(var) : 0.0;
    const double lc = K + dm - 2.0 * sig, hc = K + dm + 2.0 * sig;
    // ---- pass 2: 2-sigma clipped
    s0 = s1 = s2 = 0;
    {
        // x >= lc (double) <=> x >= the smallest float >= lc; likewise for the upper cut: the
        // comparisons run in fp32 with the same outcome
        float lcf = (float)lc, hcf = (float)hc;
        if ((double)lcf < lc) lcf = nextafterf(lcf, INFINITY);
        if ((double)hcf > hc) hcf = nextafterf(hcf, -INFINITY);
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < BKF_PX; ++k) {
            const float x = v[k];
            const bool in = (x >= lcf) && (x <= hcf);          // false for NaN
            const double d = (double)(in ? x : kf) - K;
            cnt += in ? 1 : 0;
            s1 += d;
            s2 = fma(d, d, s2);
        }
        s0 = (double)cnt;
    }
    blockf_sum3_1b(s0, s1, s2, S->rot[(rot0 + 1) % 3]);
    if (s0 < 1.0) {
        if (tid == 0) D->valid = 0;
        return;
    }
    dm = s1 / s0;
    const double mean = K + dm;
    var = s2 / s0 - dm * dm;
    sig = var > 0 ? sqrt(var) : 0.0;
    const bk_quant q = make_quant(mean, sig, s0);
    if (dbg == 2) { if (sig == 12345.0) D->valid = 7; return; }
    if (sig == 0.0) {
        // a constant mesh (flat variance maps, most of the time): qscale = 1 and every pixel
        // falls into bin 0, so backguess returns exactly qzero = (float)mean and sigma 0 -
        // no histogram, no prefix arrays, no dump
        if (tid == 0) {
                        D->q = q;
            D->mean0 = (double)(float)mean;
            D->valid = 2;
        }
        return;
    }
    // ---- histogram.  bin = (int)(x / qscale + cste) with a correctly rounded quotient:
    // y = RN(1 / qscale), q0 = RN(x y), r = RN(x - q0 qscale), q = RN(q0 + r y) is the
    // correctly rounded x / qscale (Markstein) unless the significand of qscale is all
    // ones; three instructions instead of the ten of a division.
    {
        // one exec-masked LDS add per valid pixel, nothing else in the loop's control flow (a version that
        // merged runs of equal bins paid four branches per pixel for a case - flat maps - that no longer
        // comes here); the division that does not take the three-instruction form is a loop of its own
        const float qinv = 1.0f / q.qscale;
        const bool slow_div = (__float_as_uint(q.qscale) & 0x7fffffu) == 0x7fffffu;
        if (!slow_div) {
#pragma unroll
            for (int k = 0; k < BKF_PX; ++k) {
                const float x = v[k];
                const float q0 = x * qinv;
                const float qq = fmaf(fmaf(-q0, q.qscale, x), qinv, q0);
                const int b = (int)(qq + q.cste);
                if ((x == x) && b >= 0 && b < q.nlevels) atomicAdd(&S->histo[b], 1);
            }
        } else {
#pragma unroll
            for (int k = 0; k < BKF_PX; ++k) {
                const float x = v[k];
                const int b = (int)(x / q.qscale + q.cste);
                if ((x == x) && b >= 0 && b < q.nlevels) atomicAdd(&S->histo[b], 1);
            }
        }
    }
    __syncthreads();
    if (dbg == 3) { if (S->histo[tid] == -5) D->valid = 7; return; }
    // ---- the histogram goes out as 16-bit counts, 8 bins = one 16-byte store per thread; the prefix
    // arrays and the clip iterations are k_mesh_guess's
    {
        constexpr int NB = BK_NLEVELS / BKF_THREADS;       // bins per thread: 8 (one 16-byte store) or 4
        static_assert(NB == 8 || NB == 4, "the histogram hand-over packs 8 or 4 bins per thread");
        const int4 lo = *reinterpret_cast<const int4*>(&S->histo[NB * tid]);
        if (NB == 8) {
            const int4 hi = *reinterpret_cast<const int4*>(&S->histo[NB * tid + 4]);
            int4 o;
            o.x = (int)((unsigned)lo.x | ((unsigned)lo.y << 16));
            o.y = (int)((unsigned)lo.z | ((unsigned)lo.w << 16));
            o.z = (int)((unsigned)hi.x | ((unsigned)hi.y << 16));
            o.w = (int)((unsigned)hi.z | ((unsigned)hi.w << 16));
            reinterpret_cast<int4*>(D->h)[tid] = o;
        } else {
            int2 o;
            o.x = (int)((unsigned)lo.x | ((unsigned)lo.y << 16));
            o.y = (int)((unsigned)lo.z | ((unsigned)lo.w << 16));
            reinterpret_cast<int2*>(D->h)[tid] = o;
        }
        if (tid == 0) { D->q = q; D->mean0 = (double)(float)mean; D->valid = 1; }
    }
}

// SEL 0: image statistic; 1: variance statistic (1 / weight); 2: both from one read of the two planes
// (blockIdx.z = frame; the dumps of a frame are consecutive: image, then variance).  Most weight
// maps are flat inside a mesh: then every 1 / w is the same float, the clipped mean is that value
// exactly and sigma is 0 - the result of the general path - and a min / max / count of the weights,
// taken while the pixels are loaded, replaces the divisions and the moment passes; only a mesh
// whose weights vary reads them again (SEL 2) and takes the general path a second time.
#ifndef BKF_WPS
#define BKF_WPS 4                     // waves per SIMD the register budget is set for (4: two workgroups per CU)
#endif
template <int SEL>
__global__ __launch_bounds__(BKF_THREADS, BKF_WPS) void k_mesh_stats_fast(const bk_batch B, int nx, int ny, int mesh,
                                                                 int nbx, int nby, float wthresh, int vec_ok,
                                                                 int dbg, mesh_dump* __restrict__ dump) {
    extern __shared__ char smem_raw[];
    meshf_lds* S = reinterpret_cast<meshf_lds*>(smem_raw);
    constexpr int NM = SEL == 2 ? 2 : 1;
    const int frame = blockIdx.z;
    const float* __restrict__ img = B.img[frame];
    const float* __restrict__ wgt = B.wgt[frame];
    const int mi = blockIdx.x, mj = blockIdx.y;
    const int x0 = mi * mesh, y0 = mj * mesh;
    const int w = min(mesh, nx - x0), h = min(mesh, ny - y0);
    const int area = w * h;
    const int tid = threadIdx.x;
    const float qnan = __builtin_nanf("");
    mesh_dump* D0 = dump + ((size_t)(frame * NM) * nby + mj) * nbx + mi;
    mesh_dump* D1 = D0 + (size_t)(NM - 1) * nby * nbx;              // variance statistic

    // ---- load: BKF_PX / 4 passes of BKF_ROWS rows x 128 columns, 4 px per thread per pass
    // A pixel outside the mesh or the frame gets the weight -inf, a frame without weights the weight 1 and
    // the threshold -inf: the classification below is then two comparisons per pixel, with no terms for
    // the mesh edge or the missing plane (as separate conditions they were carried as 64-bit masks per pixel,
    // spilled to lanes of a VGPR at this kernel's register budget)
    float v[BKF_PX];
    float wmn = __builtin_inff(), wmx = -__builtin_inff();
    int wcnt = 0;
    const float pinf = __builtin_inff();
    const float thr = wgt ? wthresh : -pinf;
    // the pivot of the image statistic without a search: the mesh's first pixel, read by every thread (one
    // address: a broadcast out of the cache), when it is a sample
    float khint = qnan;
    if (SEL != 1) {
        const size_t i00 = (size_t)y0 * nx + x0;
        const float p0 = img[i00], w0 = wgt ? wgt[i00] : 1.f;
        khint = (w0 > thr && p0 > -BK_BIG) ? p0 : qnan;
    }
    const int c4 = (tid & 31) * 4, r32 = tid >> 5;
#pragma unroll
    for (int k = 0; k < BKF_PX / 4; ++k) {
        const int row = BKF_ROWS * k + r32;
        float pv[4] = {qnan, qnan, qnan, qnan};
        float pw[4] = {-pinf, -pinf, -pinf, -pinf};
        if (row < h && c4 < w) {
            const size_t idx = (size_t)(y0 + row) * nx + x0 + c4;
            if (vec_ok && c4 + 3 < w) {
                if (SEL != 1) {
                    float4 a = *reinterpret_cast<const float4*>(img + idx);
                    pv[0] = a.x; pv[1] = a.y; pv[2] = a.z; pv[3] = a.w;
                }
                if (wgt) {
                    float4 b = *reinterpret_cast<const float4*>(wgt + idx);
                    pw[0] = b.x; pw[1] = b.y; pw[2] = b.z; pw[3] = b.w;
                } else {
                    pw[0] = pw[1] = pw[2] = pw[3] = 1.f;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (c4 + j < w) {
                        if (SEL != 1) pv[j] = img[idx + j];
                        pw[j] = wgt ? wgt[idx + j] : 1.f;
                    }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x = pw[j];
            const bool good = x > thr;                                  // (false for NaN)
            if (SEL != 0) {
                const bool in = good && (x > -BK_BIG);
                wmn = fminf(wmn, in ? x : pinf);
                wmx = fmaxf(wmx, in ? x : -pinf);
                wcnt += in ? 1 : 0;
                if (SEL == 1) v[4 * k + j] = in ? x : qnan;          // inverted below, unless the mesh is flat
            }
            if (SEL != 1) {
                const float val = pv[j];
                v[4 * k + j] = (good && (val > -BK_BIG)) ? val : qnan;
            }
        }
    }
    if (dbg == 1) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < BKF_PX; ++k) a += v[k];
        if (a == 12345.f) dump[0].valid = 7;
        return;
    }
    bool flat = false;
    if (SEL != 0) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            wmn = fminf(wmn, __shfl_xor(wmn, o));
            wmx = fmaxf(wmx, __shfl_xor(wmx, o));
            wcnt += __shfl_xor(wcnt, o);
        }
        if ((tid & 63) == 0) { S->red3[0][tid >> 6] = wmn; S->red3[1][tid >> 6] = wmx; S->red3[2][tid >> 6] = wcnt; }
        __syncthreads();
        double bmn = S->red3[0][0], bmx = S->red3[1][0], bc = 0.0;
#pragma unroll
        for (int w8 = 0; w8 < BKF_WAVES; ++w8) {
            bmn = fmin(bmn, S->red3[0][w8]);
            bmx = fmax(bmx, S->red3[1][w8]);
            bc += S->red3[2][w8];
        }
        const float inv = 1.0f / (float)bmn;
        flat = bc >= 1.0 && bc >= area * 0.5 && bmn == bmx && inv > -BK_BIG && inv == inv;
        if (flat && tid == 0) {
            D1->q = make_quant((double)inv, 0.0, bc);
            D1->mean0 = (double)inv;
            D1->valid = 2;
        }
        // (no barrier here: the statistics below keep their partial sums in other slots)
    }
    if (SEL != 1) mesh_general(v, area, S, D0, dbg, khint, 0);
    if (SEL == 0 || flat) return;
    if (SEL == 2) {
        // the weights of this mesh vary: read them again (the registers held the image)
        __syncthreads();
#pragma unroll
        for (int k = 0; k < BKF_PX / 4; ++k) {
            const int row = BKF_ROWS * k + r32;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x = qnan;
                if (row < h && c4 + j < w) x = wgt[(size_t)(y0 + row) * nx + x0 + c4 + j];
                v[4 * k + j] = (x > wthresh && x > -BK_BIG) ? x : qnan;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < BKF_PX; ++k) {
        const float x = v[k];
        const float val = 1.0f / x;
        const bool ok = (x == x) && (val > -BK_BIG) && (val == val);
        v[k] = ok ? val : qnan;
    }
    __syncthreads();                                       // (the slots and the histogram of the first statistic are consumed)
    mesh_general(v, area, S, D1, dbg, qnan, 2);
}

// One wave per mesh: iterated clipping on the dumped prefix arrays (staged in LDS).
__global__ __launch_bounds__(64) void k_mesh_guess(const mesh_dump* __restrict__ dump, int n, int nmode,
                                                   int nslot, float* __restrict__ raw) {
    __shared__ __attribute__((aligned(16))) unsigned short P0[BK_NLEVELS];      // 16-bit: 8 KB of LDS per mesh
    __shared__ long long B1[BK_THREADS];
    __shared__ long long B2[BK_THREADS];
    const int m = blockIdx.x, z = blockIdx.y, lane = threadIdx.x;
    const int frame = z / nmode, mode = z - frame * nmode;
    const mesh_dump* D = dump + (size_t)z * n + m;
    float* ob = raw + (size_t)frame * 4 * nslot + (size_t)mode * 2 * n + m;   // raw: [frame: 4 nslot][statistic][2 maps][n]
    float* os = ob + n;
    if (!D->valid) {
        if (lane == 0) { *ob = -BK_BIG; *os = -BK_BIG; }
        return;
    }
    if (D->valid == 2) {                 // constant mesh: see k_mesh_stats_fast
        if (lane == 0) { *ob = D->q.qzero; *os = 0.f; }
        return;
    }
    // prefix arrays of the histogram, 512 bins per pass: lane l holds bins 512 k + 8 l .. + 7 (one 16-byte
    // piece, read coalesced), sums them locally, the wave scans the lane sums.  P0: inclusive prefix of the
    // counts; B1 / B2: exclusive sums of h i and h i i per block of 16 bins (two lanes), as backguess_wave reads them
    const int4* src = reinterpret_cast<const int4*>(D->h);
    int4 piece[BK_NLEVELS / 8 / 64];
#pragma unroll
    for (int k = 0; k < BK_NLEVELS / 8 / 64; ++k) piece[k] = src[k * 64 + lane];
    unsigned run0 = 0;
    long long run1 = 0, run2 = 0;
#pragma unroll
    for (int k = 0; k < BK_NLEVELS / 8 / 64; ++k) {
        const unsigned wv[4] = {(unsigned)piece[k].x, (unsigned)piece[k].y, (unsigned)piece[k].z, (unsigned)piece[k].w};
        unsigned c[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { c[2 * j] = wv[j] & 0xffffu; c[2 * j + 1] = wv[j] >> 16; }
        unsigned s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) { s0 += c[j]; s1 += c[j] * (unsigned)j; s2 += c[j] * (unsigned)(j * j); }
        const long long base = (long long)(512 * k + 8 * lane);
        const long long a1 = base * s0 + s1, a2 = base * base * s0 + 2 * base * s1 + s2;
        unsigned e0 = s0;
        long long e1 = a1, e2 = a2;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned t0 = __shfl_up(e0, o);
            const long long t1 = __shfl_up(e1, o), t2 = __shfl_up(e2, o);
            if (lane >= o) { e0 += t0; e1 += t1; e2 += t2; }
        }
        // inclusive prefixes of this lane's 8 bins
        unsigned p = run0 + e0 - s0;
        unsigned w16[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            p += c[2 * j];
            const unsigned lo16 = p;
            p += c[2 * j + 1];
            w16[j] = lo16 | (p << 16);
        }
        reinterpret_cast<int4*>(P0)[k * 64 + lane] = make_int4((int)w16[0], (int)w16[1], (int)w16[2], (int)w16[3]);
        if ((lane & 1) == 0) {                              // first lane of a 16-bin block: sums of everything before it
            B1[32 * k + (lane >> 1)] = run1 + e1 - a1;
            B2[32 * k + (lane >> 1)] = run2 + e2 - a2;
        }
        run0 += __shfl(e0, 63);
        run1 += __shfl(e1, 63);
        run2 += __shfl(e2, 63);
    }
    __syncthreads();
    backguess_wave(P0, B1, B2, D->q, D->mean0, ob, os);
}

// Generic path (any mesh size): three passes over global memory.
__global__ __launch_bounds__(BK_THREADS) void k_mesh_stats(const float* __restrict__ img,
                                                           const float* __restrict__ wgt,
                                                           int nx, int ny, int mesh, int nbx,
                                                           int nby, float wthresh, int mode0,
                                                           float* __restrict__ raw) {
    extern __shared__ char smem_raw[];
    mesh_lds* S = reinterpret_cast<mesh_lds*>(smem_raw);
    const int mode = mode0 + blockIdx.z;
    const int mi = blockIdx.x, mj = blockIdx.y;
    const int x0 = mi * mesh, y0 = mj * mesh;
    const int w = min(mesh, nx - x0), h = min(mesh, ny - y0);
    const int area = w * h;
    const int tid = threadIdx.x;
    float* ob = raw + (size_t)blockIdx.z * 2 * nbx * nby + (size_t)mj * nbx + mi;
    float* os = ob + (size_t)nbx * nby;

    auto value = [&](int k, bool* ok) -> float {
        int yy = k / w, xx = k - yy * w;
        size_t idx = (size_t)(y0 + yy) * nx + (x0 + xx);
        float v;
        bool good = true;
        if (mode == 0) {
            v = img[idx];
            if (wgt) good = wgt[idx] > wthresh;
        } else {
            float ww = wgt[idx];
            good = ww > wthresh;
            v = good ? 1.0f / ww : 0.f;
        }
        good = good && (v > -BK_BIG) && (v == v);
        *ok = good;
        return v;
    };

    double s0 = 0, s1 = 0, s2 = 0;
    for (int k = tid; k < area; k += BK_THREADS) {
        bool ok;
        float v = value(k, &ok);
        if (ok) { s0 += 1.0; s1 += v; }
    }
    s0 = block_sum(s0, S->red);
    s1 = block_sum(s1, S->red);
    if (s0 < area * 0.5 || s0 < 1.0) {   // BACK_MINGOODFRAC
        if (tid == 0) { *ob = -BK_BIG; *os = -BK_BIG; }
        return;
    }
    double mean = s1 / s0;
    for (int k = tid; k < area; k += BK_THREADS) {
        bool ok;
        float v = value(k, &ok);
        if (ok) { double dlt = (double)v - mean; s2 += dlt * dlt; }
    }
    s2 = block_sum(s2, S->red);
    double var = s2 / s0;
    double sig = var > 0 ? sqrt(var) : 0.0;
    const double lc = mean - 2.0 * sig, hc = mean + 2.0 * sig;
    s0 = s1 = s2 = 0;
    for (int k = tid; k < area; k += BK_THREADS) {
        bool ok;
        float v = value(k, &ok);
        if (ok && v >= lc && v <= hc) { s0 += 1.0; s1 += v; }
    }
    s0 = block_sum(s0, S->red);
    s1 = block_sum(s1, S->red);
    if (s0 < 1.0) {
        if (tid == 0) { *ob = -BK_BIG; *os = -BK_BIG; }
        return;
    }
    mean = s1 / s0;
    for (int k = tid; k < area; k += BK_THREADS) {
        bool ok;
        float v = value(k, &ok);
        if (ok && v >= lc && v <= hc) { double dlt = (double)v - mean; s2 += dlt * dlt; }
    }
    s2 = block_sum(s2, S->red);
    var = s2 / s0;
    sig = var > 0 ? sqrt(var) : 0.0;
    const bk_quant q = make_quant(mean, sig, s0);
    for (int k = tid; k < BK_NLEVELS; k += BK_THREADS) S->histo[k] = 0;
    __syncthreads();
    for (int k = tid; k < area; k += BK_THREADS) {
        bool ok;
        float v = value(k, &ok);
        if (ok) {
            int b = (int)(v / q.qscale + q.cste);
            if (b >= 0 && b < q.nlevels) atomicAdd(&S->histo[b], 1);
        }
    }
    __syncthreads();
    histo_prefix(S);
    if (tid < 64) backguess_wave(S->histo, S->b1, S->b2, q, (double)(float)mean, ob, os);
}

// ---------------------------------------------------------------------------
// Median of the window [j0, j1] x [i0, i1] of a staged map without a copy of the window: the rank of an element is the
// number of smaller ones plus the equal ones before it in window order (what an insertion sort's position is);
// the median is (n & 1) ? v[n / 2] : 0.5 (v[n / 2 - 1] + v[n / 2]) of the sorted window (SExtractor's back filter).
__device__ inline float window_median(const float* m, int nbx, int j0, int j1, int i0, int i1) {
    const int c = (j1 - j0 + 1) * (i1 - i0 + 1);
    const int r0 = (c - 1) >> 1, r1 = c >> 1;
    float lo = 0.f, hi = 0.f;
    int e = 0;
    for (int jj = j0; jj <= j1; ++jj)
        for (int ii = i0; ii <= i1; ++ii, ++e) {
            const float v = m[jj * nbx + ii];
            int rank = 0, f = 0;
            for (int pj = j0; pj <= j1; ++pj)
                for (int pi = i0; pi <= i1; ++pi, ++f) {
                    const float w = m[pj * nbx + pi];
                    rank += (w < v || (w == v && f < e)) ? 1 : 0;
                }
            if (rank == r0) lo = v;
            if (rank == r1) hi = v;
        }
    return (c & 1) ? lo : 0.5f * (lo + hi);
}

// 9-element sorting network (25 compare-exchanges) and a register pick
__device__ inline void cswap(float& x, float& y) {
    float lo = fminf(x, y), hi = fmaxf(x, y);
    x = lo; y = hi;
}
__device__ inline void sort9(float (&a)[9]) {
    cswap(a[0], a[1]); cswap(a[3], a[4]); cswap(a[6], a[7]);
    cswap(a[1], a[2]); cswap(a[4], a[5]); cswap(a[7], a[8]);
    cswap(a[0], a[1]); cswap(a[3], a[4]); cswap(a[6], a[7]);
    cswap(a[0], a[3]); cswap(a[3], a[6]); cswap(a[0], a[3]);
    cswap(a[1], a[4]); cswap(a[4], a[7]); cswap(a[1], a[4]);
    cswap(a[2], a[5]); cswap(a[5], a[8]); cswap(a[2], a[5]);
    cswap(a[1], a[3]); cswap(a[5], a[7]);
    cswap(a[2], a[6]); cswap(a[4], a[6]); cswap(a[2], a[4]);
    cswap(a[2], a[3]); cswap(a[5], a[6]);
}
__device__ inline float pick9(const float (&a)[9], int idx) {
    float r = a[0];
#pragma unroll
    for (int i = 1; i < 9; ++i) {
        r = (i == idx) ? a[i] : r;
        // (kept a chain of selects: without the barrier the compiler turns it back into a[idx] - the array then
        // lives in scratch memory, 96 bytes per thread in both instances of k_mesh_filter)
        asm volatile("" : "+v"(r));
    }
    return r;
}

// natural cubic spline second derivatives / 6 along a strided line (unit spacing);
// the recurrence state stays in registers, LDS only holds the per-node results
__device__ inline void spline_line(const float* a, float* d, float* u, int n, int stride) {
    if (n < 3) {
        for (int k = 0; k < n; ++k) d[k * stride] = 0.f;
        return;
    }
    float dp = 0.f, up = 0.f;
    float am = a[0], ac = a[stride];
    d[0] = 0.f;
    for (int y = 1; y < n - 1; ++y) {
        const float an = a[(y + 1) * stride];
        const float temp = -1.f / (dp + 4.f);
        up = temp * (up - 6.f * (an + am - 2.f * ac));
        dp = temp;
        d[y * stride] = dp;
        u[y * stride] = up;
        am = ac;
        ac = an;
    }
    float dn = 0.f;
    d[(n - 1) * stride] = 0.f;
    for (int y = n - 2; y >= 1; --y) {
        dn = d[y * stride] * dn + u[y * stride];
        d[y * stride] = dn * (1.f / 6.f);
    }
}

#define BK_MAXMESH 4096

// One workgroup per statistic (blockIdx.x = mode index).  raw: [mode][2][n] mode /
// sigma maps (may hold -BIG); nodes: [mode][2 maps][4 planes][n]; stats: [mode][2]
// = {median of the filtered mode map, median of the filtered sigma map}.
// Everything is staged in LDS: the spline recurrences are latency chains.
// FAST (n <= 1024 meshes, i.e. every ZTF frame at BACK_SIZE >= 96): the two global
// medians come from one bitonic sort of both maps (55 compare-exchange rounds instead
// of n^2 rank counting), and the six spline passes of the two maps run as two phases
// (all y lines, then all x lines) instead of six.
template <bool FAST>
__global__ __launch_bounds__(1024) void k_mesh_filter(const float* __restrict__ raw_all,
                                                            int nbx, int nby, int fsize, int nmode, int nslot,
                                                            float* __restrict__ nodes_all,
                                                            float* __restrict__ stats_all) {
    extern __shared__ float mf_smem[];
    const int n = nbx * nby, tid = threadIdx.x;
    // blockIdx.x = frame * nmode + statistic; per-frame strides hold two statistics
    const int frame = blockIdx.x / nmode, zm = blockIdx.x - frame * nmode;
    const float* raw = raw_all + (size_t)frame * 4 * nslot + (size_t)zm * 2 * n;
    float* nodes = nodes_all + (size_t)frame * 16 * nslot + (size_t)zm * 8 * n;
    float* stats = stats_all + ((size_t)frame * 2 + zm) * 2;
    float* sb0 = mf_smem;            // filled maps
    float* sb1 = sb0 + n;
    float* fb0 = sb1 + n;            // filtered maps
    float* fb1 = fb0 + n;
    float* tmp = fb1 + n;            // spline scratch (FAST: 4 n, one per concurrent line set)
    float* pl = tmp + (FAST ? 4 : 1) * n;   // 4 node planes of the current map (FAST: of both)
    __shared__ int ngood;
    __shared__ float med[4];
    if (tid == 0) ngood = 0;
    __syncthreads();
    for (int k = tid; k < n; k += 1024) {
        float b = raw[k];
        sb0[k] = b;
        sb1[k] = raw[n + k];
        if (b > -BK_BIG) atomicAdd(&ngood, 1);
    }
    __syncthreads();
    // 1. fill bad meshes from the nearest good ones (ties averaged)
    for (int k = tid; k < n; k += 1024) {
        float b = sb0[k], s = sb1[k];
        if (!(b > -BK_BIG)) {
            if (ngood == 0) { b = 0.f; s = 1.f; }
            else {
                int j = k / nbx, i = k - j * nbx;
                int dmin = 0x7fffffff, cnt = 0;
                float vb = 0.f, vs = 0.f;
                for (int q = 0; q < n; ++q) {
                    float rq = sb0[q];
                    if (!(rq > -BK_BIG)) continue;
                    int qj = q / nbx, qi = q - qj * nbx;
                    int d2 = (qi - i) * (qi - i) + (qj - j) * (qj - j);
                    if (d2 < dmin) { dmin = d2; vb = rq; vs = sb1[q]; cnt = 1; }
                    else if (d2 == dmin) { vb += rq; vs += sb1[q]; ++cnt; }
                }
                b = vb / cnt; s = vs / cnt;
            }
        }
        fb0[k] = b; fb1[k] = s;      // staged here, copied back below
    }
    __syncthreads();
    for (int k = tid; k < n; k += 1024) { sb0[k] = fb0[k]; sb1[k] = fb1[k]; }
    __syncthreads();
    // 2. fsize x fsize median filter (window clipped at the borders)
    const int hb = fsize / 2;
    for (int k = tid; k < n; k += 1024) {
        int j = k / nbx, i = k - j * nbx;
        if (fsize == 3) {
            // registers only: clipped cells carry +inf and sort to the end
            float a[9], b[9];
            int c = 0;
#pragma unroll
            for (int dj = -1; dj <= 1; ++dj)
#pragma unroll
                for (int di = -1; di <= 1; ++di) {
                    const int jj = j + dj, ii = i + di;
                    const bool in = jj >= 0 && jj < nby && ii >= 0 && ii < nbx;
                    const int q = in ? jj * nbx + ii : k;
                    a[(dj + 1) * 3 + di + 1] = in ? sb0[q] : __builtin_inff();
                    b[(dj + 1) * 3 + di + 1] = in ? sb1[q] : __builtin_inff();
                    c += in ? 1 : 0;
                }
            sort9(a);
            sort9(b);
            fb0[k] = 0.5f * (pick9(a, (c - 1) >> 1) + pick9(a, c >> 1));
            fb1[k] = 0.5f * (pick9(b, (c - 1) >> 1) + pick9(b, c >> 1));
        } else if (fsize > 1) {
            // any other window (5 x 5, 7 x 7): the two middle ranks by counting, straight from the staged maps - a
            // per-thread window of 2 x 49 floats was 416 bytes of scratch in every instance of this kernel (round 6)
            const int j0 = max(j - hb, 0), j1 = min(j + hb, nby - 1), i0 = max(i - hb, 0), i1 = min(i + hb, nbx - 1);
            fb0[k] = window_median(sb0, nbx, j0, j1, i0, i1);
            fb1[k] = window_median(sb1, nbx, j0, j1, i0, i1);
        } else {
            fb0[k] = sb0[k]; fb1[k] = sb1[k];
        }
    }
    __syncthreads();
    if (FAST) {
        // 3. global medians: ascending bitonic sort of both maps, padded to 1024 with +inf
        float* srt = pl;                                   // 2 x 1024, before pl is needed
        srt[tid] = tid < n ? fb0[tid] : __builtin_inff();
        srt[1024 + tid] = tid < n ? fb1[tid] : __builtin_inff();
        __syncthreads();
        float* my = srt + (tid >> 9) * 1024;
        const int i = tid & 511;
        for (int k = 2; k <= 1024; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                const int lo = 2 * j * (i / j) + (i % j), hi = lo + j;
                const float a = my[lo], b = my[hi];
                const bool up = (lo & k) == 0;
                if ((a > b) == up) { my[lo] = b; my[hi] = a; }
                __syncthreads();
            }
        if (tid == 0) {
            stats[0] = 0.5f * (srt[(n - 1) / 2] + srt[n / 2]);
            stats[1] = 0.5f * (srt[1024 + (n - 1) / 2] + srt[1024 + n / 2]);
        }
        __syncthreads();
        // 4. node planes of both maps: V, DY (along y per column), A (along x of V), B (along x of DY)
        for (int k = tid; k < n; k += 1024) { pl[k] = fb0[k]; pl[4 * n + k] = fb1[k]; }
        __syncthreads();
        for (int e = tid; e < 2 * nbx; e += 1024) {
            const int m = e / nbx, c = e - m * nbx;
            float* P = pl + m * 4 * n;
            spline_line(P + c, P + n + c, tmp + m * n + c, nby, nbx);
        }
        __syncthreads();
        for (int e = tid; e < 4 * nby; e += 1024) {
            const int m = e / (2 * nby), r = e - m * 2 * nby;
            const int which = r / nby, row = r - which * nby;
            float* P = pl + m * 4 * n;
            spline_line(P + which * n + row * nbx, P + (2 + which) * n + row * nbx,
                        tmp + (2 * m + which) * n + row * nbx, nbx, 1);
        }
        __syncthreads();
        for (int k = tid; k < 8 * n; k += 1024) nodes[k] = pl[k];
        return;
    }
    // 3. global medians by rank counting (exact)
    for (int m = 0; m < 2; ++m) {
        const float* fb = m ? fb1 : fb0;
        for (int k = tid; k < n; k += 1024) {
            float v = fb[k];
            int less = 0, eq = 0;
            for (int q = 0; q < n; ++q) { float o = fb[q]; less += o < v; eq += o == v; }
            int i1 = (n - 1) / 2, i2 = n / 2;
            if (i1 >= less && i1 < less + eq) med[2 * m] = v;
            if (i2 >= less && i2 < less + eq) med[2 * m + 1] = v;
        }
    }
    __syncthreads();
    if (tid == 0) {
        stats[0] = 0.5f * (med[0] + med[1]);
        stats[1] = 0.5f * (med[2] + med[3]);
    }
    // 4. node planes: V, DY (along y per column), A (along x of V), B (along x of DY)
    for (int m = 0; m < 2; ++m) {
        const float* fb = m ? fb1 : fb0;
        float* V = pl;
        float* DY = pl + n;
        float* A = pl + 2 * n;
        float* B = pl + 3 * n;
        __syncthreads();
        for (int k = tid; k < n; k += 1024) V[k] = fb[k];
        __syncthreads();
        for (int i = tid; i < nbx; i += 1024) spline_line(V + i, DY + i, tmp + i, nby, nbx);
        __syncthreads();
        for (int j = tid; j < nby; j += 1024) spline_line(V + j * nbx, A + j * nbx, tmp + j * nbx, nbx, 1);
        __syncthreads();
        for (int j = tid; j < nby; j += 1024) spline_line(DY + j * nbx, B + j * nbx, tmp + j * nbx, nbx, 1);
        __syncthreads();
        float* out = nodes + (size_t)m * 4 * n;
        for (int k = tid; k < 4 * n; k += 1024) out[k] = pl[k];
    }
}

// same evaluator as in resample.hip (kept in sync; 16-term tensor-product spline)
__device__ inline float bk_eval2(const float* __restrict__ bk, int nbx, int nby, float invmesh,
                                 int x, int y) {
    size_t pl = (size_t)nbx * nby;
    float ty = (y + 0.5f) * invmesh - 0.5f;
    float tx = (x + 0.5f) * invmesh - 0.5f;
    int j0 = 0, i0 = 0;
    float dy = 0.f, dx = 0.f;
    if (nby > 1) { j0 = min(max((int)floorf(ty), 0), nby - 2); dy = ty - j0; }
    if (nbx > 1) { i0 = min(max((int)floorf(tx), 0), nbx - 2); dx = tx - i0; }
    int j1 = nby > 1 ? j0 + 1 : j0, i1 = nbx > 1 ? i0 + 1 : i0;
    float dy1 = 1.f - dy, dx1 = 1.f - dx;
    float cdy = dy * dy * dy - dy, cdy1 = dy1 * dy1 * dy1 - dy1;
    float cdx = dx * dx * dx - dx, cdx1 = dx1 * dx1 * dx1 - dx1;
    const float* V = bk; const float* DY = bk + pl; const float* A = bk + 2 * pl; const float* B = bk + 3 * pl;
    int a00 = j0 * nbx + i0, a01 = j0 * nbx + i1, a10 = j1 * nbx + i0, a11 = j1 * nbx + i1;
    float r0 = dy1 * V[a00] + dy * V[a10] + cdy1 * DY[a00] + cdy * DY[a10];
    float r1 = dy1 * V[a01] + dy * V[a11] + cdy1 * DY[a01] + cdy * DY[a11];
    float e0 = dy1 * A[a00] + dy * A[a10] + cdy1 * B[a00] + cdy * B[a10];
    float e1 = dy1 * A[a01] + dy * A[a11] + cdy1 * B[a01] + cdy * B[a11];
    return dx1 * r0 + dx * r1 + cdx1 * e0 + cdx * e1;
}

__global__ __launch_bounds__(256) void k_bk_expand(const float* __restrict__ img,
                                                   const float* __restrict__ nodes, int nx,
                                                   int ny, int nbx, int nby, float invmesh,
                                                   float* __restrict__ out_bkg,
                                                   float* __restrict__ out_rms,
                                                   float* __restrict__ out_sub) {
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= nx) return;
    size_t idx = (size_t)y * nx + x;
    float b = bk_eval2(nodes, nbx, nby, invmesh, x, y);
    if (out_bkg) out_bkg[idx] = b;
    if (out_sub) out_sub[idx] = img[idx] - b;
    if (out_rms) out_rms[idx] = bk_eval2(nodes + (size_t)4 * nbx * nby, nbx, nby, invmesh, x, y);
}

__global__ void k_var_scale(const float* __restrict__ bstats, const float* __restrict__ vstats,
                            float* __restrict__ out) {
    float backsig = bstats[1], level = vstats[0];
    out[0] = (level > 0.f && backsig > 0.f) ? backsig * backsig / level : 1.f;
}

// stats: [frame][{backmean, backsig, varlevel, varsig}]; out: [frame][4]
__global__ void k_var_scale_batch(const float* __restrict__ stats, float* __restrict__ out, int nf) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nf) return;
    const float backsig = stats[4 * f + 1], level = stats[4 * f + 2];
    out[4 * f] = (level > 0.f && backsig > 0.f) ? backsig * backsig / level : 1.f;
}

// ---------------------------------------------------------------------------
// Background of one frame in two enqueue steps, so that a caller can put the
// (two-workgroup, latency-bound) filter on another stream than the statistics:
//   zm_frame_stats   `nmode` statistics starting at `mode0` (0 = image background,
//                    1 = variance level of 1 / wgt) -> raw mesh maps [mode][2][n]
//   zm_frame_filter  raw maps -> node planes [mode][2 maps][4][n] and stats [mode][2]
// Raw maps, node planes and stats are kept per frame (`index` of `count`) in the
// scratch slots "<slot>_raw" / "<slot>_nodes" / "<slot>_stats"; the histogram dumps
// are consumed in order on the statistics stream.
// Frame `index` of `count` owns a slot of `nslot` meshes (>= its own mesh count; the largest
// of a ragged stack) in each scratch buffer, so that every frame's products can coexist.
static int frame_slots(zm_ctx* ctx, int nx, int ny, int mesh, const char* slot, int index, int count,
                       int nslot, int* nbx_out, int* nby_out, float** raw, float** nodes, float** stats) {
    const int nbx = (nx - 1) / mesh + 1, nby = (ny - 1) / mesh + 1;
    const int n = nbx * nby;
    ZM_CHECK(n <= BK_MAXMESH, "background: %d x %d meshes exceed %d; raise BACK_SIZE", nbx, nby,
             BK_MAXMESH);
    if (nslot < n) nslot = n;
    std::string s(slot);
    ZM_TRY(ctx->get((s + "_raw").c_str(), sizeof(float) * 2 * 2 * nslot * (size_t)count, (void**)raw));
    ZM_TRY(ctx->get((s + "_nodes").c_str(), sizeof(float) * 2 * 8 * nslot * (size_t)count, (void**)nodes));
    ZM_TRY(ctx->get((s + "_stats").c_str(), sizeof(float) * 4 * (size_t)count, (void**)stats));
    *raw += (size_t)index * 2 * 2 * nslot;
    *nodes += (size_t)index * 2 * 8 * nslot;
    *stats += (size_t)index * 4;
    *nbx_out = nbx;
    *nby_out = nby;
    return 0;
}

// nf equally sized frames (slots index0 .. index0 + nf - 1 of `count`) per call; launches
// carry up to BK_BATCH frames each, so a 32-deep stack is 1 + 1 launches instead of 64 and
// the statistics grid (1152 workgroups per frame against 1024 resident) loses its tail.
int zm_batch_stats(zm_ctx* ctx, int nf, const float* const* imgs, const float* const* wgts, int nx,
                   int ny, int mesh, float wthresh, int mode0, int nmode, const char* slot, int index0,
                   int count, int nslot) {
    ZM_CHECK(mesh >= 8 && mesh <= 4096, "background: BACK_SIZE %d out of range [8, 4096]", mesh);
    ZM_CHECK(nmode >= 1 && mode0 >= 0 && mode0 + nmode <= 2, "background: bad statistic selection");
    ZM_CHECK(nf >= 1 && index0 >= 0 && index0 + nf <= count, "background: bad frame range");
    for (int f = 0; f < nf; ++f) {
        ZM_CHECK(mode0 + nmode < 2 || wgts[f] != nullptr, "background: the variance level needs a weight map");
        ZM_CHECK(mode0 > 0 || imgs[f] != nullptr, "background: image is NULL");
    }
    int nbx, nby;
    float *raw = nullptr, *nodes = nullptr, *stats = nullptr;
    ZM_TRY(frame_slots(ctx, nx, ny, mesh, slot, index0, count, nslot, &nbx, &nby, &raw, &nodes, &stats));
    const int n = nbx * nby;
    if (nslot < n) nslot = n;
    if (!ctx->bk_stats_set) {
        ZM_HIP(hipFuncSetAttribute((const void*)k_mesh_stats, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)sizeof(mesh_lds)));
        ctx->bk_stats_set = true;
    }
    zm_scope_timer t(ctx, "mesh_stats");
    if (mesh <= 128) {
        static const int dbg = ZM_DEVENV("ZM_DBG_BK") ? atoi(ZM_DEVENV("ZM_DBG_BK")) : 0;
        for (int f0 = 0; f0 < nf; f0 += BK_BATCH) {
            const int nb = std::min(BK_BATCH, nf - f0);
            bk_batch B;
            memset(&B, 0, sizeof(B));
            int vec_ok = (mesh % 4 == 0) && (nx % 4 == 0);
            for (int f = 0; f < nb; ++f) {
                B.img[f] = imgs[f0 + f];
                B.wgt[f] = wgts[f0 + f];
                if (((uintptr_t)B.img[f] & 15) || ((uintptr_t)B.wgt[f] & 15)) vec_ok = 0;
            }
            mesh_dump* dump = nullptr;
            ZM_TRY(ctx->get((std::string(slot) + "_dump").c_str(),
                            sizeof(mesh_dump) * 2 * (size_t)n * std::min(nf, BK_BATCH), (void**)&dump));
            const dim3 grid(nbx, nby, nb), block(BKF_THREADS, 1, 1);
            if (nmode == 2)
                hipLaunchKernelGGL(k_mesh_stats_fast<2>, grid, block, sizeof(meshf_lds), ctx->stream, B, nx, ny, mesh,
                                   nbx, nby, wthresh, vec_ok, dbg, dump);
            else if (mode0 == 0)
                hipLaunchKernelGGL(k_mesh_stats_fast<0>, grid, block, sizeof(meshf_lds), ctx->stream, B, nx, ny, mesh,
                                   nbx, nby, wthresh, vec_ok, dbg, dump);
            else
                hipLaunchKernelGGL(k_mesh_stats_fast<1>, grid, block, sizeof(meshf_lds), ctx->stream, B, nx, ny, mesh,
                                   nbx, nby, wthresh, vec_ok, dbg, dump);
            // (the statistics are through: what wants to run beside the small kernels that follow can start)
            if (!ctx->bk_stats_event) ZM_HIP(hipEventCreateWithFlags(&ctx->bk_stats_event, hipEventDisableTiming));
            ZM_HIP(hipEventRecord(ctx->bk_stats_event, ctx->stream));
            ctx->bk_stats_event_valid = true;
            hipLaunchKernelGGL(k_mesh_guess, dim3(n, nmode * nb, 1), dim3(64, 1, 1), 0, ctx->stream,
                               dump, n, nmode, nslot, raw + (size_t)f0 * 4 * nslot);
        }
    } else {
        for (int f = 0; f < nf; ++f)
            hipLaunchKernelGGL(k_mesh_stats, dim3(nbx, nby, nmode), dim3(BK_THREADS, 1, 1),
                               sizeof(mesh_lds), ctx->stream, imgs[f], wgts[f], nx, ny, mesh, nbx, nby,
                               wthresh, mode0, raw + (size_t)f * 4 * nslot);
    }
    ZM_HIP(hipGetLastError());
    return 0;
}

int zm_batch_filter(zm_ctx* ctx, int nf, int nx, int ny, int mesh, int fsize, int nmode,
                    const char* slot, int index0, int count, int nslot) {
    ZM_CHECK(fsize >= 1 && fsize <= 7, "background: BACK_FILTERSIZE %d out of range [1, 7]", fsize);
    int nbx, nby;
    float *raw = nullptr, *nodes = nullptr, *stats = nullptr;
    ZM_TRY(frame_slots(ctx, nx, ny, mesh, slot, index0, count, nslot, &nbx, &nby, &raw, &nodes, &stats));
    const int n = nbx * nby;
    if (nslot < n) nslot = n;
    const bool fast = n <= 1024;
    // FAST: 4 n maps + 4 n spline scratch + max(8 n planes, 2 x 1024 sort buffer)
    const size_t fsh = sizeof(float) * (fast ? (size_t)8 * n + std::max(8 * n, 2048) : (size_t)9 * n);
    if (!ctx->bk_filter_set) {
        ZM_HIP(hipFuncSetAttribute((const void*)k_mesh_filter<false>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
        ZM_HIP(hipFuncSetAttribute((const void*)k_mesh_filter<true>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
        ctx->bk_filter_set = true;
    }
    zm_scope_timer t(ctx, "mesh_filter");
    if (fast)
        hipLaunchKernelGGL(k_mesh_filter<true>, dim3(nmode * nf, 1, 1), dim3(1024, 1, 1), fsh, ctx->stream,
                           raw, nbx, nby, fsize, nmode, nslot, nodes, stats);
    else
        hipLaunchKernelGGL(k_mesh_filter<false>, dim3(nmode * nf, 1, 1), dim3(1024, 1, 1), fsh, ctx->stream,
                           raw, nbx, nby, fsize, nmode, nslot, nodes, stats);
    ZM_HIP(hipGetLastError());
    return 0;
}

// device pointers of frame `index`: node planes [2 statistics][2 maps][4][n], stats [2][2]
int zm_frame_products(zm_ctx* ctx, int nx, int ny, int mesh, const char* slot, int index, int count,
                      int nslot, float** nodes_dev, float** stats_dev, int* nbx_out, int* nby_out) {
    float* raw = nullptr;
    return frame_slots(ctx, nx, ny, mesh, slot, index, count, nslot, nbx_out, nby_out, &raw, nodes_dev,
                       stats_dev);
}

// variance rescale factors of nf frames: out[4 f] = backsig^2 / variance level
int zm_batch_var_scale(zm_ctx* ctx, int nf, const float* stats, float* out) {
    hipLaunchKernelGGL(k_var_scale_batch, dim3(zm_div_up(nf, 64)), dim3(64), 0, ctx->stream, stats, out, nf);
    ZM_HIP(hipGetLastError());
    return 0;
}

int zm_frame_stats(zm_ctx* ctx, const float* img, const float* wgt, int nx, int ny, int mesh,
                   float wthresh, int mode0, int nmode, const char* slot, int index, int count) {
    return zm_batch_stats(ctx, 1, &img, &wgt, nx, ny, mesh, wthresh, mode0, nmode, slot, index, count, 0);
}

int zm_frame_filter(zm_ctx* ctx, int nx, int ny, int mesh, int fsize, int nmode, float** nodes_dev,
                    float** stats_dev, int* nbx_out, int* nby_out, const char* slot, int index,
                    int count) {
    ZM_TRY(zm_batch_filter(ctx, 1, nx, ny, mesh, fsize, nmode, slot, index, count, 0));
    return zm_frame_products(ctx, nx, ny, mesh, slot, index, count, 0, nodes_dev, stats_dev, nbx_out, nby_out);
}

int zm_frame_background(zm_ctx* ctx, const float* img, const float* wgt, int nx, int ny,
                        int mesh, int fsize, float wthresh, int mode0, int nmode,
                        float** nodes_dev, float** stats_dev, int* nbx_out, int* nby_out,
                        const char* slot, int index, int count) {
    ZM_CHECK(fsize >= 1 && fsize <= 7, "background: BACK_FILTERSIZE %d out of range [1, 7]", fsize);
    ZM_TRY(zm_frame_stats(ctx, img, wgt, nx, ny, mesh, wthresh, mode0, nmode, slot, index, count));
    return zm_frame_filter(ctx, nx, ny, mesh, fsize, nmode, nodes_dev, stats_dev, nbx_out, nby_out,
                           slot, index, count);
}

int zm_launch_var_scale(zm_ctx* ctx, const float* bstats, const float* vstats, float* out) {
    hipLaunchKernelGGL(k_var_scale, dim3(1), dim3(1), 0, ctx->stream, bstats, vstats, out);
    ZM_HIP(hipGetLastError());
    return 0;
}

extern "C" int zm_background_dev(zm_ctx* ctx, const float* img, const float* wgt, int nx, int ny,
                                 int mesh, int filtersize, float* out_bkg, float* out_rms,
                                 float* out_sub, double* out_stats_host) {
    ZM_CHECK(ctx && img, "zm_background_dev: null argument");
    ZM_CHECK(nx > 0 && ny > 0, "zm_background_dev: empty image");
    ZM_HIP(hipSetDevice(ctx->device));
    float *nodes = nullptr, *stats = nullptr;
    int nbx = 0, nby = 0;
    ZM_TRY(zm_frame_background(ctx, img, wgt, nx, ny, mesh, filtersize, 1e-30f, 0, 1, &nodes, &stats,
                               &nbx, &nby, "bkg"));
    if (out_bkg || out_rms || out_sub) {
        zm_scope_timer t(ctx, "bk_expand");
        hipLaunchKernelGGL(k_bk_expand, dim3(zm_div_up(nx, 256), ny, 1), dim3(256, 1, 1), 0,
                           ctx->stream, img, nodes, nx, ny, nbx, nby, 1.0f / mesh, out_bkg,
                           out_rms, out_sub);
        ZM_HIP(hipGetLastError());
    }
    if (out_stats_host) {
        float hs[2];
        ZM_HIP(hipMemcpyAsync(hs, stats, sizeof(hs), hipMemcpyDeviceToHost, ctx->stream));
        ZM_HIP(hipStreamSynchronize(ctx->stream));
        out_stats_host[0] = hs[0];
        out_stats_host[1] = hs[1];
    }
    return 0;
}

extern "C" int zm_background(zm_ctx* ctx, const float* img, const float* wgt, int nx, int ny,
                             int mesh, int filtersize, float* out_bkg, float* out_rms,
                             float* out_sub, double* out_stats) {
    ZM_CHECK(ctx && img, "zm_background: null argument");
    ZM_CHECK(nx > 0 && ny > 0, "zm_background: empty image");
    ZM_HIP(hipSetDevice(ctx->device));
    const size_t np = (size_t)nx * ny;
    float *d_img = nullptr, *d_wgt = nullptr, *d_b = nullptr, *d_r = nullptr, *d_s = nullptr;
    ZM_TRY(ctx->get("h_img", np * 4, (void**)&d_img));
    ZM_HIP(hipMemcpyAsync(d_img, img, np * 4, hipMemcpyHostToDevice, ctx->stream));
    if (wgt) {
        ZM_TRY(ctx->get("h_wgt", np * 4, (void**)&d_wgt));
        ZM_HIP(hipMemcpyAsync(d_wgt, wgt, np * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    if (out_bkg) ZM_TRY(ctx->get("h_oimg", np * 4, (void**)&d_b));
    if (out_rms) ZM_TRY(ctx->get("h_owgt", np * 4, (void**)&d_r));
    if (out_sub) ZM_TRY(ctx->get("h_osub", np * 4, (void**)&d_s));
    ZM_TRY(zm_background_dev(ctx, d_img, d_wgt, nx, ny, mesh, filtersize, d_b, d_r, d_s, out_stats));
    if (out_bkg) ZM_HIP(hipMemcpyAsync(out_bkg, d_b, np * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (out_rms) ZM_HIP(hipMemcpyAsync(out_rms, d_r, np * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (out_sub) ZM_HIP(hipMemcpyAsync(out_sub, d_s, np * 4, hipMemcpyDeviceToHost, ctx->stream));
    ZM_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}
