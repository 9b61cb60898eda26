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

struct mesh_lds {
    int histo[BK_NLEVELS];          // counts, then inclusive prefix P0
    long long p1[BK_NLEVELS];       // inclusive prefix of h * i
    long long p2[BK_NLEVELS];       // inclusive prefix of h * i * i
    double red[4];
    long long sred[3][BK_THREADS];
};

// mode 0: statistic of img; mode 1: statistic of 1 / wgt (variance level)
__global__ __launch_bounds__(BK_THREADS) void k_mesh_stats(const float* __restrict__ img,
                                                           const float* __restrict__ wgt,
                                                           int nx, int ny, int mesh, int nbx,
                                                           float wthresh, int mode,
                                                           float* __restrict__ back,
                                                           float* __restrict__ sigm) {
    extern __shared__ char smem_raw[];
    mesh_lds* S = reinterpret_cast<mesh_lds*>(smem_raw);
    const int mi = blockIdx.x, mj = blockIdx.y;
    const int x0 = mi * mesh, y0 = mj * mesh;
    const int w = min(mesh, nx - x0), h = min(mesh, ny - y0);
    const int area = w * h;
    const int tid = threadIdx.x;

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

    // pass 1
    double s0 = 0, s1 = 0, s2 = 0;
    for (int k = tid; k < area; k += BK_THREADS) {
        bool ok;
        float v = value(k, &ok);
        if (ok) { s0 += 1.0; s1 += v; s2 += (double)v * v; }
    }
    s0 = block_sum(s0, S->red);
    s1 = block_sum(s1, S->red);
    s2 = block_sum(s2, S->red);
    float* ob = back + (size_t)mj * nbx + mi;
    float* os = sigm + (size_t)mj * nbx + mi;
    if (s0 < area * 0.5 || s0 < 1.0) {   // BACK_MINGOODFRAC
        if (tid == 0) { *ob = -BK_BIG; *os = -BK_BIG; }
        return;
    }
    double mean = s1 / s0;
    double var = s2 / s0 - mean * mean;
    double sig = var > 0 ? sqrt(var) : 0.0;
    const double lc = mean - 2.0 * sig, hc = mean + 2.0 * sig;
    // pass 2
    s0 = s1 = s2 = 0;
    for (int k = tid; k < area; k += BK_THREADS) {
        bool ok;
        float v = value(k, &ok);
        if (ok && v >= lc && v <= hc) { s0 += 1.0; s1 += v; s2 += (double)v * v; }
    }
    s0 = block_sum(s0, S->red);
    s1 = block_sum(s1, S->red);
    s2 = block_sum(s2, S->red);
    if (s0 < 1.0) {
        if (tid == 0) { *ob = -BK_BIG; *os = -BK_BIG; }
        return;
    }
    mean = s1 / s0;
    var = s2 / s0 - mean * mean;
    sig = var > 0 ? sqrt(var) : 0.0;
    int nlevels = (int)(0.7978845608028654 * 5.0 / 4.0 * s0 + 1.0);
    if (nlevels > BK_NLEVELS) nlevels = BK_NLEVELS;
    const double qscale = sig > 0 ? 2.0 * 5.0 * sig / nlevels : 1.0;
    const double qzero = mean - 5.0 * sig;

    // histogram
    for (int k = tid; k < BK_NLEVELS; k += BK_THREADS) S->histo[k] = 0;
    __syncthreads();
    for (int k = tid; k < area; k += BK_THREADS) {
        bool ok;
        float v = value(k, &ok);
        if (ok) {
            double b = floor(((double)v - qzero) / qscale + 0.5);
            if (b >= 0.0 && b < (double)nlevels) atomicAdd(&S->histo[(int)b], 1);
        }
    }
    __syncthreads();
    // inclusive prefix sums of h, h*i, h*i^2 (16 bins per thread + block scan)
    {
        const int per = BK_NLEVELS / BK_THREADS;
        long long a0 = 0, a1 = 0, a2 = 0;
        for (int k = 0; k < per; ++k) {
            int i = tid * per + k;
            long long hh = S->histo[i];
            a0 += hh; a1 += hh * i; a2 += hh * i * (long long)i;
        }
        S->sred[0][tid] = a0; S->sred[1][tid] = a1; S->sred[2][tid] = a2;
        __syncthreads();
        if (tid < 3) {   // three serial 256-element exclusive scans
            long long run = 0;
            for (int t = 0; t < BK_THREADS; ++t) {
                long long v = S->sred[tid][t];
                S->sred[tid][t] = run;
                run += v;
            }
        }
        __syncthreads();
        a0 = S->sred[0][tid]; a1 = S->sred[1][tid]; a2 = S->sred[2][tid];
        for (int k = 0; k < per; ++k) {
            int i = tid * per + k;
            long long hh = S->histo[i];
            a0 += hh; a1 += hh * i; a2 += hh * i * (long long)i;
            S->p1[i] = a1; S->p2[i] = a2;
            S->histo[i] = (int)a0;
        }
    }
    __syncthreads();
    if (tid != 0) return;
    // ---- backguess, thread 0, exact integer arithmetic on the prefix arrays ----
    const int* P0 = S->histo;
    auto p0 = [&](int i) -> long long { return i < 0 ? 0 : (long long)P0[i]; };
    auto q1 = [&](int i) -> long long { return i < 0 ? 0 : S->p1[i]; };
    auto q2 = [&](int i) -> long long { return i < 0 ? 0 : S->p2[i]; };
    auto hbin = [&](int i) -> long long { return p0(i) - p0(i - 1); };
    if (p0(nlevels - 1) == 0) { *ob = -BK_BIG; *os = -BK_BIG; return; }
    const int nlm1 = nlevels - 1;
    int lcut = 0, hcut = nlm1;
    double sg = 10.0 * nlm1, sg1 = 1.0, mea = mean, med = mean;
    for (int n = 100; n-- && sg >= 0.1 && fabs(sg / sg1 - 1.0) > 1e-4;) {
        sg1 = sg;
        const long long sum = p0(hcut) - p0(lcut - 1);
        mea = (double)(q1(hcut) - q1(lcut - 1));
        sg = (double)(q2(hcut) - q2(lcut - 1));
        // two-pointer walk == merge path: largest a with a == 0 or L(a-1) < H(T-a)
        const int T = hcut - lcut + 1;
        int lo = 0, hi = T;
        while (lo < hi) {
            int a = (lo + hi + 1) >> 1;
            long long La = p0(lcut + a - 2) - p0(lcut - 1);       // L(a-1)
            long long Hb = p0(hcut) - p0(hcut - (T - a));         // H(T-a)
            if (La < Hb) lo = a; else hi = a - 1;
        }
        const int a = lo, b = T - a;
        const long long lowsum = p0(lcut + a - 1) - p0(lcut - 1);
        const long long highsum = p0(hcut) - p0(hcut - b);
        const int ihigh = hcut - b, ilow = lcut + a;
        if (ihigh >= 0) {
            long long ha = hbin(ilow), hb = hbin(ihigh);
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
    double modev;
    if (sg > 0.0)
        modev = fabs((mea - med) / sg) < 0.3 ? qzero + (2.5 * med - 1.5 * mea) * qscale
                                             : qzero + med * qscale;
    else
        modev = qzero + mea * qscale;
    *ob = (float)modev;
    *os = (float)(sg * qscale);
}

// ---------------------------------------------------------------------------
__device__ inline float small_median(float* v, int n) {
    for (int i = 1; i < n; ++i) {
        float t = v[i];
        int j = i - 1;
        while (j >= 0 && v[j] > t) { v[j + 1] = v[j]; --j; }
        v[j + 1] = t;
    }
    return (n & 1) ? v[n / 2] : 0.5f * (v[n / 2 - 1] + v[n / 2]);
}

// natural cubic spline second derivatives / 6 along a strided line (unit spacing)
__device__ inline void spline_line(const float* a, float* d, float* u, int n, int stride) {
    for (int k = 0; k < n; ++k) d[k * stride] = 0.f;
    if (n < 3) return;
    u[0] = 0.f;
    for (int y = 1; y < n - 1; ++y) {
        float temp = -1.f / (d[(y - 1) * stride] + 4.f);
        d[y * stride] = temp;
        u[y * stride] = temp * (u[(y - 1) * stride]
                                - 6.f * (a[(y + 1) * stride] + a[(y - 1) * stride]
                                         - 2.f * a[y * stride]));
    }
    d[(n - 1) * stride] = 0.f;
    for (int y = n - 2; y >= 1; --y)
        d[y * stride] = d[y * stride] * d[(y + 1) * stride] + u[y * stride];
    d[0] = 0.f;
    for (int k = 0; k < n; ++k) d[k * stride] *= (1.f / 6.f);
}

#define BK_MAXMESH 4096

// raw[0..n) mode map, raw[n..2n) sigma map (may hold -BIG); nodes: 2 maps x 4 planes;
// stats: {backmean, backsig}
__global__ __launch_bounds__(BK_THREADS) void k_mesh_filter(const float* __restrict__ raw,
                                                            int nbx, int nby, int fsize,
                                                            float* __restrict__ nodes,
                                                            float* __restrict__ stats) {
    __shared__ float sb[2][BK_MAXMESH];   // filled maps
    __shared__ float fb[2][BK_MAXMESH];   // filtered maps
    __shared__ float tmp[BK_MAXMESH];     // spline scratch
    __shared__ int ngood;
    const int n = nbx * nby, tid = threadIdx.x;
    if (tid == 0) ngood = 0;
    __syncthreads();
    for (int k = tid; k < n; k += BK_THREADS)
        if (raw[k] > -BK_BIG) atomicAdd(&ngood, 1);
    __syncthreads();
    // 1. fill bad meshes from the nearest good ones (ties averaged)
    for (int k = tid; k < n; k += BK_THREADS) {
        float b = raw[k], s = raw[n + k];
        if (!(b > -BK_BIG)) {
            if (ngood == 0) { b = 0.f; s = 1.f; }
            else {
                int j = k / nbx, i = k - j * nbx;
                int dmin = 0x7fffffff, cnt = 0;
                float vb = 0.f, vs = 0.f;
                for (int q = 0; q < n; ++q) {
                    if (!(raw[q] > -BK_BIG)) continue;
                    int qj = q / nbx, qi = q - qj * nbx;
                    int d2 = (qi - i) * (qi - i) + (qj - j) * (qj - j);
                    if (d2 < dmin) { dmin = d2; vb = raw[q]; vs = raw[n + q]; cnt = 1; }
                    else if (d2 == dmin) { vb += raw[q]; vs += raw[n + q]; ++cnt; }
                }
                b = vb / cnt; s = vs / cnt;
            }
        }
        sb[0][k] = b; sb[1][k] = s;
    }
    __syncthreads();
    // 2. fsize x fsize median filter (window clipped at the borders)
    const int hb = fsize / 2;
    for (int k = tid; k < n; k += BK_THREADS) {
        int j = k / nbx, i = k - j * nbx;
        if (fsize > 1) {
            float wv[2][49];
            int c = 0;
            for (int jj = max(j - hb, 0); jj <= min(j + hb, nby - 1); ++jj)
                for (int ii = max(i - hb, 0); ii <= min(i + hb, nbx - 1); ++ii) {
                    if (c < 49) { wv[0][c] = sb[0][jj * nbx + ii]; wv[1][c] = sb[1][jj * nbx + ii]; ++c; }
                }
            fb[0][k] = small_median(wv[0], c);
            fb[1][k] = small_median(wv[1], c);
        } else {
            fb[0][k] = sb[0][k]; fb[1][k] = sb[1][k];
        }
    }
    __syncthreads();
    // 3. global medians by rank counting (exact, O(n^2 / threads))
    for (int m = 0; m < 2; ++m) {
        for (int k = tid; k < n; k += BK_THREADS) {
            float v = fb[m][k];
            int less = 0, eq = 0;
            for (int q = 0; q < n; ++q) { less += fb[m][q] < v; eq += fb[m][q] == v; }
            // v occupies sorted positions [less, less + eq)
            int i1 = (n - 1) / 2, i2 = n / 2;
            if (i1 >= less && i1 < less + eq) tmp[2 * m] = v;
            if (i2 >= less && i2 < less + eq) tmp[2 * m + 1] = v;
        }
        __syncthreads();
    }
    if (tid == 0) {
        stats[0] = 0.5f * (tmp[0] + tmp[1]);
        stats[1] = 0.5f * (tmp[2] + tmp[3]);
    }
    __syncthreads();
    // 4. node planes: V, DY (along y per column), A (along x of V), B (along x of DY)
    for (int m = 0; m < 2; ++m) {
        float* V = nodes + (size_t)m * 4 * n;
        float* DY = V + n;
        float* A = V + 2 * n;
        float* B = V + 3 * n;
        for (int k = tid; k < n; k += BK_THREADS) V[k] = fb[m][k];
        __syncthreads();
        for (int i = tid; i < nbx; i += BK_THREADS) spline_line(V + i, DY + i, tmp + i, nby, nbx);
        __syncthreads();
        for (int j = tid; j < nby; j += BK_THREADS) spline_line(V + j * nbx, A + j * nbx, tmp + j * nbx, nbx, 1);
        __syncthreads();
        for (int j = tid; j < nby; j += BK_THREADS) spline_line(DY + j * nbx, B + j * nbx, tmp + j * nbx, nbx, 1);
        __syncthreads();
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

// ---------------------------------------------------------------------------
// Runs stats + filter for one frame.  Node planes (bkg map then sigma map, 4
// planes each) and the {backmean, backsig} pair stay on the device in scratch
// slots "<slot>_nodes" / "<slot>_stats".
int zm_frame_background(zm_ctx* ctx, const float* img, const float* wgt, int nx, int ny,
                        int mesh, int fsize, float wthresh, int mode, float** nodes_dev,
                        float** stats_dev, int* nbx_out, int* nby_out, const char* slot) {
    ZM_CHECK(mesh >= 8 && mesh <= 4096, "background: BACK_SIZE %d out of range [8, 4096]", mesh);
    ZM_CHECK(fsize >= 1 && fsize <= 7, "background: BACK_FILTERSIZE %d out of range [1, 7]", fsize);
    const int nbx = (nx - 1) / mesh + 1, nby = (ny - 1) / mesh + 1;
    ZM_CHECK(nbx * nby <= BK_MAXMESH, "background: %d x %d meshes exceed %d; raise BACK_SIZE",
             nbx, nby, BK_MAXMESH);
    std::string s(slot);
    float *raw = nullptr, *nodes = nullptr, *stats = nullptr;
    ZM_TRY(ctx->get((s + "_raw").c_str(), sizeof(float) * 2 * nbx * nby, (void**)&raw));
    ZM_TRY(ctx->get((s + "_nodes").c_str(), sizeof(float) * 8 * nbx * nby, (void**)&nodes));
    ZM_TRY(ctx->get((s + "_stats").c_str(), sizeof(float) * 4, (void**)&stats));
    static bool attr_set = false;
    if (!attr_set) {
        ZM_HIP(hipFuncSetAttribute((const void*)k_mesh_stats,
                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)sizeof(mesh_lds)));
        attr_set = true;
    }
    {
        zm_scope_timer t(ctx, "mesh_stats");
        hipLaunchKernelGGL(k_mesh_stats, dim3(nbx, nby, 1), dim3(BK_THREADS, 1, 1),
                           sizeof(mesh_lds), ctx->stream, img, wgt, nx, ny, mesh, nbx, wthresh,
                           mode, raw, raw + nbx * nby);
        ZM_HIP(hipGetLastError());
    }
    {
        zm_scope_timer t(ctx, "mesh_filter");
        hipLaunchKernelGGL(k_mesh_filter, dim3(1, 1, 1), dim3(BK_THREADS, 1, 1), 0, ctx->stream,
                           raw, nbx, nby, fsize, nodes, stats);
        ZM_HIP(hipGetLastError());
    }
    *nodes_dev = nodes;
    *stats_dev = stats;
    *nbx_out = nbx;
    *nby_out = nby;
    return 0;
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
    ZM_TRY(zm_frame_background(ctx, img, wgt, nx, ny, mesh, filtersize, 1e-30f, 0, &nodes, &stats,
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
