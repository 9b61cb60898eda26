/* CPU restatement in C of the subtraction leg of the hot path (oracle; TEST INFRASTRUCTURE, not product code:
 * only tests/ and the cpu_baseline leg of bench.py may build, load or call this).
 *
 * A port of oracle/hotpants.py - the operator the reference defines through zuds/hotpants.py:77-93 (hotpants -c t
 * -n i -r 2.5 SEEING -rss 6 SEEING -nsx / -nsy / -nrx / -nry -ko 4 -bgo 0 -tni / -ini / -imi / -oni / -fin, call site
 * zuds/subtraction.py:162) - function by function, with its conventions: validity mask, regions, greedy brightest-first
 * substamps, Gaussian x polynomial basis in hotpants' normalisation, unweighted least squares with Jacobi scaling and
 * a 1e-10 ridge, Cholesky in fp64, sigma-clipped stamp rejection (at most 8 rounds), kernel re-evaluated per
 * (2 hwk + 1)^2 block, D = I - (T (x) K + bg), noise = sqrt(sI^2 + sT^2 (x) K^2), fill values.
 *
 * One difference of FORM, none of definition: the basis vectors of a substamp are built with the separable passes the
 * basis allows (row pass per 1-D filter, column pass per term, as hotpants' own xy_conv_stamp does and as
 * csrc/hp_vectors.hip does) instead of 49 direct two-dimensional correlations; the sums differ from the numpy
 * oracle's in rounding only (tests/test_oracle_cport.py: 1e-9 of the difference image's scale, the same stamps,
 * rounds and counts).  fp64 throughout; OpenMP over substamps, stamp cells and output blocks.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAX_ROUNDS 8
#define RIDGE 1e-10
#define MAXG 4
#define MAXDEG 8
#define MAXNSS 8

typedef struct {
    double tu, tl, iu, il, r, rss, fin, fi, ft, ks;
    int32_t nsx, nsy, nrx, nry, ko, bgo, nss, normalize, ngauss;
    int32_t deg[MAXG];
    double sigma[MAXG];
} zo_hp_params;

typedef struct {
    int32_t solved, nstamps_total, nstamps_used, niter, ncoeff, pad_;
    double kernel_sum, chi2;
} zo_hp_region;

static double ipw(double x, int n) { double r = 1.0; while (n-- > 0) r *= x; return r; }

/* clipped_moments of oracle/hotpants.py: mean / population std, `passes` rounds of |v - m| <= nsig s */
static void clipped_moments(const double* v, int64_t n, double nsig, int passes, double* mo, double* so) {
    if (n == 0) { *mo = 0.0; *so = 0.0; return; }
    double m = 0.0, s = 0.0;
    for (int64_t i = 0; i < n; ++i) m += v[i];
    m /= (double)n;
    for (int64_t i = 0; i < n; ++i) s += (v[i] - m) * (v[i] - m);
    s = sqrt(s / (double)n);
    for (int p = 0; p < passes; ++p) {
        double sm = 0.0; int64_t c = 0;
        for (int64_t i = 0; i < n; ++i) if (fabs(v[i] - m) <= nsig * s) { sm += v[i]; ++c; }
        if (c == 0) break;
        const double m2 = sm / (double)c;
        double ss = 0.0;
        for (int64_t i = 0; i < n; ++i) if (fabs(v[i] - m) <= nsig * s) ss += (v[i] - m2) * (v[i] - m2);
        m = m2;
        s = sqrt(ss / (double)c);
    }
    *mo = m; *so = s;
}

/* box_any: 1 where any bad pixel lies within the (2 hw + 1)^2 box (clipped at the frame) */
static void box_any(const uint8_t* bad, int nx, int ny, int hw, uint8_t* out) {
    int32_t* c = (int32_t*)calloc((size_t)(nx + 1) * (ny + 1), sizeof(int32_t));
    for (int y = 0; y < ny; ++y) {
        int32_t run = 0;
        for (int x = 0; x < nx; ++x) {
            run += bad[(size_t)y * nx + x] ? 1 : 0;
            c[(size_t)(y + 1) * (nx + 1) + x + 1] = c[(size_t)y * (nx + 1) + x + 1] + run;
        }
    }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < ny; ++y) {
        const int y0 = y - hw < 0 ? 0 : y - hw, y1 = y + hw + 1 > ny ? ny : y + hw + 1;
        for (int x = 0; x < nx; ++x) {
            const int x0 = x - hw < 0 ? 0 : x - hw, x1 = x + hw + 1 > nx ? nx : x + hw + 1;
            const int32_t s = c[(size_t)y1 * (nx + 1) + x1] - c[(size_t)y0 * (nx + 1) + x1] - c[(size_t)y1 * (nx + 1) + x0] +
                              c[(size_t)y0 * (nx + 1) + x0];
            out[(size_t)y * nx + x] = s > 0;
        }
    }
    free(c);
}

typedef struct {
    int hwk, kw, nc, nf1;                 /* kw = 2 hwk + 1; nc terms; nf1 distinct 1-D filters */
    int tg[64], ta[64], tb[64], tee[64];  /* term: gaussian, x degree, y degree, even-even */
    int fbase[MAXG];                      /* first 1-D filter of a gaussian */
    double* f1;                           /* [nf1][kw] raw filters ga * u^a */
    double* f1n;                          /* the same, unit sum (used by even-even terms) */
    double* k2;                           /* [nc][kw][kw] 2-D basis, hotpants normalisation */
} basis_t;

static int make_basis(const zo_hp_params* p, basis_t* B) {
    const int hwk = (int)p->r, kw = 2 * hwk + 1;
    memset(B, 0, sizeof(*B));
    B->hwk = hwk; B->kw = kw;
    int nf = 0, nc = 0;
    for (int g = 0; g < p->ngauss; ++g) { B->fbase[g] = nf; nf += p->deg[g] + 1; }
    B->nf1 = nf;
    B->f1 = (double*)malloc(sizeof(double) * nf * kw);
    B->f1n = (double*)malloc(sizeof(double) * nf * kw);
    for (int g = 0; g < p->ngauss; ++g)
        for (int a = 0; a <= p->deg[g]; ++a) {
            double* f = B->f1 + (size_t)(B->fbase[g] + a) * kw;
            double s = 0.0;
            for (int i = 0; i < kw; ++i) {
                const double u = (double)(i - hwk);
                f[i] = exp(-u * u / (2.0 * p->sigma[g] * p->sigma[g])) * ipw(u, a);
                s += f[i];
            }
            double* fn = B->f1n + (size_t)(B->fbase[g] + a) * kw;
            for (int i = 0; i < kw; ++i) fn[i] = f[i] / s;
        }
    for (int g = 0; g < p->ngauss; ++g)
        for (int a = 0; a <= p->deg[g]; ++a)
            for (int b = 0; b <= p->deg[g] - a; ++b) {
                if (nc >= 64) return 1;
                B->tg[nc] = g; B->ta[nc] = a; B->tb[nc] = b; B->tee[nc] = (a % 2 == 0 && b % 2 == 0);
                ++nc;
            }
    B->nc = nc;
    B->k2 = (double*)malloc(sizeof(double) * (size_t)nc * kw * kw);
    for (int n = 0; n < nc; ++n) {
        const double* fx = (B->tee[n] ? B->f1n : B->f1) + (size_t)(B->fbase[B->tg[n]] + B->ta[n]) * kw;
        const double* fy = (B->tee[n] ? B->f1n : B->f1) + (size_t)(B->fbase[B->tg[n]] + B->tb[n]) * kw;
        double* k = B->k2 + (size_t)n * kw * kw;
        for (int v = 0; v < kw; ++v)
            for (int u = 0; u < kw; ++u) k[v * kw + u] = fy[v] * fx[u];
        if (B->tee[n] && n > 0)
            for (int i = 0; i < kw * kw; ++i) k[i] -= B->k2[i];
    }
    return 0;
}

static void free_basis(basis_t* B) { free(B->f1); free(B->f1n); free(B->k2); }

typedef struct {
    double *Q, *b;             /* [ne][ne], [ne] */
    double ii, vbar, fx, fy;
    int npix, cx, cy, have;
} stamp_sys;

/* substamp_system: extended vectors of one substamp -> Gram pieces */
static void substamp_system(const double* sci, const double* ref, const double* svar, const double* tvar, int nx,
                            int cx, int cy, const basis_t* B, const int reg[4], const zo_hp_params* p, int nbg,
                            const int* bgi, const int* bgj, stamp_sys* S) {
    const int hwk = B->hwk, kw = B->kw, hwss = (int)p->rss, hw = hwk + hwss, pw = 2 * hw + 1, sw = 2 * hwss + 1;
    const int npix = sw * sw, nc = B->nc, ne = nc + nbg;
    double* E = (double*)malloc(sizeof(double) * (size_t)ne * npix);
    double* R = (double*)malloc(sizeof(double) * (size_t)pw * sw);      /* row pass of one 1-D filter */
    /* true convolution W(y, x) = sum_dv fy[hwk + dv] sum_du fx[hwk + du] T(y - dv, x - du) */
    for (int g = 0; g < p->ngauss; ++g)
        for (int a = 0; a <= p->deg[g]; ++a)
            for (int norm = 0; norm < 2; ++norm) {
                /* which terms use x filter (g, a) in this normalisation? */
                int any = 0;
                for (int n = 0; n < nc; ++n) any |= (B->tg[n] == g && B->ta[n] == a && B->tee[n] == norm);
                if (!any) continue;
                const double* fx = (norm ? B->f1n : B->f1) + (size_t)(B->fbase[g] + a) * kw;
                for (int yy = 0; yy < pw; ++yy) {
                    const double* row = ref + (size_t)(cy - hw + yy) * nx + (cx - hw);
                    for (int xx = 0; xx < sw; ++xx) {
                        double s = 0.0;
                        for (int du = -hwk; du <= hwk; ++du) s += fx[hwk + du] * row[xx + hwk - du];
                        R[yy * sw + xx] = s;
                    }
                }
                for (int n = 0; n < nc; ++n) {
                    if (!(B->tg[n] == g && B->ta[n] == a && B->tee[n] == norm)) continue;
                    const double* fy = (norm ? B->f1n : B->f1) + (size_t)(B->fbase[g] + B->tb[n]) * kw;
                    double* W = E + (size_t)n * npix;
                    for (int yy = 0; yy < sw; ++yy)
                        for (int xx = 0; xx < sw; ++xx) {
                            double s = 0.0;
                            for (int dv = -hwk; dv <= hwk; ++dv) s += fy[hwk + dv] * R[(yy + hwk - dv) * sw + xx];
                            W[yy * sw + xx] = s;
                        }
                }
            }
    for (int n = 1; n < nc; ++n)
        if (B->tee[n])
            for (int i = 0; i < npix; ++i) E[(size_t)n * npix + i] -= E[i];
    const double xc = reg[0] + (reg[1] - reg[0]) / 2.0, hx = (reg[1] - reg[0]) / 2.0;
    const double yc = reg[2] + (reg[3] - reg[2]) / 2.0, hy = (reg[3] - reg[2]) / 2.0;
    for (int q = 0; q < nbg; ++q)
        for (int yy = 0; yy < sw; ++yy)
            for (int xx = 0; xx < sw; ++xx) {
                const double xf = ((cx - hwss + xx) - xc) / hx, yf = ((cy - hwss + yy) - yc) / hy;
                E[(size_t)(nc + q) * npix + yy * sw + xx] = ipw(xf, bgi[q]) * ipw(yf, bgj[q]);
            }
    double* I = (double*)malloc(sizeof(double) * npix);
    double vs = 0.0, ii = 0.0;
    for (int yy = 0; yy < sw; ++yy)
        for (int xx = 0; xx < sw; ++xx) {
            const size_t k = (size_t)(cy - hwss + yy) * nx + (cx - hwss + xx);
            I[yy * sw + xx] = sci[k];
            vs += svar[k] + tvar[k];
            ii += sci[k] * sci[k];
        }
    for (int i = 0; i < ne; ++i) {
        const double* ei = E + (size_t)i * npix;
        for (int j = 0; j <= i; ++j) {
            const double* ej = E + (size_t)j * npix;
            double s = 0.0;
            for (int k = 0; k < npix; ++k) s += ei[k] * ej[k];
            S->Q[i * ne + j] = S->Q[j * ne + i] = s;
        }
        double s = 0.0;
        for (int k = 0; k < npix; ++k) s += ei[k] * I[k];
        S->b[i] = s;
    }
    S->ii = ii;
    S->vbar = vs / npix;
    S->npix = npix;
    S->fx = (cx - xc) / hx;
    S->fy = (cy - yc) / hy;
    S->cx = cx; S->cy = cy; S->have = 1;
    free(E); free(R); free(I);
}

/* in-place lower Cholesky of the n x n matrix A (row-major); returns 0, or 1 when a pivot is not positive */
static int cholesky(double* A, int n) {
    for (int j = 0; j < n; ++j) {
        double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
        if (!(d > 0.0)) return 1;
        d = sqrt(d);
        A[(size_t)j * n + j] = d;
#pragma omp parallel for schedule(static) if (n - j > 256)
        for (int i = j + 1; i < n; ++i) {
            double s = A[(size_t)i * n + j];
            const double *ai = A + (size_t)i * n, *aj = A + (size_t)j * n;
            for (int k = 0; k < j; ++k) s -= ai[k] * aj[k];
            A[(size_t)i * n + j] = s / d;
        }
    }
    return 0;
}

typedef struct { int *cols, *src; int n; } design_t;

static void make_design(int nc, int nbg, int nkp, design_t* D) {
    D->n = 1 + (nc - 1) * nkp + nbg;
    D->cols = (int*)malloc(sizeof(int) * D->n);
    D->src = (int*)malloc(sizeof(int) * D->n);
    int k = 0;
    D->cols[k] = 0; D->src[k] = 0; ++k;
    for (int n = 1; n < nc; ++n)
        for (int q = 0; q < nkp; ++q) { D->cols[k] = 1 + (n - 1) * nkp + q; D->src[k] = n; ++k; }
    for (int q = 0; q < nbg; ++q) { D->cols[k] = 1 + (nc - 1) * nkp + q; D->src[k] = nc + q; ++k; }
}

static void design_weights(const stamp_sys* S, int nc, int nbg, int nkp, const int* kpi, const int* kpj, double* w) {
    int k = 0;
    w[k++] = 1.0;
    for (int n = 1; n < nc; ++n)
        for (int q = 0; q < nkp; ++q) w[k++] = ipw(S->fx, kpi[q]) * ipw(S->fy, kpj[q]);
    for (int q = 0; q < nbg; ++q) w[k++] = 1.0;
}

static int poly_terms(int order, int* pi, int* pj) {
    int n = 0;
    for (int i = 0; i <= order; ++i)
        for (int j = 0; j <= order - i; ++j) { pi[n] = i; pj[n] = j; ++n; }
    return n;
}

/* fit_region: x[nunk] or solved = 0 */
static void fit_region(const double* sci, const double* ref, const double* svar, const double* tvar,
                       const uint8_t* ok, const uint8_t* elig_all, int nx, int ny, const int reg[4], const basis_t* B,
                       const zo_hp_params* p, double* x, zo_hp_region* info) {
    const int hwss = (int)p->rss;
    const int nc = B->nc;
    int bgi[64], bgj[64], kpi[64], kpj[64];
    const int nbg = poly_terms(p->bgo, bgi, bgj), nkp = poly_terms(p->ko, kpi, kpj);
    const int ne = nc + nbg, nunk = 1 + (nc - 1) * nkp + nbg;
    const int ncell = p->nsx * p->nsy, nss = p->nss;
    const int cw = (reg[1] - reg[0]) / p->nsx, ch = (reg[3] - reg[2]) / p->nsy;
    int* cand = (int*)malloc(sizeof(int) * 2 * ncell * nss);       /* centres (x, y) */
    int* ncand = (int*)calloc(ncell, sizeof(int));
    memset(info, 0, sizeof(*info));
    /* ---- find_substamps ---- */
#pragma omp parallel for schedule(dynamic)
    for (int cell = 0; cell < ncell; ++cell) {
        const int sy = cell / p->nsx, sx = cell % p->nsx;
        const int cx0 = reg[0] + sx * cw, cy0 = reg[2] + sy * ch;
        double* vals = (double*)malloc(sizeof(double) * (size_t)cw * ch);
        uint8_t* el = (uint8_t*)malloc((size_t)cw * ch);
        int64_t nv = 0;
        for (int yy = 0; yy < ch; ++yy)
            for (int xx = 0; xx < cw; ++xx) {
                const size_t k = (size_t)(cy0 + yy) * nx + cx0 + xx;
                if (ok[k]) vals[nv++] = ref[k];
            }
        double sky, sig;
        clipped_moments(vals, nv, 3.0, 3, &sky, &sig);
        const double thr = sky + p->ft * sig;
        for (int yy = 0; yy < ch; ++yy)
            for (int xx = 0; xx < cw; ++xx) {
                const size_t k = (size_t)(cy0 + yy) * nx + cx0 + xx;
                el[yy * cw + xx] = elig_all[k] && (ref[k] >= thr);
            }
        for (int s = 0; s < nss; ++s) {
            double best = -INFINITY; int bj = -1;
            for (int j = 0; j < cw * ch; ++j)
                if (el[j]) {
                    const double t = ref[(size_t)(cy0 + j / cw) * nx + cx0 + j % cw];
                    if (t > best) { best = t; bj = j; }
                }
            if (bj < 0) break;
            const int yy = bj / cw, xx = bj % cw;
            cand[2 * (cell * nss + s)] = cx0 + xx;
            cand[2 * (cell * nss + s) + 1] = cy0 + yy;
            ncand[cell] = s + 1;
            for (int y2 = (yy - hwss < 0 ? 0 : yy - hwss); y2 < yy + hwss + 1 && y2 < ch; ++y2)
                for (int x2 = (xx - hwss < 0 ? 0 : xx - hwss); x2 < xx + hwss + 1 && x2 < cw; ++x2) el[y2 * cw + x2] = 0;
        }
        free(vals); free(el);
    }
    int* active = (int*)malloc(sizeof(int) * ncell);
    int ntotal = 0;
    for (int c = 0; c < ncell; ++c) { active[c] = ncand[c] ? 0 : -1; ntotal += ncand[c] ? 1 : 0; }
    stamp_sys* cache = (stamp_sys*)calloc((size_t)ncell * nss, sizeof(stamp_sys));
    design_t D;
    make_design(nc, nbg, nkp, &D);
    double* A = (double*)malloc(sizeof(double) * (size_t)nunk * nunk);
    double* rhs = (double*)malloc(sizeof(double) * nunk);
    double* dsc = (double*)malloc(sizeof(double) * nunk);
    double* merits = (double*)malloc(sizeof(double) * ncell);
    int* live = (int*)malloc(sizeof(int) * ncell);
    int rounds = 0, nfit = 0, nm = 0, solved = 0;
    for (rounds = 1; rounds <= MAX_ROUNDS; ++rounds) {
        int nl = 0;
        for (int c = 0; c < ncell; ++c) if (active[c] >= 0) live[nl++] = c;
        if (nl == 0) { solved = 0; break; }
        /* the systems of the live stamps (cached per substamp) */
#pragma omp parallel for schedule(dynamic)
        for (int k = 0; k < nl; ++k) {
            stamp_sys* S = cache + (size_t)live[k] * nss + active[live[k]];
            if (S->have) continue;
            S->Q = (double*)malloc(sizeof(double) * ne * ne);
            S->b = (double*)malloc(sizeof(double) * ne);
            substamp_system(sci, ref, svar, tvar, nx, cand[2 * (live[k] * nss + active[live[k]])],
                            cand[2 * (live[k] * nss + active[live[k]]) + 1], B, reg, p, nbg, bgi, bgj, S);
        }
        nfit = nl;
        /* solve_region */
        memset(A, 0, sizeof(double) * (size_t)nunk * nunk);
        memset(rhs, 0, sizeof(double) * nunk);
        double* w = (double*)malloc(sizeof(double) * D.n);
        for (int k = 0; k < nl; ++k) {
            const stamp_sys* S = cache + (size_t)live[k] * nss + active[live[k]];
            design_weights(S, nc, nbg, nkp, kpi, kpj, w);
#pragma omp parallel for schedule(static)
            for (int i = 0; i < D.n; ++i) {
                double* ar = A + (size_t)D.cols[i] * nunk;
                const double* qr = S->Q + (size_t)D.src[i] * ne;
                const double wi = w[i];
                for (int j = 0; j < D.n; ++j) ar[D.cols[j]] += qr[D.src[j]] * (wi * w[j]);
            }
            for (int i = 0; i < D.n; ++i) rhs[D.cols[i]] += w[i] * S->b[D.src[i]];
        }
        for (int i = 0; i < nunk; ++i) { const double d = A[(size_t)i * nunk + i]; dsc[i] = sqrt(d > 0 ? d : 1.0); }
        for (int i = 0; i < nunk; ++i)
            for (int j = 0; j < nunk; ++j) A[(size_t)i * nunk + j] /= dsc[i] * dsc[j];
        for (int i = 0; i < nunk; ++i) A[(size_t)i * nunk + i] += RIDGE;
        if (cholesky(A, nunk)) { free(w); solved = 0; break; }
        for (int i = 0; i < nunk; ++i) {                       /* L y = rhs / d */
            double s = rhs[i] / dsc[i];
            for (int k = 0; k < i; ++k) s -= A[(size_t)i * nunk + k] * x[k];
            x[i] = s / A[(size_t)i * nunk + i];
        }
        for (int i = nunk - 1; i >= 0; --i) {                  /* L^T x = y */
            double s = x[i];
            for (int k = i + 1; k < nunk; ++k) s -= A[(size_t)k * nunk + i] * x[k];
            x[i] = s / A[(size_t)i * nunk + i];
        }
        for (int i = 0; i < nunk; ++i) x[i] /= dsc[i];
        solved = 1;
        /* stamp_merit */
        nm = nl;
        for (int k = 0; k < nl; ++k) {
            const stamp_sys* S = cache + (size_t)live[k] * nss + active[live[k]];
            design_weights(S, nc, nbg, nkp, kpi, kpj, w);
            double c[128];
            memset(c, 0, sizeof(c));
            for (int i = 0; i < D.n; ++i) c[D.src[i]] += w[i] * x[D.cols[i]];
            double cb = 0.0, cqc = 0.0;
            for (int i = 0; i < ne; ++i) {
                cb += c[i] * S->b[i];
                double s = 0.0;
                for (int j = 0; j < ne; ++j) s += S->Q[i * ne + j] * c[j];
                cqc += c[i] * s;
            }
            merits[k] = (S->ii - 2.0 * cb + cqc) / (S->npix * S->vbar);
        }
        free(w);
        double m, s;
        clipped_moments(merits, nl, 3.0, 3, &m, &s);
        int nrej = 0;
        for (int k = 0; k < nl; ++k)
            if (merits[k] > m + p->ks * s) {
                ++nrej;
                const int c = live[k];
                active[c] += 1;
                if (active[c] >= ncand[c]) active[c] = -1;
            }
        if (!nrej) break;
    }
    if (rounds > MAX_ROUNDS) rounds = MAX_ROUNDS;
    info->solved = solved;
    if (solved) {
        double cm = 0.0;
        for (int k = 0; k < nm; ++k) cm += merits[k];
        info->nstamps_total = ntotal;
        info->nstamps_used = nfit;
        info->niter = rounds;
        info->ncoeff = nunk;
        info->kernel_sum = x[0];
        info->chi2 = nm ? cm / nm : 0.0;
    }
    for (size_t k = 0; k < (size_t)ncell * nss; ++k) if (cache[k].have) { free(cache[k].Q); free(cache[k].b); }
    free(cache); free(cand); free(ncand); free(active); free(A); free(rhs); free(dsc); free(merits); free(live);
    free(D.cols); free(D.src);
}

/* The whole difference image (oracle/hotpants.py::subtract).  only_region >= 0: fit and apply that region alone
 * (the cpu_baseline leg times one region per process / call).  Planes: float64, row-major [ny][nx]; bpm may be NULL.
 * diff / noise: [ny][nx] float64, filled with fi / fin where nothing is computed.  regions: nrx nry records. */
int zo_hotpants(const double* sci, const double* ref, const double* srms, const double* trms, const uint8_t* bpm, int nx,
                int ny, const zo_hp_params* p, int only_region, double* diff, double* noise, zo_hp_region* regions,
                int64_t* nmasked) {
    basis_t B;
    if (make_basis(p, &B)) return 1;
    const int hwk = B.hwk, kw = B.kw, hwss = (int)p->rss, hw = hwk + hwss, step = kw;
    const size_t np = (size_t)nx * ny;
    double* svar = (double*)malloc(sizeof(double) * np);
    double* tvar = (double*)malloc(sizeof(double) * np);
    double* refz = (double*)malloc(sizeof(double) * np);
    double* tvz = (double*)malloc(sizeof(double) * np);
    uint8_t* ok = (uint8_t*)malloc(np);
    uint8_t* nok = (uint8_t*)malloc(np);
    uint8_t* dirty = (uint8_t*)malloc(np);
    uint8_t* outbad = (uint8_t*)malloc(np);
    uint8_t* elig = (uint8_t*)malloc(np);
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < (int64_t)np; ++k) {
        svar[k] = srms[k] * srms[k];
        tvar[k] = trms[k] * trms[k];
        const int fin = isfinite(sci[k]) && isfinite(ref[k]);
        ok[k] = fin && sci[k] >= p->il && sci[k] <= p->iu && ref[k] >= p->tl && ref[k] <= p->tu && (!bpm || bpm[k] == 0);
        nok[k] = !ok[k];
        refz[k] = isfinite(ref[k]) ? ref[k] : 0.0;
        tvz[k] = isfinite(tvar[k]) ? tvar[k] : 0.0;
        diff[k] = p->fi;
        noise[k] = p->fin;
    }
    box_any(nok, nx, ny, hw, dirty);
    box_any(nok, nx, ny, hwk, outbad);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < ny; ++y)
        for (int x = 0; x < nx; ++x) {
            const size_t k = (size_t)y * nx + x;
            const int inside = y >= hw && y < ny - hw && x >= hw && x < nx - hw;
            elig[k] = !dirty[k] && inside;
            if (y < hwk || y >= ny - hwk || x < hwk || x >= nx - hwk) outbad[k] = 1;
        }
    int bgi[64], bgj[64], kpi[64], kpj[64];
    const int nbg = poly_terms(p->bgo, bgi, bgj), nkp = poly_terms(p->ko, kpi, kpj);
    const int nc = B.nc, nunk = 1 + (nc - 1) * nkp + nbg;
    double* x = (double*)malloc(sizeof(double) * nunk);
    int ri = 0;
    for (int ry = 0; ry < p->nry; ++ry)
        for (int rx = 0; rx < p->nrx; ++rx, ++ri) {
            if (only_region >= 0 && ri != only_region) { memset(&regions[ri], 0, sizeof(zo_hp_region)); continue; }
            int reg[4];
            reg[0] = rx * (nx / p->nrx);
            reg[1] = rx == p->nrx - 1 ? nx : (rx + 1) * (nx / p->nrx);
            reg[2] = ry * (ny / p->nry);
            reg[3] = ry == p->nry - 1 ? ny : (ry + 1) * (ny / p->nry);
            fit_region(sci, ref, svar, tvar, ok, elig, nx, ny, reg, &B, p, x, &regions[ri]);
            if (!regions[ri].solved) continue;
            const double xc = reg[0] + (reg[1] - reg[0]) / 2.0, hx = (reg[1] - reg[0]) / 2.0;
            const double yc = reg[2] + (reg[3] - reg[2]) / 2.0, hy = (reg[3] - reg[2]) / 2.0;
            const double* bgc = x + 1 + (size_t)(nc - 1) * nkp;
            const double norm = p->normalize ? 1.0 / x[0] : 1.0;
            const int nby = (reg[3] - reg[2] + step - 1) / step, nbx = (reg[1] - reg[0] + step - 1) / step;
#pragma omp parallel for schedule(dynamic) collapse(2)
            for (int jb = 0; jb < nby; ++jb)
                for (int ib = 0; ib < nbx; ++ib) {
                    const int gy = reg[2] + jb * step, gx = reg[0] + ib * step;
                    const int by = gy > hwk ? gy : hwk, bx = gx > hwk ? gx : hwk;
                    int ey = gy + step, ex = gx + step;
                    if (ey > reg[3]) ey = reg[3];
                    if (ey > ny - hwk) ey = ny - hwk;
                    if (ex > reg[1]) ex = reg[1];
                    if (ex > nx - hwk) ex = nx - hwk;
                    if (ey <= by || ex <= bx) continue;
                    const double fx = ((gx + hwk) - xc) / hx, fy = ((gy + hwk) - yc) / hy;
                    double c[64], K[64 * 64], K2[64 * 64];
                    c[0] = x[0];
                    for (int n = 1; n < nc; ++n) {
                        double s = 0.0;
                        for (int q = 0; q < nkp; ++q) s += x[1 + (size_t)(n - 1) * nkp + q] * (ipw(fx, kpi[q]) * ipw(fy, kpj[q]));
                        c[n] = s;
                    }
                    for (int i = 0; i < kw * kw; ++i) {
                        double s = 0.0;
                        for (int n = 0; n < nc; ++n) s += c[n] * B.k2[(size_t)n * kw * kw + i];
                        K[i] = s;
                        K2[i] = s * s;
                    }
                    for (int yy = by; yy < ey; ++yy)
                        for (int xx = bx; xx < ex; ++xx) {
                            const size_t k = (size_t)yy * nx + xx;
                            if (outbad[k]) continue;           /* (fill values are in place) */
                            double conv = 0.0, cvar = 0.0;
                            for (int dv = -hwk; dv <= hwk; ++dv) {
                                const double* tr = refz + (size_t)(yy - dv) * nx + xx;
                                const double* vr = tvz + (size_t)(yy - dv) * nx + xx;
                                const double* kr = K + (hwk + dv) * kw + hwk;
                                const double* k2r = K2 + (hwk + dv) * kw + hwk;
                                for (int du = -hwk; du <= hwk; ++du) {
                                    conv += kr[du] * tr[-du];
                                    cvar += k2r[du] * vr[-du];
                                }
                            }
                            const double xf = (xx - xc) / hx, yf = (yy - yc) / hy;
                            double bg = 0.0;
                            for (int q = 0; q < nbg; ++q) bg += bgc[q] * ipw(xf, bgi[q]) * ipw(yf, bgj[q]);
                            const double v = svar[k] + cvar;
                            diff[k] = (sci[k] - conv - bg) * norm;
                            noise[k] = sqrt(v > 0.0 ? v : 0.0) * fabs(norm);
                        }
                }
        }
    int64_t nmk = 0;
    for (size_t k = 0; k < np; ++k) nmk += diff[k] == p->fi;
    *nmasked = nmk;
    free(x); free(svar); free(tvar); free(refz); free(tvz); free(ok); free(nok); free(dirty); free(outbad); free(elig);
    free_basis(&B);
    return 0;
}
