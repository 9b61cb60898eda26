/* CPU restatement in C of the resample -> coadd leg of the hot path (oracle; TEST
 * INFRASTRUCTURE, not product code: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may build, load or call this).
 *
 * A line-by-line port of oracle/wcs.py (TAN / TPV, Newton inverse), oracle/resample.py
 * (Lanczos-3 / bilinear with SWarp's conventions) and oracle/combine.py (WEIGHTED /
 * CLIPPED / MEDIAN), i.e. of the operator the reference defines through
 * zuds/swarp.py:20-80 and zuds/astromatic/makecoadd/default.swarp:1-118 (call sites
 * zuds/coadd.py:133,156).  fp64 throughout, one OpenMP thread per output row block.
 * tests/test_oracle_cport.py pins it against the numpy oracle (1e-12).
 *
 *   gcc -O3 -fopenmp -shared -fPIC zm_oracle.c -o libzmoracle.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NPV 40
#define D2R (3.14159265358979323846 / 180.0)
#define SNAP 1e-5
#define BIGVAR 1e30
#define WEIGHT_THRESH 1e-30

typedef struct {
    double crpix[2], crval[2], cd[4], pv1[NPV], pv2[NPV];
    int32_t naxis[2], has_pv, pad_;
} zo_wcs;

/* exponents (x, y, r) of the TPV terms of axis 1; axis 2 swaps x and y */
static const int TPV[NPV][3] = {
    {0,0,0},{1,0,0},{0,1,0},{0,0,1},{2,0,0},{1,1,0},{0,2,0},{3,0,0},{2,1,0},{1,2,0},{0,3,0},{0,0,3},
    {4,0,0},{3,1,0},{2,2,0},{1,3,0},{0,4,0},{5,0,0},{4,1,0},{3,2,0},{2,3,0},{1,4,0},{0,5,0},{0,0,5},
    {6,0,0},{5,1,0},{4,2,0},{3,3,0},{2,4,0},{1,5,0},{0,6,0},{7,0,0},{6,1,0},{5,2,0},{4,3,0},{3,4,0},
    {2,5,0},{1,6,0},{0,7,0},{0,0,7}};

static double ipow(double x, int n) { double r = 1.0; while (n-- > 0) r *= x; return r; }

static void tpv_eval(const double* pv, double x, double y, double* f, double* fx, double* fy) {
    double r = sqrt(x * x + y * y), rs = r > 0 ? r : 1.0;
    double s = 0, sx = 0, sy = 0;
    for (int k = 0; k < NPV; ++k) {
        double p = pv[k];
        if (p == 0.0) continue;
        int a = TPV[k][0], b = TPV[k][1], c = TPV[k][2];
        if (c) {
            s += p * ipow(r, c);
            double g = p * c * ipow(r, c - 1) / rs;
            sx += g * x;
            sy += g * y;
        } else {
            s += p * ipow(x, a) * ipow(y, b);
            if (a) sx += p * a * ipow(x, a - 1) * ipow(y, b);
            if (b) sy += p * b * ipow(x, a) * ipow(y, b - 1);
        }
    }
    *f = s; *fx = sx; *fy = sy;
}

static void frame(const zo_wcs* w, double fr[9]) {
    double a0 = w->crval[0] * D2R, d0 = w->crval[1] * D2R;
    double sa = sin(a0), ca = cos(a0), sd = sin(d0), cd = cos(d0);
    fr[0] = -sa; fr[1] = ca; fr[2] = 0;
    fr[3] = -sd * ca; fr[4] = -sd * sa; fr[5] = cd;
    fr[6] = cd * ca; fr[7] = cd * sa; fr[8] = sd;
}

static void pix2plane(const zo_wcs* w, double x, double y, double* xi, double* eta) {
    double dx = x - w->crpix[0], dy = y - w->crpix[1];
    double u = w->cd[0] * dx + w->cd[1] * dy, v = w->cd[2] * dx + w->cd[3] * dy;
    if (!w->has_pv) { *xi = u; *eta = v; return; }
    double t1, t2;
    tpv_eval(w->pv1, u, v, xi, &t1, &t2);
    tpv_eval(w->pv2, v, u, eta, &t1, &t2);
}

static void plane2pix(const zo_wcs* w, double xi, double eta, double* x, double* y) {
    double u = xi, v = eta;
    if (w->has_pv) {
        u = (xi - w->pv1[0]) / w->pv1[1];
        v = (eta - w->pv2[0]) / w->pv2[1];
        for (int it = 0; it < 20; ++it) {
            double f, fu, fv, g, gv, gu;
            tpv_eval(w->pv1, u, v, &f, &fu, &fv);
            tpv_eval(w->pv2, v, u, &g, &gv, &gu);
            double rf = f - xi, rg = g - eta, det = fu * gv - fv * gu;
            double du = (rf * gv - rg * fv) / det, dv = (rg * fu - rf * gu) / det;
            u -= du; v -= dv;
            if (fabs(du) < 1e-13 && fabs(dv) < 1e-13) break;
        }
    }
    double det = w->cd[0] * w->cd[3] - w->cd[1] * w->cd[2];
    *x = (w->cd[3] * u - w->cd[1] * v) / det + w->crpix[0];
    *y = (-w->cd[2] * u + w->cd[0] * v) / det + w->crpix[1];
}

/* 0-based input positions of every output pixel (oracle.resample.positions) */
void zo_positions(const zo_wcs* wout, const zo_wcs* win, int onx, int ony, double* px, double* py) {
    double fo[9], fi[9], m[9];
    frame(wout, fo); frame(win, fi);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            m[3 * i + j] = fi[3 * i] * fo[3 * j] + fi[3 * i + 1] * fo[3 * j + 1] + fi[3 * i + 2] * fo[3 * j + 2];
#pragma omp parallel for schedule(static)
    for (int y = 0; y < ony; ++y)
        for (int x = 0; x < onx; ++x) {
            double xi, eta, xo, yo;
            pix2plane(wout, x + 1.0, y + 1.0, &xi, &eta);
            double xr = xi * D2R, er = eta * D2R;
            double a = m[0] * xr + m[1] * er + m[2], b = m[3] * xr + m[4] * er + m[5],
                   c = m[6] * xr + m[7] * er + m[8];
            plane2pix(win, a / c / D2R, b / c / D2R, &xo, &yo);
            px[(size_t)y * onx + x] = xo - 1.0;
            py[(size_t)y * onx + x] = yo - 1.0;
        }
}

static void split_pos(double p, long* i, double* d, int* delta) {
    double f = floor(p), fr = p - f;
    if (fr > 1.0 - SNAP) { f += 1.0; fr = 0.0; }
    *delta = fr < SNAP;
    *i = (long)f;
    *d = *delta ? 0.0 : fr;
}

static void lanczos3(double d, int delta, double t[6]) {
    if (delta) { for (int k = 0; k < 6; ++k) t[k] = (k == 2); return; }
    double s = 0;
    for (int k = 0; k < 6; ++k) {
        double x = d - (k - 2);
        t[k] = x == 0.0 ? M_PI * M_PI / 3.0 : sin(M_PI * x) * sin(M_PI * x / 3.0) / (x * x);
        s += t[k];
    }
    for (int k = 0; k < 6; ++k) t[k] /= s;
}

/* kind 3 = LANCZOS3, 1 = BILINEAR.  img / wgt float32 (wgt may be NULL), mask int32 or NULL.
 * out_img / out_wgt float64, out_mask int64 (oracle.resample.resample). */
void zo_resample(const float* img, const float* wgt, const int32_t* mask, int nx, int ny,
                 const double* px, const double* py, int onx, int ony, int kind, double fscale,
                 double* out_img, double* out_wgt, int64_t* out_mask) {
    const int nt = kind == 3 ? 6 : 2, off = kind == 3 ? -2 : 0;
#pragma omp parallel for schedule(dynamic, 8)
    for (int oy = 0; oy < ony; ++oy)
        for (int ox = 0; ox < onx; ++ox) {
            size_t o = (size_t)oy * onx + ox;
            long ix, iy;
            double dx, dy, tx[6], ty[6];
            int ddx, ddy;
            split_pos(px[o], &ix, &dx, &ddx);
            split_pos(py[o], &iy, &dy, &ddy);
            if (kind == 3) {
                lanczos3(ddx ? 0.5 : dx, ddx, tx);
                lanczos3(ddy ? 0.5 : dy, ddy, ty);
            } else {
                tx[0] = 1.0 - dx; tx[1] = dx; ty[0] = 1.0 - dy; ty[1] = dy;
            }
            long x0 = ix + off, y0 = iy + off;
            /* non-zero taps on the frame, per axis: the whole footprint, or the centre pixel of a
             * delta kernel (oracle.resample.on_frame) */
            int inbx = ddx ? (ix >= 0 && ix < nx) : (x0 >= 0 && x0 + nt <= nx);
            int inby = ddy ? (iy >= 0 && iy < ny) : (y0 >= 0 && y0 + nt <= ny);
            int inb = inbx && inby;
            double acc = 0, vacc = 0;
            int anybad = 0;
            int64_t macc = 0;
            if (inb)
                for (int r = 0; r < nt; ++r)
                    for (int c = 0; c < nt; ++c) {
                        double wt = ty[r] * tx[c];
                        if (wt == 0.0) continue;          /* (may lie off the frame next to a delta axis) */
                        size_t q = (size_t)(y0 + r) * nx + (x0 + c);
                        double var = 1.0;
                        int bad = 0;
                        if (wgt) {
                            double w = wgt[q];
                            if (w > WEIGHT_THRESH) var = 1.0 / w; else { var = BIGVAR; bad = 1; }
                        }
                        acc += wt * (double)img[q];
                        vacc += wt * (bad ? 0.0 : var);
                        if (wt != 0.0) {
                            anybad |= bad;
                            if (mask) macc |= mask[q];
                        }
                    }
            int good = inb && !anybad && vacc > 0;
            out_img[o] = good ? acc * fscale : 0.0;
            out_wgt[o] = good ? 1.0 / (vacc * fscale * fscale) : 0.0;
            if (out_mask) out_mask[o] = inb ? macc : 0;
        }
}

static int cmp_d(const void* a, const void* b) {
    double x = *(const double*)a, y = *(const double*)b;
    return (x > y) - (x < y);
}

/* kind 0 WEIGHTED, 1 MEDIAN, 2 CLIPPED, 3 AVERAGE; vals / wgts [n][npix] float64
 * (oracle.combine.combine) */
void zo_combine(const double* vals, const double* wgts, int n, int64_t npix, int kind,
                double clip_sigma, double clip_ampfrac, double* out, double* outw) {
#pragma omp parallel
    {
        double* buf = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
#pragma omp for schedule(static)
        for (int64_t p = 0; p < npix; ++p) {
            double s0 = 0, s1 = 0, wsum = 0;
            int m = 0;
            for (int i = 0; i < n; ++i) {
                double w = wgts[(size_t)i * npix + p];
                if (w > 0) { buf[m++] = vals[(size_t)i * npix + p]; wsum += w; }
            }
            if (kind == 0 || kind == 3) {
                for (int i = 0; i < n; ++i) {
                    double w = wgts[(size_t)i * npix + p];
                    if (w > 0) { double ww = kind == 0 ? w : 1.0; s0 += ww; s1 += ww * vals[(size_t)i * npix + p]; }
                }
                out[p] = s0 > 0 ? s1 / s0 : 0.0;
                outw[p] = wsum;
                continue;
            }
            double med = 0.0;
            if (m) {
                qsort(buf, (size_t)m, sizeof(double), cmp_d);
                med = 0.5 * (buf[(m - 1) / 2] + buf[m / 2]);
            }
            if (kind == 1) { out[p] = med; outw[p] = wsum; continue; }
            for (int i = 0; i < n; ++i) {
                double w = wgts[(size_t)i * npix + p];
                if (!(w > 0)) continue;
                double v = vals[(size_t)i * npix + p];
                if (fabs(v - med) <= clip_sigma / sqrt(w) + clip_ampfrac * fabs(med)) { s0 += w; s1 += w * v; }
            }
            out[p] = s0 > 0 ? s1 / s0 : 0.0;
            outw[p] = s0;
        }
        free(buf);
    }
}

void zo_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int zo_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---- mesh background (oracle/background.py, line by line) ------------------------------------
 * SWarp SUBTRACT_BACK Y / BACK_SIZE 128 / BACK_FILTERSIZE 3 (zuds/swarp.py:69,
 * zuds/astromatic/makecoadd/default.swarp:77-88) and SExtractor's -BACKGROUND / BACKGROUND_RMS
 * check-images (zuds/sextractor.py:21-26): per mesh 2-sigma pre-clip, quantised histogram over
 * +-5 sigma, iterated +-3 sigma clipping around the histogram median, mode; bad meshes filled
 * from the nearest good ones; fsize x fsize median; natural bicubic spline to full resolution. */
#define ZB_BIG 1e30
#define ZB_NMAXLEVELS 4096

static double zb_median_walk(const int64_t* histo, int nlevels, int lcut, int hcut) {
    int64_t lowsum = 0, highsum = 0;
    int lo = lcut, hi = hcut;
    for (int k = lcut; k <= hcut; ++k) {
        if (lowsum < highsum) lowsum += histo[lo++];
        else highsum += histo[hi--];
    }
    if (hi < 0) return 0.0;
    int64_t a = lo < nlevels ? histo[lo] : 0, b = histo[hi];
    double den = 2.0 * (double)(a > b ? a : b);
    double frac = den > 0 ? (double)(highsum - lowsum) / den : 0.0;
    return hi + 0.5 + frac;
}

/* one mesh: pix = its valid pixels (n of them); returns 0 when the mesh has no estimate */
static int zb_mesh(const double* pix, int n, int64_t* histo, double* mode, double* sigma) {
    if (n == 0) return 0;
    double s = 0, s2 = 0;
    for (int i = 0; i < n; ++i) s += pix[i];
    double mean = s / n;
    for (int i = 0; i < n; ++i) s2 += (pix[i] - mean) * (pix[i] - mean);
    double var = s2 / n, sig = var > 0 ? sqrt(var) : 0.0;
    double lcut = mean - 2.0 * sig, hcut = mean + 2.0 * sig;
    int npix = 0;
    s = 0;
    for (int i = 0; i < n; ++i) if (pix[i] >= lcut && pix[i] <= hcut) { s += pix[i]; ++npix; }
    if (npix == 0) return 0;
    mean = s / npix;
    s2 = 0;
    for (int i = 0; i < n; ++i) if (pix[i] >= lcut && pix[i] <= hcut) s2 += (pix[i] - mean) * (pix[i] - mean);
    var = s2 / npix;
    sig = var > 0 ? sqrt(var) : 0.0;
    double step = sqrt(2.0 / 3.14159265358979323846) * 5 / 4;
    int nlevels = (int)(step * npix + 1);
    if (nlevels > ZB_NMAXLEVELS) nlevels = ZB_NMAXLEVELS;
    float mean32 = (float)mean, sig32 = (float)sig;
    float qscale = sig32 > 0 ? (float)(2.0 * 5 * (double)sig32 / nlevels) : 1.0f;
    float qzero = (float)((double)mean32 - 5 * (double)sig32);
    float cste = (float)(0.499999 - (double)(qzero / qscale));
    memset(histo, 0, sizeof(int64_t) * (size_t)nlevels);
    int64_t total = 0;
    for (int i = 0; i < n; ++i) {
        float q = (float)pix[i] / qscale + cste;
        double t = trunc((double)q);
        if (t >= 0 && t < nlevels) { histo[(int)t]++; ++total; }
    }
    if (total == 0) return 0;
    /* backguess */
    int nlm1 = nlevels - 1, lc = 0, hc = nlm1;
    double sg = 10.0 * nlm1, sg1 = 1.0, mea = mean32, med = mean32;
    for (int it = 100; it > 0 && sg >= 0.1 && fabs(sg / sg1 - 1.0) > 1e-4; --it) {
        sg1 = sg;
        double hs = 0, hm = 0, hq = 0;
        for (int k = lc; k <= hc; ++k) { double h = (double)histo[k]; hs += h; hm += h * k; hq += h * k * (double)k; }
        mea = hm;
        sg = hq;
        med = zb_median_walk(histo, nlevels, lc, hc);
        if (hs > 0) { mea /= hs; sg = sg / hs - mea * mea; }
        sg = sg > 0 ? sqrt(sg) : 0.0;
        double ft = med - 3.0 * sg;
        lc = ft > 0 ? (int)(ft + 0.5) : 0;
        ft = med + 3.0 * sg;
        hc = ft < nlm1 ? (ft > 0 ? (int)(ft + 0.5) : (int)(ft - 0.5)) : nlm1;
    }
    double qz = qzero, qs = qscale;
    if (sg > 0) *mode = fabs((mea - med) / sg) < 0.3 ? qz + (2.5 * med - 1.5 * mea) * qs : qz + med * qs;
    else *mode = qz + mea * qs;
    *sigma = sg * qs;
    return 1;
}

static double zb_fqmedian(double* v, int n) {
    if (n == 0) return 0.0;
    qsort(v, (size_t)n, sizeof(double), cmp_d);
    return (n & 1) ? v[n / 2] : 0.5 * (v[n / 2 - 1] + v[n / 2]);
}

/* natural cubic spline second derivatives / 6 of `n` values with stride `st` */
static void zb_derivs(const double* a, int n, int st, double* d) {
    for (int i = 0; i < n; ++i) d[i] = 0.0;
    if (n < 3) return;
    double* u = (double*)calloc((size_t)n, sizeof(double));
    for (int y = 1; y < n - 1; ++y) {
        double temp = -1.0 / (d[y - 1] + 4.0);
        d[y] = temp;
        u[y] = temp * (u[y - 1] - 6.0 * (a[(size_t)(y + 1) * st] + a[(size_t)(y - 1) * st] - 2.0 * a[(size_t)y * st]));
    }
    d[n - 1] = 0.0;
    for (int y = n - 2; y > 0; --y) d[y] = d[y] * d[y + 1] + u[y];
    d[0] = 0.0;
    for (int i = 0; i < n; ++i) d[i] /= 6.0;
    free(u);
}

static void zb_expand(const double* nodes, int nbx, int nby, int nx, int ny, int mesh, double* out) {
    /* spline along y per node column -> rows [ny][nbx]; then along x per image row */
    double* rows = (double*)malloc(sizeof(double) * (size_t)ny * nbx);
    double* d = (double*)malloc(sizeof(double) * (size_t)(nby > nbx ? nby : nbx));
    for (int i = 0; i < nbx; ++i) {
        zb_derivs(nodes + i, nby, nbx, d);
        for (int y = 0; y < ny; ++y) {
            if (nby < 2) { rows[(size_t)y * nbx + i] = nodes[i]; continue; }
            double t = (y + 0.5) / mesh - 0.5;
            int j0 = (int)floor(t);
            if (j0 < 0) j0 = 0;
            if (j0 > nby - 2) j0 = nby - 2;
            double dy = t - j0, dy1 = 1.0 - dy, cdy = dy * dy * dy - dy, cdy1 = dy1 * dy1 * dy1 - dy1;
            rows[(size_t)y * nbx + i] = dy1 * nodes[(size_t)j0 * nbx + i] + dy * nodes[(size_t)(j0 + 1) * nbx + i] +
                                        cdy1 * d[j0] + cdy * d[j0 + 1];
        }
    }
    free(d);
#pragma omp parallel
    {
        double* dx = (double*)malloc(sizeof(double) * (size_t)nbx);
#pragma omp for schedule(static)
        for (int y = 0; y < ny; ++y) {
            const double* r = rows + (size_t)y * nbx;
            zb_derivs(r, nbx, 1, dx);
            for (int x = 0; x < nx; ++x) {
                if (nbx < 2) { out[(size_t)y * nx + x] = r[0]; continue; }
                double t = (x + 0.5) / mesh - 0.5;
                int i0 = (int)floor(t);
                if (i0 < 0) i0 = 0;
                if (i0 > nbx - 2) i0 = nbx - 2;
                double ddx = t - i0, dx1 = 1.0 - ddx, cdx = ddx * ddx * ddx - ddx, cdx1 = dx1 * dx1 * dx1 - dx1;
                out[(size_t)y * nx + x] = dx1 * r[i0] + ddx * r[i0 + 1] + cdx1 * dx[i0] + cdx * dx[i0 + 1];
            }
        }
        free(dx);
    }
    free(rows);
}

/* oracle.background.background: bkg / rms [ny][nx] float64 (either may be NULL), stats = {backmean,
 * backsig}, nodes_b / nodes_s [nby][nbx] (may be NULL).  img float64 [ny][nx]; wgt float32 or NULL. */
void zo_background(const double* img, const float* wgt, int nx, int ny, int mesh, int fsize, double* bkg,
                   double* rms, double* stats, double* nodes_b, double* nodes_s) {
    const int nbx = (nx - 1) / mesh + 1, nby = (ny - 1) / mesh + 1, nm = nbx * nby;
    double* back = (double*)malloc(sizeof(double) * (size_t)nm);
    double* sigm = (double*)malloc(sizeof(double) * (size_t)nm);
#pragma omp parallel
    {
        double* pix = (double*)malloc(sizeof(double) * (size_t)mesh * mesh);
        int64_t* histo = (int64_t*)malloc(sizeof(int64_t) * ZB_NMAXLEVELS);
#pragma omp for schedule(dynamic)
        for (int m = 0; m < nm; ++m) {
            const int j = m / nbx, i = m - j * nbx;
            const int y0 = j * mesh, y1 = (j + 1) * mesh < ny ? (j + 1) * mesh : ny;
            const int x0 = i * mesh, x1 = (i + 1) * mesh < nx ? (i + 1) * mesh : nx;
            int n = 0;
            for (int y = y0; y < y1; ++y)
                for (int x = x0; x < x1; ++x) {
                    const double p = img[(size_t)y * nx + x];
                    if (p > -ZB_BIG && (!wgt || wgt[(size_t)y * nx + x] > (float)WEIGHT_THRESH)) pix[n++] = p;
                }
            back[m] = sigm[m] = -ZB_BIG;
            if (n < (y1 - y0) * (x1 - x0) * 0.5) continue;
            double mo, sg;
            if (zb_mesh(pix, n, histo, &mo, &sg)) { back[m] = mo; sigm[m] = sg; }
        }
        free(pix);
        free(histo);
    }
    /* fill bad meshes from the nearest good ones */
    double* b2 = (double*)malloc(sizeof(double) * (size_t)nm);
    double* s2 = (double*)malloc(sizeof(double) * (size_t)nm);
    int ngood = 0;
    for (int m = 0; m < nm; ++m) ngood += back[m] > -ZB_BIG;
    for (int j = 0; j < nby; ++j)
        for (int i = 0; i < nbx; ++i) {
            const int m = j * nbx + i;
            b2[m] = back[m];
            s2[m] = sigm[m];
            if (back[m] > -ZB_BIG) continue;
            if (!ngood) { b2[m] = 0.0; s2[m] = 1.0; continue; }
            long best = -1;
            for (int q = 0; q < nm; ++q)
                if (back[q] > -ZB_BIG) {
                    long dd = (long)(q % nbx - i) * (q % nbx - i) + (long)(q / nbx - j) * (q / nbx - j);
                    if (best < 0 || dd < best) best = dd;
                }
            double sb = 0, ss = 0;
            int cnt = 0;
            for (int q = 0; q < nm; ++q)
                if (back[q] > -ZB_BIG) {
                    long dd = (long)(q % nbx - i) * (q % nbx - i) + (long)(q / nbx - j) * (q / nbx - j);
                    if (dd == best) { sb += back[q]; ss += sigm[q]; ++cnt; }
                }
            b2[m] = sb / cnt;
            s2[m] = ss / cnt;
        }
    /* fsize x fsize median */
    double* bo = (double*)malloc(sizeof(double) * (size_t)nm);
    double* so = (double*)malloc(sizeof(double) * (size_t)nm);
    const int hb = fsize / 2;
    double* tmp = (double*)malloc(sizeof(double) * (size_t)(fsize * fsize > nm ? fsize * fsize : nm));
    for (int j = 0; j < nby; ++j)
        for (int i = 0; i < nbx; ++i) {
            if (fsize <= 1) { bo[j * nbx + i] = b2[j * nbx + i]; so[j * nbx + i] = s2[j * nbx + i]; continue; }
            int n = 0;
            for (int y = (j - hb > 0 ? j - hb : 0); y <= (j + hb < nby - 1 ? j + hb : nby - 1); ++y)
                for (int x = (i - hb > 0 ? i - hb : 0); x <= (i + hb < nbx - 1 ? i + hb : nbx - 1); ++x) tmp[n++] = b2[y * nbx + x];
            bo[j * nbx + i] = zb_fqmedian(tmp, n);
            n = 0;
            for (int y = (j - hb > 0 ? j - hb : 0); y <= (j + hb < nby - 1 ? j + hb : nby - 1); ++y)
                for (int x = (i - hb > 0 ? i - hb : 0); x <= (i + hb < nbx - 1 ? i + hb : nbx - 1); ++x) tmp[n++] = s2[y * nbx + x];
            so[j * nbx + i] = zb_fqmedian(tmp, n);
        }
    if (bkg) zb_expand(bo, nbx, nby, nx, ny, mesh, bkg);
    if (rms) zb_expand(so, nbx, nby, nx, ny, mesh, rms);
    if (nodes_b) memcpy(nodes_b, bo, sizeof(double) * (size_t)nm);
    if (nodes_s) memcpy(nodes_s, so, sizeof(double) * (size_t)nm);
    if (stats) {
        memcpy(tmp, bo, sizeof(double) * (size_t)nm);
        stats[0] = zb_fqmedian(tmp, nm);
        memcpy(tmp, so, sizeof(double) * (size_t)nm);
        stats[1] = zb_fqmedian(tmp, nm);
    }
    free(tmp); free(bo); free(so); free(b2); free(s2); free(back); free(sigm);
}
