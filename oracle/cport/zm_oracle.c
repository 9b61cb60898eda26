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
