// Host-side fp64 WCS helpers of the C-ABI (no device work).
#include <algorithm>
#include <cmath>
#include <vector>

#include "zm_internal.h"
#include "wcs_math.h"

static const double D2R = 0.017453292519943295;

void zm_wcs_frame(const zm_wcs* w, double fr[9]) {
    double a0 = w->crval[0] * D2R, d0 = w->crval[1] * D2R;
    double sa = sin(a0), ca = cos(a0), sd = sin(d0), cd = cos(d0);
    fr[0] = -sa;      fr[1] = ca;       fr[2] = 0.0;   // east
    fr[3] = -sd * ca; fr[4] = -sd * sa; fr[5] = cd;    // north
    fr[6] = cd * ca;  fr[7] = cd * sa;  fr[8] = sd;    // pole (towards CRVAL)
}

// the polynomial order of a TPV WCS into bits 8 - 11 of the flags of a PRIVATE copy (bit 12: radial terms): what
// zm_tpv_eval leaves out (zm_wcs.flags of a caller's struct only ever carries bit 0)
static void mark_order(zm_wcs* w) {
    w->flags &= 1;
    if (!(w->flags & 1)) return;
    const int o1 = zm_tpv_order(w->pv1), o2 = zm_tpv_order(w->pv2);
    const int order = o1 > o2 ? o1 : o2;
    const bool radial = w->pv1[3] != 0.0 || w->pv1[11] != 0.0 || w->pv1[23] != 0.0 || w->pv1[39] != 0.0 ||
                        w->pv2[3] != 0.0 || w->pv2[11] != 0.0 || w->pv2[23] != 0.0 || w->pv2[39] != 0.0;
    w->flags |= (order << 8) | (radial ? 1 << 12 : 0);
}

void zm_make_map(const zm_wcs* wout, const zm_wcs* win, zm_map_params* mp) {
    mp->wout = *wout;
    mp->win = *win;
    mark_order(&mp->wout);
    mark_order(&mp->win);
    double fo[9], fi[9];
    zm_wcs_frame(wout, fo);
    zm_wcs_frame(win, fi);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += fi[r * 3 + k] * fo[c * 3 + k];
            mp->rot[r * 3 + c] = s;
        }
}

void zm_map_point(const zm_map_params* mp, double xo, double yo, double* xi, double* yi) {
    zm_map_out_to_in(&mp->wout, &mp->win, mp->rot, xo, yo, xi, yi);
}

double zm_pixel_area(const zm_wcs* w, double x, double y) {
    const double h = 0.5;
    double x1, e1, x0, e0, x3, e3, x2, e2;
    zm_pix2plane(w, x + h, y, &x1, &e1);
    zm_pix2plane(w, x - h, y, &x0, &e0);
    zm_pix2plane(w, x, y + h, &x3, &e3);
    zm_pix2plane(w, x, y - h, &x2, &e2);
    return fabs((x1 - x0) * (e3 - e2) - (x3 - x2) * (e1 - e0));
}

static void pix2vec(const zm_wcs* w, const double* fr, double x, double y, double v[3]) {
    double xi, eta;
    zm_pix2plane(w, x, y, &xi, &eta);
    double xr = xi * D2R, er = eta * D2R;
    double n = 0.0;
    for (int k = 0; k < 3; ++k) {
        v[k] = xr * fr[k] + er * fr[3 + k] + fr[6 + k];
        n += v[k] * v[k];
    }
    n = sqrt(n);
    for (int k = 0; k < 3; ++k) v[k] /= n;
}

static void vec2pix(const zm_wcs* w, const double* fr, const double v[3], double* x, double* y) {
    double a = 0, b = 0, c = 0;
    for (int k = 0; k < 3; ++k) {
        a += v[k] * fr[k];
        b += v[k] * fr[3 + k];
        c += v[k] * fr[6 + k];
    }
    zm_plane2pix(w, a / c / D2R, b / c / D2R, x, y);
}

extern "C" int zm_wcs_pix2sky(const zm_wcs* w, int n, const double* x, const double* y,
                              double* ra, double* dec) {
    ZM_CHECK(w && x && y && ra && dec, "zm_wcs_pix2sky: null argument");
    double fr[9];
    zm_wcs_frame(w, fr);
    for (int i = 0; i < n; ++i) {
        double v[3];
        pix2vec(w, fr, x[i], y[i], v);
        double a = atan2(v[1], v[0]) / D2R;
        if (a < 0) a += 360.0;
        ra[i] = a;
        dec[i] = asin(std::max(-1.0, std::min(1.0, v[2]))) / D2R;
    }
    return 0;
}

extern "C" int zm_wcs_sky2pix(const zm_wcs* w, int n, const double* ra, const double* dec,
                              double* x, double* y) {
    ZM_CHECK(w && x && y && ra && dec, "zm_wcs_sky2pix: null argument");
    double fr[9];
    zm_wcs_frame(w, fr);
    for (int i = 0; i < n; ++i) {
        double a = ra[i] * D2R, d = dec[i] * D2R;
        double v[3] = {cos(d) * cos(a), cos(d) * sin(a), sin(d)};
        vec2pix(w, fr, v, &x[i], &y[i]);
    }
    return 0;
}

extern "C" int zm_wcs_map(const zm_wcs* wout, const zm_wcs* win, int n, const double* xo,
                          const double* yo, double* xi, double* yi) {
    ZM_CHECK(wout && win && xo && yo && xi && yi, "zm_wcs_map: null argument");
    zm_map_params mp;
    zm_make_map(wout, win, &mp);
    for (int i = 0; i < n; ++i) zm_map_point(&mp, xo[i], yo[i], &xi[i], &yi[i]);
    return 0;
}

extern "C" int zm_flux_scale(const zm_wcs* win, const zm_wcs* wout, double flxscale,
                             double* out) {
    ZM_CHECK(win && wout && out, "zm_flux_scale: null argument");
    double cx = (win->naxis[0] + 1) / 2.0, cy = (win->naxis[1] + 1) / 2.0;
    double a_in = zm_pixel_area(win, cx, cy);
    zm_map_params mp;
    zm_make_map(win, wout, &mp);   // input pixel -> output pixel
    double xo, yo;
    zm_map_point(&mp, cx, cy, &xo, &yo);
    double a_out = zm_pixel_area(wout, xo, yo);
    ZM_CHECK(a_in > 0.0, "zm_flux_scale: degenerate input WCS");
    *out = flxscale * a_out / a_in;
    return 0;
}

// Automatic output grid; convention documented in oracle/grid.py.
extern "C" int zm_autogrid(int nframes, const zm_wcs* wcs, zm_wcs* out) {
    ZM_CHECK(nframes > 0 && wcs && out, "zm_autogrid: need at least one frame");
    std::vector<double> ras, decs, scales;
    double ra0 = wcs[0].crval[0];
    for (int f = 0; f < nframes; ++f) {
        const zm_wcs* w = &wcs[f];
        int nx = w->naxis[0], ny = w->naxis[1];
        ZM_CHECK(nx > 0 && ny > 0, "zm_autogrid: frame %d has no NAXIS", f);
        double fr[9];
        zm_wcs_frame(w, fr);
        std::vector<double> xs, ys;
        for (double x = 0.5; x < nx + 0.5; x += 64.0) xs.push_back(x);
        xs.push_back(nx + 0.5);
        for (double y = 0.5; y < ny + 0.5; y += 64.0) ys.push_back(y);
        ys.push_back(ny + 0.5);
        auto add = [&](double x, double y) {
            double v[3];
            pix2vec(w, fr, x, y, v);
            double a = atan2(v[1], v[0]) / D2R;
            a = fmod(a - ra0 + 180.0 + 720.0, 360.0) - 180.0 + ra0;
            ras.push_back(a);
            decs.push_back(asin(std::max(-1.0, std::min(1.0, v[2]))) / D2R);
        };
        for (double x : xs) { add(x, 0.5); add(x, ny + 0.5); }
        for (double y : ys) { add(0.5, y); add(nx + 0.5, y); }
        scales.push_back(sqrt(zm_pixel_area(w, (nx + 1) / 2.0, (ny + 1) / 2.0)));
    }
    double ramin = *std::min_element(ras.begin(), ras.end());
    double ramax = *std::max_element(ras.begin(), ras.end());
    double demin = *std::min_element(decs.begin(), decs.end());
    double demax = *std::max_element(decs.begin(), decs.end());
    std::sort(scales.begin(), scales.end());
    size_t ns = scales.size();
    double s = (ns & 1) ? scales[ns / 2] : 0.5 * (scales[ns / 2 - 1] + scales[ns / 2]);
    memset(out, 0, sizeof(*out));
    double cra = fmod(0.5 * (ramin + ramax) + 360.0, 360.0);
    out->crval[0] = cra;
    out->crval[1] = 0.5 * (demin + demax);
    out->cd[0] = -s; out->cd[1] = 0.0; out->cd[2] = 0.0; out->cd[3] = s;
    out->pv1[1] = 1.0;
    out->pv2[1] = 1.0;
    double fr[9];
    zm_wcs_frame(out, fr);
    double xmin = 1e300, xmax = -1e300, ymin = 1e300, ymax = -1e300;
    for (size_t i = 0; i < ras.size(); ++i) {
        double a = ras[i] * D2R, d = decs[i] * D2R;
        double v[3] = {cos(d) * cos(a), cos(d) * sin(a), sin(d)};
        double x, y;
        vec2pix(out, fr, v, &x, &y);
        xmin = std::min(xmin, x); xmax = std::max(xmax, x);
        ymin = std::min(ymin, y); ymax = std::max(ymax, y);
    }
    double ex = xmax - xmin, ey = ymax - ymin;
    int nx = std::max((int)ceil(ex - 1e-9), 1), ny = std::max((int)ceil(ey - 1e-9), 1);
    out->crpix[0] = 0.5 - xmin + 0.5 * (nx - ex);
    out->crpix[1] = 0.5 - ymin + 0.5 * (ny - ey);
    out->naxis[0] = nx;
    out->naxis[1] = ny;
    return 0;
}
