// TAN / TPV projection math in fp64, usable from host and device code.
// Pixel coordinates are FITS 1-based.  See oracle/wcs.py for the restatement
// these follow (Calabretta & Greisen 2002; TPV registry convention).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include "../../include/zudsmi.h"

#define ZM_HD __host__ __device__ inline

// One TPV polynomial: value and partials wrt (x, y).  Axis 2 is evaluated by
// swapping the roles of x and y at the call site.
// `order` (round 4): the highest total degree with a non-zero coefficient, 1 .. 7 (0: not known, all forty terms);
// zm_make_map works it out once per WCS and keeps it in bits 8 - 11 of the flags of its private copies (bit 12:
// radial terms present).  The terms are added in the order of the full expression, degree by degree, so leaving
// out the degrees whose coefficients are all zero leaves the sums as they were (x + 0 = x): a ZTF header stops at
// the third degree - ten terms per polynomial instead of forty in every Newton step of every lattice node.
ZM_HD void zm_tpv_eval(const double* pv, double x, double y, double* f,
                       double* fx, double* fy, int order = 0, bool radial = true) {
    if (order <= 0) order = 7;
    double x2 = x * x, y2 = y * y, xy = x * y;
    double v = pv[0] + pv[1] * x + pv[2] * y;
    double dx = pv[1];
    double dy = pv[2];
    if (order >= 2) {
        v = v + pv[4] * x2 + pv[5] * xy + pv[6] * y2;
        dx = dx + 2 * pv[4] * x + pv[5] * y;
        dy = dy + pv[5] * x + 2 * pv[6] * y;
    }
    double x3 = x2 * x, y3 = y2 * y;
    if (order >= 3) {
        v = v + pv[7] * x3 + pv[8] * x2 * y + pv[9] * x * y2 + pv[10] * y3;
        dx = dx + 3 * pv[7] * x2 + 2 * pv[8] * xy + pv[9] * y2;
        dy = dy + pv[8] * x2 + 2 * pv[9] * xy + 3 * pv[10] * y2;
    }
    if (order >= 4) {
        double x4 = x2 * x2, y4 = y2 * y2;
        v = v + pv[12] * x4 + pv[13] * x3 * y + pv[14] * x2 * y2 + pv[15] * x * y3 + pv[16] * y4;
        dx = dx + 4 * pv[12] * x3 + 3 * pv[13] * x2 * y + 2 * pv[14] * x * y2 + pv[15] * y3;
        dy = dy + pv[13] * x3 + 2 * pv[14] * x2 * y + 3 * pv[15] * x * y2 + 4 * pv[16] * y3;
        if (order >= 5) {
            double x5 = x4 * x, y5 = y4 * y;
            v = v + pv[17] * x5 + pv[18] * x4 * y + pv[19] * x3 * y2 + pv[20] * x2 * y3
                  + pv[21] * x * y4 + pv[22] * y5;
            dx = dx + 5 * pv[17] * x4 + 4 * pv[18] * x3 * y + 3 * pv[19] * x2 * y2
                    + 2 * pv[20] * x * y3 + pv[21] * y4;
            dy = dy + pv[18] * x4 + 2 * pv[19] * x3 * y + 3 * pv[20] * x2 * y2
                    + 4 * pv[21] * x * y3 + 5 * pv[22] * y4;
            if (order >= 6) {
                double x6 = x3 * x3, y6 = y3 * y3;
                v = v + pv[24] * x6 + pv[25] * x5 * y + pv[26] * x4 * y2 + pv[27] * x3 * y3
                      + pv[28] * x2 * y4 + pv[29] * x * y5 + pv[30] * y6;
                dx = dx + 6 * pv[24] * x5 + 5 * pv[25] * x4 * y + 4 * pv[26] * x3 * y2
                        + 3 * pv[27] * x2 * y3 + 2 * pv[28] * x * y4 + pv[29] * y5;
                dy = dy + pv[25] * x5 + 2 * pv[26] * x4 * y + 3 * pv[27] * x3 * y2
                        + 4 * pv[28] * x2 * y3 + 5 * pv[29] * x * y4 + 6 * pv[30] * y5;
                if (order >= 7) {
                    double x7 = x6 * x, y7 = y6 * y;
                    v = v + pv[31] * x7 + pv[32] * x6 * y + pv[33] * x5 * y2 + pv[34] * x4 * y3
                          + pv[35] * x3 * y4 + pv[36] * x2 * y5 + pv[37] * x * y6 + pv[38] * y7;
                    dx = dx + 7 * pv[31] * x6 + 6 * pv[32] * x5 * y + 5 * pv[33] * x4 * y2
                            + 4 * pv[34] * x3 * y3 + 3 * pv[35] * x2 * y4 + 2 * pv[36] * x * y5 + pv[37] * y6;
                    dy = dy + pv[32] * x6 + 2 * pv[33] * x5 * y + 3 * pv[34] * x4 * y2
                            + 4 * pv[35] * x3 * y3 + 5 * pv[36] * x2 * y4 + 6 * pv[37] * x * y5 + 7 * pv[38] * y6;
                }
            }
        }
    }
    if (radial && (pv[3] != 0.0 || pv[11] != 0.0 || pv[23] != 0.0 || pv[39] != 0.0)) {
        double r2 = x2 + y2;
        double r = sqrt(r2);
        double rs = r > 0.0 ? r : 1.0;
        double r4 = r2 * r2, r6 = r4 * r2;
        v += pv[3] * r + pv[11] * r2 * r + pv[23] * r4 * r + pv[39] * r6 * r;
        double g = (pv[3] + 3 * pv[11] * r2 + 5 * pv[23] * r4 + 7 * pv[39] * r6) / rs;
        dx += g * x;
        dy += g * y;
    }
    *f = v;
    *fx = dx;
    *fy = dy;
}

// polynomial order of a TPV axis (see zm_tpv_eval): highest degree with a non-zero coefficient, at least 1
ZM_HD int zm_tpv_order(const double* pv) {
    const int first[8] = {0, 1, 4, 7, 12, 17, 24, 31}, last[8] = {0, 2, 6, 10, 16, 22, 30, 38};
    int order = 1;
    for (int d = 2; d <= 7; ++d)
        for (int k = first[d]; k <= last[d]; ++k)
            if (pv[k] != 0.0) order = d;
    return order;
}
#define ZM_WCS_ORDER(w) (((w)->flags >> 8) & 15)
#define ZM_WCS_RADIAL(w) ((((w)->flags >> 8) & 15) == 0 || (((w)->flags >> 12) & 1))

ZM_HD void zm_pix2plane(const zm_wcs* w, double x, double y, double* xi, double* eta) {
    double dx = x - w->crpix[0], dy = y - w->crpix[1];
    double u = w->cd[0] * dx + w->cd[1] * dy;
    double v = w->cd[2] * dx + w->cd[3] * dy;
    if (w->flags & 1) {
        double a, b, c;
        zm_tpv_eval(w->pv1, u, v, xi, &a, &b, ZM_WCS_ORDER(w), ZM_WCS_RADIAL(w));
        zm_tpv_eval(w->pv2, v, u, eta, &c, &a, ZM_WCS_ORDER(w), ZM_WCS_RADIAL(w));
    } else {
        *xi = u;
        *eta = v;
    }
}

ZM_HD void zm_plane2pix(const zm_wcs* w, double xi, double eta, double* x, double* y) {
    double u = xi, v = eta;
    if (w->flags & 1) {
        u = (xi - w->pv1[0]) / w->pv1[1];
        v = (eta - w->pv2[0]) / w->pv2[1];
        for (int it = 0; it < 20; ++it) {
            double f, fu, fv, g, gv, gu;
            zm_tpv_eval(w->pv1, u, v, &f, &fu, &fv, ZM_WCS_ORDER(w), ZM_WCS_RADIAL(w));
            zm_tpv_eval(w->pv2, v, u, &g, &gv, &gu, ZM_WCS_ORDER(w), ZM_WCS_RADIAL(w));
            double rf = f - xi, rg = g - eta;
            double det = fu * gv - fv * gu;
            double du = (rf * gv - rg * fv) / det;
            double dv = (rg * fu - rf * gu) / det;
            u -= du;
            v -= dv;
            if (fabs(du) < 1e-13 && fabs(dv) < 1e-13) break;
        }
    }
    double det = w->cd[0] * w->cd[3] - w->cd[1] * w->cd[2];
    *x = (w->cd[3] * u - w->cd[1] * v) / det + w->crpix[0];
    *y = (-w->cd[2] * u + w->cd[0] * v) / det + w->crpix[1];
}

// Output pixel (1-based) -> input pixel (1-based) through the two tangent
// frames; rot = in-frame axes in the out-frame basis (row major 3x3).
ZM_HD void zm_map_out_to_in(const zm_wcs* wout, const zm_wcs* win, const double* rot,
                            double xo, double yo, double* xi_pix, double* yi_pix) {
    const double d2r = 0.017453292519943295;
    double xi, eta;
    zm_pix2plane(wout, xo, yo, &xi, &eta);
    double xr = xi * d2r, er = eta * d2r;
    double a = rot[0] * xr + rot[1] * er + rot[2];
    double b = rot[3] * xr + rot[4] * er + rot[5];
    double c = rot[6] * xr + rot[7] * er + rot[8];
    zm_plane2pix(win, a / c / d2r, b / c / d2r, xi_pix, yi_pix);
}
