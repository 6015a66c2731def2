// diffusive_core.hpp -- diffusive-wave routing of a mainstem network (Crank-Nicolson + Hermite interpolation
// for the flow, a Newton/bisection depth solve per node), precision double, host/device neutral.
//
// Semantics follow the reference Fortran (paths relative to the T-Route tree), synthetic cross sections only:
//   src/kernel/diffusive/diffusive.f90
//     diffnw :75-940 (set-up :261-560, ordered time loop :632-870), calculateDT :942-991,
//     mesh_diffusive_forward :1108-1355, mesh_diffusive_backward :1357-1553, rtsafe :1555-1662,
//     funcd_diffdepth :1664-1711, intp_xsec_tab :1713-1754, readXsection :2093-2443 (RouteLink trapezoid +
//     flood plain), the geometry helpers :2445-2520, r_interpol :2553-2594, LInterpol / intp_y / locate :2650-2755
// The boundary is the bind(c) symbol c_diffnw (src/kernel/diffusive/pydiffusive.f90:8-55): same arguments,
// Fortran (column-major) arrays.
//
// Natural cross sections (mxnbathy_g > 0, readXsection_natural_mann_vertices :1756-2091) are covered as well.
// The mapping of results from a refactored hydrofabric back to the original one (cwnrow_g > 0, :849-920) is
// crosswalk_instant() at the end of this file.
// The streamflow-DA branch is commented out in the reference itself (:1301-1327).
//
// Layout of the computation (not of the reference's call tree):
//   * tables     per node, 501 water elevations x the 8 hydraulic columns the solver reads; independent per
//                (node, level): table_row() -- the only data-parallel part, one thread per (node, level);
//   * the ordered time loop is a recurrence over (sub-step, reach, node): solve() runs it in one thread.
// Single-precision literals of the Fortran source that are not exactly representable are written as float
// literals here (0.01f, 0.3f, ...): they enter the double arithmetic with their single-precision value.
#pragma once

#include <math.h>
#include <stdint.h>

#include "det_pow64.h"

#if defined(__HIPCC__)
#define DW_HD __host__ __device__
#else
#define DW_HD
#endif

// x**y for the solver's fractional exponents: the C library's pow on the host (what the reference Fortran calls);
// on the device the restatement of that library's algorithm (det_pow64.h), so that both give the same bits.
#if defined(__HIP_DEVICE_COMPILE__)
#define DW_POW(x, y) det_pow64((x), (y))
#else
#define DW_POW(x, y) pow((x), (y))
#endif

namespace trdw {

constexpr int kNel = 501;   // rows of every lookup table (diffnw :275)
constexpr int kCols = 8;    // columns kept per table row
// column ids of the reference's xsec_tab (1 elevation, 2 area, 3 perimeter, 5 conveyance, 6 top width,
// 9 dK/dA, 10 uniform flow, 11 compound 1/n) -> slot in this table
enum { C_ELEV = 0, C_AREA, C_PERI, C_CONV, C_TOPW, C_DKDA, C_UNIF, C_SKK };

struct Problem {
    // ---- the c_diffnw arguments (column-major) -------------------------------------------------------------
    const double *timestep_ar;
    int nts_ql, nts_ub, nts_db, ntss_ev, nts_qtrib, nts_da, mxncomp, nrch;
    const double *z_ar, *bo_ar, *traps_ar, *tw_ar, *twcc_ar, *mann_ar, *manncc_ar, *dx_ar, *iniq;
    int frnw_col;
    const int32_t *frnw;
    const double *qlat, *dbcd, *qtrib;
    int mxnbathy;                               // > 0: natural cross sections (bathymetry stations per node)
    const double *x_bathy, *z_bathy, *mann_bathy; // [mxnbathy][mxncomp][nrch]
    const int32_t *size_bathy;                  // [mxncomp][nrch]
    double *nat_v;                              // work: vertex lists [mxncomp*nrch][3][mxnbathy + 2]
    const double *para_ar;
    // refactored hydrofabric: results are mapped back to the original one at the end (cwnrow > 0)
    const double *rdx_ar, *crosswalk, *z_thalweg; // [mxncomp][nrch], [cwnrow][cwncol], [mxncomp][nrch]
    int cwnrow, cwncol;
    double *q_ev, *elv_ev, *depth_ev;
    // ---- work space (caller allocated; sizes in work_doubles()) ---------------------------------------------
    double *tab;            // [nrch*mxncomp][kCols][kNel]
    double *z, *dx, *bo, *sk, *pere, *celerity, *diffusivity, *qpx, *qp, *oldQ, *newQ, *oldArea, *newArea, *oldY, *newY,
        *lateralFlow;                                        // [mxncomp*nrch] each
    double *eei, *ffi, *exi, *fxi, *celerity2, *diffusivity2, *co; // [mxncomp] each
    double *tarr_ql, *varr_ql, *tarr_qtrib, *varr_qtrib, *tarr_db, *varr_db;
    int32_t *mstem_frj;     // [nrch] mainstem reaches, upstream first; then [nrch + 1] flags "reach j is mainstem"
    int32_t *is_main;
    int64_t *counters;      // optional [4]: sub-steps, node sweeps, depth-solve function evaluations (diagnostics)
    int nmstem;
    // ---- scalars of diffnw ------------------------------------------------------------------------------------
    double dtini, dtini_min, cfl, C_llm, D_llm, D_ulm, q_llm, so_llm, theta;
    int dsbc_option;
};

DW_HD inline int64_t work_doubles(int mxncomp, int nrch, int nts_ql, int nts_qtrib, int nts_db, int mxnbathy)
{
    const int64_t nn = (int64_t)mxncomp * nrch;
    return nn * kCols * kNel + 16 * nn + 7 * (int64_t)mxncomp + 2 * ((int64_t)nts_ql + 1) + 2 * (int64_t)nts_qtrib
           + 2 * (int64_t)nts_db + 8 + (mxnbathy > 0 ? nn * 3 * (int64_t)(mxnbathy + 3) : 0);
}

// carve the work space out of one allocation
DW_HD inline void bind_work(Problem &p, double *w)
{
    const int64_t nn = (int64_t)p.mxncomp * p.nrch;
    p.tab = w; w += nn * kCols * kNel;
    double **grid[] = {&p.z, &p.dx, &p.bo, &p.sk, &p.pere, &p.celerity, &p.diffusivity, &p.qpx, &p.qp, &p.oldQ, &p.newQ,
                       &p.oldArea, &p.newArea, &p.oldY, &p.newY, &p.lateralFlow};
    for (int k = 0; k < 16; ++k) { *grid[k] = w; w += nn; }
    double **line[] = {&p.eei, &p.ffi, &p.exi, &p.fxi, &p.celerity2, &p.diffusivity2, &p.co};
    for (int k = 0; k < 7; ++k) { *line[k] = w; w += p.mxncomp; }
    p.tarr_ql = w; w += p.nts_ql + 1;
    p.varr_ql = w; w += p.nts_ql + 1;
    p.tarr_qtrib = w; w += p.nts_qtrib;
    p.varr_qtrib = w; w += p.nts_qtrib;
    p.tarr_db = w; w += p.nts_db;
    p.varr_db = w; w += p.nts_db;
    p.nat_v = w;
}

// 1-based accessors in the reference's index order
#define DW_G(a, i, j) (a)[((i) - 1) + (int64_t)((j) - 1) * p.mxncomp]
// the per-node state of the sweeps and the per-reach scratch lines, through the Scan policy: Scan::state(ptr) is the
// pointer itself on the host and for a device run whose state stays in global memory, and the same address as an LDS
// pointer when the kernel has moved the state into LDS (ds_read/ds_write instead of flat accesses)
#define DW_S(a, i, j) Scan::state(a)[((i) - 1) + (int64_t)((j) - 1) * p.mxncomp]
#define DW_L(a) Scan::state(a)
#define DW_FRNW(j, c) p.frnw[((j) - 1) + (int64_t)((c) - 1) * p.nrch]
#define DW_TAB(col, iel, i, j) p.tab[((((int64_t)((j) - 1) * p.mxncomp + ((i) - 1)) * kCols + (col)) * kNel) + ((iel) - 1)]
#define DW_EV(a, ts, i, j) (a)[((ts) - 1) + (int64_t)p.ntss_ev * (((i) - 1) + (int64_t)p.mxncomp * ((j) - 1))]

DW_HD inline double dmax(double a, double b) { return a > b ? a : b; }
DW_HD inline double dmin(double a, double b) { return a < b ? a : b; }

// ------------------------------------------------------------------------------------------------ interpolation
// Numerical Recipes' locate (:2715-2753): j with x between xx(j) and xx(j+1); 0 / n out of range.  xx 0-based here.
DW_HD inline int locate(const double *xx, int n, double x)
{
    const bool ascnd = xx[n - 1] >= xx[0];
    int jl = 0, ju = n + 1;
    while (ju - jl > 1) {
        const int jm = (ju + jl) / 2;
        if (ascnd == (x >= xx[jm - 1])) jl = jm; else ju = jm;
    }
    if (x == xx[0]) return 1;
    if (x == xx[n - 1]) return n - 1;
    return jl;
}
DW_HD inline double linterpol(double x1, double y1, double x2, double y2, double x)
{
    if (fabs(x2 - x1) < (double)0.0001f) return 0.5 * (y1 + y2);       // :2660-2666
    return (y2 - y1) / (x2 - x1) * (x - x1) + y1;
}
DW_HD inline double intp_y(int n, const double *xarr, const double *yarr, double x)
{
    int irow = locate(xarr, n, x);
    if (irow == 0) irow = 1;
    if (irow == n) irow = n - 1;
    return linterpol(xarr[irow - 1], yarr[irow - 1], xarr[irow], yarr[irow], x);
}
// r_interpol (:2553-2594): first bracketing interval from the left; beyond the top: extrapolate; below: min(y)
DW_HD inline double r_interpol(const double *x, const double *y, int kk, double xrt)
{
    double xmax = x[0], xmin = x[0];
    for (int k = 1; k < kk; ++k) { xmax = dmax(xmax, x[k]); xmin = dmin(xmin, x[k]); }
    if (xrt <= xmax && xrt >= xmin) {
        for (int k = 0; k < kk - 1; ++k)
            if ((x[k] - xrt) * (x[k + 1] - xrt) <= 0.0) return (xrt - x[k]) / (x[k + 1] - x[k]) * (y[k + 1] - y[k]) + y[k];
        return 0.0; // (not reached for a table that brackets xrt)
    }
    if (xrt >= xmax) return (xrt - x[kk - 2]) / (x[kk - 1] - x[kk - 2]) * (y[kk - 1] - y[kk - 2]) + y[kk - 2];
    double ymin = y[0];
    for (int k = 1; k < kk; ++k) ymin = dmin(ymin, y[k]);
    return ymin;
}
// r_interpol's search separated from its arithmetic, so that one search serves every column looked up at the same
// abscissa (mesh_diffusive_backward reads four columns at the same water elevation).  mode 0: interval k brackets,
// 1: above the table (extrapolate from the last interval), 2: below (minimum of the ordinate), 3: no bracket found.
struct Bracket {
    int mode, k;
    double xk, xk1; // abscissae of the interval used
};
// X: k -> abscissa (the elevation column, or its squared depth above the bed)
template <class X> DW_HD inline Bracket bracket_serial(const X &xf, int kk, double xrt)
{
    double xmax = xf(0), xmin = xf(0);
    for (int k = 1; k < kk; ++k) { const double v = xf(k); xmax = dmax(xmax, v); xmin = dmin(xmin, v); }
    Bracket b;
    if (xrt <= xmax && xrt >= xmin) {
        for (int k = 0; k < kk - 1; ++k) {
            const double xk = xf(k), xk1 = xf(k + 1);
            if ((xk - xrt) * (xk1 - xrt) <= 0.0) { b.mode = 0; b.k = k; b.xk = xk; b.xk1 = xk1; return b; }
        }
        b.mode = 3; b.k = 0; b.xk = b.xk1 = 0.0;
        return b;
    }
    if (xrt >= xmax) { b.mode = 1; b.k = kk - 2; b.xk = xf(kk - 2); b.xk1 = xf(kk - 1); return b; }
    b.mode = 2; b.k = 0; b.xk = b.xk1 = 0.0;
    return b;
}
struct ElevAt { const double *x; DW_HD double operator()(int k) const { return x[k]; } };
struct SqDepthAt { const double *e; double zz; DW_HD double operator()(int k) const { return (e[k] - zz) * (e[k] - zz); } };

// the [kCols][kNel] table block of node i of reach j
DW_HD inline const double *node_block(const Problem &p, int i, int j) { return &DW_TAB(0, 1, i, j); }
struct SerialScan {
    DW_HD static double *state(double *g) { return g; }
    DW_HD void begin_node(const Problem &, int, int) const {}
    DW_HD const double *table(const Problem &p, int i, int j) const { return node_block(p, i, j); }
    DW_HD int locate_row(const double *xx, int n, double x) const { return locate(xx, n, x); }
    DW_HD Bracket bracket(const double *elev, bool squared, double zz, int kk, double xrt) const
    {
        return squared ? bracket_serial(SqDepthAt{elev, zz}, kk, xrt) : bracket_serial(ElevAt{elev}, kk, xrt);
    }
    // r_interpol's value for ordinate column y at a bracket (:2567-2591)
    DW_HD double apply(const Bracket &b, const double *y, int kk, double xrt) const
    {
        if (b.mode <= 1) return (xrt - b.xk) / (b.xk1 - b.xk) * (y[b.k + 1] - y[b.k]) + y[b.k];
        if (b.mode == 3) return 0.0;
        double ymin = y[0];
        for (int k = 1; k < kk; ++k) ymin = dmin(ymin, y[k]);
        return ymin;
    }
};

DW_HD inline double intp_blk(const double *tb, int xcol, int ycol, double x)
{
    return intp_y(kNel, tb + xcol * kNel, tb + ycol * kNel, x);
}
// intp_xsec_tab split in two: the row (search left to the scan policy) and the interpolation in a column at that
// row, so that columns looked up at the same abscissa share the search
template <class Scan> DW_HD inline int row_blk(const Scan &scan, const double *tb, int xcol, double x)
{
    int irow = scan.locate_row(tb + xcol * kNel, kNel, x);
    if (irow == 0) irow = 1;
    if (irow == kNel) irow = kNel - 1;
    return irow;
}
DW_HD inline double at_row(const double *tb, int xcol, int ycol, int irow, double x)
{
    const double *xarr = tb + xcol * kNel, *yarr = tb + ycol * kNel;
    return linterpol(xarr[irow - 1], yarr[irow - 1], xarr[irow], yarr[irow], x);
}
template <class Scan> DW_HD inline double intp_blk(const Scan &scan, const double *tb, int xcol, int ycol, double x)
{
    return at_row(tb, xcol, ycol, row_blk(scan, tb, xcol, x), x);
}
DW_HD inline double intp_tab(const Problem &p, int i, int j, int xcol, int ycol, double x)
{
    return intp_blk(node_block(p, i, j), xcol, ycol, x);
}

// ------------------------------------------------------------------------------------------------ cross sections
DW_HD inline double cal_tri_area(double el, double x0, double x1, double y1) { return fabs(0.5 * (x1 - x0) * (el - y1)); }
DW_HD inline double cal_dist(double x1, double y1, double x2, double y2)
{
    return sqrt((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2) + (double)1.e-32f);
}
// vertices (1-based arrays of 8) i1..i2
DW_HD inline double cal_multi_area(double el, const double *xx, const double *yy, int i1, int i2)
{
    double area = 0.0;
    for (int i = i1; i <= i2 - 1; ++i) area = area + fabs(0.5 * (xx[i + 1] - xx[i]) * (el - yy[i] + el - yy[i + 1]));
    return area;
}
DW_HD inline double cal_perimeter(const double *xx, const double *yy, int i1, int i2)
{
    double pp = 0.0;
    for (int i = i1; i <= i2 - 1; ++i) pp = pp + cal_dist(xx[i], yy[i], xx[i + 1], yy[i + 1]);
    return pp;
}

// The three sub-sections (left flood plain, main channel, right flood plain) of the RouteLink trapezoid with
// compound channel, as vertex lists, and the elevation grid of the table (readXsection :2148-2262).
struct Section {
    double xs[3][9], ys[3][9]; // 1-based vertices per sub-section
    int num[3];
    double mann[3];
    double el_min;             // lowest point (the 1 cm notch): becomes the node's bed elevation z
    double el_range, el_incr, elev5;
};
DW_HD inline void make_section(const Problem &p, int k, int jr, Section &s)
{
    const double timesDepth = 4.0;
    const double z_g = DW_G(p.z_ar, k, jr), bo_g = DW_G(p.bo_ar, k, jr), traps_g = DW_G(p.traps_ar, k, jr);
    const double tw_g = DW_G(p.tw_ar, k, jr), twcc_g = DW_G(p.twcc_ar, k, jr);
    const double hbf = (tw_g - bo_g) / (2.0 * traps_g);
    double xcs[9], ycs[9];
    xcs[1] = 0.0;                       ycs[1] = z_g + timesDepth * hbf;
    xcs[2] = 0.0;                       ycs[2] = z_g + hbf;
    xcs[3] = (twcc_g - tw_g) / 2.0;     ycs[3] = z_g + hbf;
    xcs[4] = xcs[3] + traps_g * hbf;    ycs[4] = z_g;
    xcs[5] = xcs[4] + bo_g;             ycs[5] = z_g;
    xcs[6] = xcs[5] + traps_g * hbf;    ycs[6] = z_g + hbf;
    xcs[7] = twcc_g;                    ycs[7] = z_g + hbf;
    xcs[8] = xcs[7];                    ycs[8] = z_g + timesDepth * hbf;
    double el_min = (double)99999.f, el_max = -(double)99999.f;
    for (int i = 2; i <= 8; ++i) {   // (the reference's loop bound is its vertex counter after the loop: 2..8)
        if (ycs[i] < el_min) el_min = ycs[i];
        if (ycs[i] > el_max) el_max = ycs[i];
    }
    const double el_range = (el_max - el_min) * 2.0;
    const double top = el_min + el_range + 1.0;
    // left flood plain
    s.xs[0][1] = xcs[1]; s.ys[0][1] = top;
    for (int i = 1; i <= 3; ++i) { s.xs[0][i + 1] = xcs[i]; s.ys[0][i + 1] = ycs[i]; }
    s.xs[0][5] = xcs[3]; s.ys[0][5] = top;
    // main channel (with the 1 cm notch at mid-bottom)
    s.xs[1][1] = xcs[3]; s.ys[1][1] = top;
    for (int i = 3; i <= 4; ++i) { s.xs[1][i - 1] = xcs[i]; s.ys[1][i - 1] = ycs[i]; }
    for (int i = 5; i <= 6; ++i) { s.xs[1][i] = xcs[i]; s.ys[1][i] = ycs[i]; }
    s.xs[1][7] = xcs[6]; s.ys[1][7] = top;
    s.xs[1][4] = (s.xs[1][3] + s.xs[1][5]) / 2.0;
    s.ys[1][4] = s.ys[1][3] - (double)0.01f;
    // right flood plain
    for (int i = 6; i <= 8; ++i) { s.xs[2][i - 4] = xcs[i]; s.ys[2][i - 4] = ycs[i]; }
    s.xs[2][1] = s.xs[2][2]; s.ys[2][1] = top;
    s.xs[2][5] = s.xs[2][4]; s.ys[2][5] = top;
    s.num[0] = 5; s.num[1] = 7; s.num[2] = 5;
    s.mann[0] = 1.0 / (1.0 / DW_G(p.manncc_ar, k, jr)); // lftBnkMann = 1/skLeft, skLeft = 1/manncc (:445-452)
    s.mann[1] = 1.0 / (1.0 / DW_G(p.mann_ar, k, jr));
    s.mann[2] = s.mann[0];
    s.el_min = s.ys[1][4];
    s.el_range = el_range;
    s.elev5 = s.el_min + (double)0.01f;
    s.el_incr = el_range / (double)(float)(kNel - 6.0f);
}
DW_HD inline double table_elev(const Section &s, int j) // elev(j), j = 1..nel (:2250-2262)
{
    if (j == 1) return s.el_min;
    if (j == 2) return s.el_min + (double)(0.01f / 4.f);
    if (j == 3) return s.el_min + (double)(0.01f / 4.f * 2.f);
    if (j == 4) return s.el_min + (double)(0.01f / 4.f * 3.f);
    if (j == 5) return s.elev5;
    return s.elev5 + s.el_incr * (double)(float)(j - 5);
}
// wetted area, perimeter, conveyance and top width of sub-section kkk at table level j (:2270-2352)
DW_HD inline void subsection_at(const Section &s, int kkk, int j, double &el_out, double &area, double &peri, double &conv,
                                double &topw)
{
    const double TOL = (double)1e-8f;
    const double *xcs = s.xs[kkk], *ycs = s.ys[kkk];
    const int num = s.num[kkk];
    double el_now = table_elev(s, j);
    if (fabs(el_now - s.el_min) < TOL) el_now = el_now + (double)0.00001f;
    int i_start[8], i_end[8], i_area = 0, i_find = 0;
    i_start[1] = -999; i_end[1] = -999;
    for (int i = 1; i <= num - 1; ++i) {
        const double y1 = ycs[i], y2 = ycs[i + 1];
        if (el_now <= y1 && el_now > y2 && i_find == 0) { i_find = 1; ++i_area; i_start[i_area] = i; }
        if (el_now > y1 && el_now <= y2 && i_find == 1) { i_find = 0; i_end[i_area] = i; }
    }
    double cal_area = 0.0, cal_peri = 0.0, cal_topW = 0.0;
    for (int i = 1; i <= i_area; ++i) {
        double x1 = xcs[i_start[i]], x2 = xcs[i_start[i] + 1], y1 = ycs[i_start[i]], y2 = ycs[i_start[i] + 1];
        const double x_start = (y1 == y2) ? x1 : x1 + (el_now - y1) / (y2 - y1) * (x2 - x1);
        x1 = xcs[i_end[i]]; x2 = xcs[i_end[i] + 1]; y1 = ycs[i_end[i]]; y2 = ycs[i_end[i] + 1];
        const double x_end = (y1 == y2) ? x1 : x1 + (el_now - y1) / (y2 - y1) * (x2 - x1);
        cal_topW = x_end - x_start + cal_topW;
        const int i1 = i_start[i], i2 = i_end[i];
        cal_area = cal_area + cal_tri_area(el_now, x_start, xcs[i1 + 1], ycs[i1 + 1]) + cal_multi_area(el_now, xcs, ycs, i1 + 1, i2)
                   + cal_tri_area(el_now, x_end, xcs[i2], ycs[i2]);
        cal_peri = cal_peri + cal_dist(x_start, el_now, xcs[i1 + 1], ycs[i1 + 1]) + cal_perimeter(xcs, ycs, i1 + 1, i2)
                   + cal_dist(x_end, el_now, xcs[i2], ycs[i2]);
        if (i1 == 1) cal_peri = cal_peri - cal_dist(x_start, el_now, xcs[i1 + 1], ycs[i1 + 1]);
        if (i2 == num - 1) cal_peri = cal_peri - cal_dist(x_end, el_now, xcs[i2], ycs[i2]);
    }
    el_out = el_now;
    area = cal_area;
    peri = cal_peri;
    double redi = area / peri;
    conv = 1.0 / s.mann[kkk] * area * DW_POW(redi, (double)(2.f / 3.f));
    if (peri <= TOL) conv = 0.0;
    topw = cal_topW;
}

// ---- natural cross sections (readXsection_natural_mann_vertices, diffusive.f90:1756-2091) ----------------------
// Vertices of node (k, jr): the bathymetry stations shifted to start at x = 0, Manning's n capped at 0.15, and one
// vertex on an infinite vertical wall at either end; 1-based arrays xcs/ycs/mcs[1..num] inside p.nat_v.
struct NatSection {
    const double *xcs, *ycs, *mcs; // 1-based views (index 0 unused)
    int num;
    double el_min, el_range, el_incr;
};
DW_HD inline double *nat_block(const Problem &p, int k, int jr)
{
    return p.nat_v + (((int64_t)(jr - 1) * p.mxncomp + (k - 1)) * 3) * (int64_t)(p.mxnbathy + 3);
}
#define DW_BATHY(a, ic, k, jr) (a)[((ic) - 1) + (int64_t)p.mxnbathy * (((k) - 1) + (int64_t)p.mxncomp * ((jr) - 1))]
DW_HD inline void nat_vertices(Problem &p, int k, int jr)
{
    const int nb = p.size_bathy[(k - 1) + (int64_t)(jr - 1) * p.mxncomp];
    const int num = nb + 2;
    double *xcs = nat_block(p, k, jr), *ycs = xcs + (p.mxnbathy + 3), *mcs = ycs + (p.mxnbathy + 3);
    const double x0 = DW_BATHY(p.x_bathy, 1, k, jr);
    for (int ic = 2; ic <= nb + 1; ++ic) {
        xcs[ic] = (-x0 + DW_BATHY(p.x_bathy, ic - 1, k, jr)) * 1.0;
        ycs[ic] = DW_BATHY(p.z_bathy, ic - 1, k, jr) * 1.0;
        double m = DW_BATHY(p.mann_bathy, ic - 1, k, jr);
        if (m > (double)0.15f) m = (double)0.15f;
        mcs[ic] = m;
    }
    double el_min = (double)99999.f, el_max = -(double)99999.f;
    for (int ic = 2; ic <= num - 1; ++ic) {
        if (ycs[ic] < el_min) el_min = ycs[ic];
        if (ycs[ic] > el_max) el_max = ycs[ic];
    }
    const double el_range = (el_max - el_min) * 4.0; // timesDepth
    xcs[1] = xcs[2];
    ycs[1] = el_min + el_range + 1.0;
    xcs[num] = xcs[num - 1];
    ycs[num] = el_min + el_range + 1.0;
    mcs[1] = 0.0;
    mcs[num - 1] = 0.0;
    mcs[num] = 0.0;
    xcs[0] = el_min;   // slot 0 of the three lists carries the scalars of the section
    ycs[0] = el_range;
    mcs[0] = (double)num;
}
DW_HD inline NatSection nat_section(const Problem &p, int k, int jr)
{
    NatSection s;
    const double *b = nat_block(p, k, jr);
    s.xcs = b; s.ycs = b + (p.mxnbathy + 3); s.mcs = s.ycs + (p.mxnbathy + 3);
    s.el_min = s.xcs[0]; s.el_range = s.ycs[0]; s.num = (int)s.mcs[0];
    s.el_incr = s.el_range / (double)(float)(kNel - 1.0f);
    return s;
}
// one table level of a natural section (:1840-1927): elevation, area, perimeter, top width, conveyance with the
// equivalent Manning's n of the wetted vertices, 1/n; sub-areas are accumulated as they close (the order the reference
// adds them in)
DW_HD inline void nat_row(Problem &p, const NatSection &s, int k, int jr, int iel)
{
    const double TOL = (double)1e-8f;
    const double *xcs = s.xcs, *ycs = s.ycs, *mcs = s.mcs;
    const int num = s.num;
    double el_now = s.el_min + (double)(float)(iel - 1) * s.el_incr;
    if (fabs(el_now - s.el_min) < TOL) el_now = el_now + (double)0.00001f;
    double cal_area = 0.0, cal_peri = 0.0, cal_topW = 0.0, cal_mann = 0.0;
    int i_find = 0, i1 = -999;
    for (int ic = 1; ic <= num - 1; ++ic) {
        const double y1 = ycs[ic], y2 = ycs[ic + 1];
        if (el_now <= y1 && el_now > y2 && i_find == 0) { i_find = 1; i1 = ic; }
        if (el_now > y1 && el_now <= y2 && i_find == 1) {
            i_find = 0;
            const int i2 = ic;
            double xa = xcs[i1], xb = xcs[i1 + 1], ya = ycs[i1], yb = ycs[i1 + 1];
            const double x_start = (ya == yb) ? xa : xa + (el_now - ya) / (yb - ya) * (xb - xa);
            xa = xcs[i2]; xb = xcs[i2 + 1]; ya = ycs[i2]; yb = ycs[i2 + 1];
            const double x_end = (ya == yb) ? xa : xa + (el_now - ya) / (yb - ya) * (xb - xa);
            cal_topW = x_end - x_start + cal_topW;
            cal_area = cal_area + cal_tri_area(el_now, x_start, xcs[i1 + 1], ycs[i1 + 1]) + cal_multi_area(el_now, xcs, ycs, i1 + 1, i2)
                       + cal_tri_area(el_now, x_end, xcs[i2], ycs[i2]);
            cal_peri = cal_peri + cal_dist(x_start, el_now, xcs[i1 + 1], ycs[i1 + 1]) + cal_perimeter(xcs, ycs, i1 + 1, i2)
                       + cal_dist(x_end, el_now, xcs[i2], ycs[i2]);
            double pm = 0.0;
            for (int i = i1 + 1; i <= i2 - 1; ++i) pm = pm + cal_dist(xcs[i], ycs[i], xcs[i + 1], ycs[i + 1]) * DW_POW(mcs[i], 1.5);
            cal_mann = cal_mann + cal_dist(x_start, el_now, xcs[i1 + 1], ycs[i1 + 1]) * DW_POW(mcs[i1], 1.5) + pm
                       + cal_dist(x_end, el_now, xcs[i2], ycs[i2]) * DW_POW(mcs[i2], 1.5);
            if (i1 == 1) cal_peri = cal_peri - cal_dist(x_start, el_now, xcs[i1 + 1], ycs[i1 + 1]);
            if (i2 == num - 1) cal_peri = cal_peri - cal_dist(x_end, el_now, xcs[i2], ycs[i2]);
        }
    }
    const double redi = cal_area / cal_peri;
    const double equiv_mann = DW_POW(cal_mann / cal_peri, (double)(2.0f / 3.0f));
    double conv = (1.0 / equiv_mann) * cal_area * DW_POW(redi, (double)(2.0f / 3.0f));
    if (cal_peri <= TOL) conv = 0.0;
    DW_TAB(C_ELEV, iel, k, jr) = el_now;
    DW_TAB(C_AREA, iel, k, jr) = cal_area;
    DW_TAB(C_PERI, iel, k, jr) = cal_peri;
    DW_TAB(C_CONV, iel, k, jr) = conv;
    DW_TAB(C_TOPW, iel, k, jr) = cal_topW;
    DW_TAB(C_SKK, iel, k, jr) = 1.0 / equiv_mann;
}
// dK/dA and the two monotonicity passes over the levels of one node (:1929-1990) -- sequential in the level
DW_HD inline void nat_smooth(Problem &p, int k, int jr)
{
    double *conv = &DW_TAB(C_CONV, 1, k, jr) - 1, *dkda = &DW_TAB(C_DKDA, 1, k, jr) - 1; // 1-based views
    const double *a1 = &DW_TAB(C_AREA, 1, k, jr) - 1, *el1 = &DW_TAB(C_ELEV, 1, k, jr) - 1;
    for (int iel = 1; iel <= kNel; ++iel)
        dkda[iel] = (iel == 1) ? conv[iel] / a1[iel] : (conv[iel] - conv[iel - 1]) / (a1[iel] - a1[iel - 1]);
    const double incr_rate = (double)0.01f;
    // (the reference assigns to its loop variable's start inside the loop, which a Fortran do-loop ignores: every
    // level from 2 to nel is visited)
    for (int iel = 2; iel <= kNel; ++iel) {
        if (conv[iel] <= conv[iel - 1]) {
            int ii = iel;
            while (conv[ii] < conv[iel - 1] && ii < kNel) ii = ii + 1;
            const int inc = ii;
            if (inc >= kNel && conv[inc] < conv[iel - 1]) conv[inc] = (1.0 + incr_rate) * conv[iel - 1];
            const double pos_slope = (conv[inc] - conv[iel - 1]) / (el1[inc] - el1[iel - 1]);
            for (int q = iel; q <= inc - 1; ++q) conv[q] = conv[iel - 1] + pos_slope * (el1[q] - el1[iel - 1]);
            for (int q = iel; q <= inc - 1; ++q)
                dkda[q] = (q == 1) ? conv[q] / a1[q] : (conv[q] - conv[q - 1]) / (a1[q] - a1[q - 1]);
        }
    }
    for (int iel = 2; iel <= kNel; ++iel) {
        if (dkda[iel] <= dkda[iel - 1]) {
            int ii = iel;
            while (dkda[ii] < dkda[iel - 1] && ii < kNel) ii = ii + 1;
            const int inc = ii;
            if (inc >= kNel && dkda[inc] < dkda[iel - 1]) dkda[inc] = (1.0 + incr_rate) * dkda[iel - 1];
            const double pos_slope = (dkda[inc] - dkda[iel - 1]) / (el1[inc] - el1[iel - 1]);
            for (int q = iel; q <= inc - 1; ++q) dkda[q] = dkda[iel - 1] + pos_slope * (el1[q] - el1[iel - 1]);
        }
    }
}
// the uniform-flow column of a natural-section node (:470-486; dK/dA is already final)
DW_HD inline void nat_row_finish(Problem &p, int k, int jr, int j)
{
    const int ncomp = DW_FRNW(jr, 1);
    double slope = (k < ncomp) ? (DW_G(p.z, k, jr) - DW_G(p.z, k + 1, jr)) / DW_G(p.dx, k, jr)
                               : (DW_G(p.z, k - 1, jr) - DW_G(p.z, k, jr)) / DW_G(p.dx, k - 1, jr);
    if (slope <= p.so_llm) slope = p.so_llm;
    DW_TAB(C_UNIF, j, k, jr) = DW_TAB(C_CONV, j, k, jr) * sqrt(slope);
}

// One table row (node k of reach jr, level j): the columns that depend on this level only; dK/dA needs level
// j-1 too and is formed from the sums stored here (table_row_finish).  sums[0..2] = total area, perimeter, conveyance.
DW_HD inline void table_row(Problem &p, const Section &s, int k, int jr, int j)
{
    double el1 = 0.0, a[3], pe[3], cv[3], tw[3];
    for (int kkk = 0; kkk < 3; ++kkk) {
        double el;
        subsection_at(s, kkk, j, el, a[kkk], pe[kkk], cv[kkk], tw[kkk]);
        if (kkk == 0) el1 = el;
    }
    const double sa = (a[0] + a[1]) + a[2], sp = (pe[0] + pe[1]) + pe[2], sc = (cv[0] + cv[1]) + cv[2];
    const double lm = s.mann[0], mm = s.mann[1], rm = s.mann[2];
    const double compoundMann = sqrt((fabs(pe[0]) * (lm * lm) + fabs(pe[1]) * (mm * mm) + fabs(pe[2]) * (rm * rm))
                                     / (fabs(pe[0]) + fabs(pe[1]) + fabs(pe[2])));
    DW_TAB(C_ELEV, j, k, jr) = el1;
    DW_TAB(C_AREA, j, k, jr) = sa;
    DW_TAB(C_PERI, j, k, jr) = sp;
    DW_TAB(C_CONV, j, k, jr) = sc;
    DW_TAB(C_TOPW, j, k, jr) = fabs(tw[0]) + fabs(tw[1]) + fabs(tw[2]);
    DW_TAB(C_SKK, j, k, jr) = 1.0 / compoundMann;
}
// dK/dA (:2395-2401) and the uniform-flow column (:470-486), after every row of the node exists and z is final
DW_HD inline void table_row_finish(Problem &p, int k, int jr, int j)
{
    const double sa = DW_TAB(C_AREA, j, k, jr), sc = DW_TAB(C_CONV, j, k, jr);
    DW_TAB(C_DKDA, j, k, jr) = (j == 1) ? sc / sa : (sc - DW_TAB(C_CONV, j - 1, k, jr)) / (sa - DW_TAB(C_AREA, j - 1, k, jr));
    const int ncomp = DW_FRNW(jr, 1);
    double slope = (k < ncomp) ? (DW_G(p.z, k, jr) - DW_G(p.z, k + 1, jr)) / DW_G(p.dx, k, jr)
                               : (DW_G(p.z, k - 1, jr) - DW_G(p.z, k, jr)) / DW_G(p.dx, k - 1, jr);
    if (slope <= p.so_llm) slope = p.so_llm;
    DW_TAB(C_UNIF, j, k, jr) = sc * sqrt(slope);
}

// ------------------------------------------------------------------------------------------------ the solver
DW_HD inline bool is_mainstem(const Problem &p, int j) { return p.is_main[j] != 0; }

// scalars, the mainstem list, dx / minDx (diffnw :261-412); returns minDx
DW_HD inline double setup_scalars(Problem &p)
{
    p.dtini = p.timestep_ar[0];
    p.dtini_min = p.dtini / p.timestep_ar[9];
    p.cfl = p.para_ar[0];
    p.C_llm = p.para_ar[1];
    p.D_llm = p.para_ar[2];
    p.D_ulm = p.para_ar[3];
    p.q_llm = p.para_ar[7];
    p.so_llm = p.para_ar[8];
    p.theta = p.para_ar[9];
    p.dsbc_option = (int)p.para_ar[10];
    const int64_t nn = (int64_t)p.mxncomp * p.nrch;
    for (int64_t e = 0; e < nn; ++e) {
        p.z[e] = p.z_ar[e];
        p.oldQ[e] = p.iniq[e];
        p.newQ[e] = p.iniq[e];
        p.qp[e] = p.iniq[e];
        p.dx[e] = 0.0;
        p.qpx[e] = 0.0;
        p.newY[e] = -999.0;
        p.oldY[e] = 0.0;   // (the reference leaves the unset entries undefined; none is read before being written)
        p.oldArea[e] = 0.0; p.newArea[e] = 0.0; p.bo[e] = 0.0; p.sk[e] = 0.0; p.pere[e] = 0.0;
        p.celerity[e] = 0.0; p.diffusivity[e] = 0.0; p.lateralFlow[e] = 0.0;
    }
    p.nmstem = 0;
    for (int j = 1; j <= p.nrch; ++j) {
        const int nus = DW_FRNW(j, 3);
        p.is_main[j] = DW_FRNW(j, 3 + nus + 1) == 555;
        if (p.is_main[j]) p.mstem_frj[p.nmstem++] = j;
    }
    double minDx = 1e10;
    for (int m = 0; m < p.nmstem; ++m) {
        const int j = p.mstem_frj[m], ncomp = DW_FRNW(j, 1);
        for (int i = 1; i <= ncomp - 1; ++i) {
            DW_G(p.dx, i, j) = DW_G(p.dx_ar, i, j);
            minDx = dmin(minDx, DW_G(p.dx, i, j));
        }
    }
    return minDx;
}

struct Depth {  // funcd_diffdepth (:1664-1711): f and df/dy at depth y_cur of node i
    double f, df;
};
// rtsafe (:1555-1662): Newton-Raphson safeguarded by bisection for the depth of node i given node i+1 -- in two parts.
// Everything that does not depend on the node below (the bracket from the normal depth and the old depth, the table
// look-ups of the three evaluations the iteration starts with) is DepthPre; depth_solve() is what is left of the
// recurrence from node to node.  The host runs one after the other; the device forms DepthPre for all nodes of a
// sub-step at once, one node per thread, and walks the chain with one wavefront (diffusive.hip).
struct DepthPre {
    double y_norm, x1, x2;   // normal depth for Q_cur, the bracket 0.05 (y_norm + y_old) .. (y_norm + y_old)
    double sf[3];            // |Q| Q / K**2 at x1, x2 and the midpoint
    double df_mid;           // df/dy at the midpoint
    double qq;               // |Q_cur| Q_cur
    double slope, dxi, z_cur;
};
// f and df/dy from the table values at y_cur (funcd_diffdepth's arithmetic)
DW_HD inline Depth funcd_arith(double Q_cur, double sf_ds, double conv_cur, double dKdA, double topw, double slope, double dxi,
                               double y_cur, double y_ds)
{
    const double sf_cur = fabs(Q_cur) * Q_cur / (conv_cur * conv_cur);
    Depth r;
    r.f = y_cur - y_ds + slope * dxi - 0.50 * (sf_cur + sf_ds) * dxi;
    r.df = 1.0 + (fabs(Q_cur) * Q_cur / (conv_cur * conv_cur * conv_cur)) * dxi * topw * dKdA;
    return r;
}
template <class Scan>
DW_HD inline DepthPre depth_pre(const Problem &p, const Scan &scan, const double *tb, int i, int j, double Q_cur, double z_cur)
{
    if (p.counters) p.counters[2] += 3;
    DepthPre d;
    const int row_norm = row_blk(scan, tb, C_UNIF, fabs(Q_cur));
    const double elv_norm = at_row(tb, C_UNIF, C_ELEV, row_norm, fabs(Q_cur));
    d.y_norm = elv_norm - DW_S(p.z, i, j);
    const double y_old = DW_S(p.oldY, i, j) - DW_S(p.z, i, j);
    d.x1 = 0.5 * (d.y_norm + y_old) * (double)0.1f;
    d.x2 = 0.5 * (d.y_norm + y_old) * 2.0;
    const double rt = 0.50 * (d.x1 + d.x2);
    // the three evaluations rtsafe starts with -- both ends of its bracket and the midpoint -- stage by stage
    // (searches, then table rows, then arithmetic): they do not depend on each other
    const double y[3] = {d.x1, d.x2, rt};
    double elv[3], conv[3], dKdA[3], topw[3];
    int irow[3];
    for (int k = 0; k < 3; ++k) elv[k] = y[k] + z_cur;
    for (int k = 0; k < 3; ++k) irow[k] = row_blk(scan, tb, C_ELEV, elv[k]);
    for (int k = 0; k < 3; ++k) {
        conv[k] = at_row(tb, C_ELEV, C_CONV, irow[k], elv[k]);
        dKdA[k] = at_row(tb, C_ELEV, C_DKDA, irow[k], elv[k]);
        topw[k] = at_row(tb, C_ELEV, C_TOPW, irow[k], elv[k]);
    }
    d.dxi = DW_S(p.dx, i, j);
    double slope = (DW_S(p.z, i, j) - DW_S(p.z, i + 1, j)) / d.dxi;
    d.slope = dmax(slope, p.so_llm);
    d.z_cur = z_cur;
    d.qq = fabs(Q_cur) * Q_cur;
    for (int k = 0; k < 3; ++k) d.sf[k] = fabs(Q_cur) * Q_cur / (conv[k] * conv[k]);
    d.df_mid = 1.0 + (fabs(Q_cur) * Q_cur / (conv[2] * conv[2] * conv[2])) * d.dxi * topw[2] * dKdA[2];
    return d;
}
// A condition every lane of the wavefront agrees on (depth_solve is only ever run by all lanes in step: the serial host
// loop, the one-wavefront kernel, the chain of the parallel kernel): on the device it is made a SCALAR condition, so that the
// branch is a scalar branch instead of an exec-mask region.
#if defined(__HIP_DEVICE_COMPILE__)
#define DW_UNIFORM(c) (__builtin_amdgcn_ballot_w64(c) != 0)
#else
#define DW_UNIFORM(c) (c)
#endif
// F: (y_cur) -> Depth, the function evaluation of an iteration (a table search and three rows)
template <class F> DW_HD inline double depth_solve(const DepthPre &d, double sf_ds, double y_ds, const F &eval)
{
    const int maxit = 40;
    const double xacc = (double)1e-4f;
    const double x1 = d.x1, x2 = d.x2;
    double rt = 0.50 * (x1 + x2);
    // f at both ends of the bracket and (f, df) at the midpoint the iteration starts from (the reference evaluates the
    // midpoint only when the bracket holds; its value is not used otherwise)
    const double fl = x1 - y_ds + d.slope * d.dxi - 0.50 * (d.sf[0] + sf_ds) * d.dxi;
    const double fh = x2 - y_ds + d.slope * d.dxi - 0.50 * (d.sf[1] + sf_ds) * d.dxi;
    if (DW_UNIFORM((fl > 0.0 && fh > 0.0) || (fl < 0.0 && fh < 0.0))) return d.y_norm;
    if (DW_UNIFORM(fl == 0.0)) return x1;
    if (DW_UNIFORM(fh == 0.0)) return x2;
    const bool up = fl < 0.0;
    double xl = up ? x1 : x2, xh = up ? x2 : x1;
    double dxold = fabs(x2 - x1), dxx = dxold;
    Depth dd;
    dd.f = rt - y_ds + d.slope * d.dxi - 0.50 * (d.sf[2] + sf_ds) * d.dxi;
    dd.df = d.df_mid;
    for (int iter = 1; iter <= maxit; ++iter) {
        // bisection step or Newton step: both candidates are formed (the quotient is needed nearly every time), one selected
        const bool bis = ((rt - xh) * dd.df - dd.f) * ((rt - xl) * dd.df - dd.f) > 0.0 || fabs(2.0 * dd.f) > fabs(dxold * dd.df);
        const double dx_b = 0.50 * (xh - xl), dx_n = dd.f / dd.df;
        dxold = dxx;
        dxx = bis ? dx_b : dx_n;
        const double rt_new = bis ? xl + dxx : rt - dxx;
        const bool stuck = bis ? xl == rt_new : rt == rt_new;
        rt = rt_new;
        if (DW_UNIFORM(stuck || fabs(dxx) < xacc)) return rt;
        dd = eval(rt);
        const bool neg = dd.f < 0.0;
        xl = neg ? rt : xl;
        xh = neg ? xh : rt;
    }
    return d.y_norm;
}
template <class Scan>
DW_HD inline double rtsafe(const Problem &p, const Scan &scan, const double *tb, const double *tb_ds, int i, int j, double Q_cur,
                          double Q_ds, double z_cur, double z_ds, double y_ds)
{
    const double elv_ds = y_ds + z_ds;
    const int row_ds = row_blk(scan, tb_ds, C_ELEV, elv_ds);
    const double conv_ds = at_row(tb_ds, C_ELEV, C_CONV, row_ds, elv_ds);
    const double sf_ds = fabs(Q_ds) * Q_ds / (conv_ds * conv_ds);
    const DepthPre d = depth_pre(p, scan, tb, i, j, Q_cur, z_cur);
    return depth_solve(d, sf_ds, y_ds, [&](double y_cur) {
        if (p.counters) p.counters[2] += 1;
        const double elv_cur = y_cur + z_cur;
        const int irow = row_blk(scan, tb, C_ELEV, elv_cur);       // one search: conveyance, dK/dA and top width
        const double conv_cur = at_row(tb, C_ELEV, C_CONV, irow, elv_cur);
        const double dKdA = at_row(tb, C_ELEV, C_DKDA, irow, elv_cur);
        const double topw = at_row(tb, C_ELEV, C_TOPW, irow, elv_cur);
        return funcd_arith(Q_cur, sf_ds, conv_cur, dKdA, topw, d.slope, d.dxi, y_cur, y_ds);
    });
}

// mesh_diffusive_forward (:1108-1355): Crank-Nicolson flow along reach j (Thomas recurrences eei/ffi, exi/fxi).
// The coefficients of node i (2..ncomp) depend on the old time level only: FwdCoef, one node at a time.
struct FwdCoef {
    double ppi, qqi, rri, ssi, sxi;
};
template <class Scan> DW_HD inline FwdCoef forward_coef(const Problem &p, int i, int j, int ncomp)
{
    const double dtini = p.dtini, theta = p.theta;
    const double dxm = DW_S(p.dx, i - 1, j);
    const double cour = dtini / dxm;
    const double cour2 = fabs(DW_S(p.celerity, i, j)) * cour;
    const double c2 = cour2 * cour2, c3 = cour2 * cour2 * cour2;
    const double a1 = 3.0 * c2 - 2.0 * c3;
    const double a2 = 1 - a1;
    const double a3 = (c2 - c3) * dxm;
    const double a4 = (-1.0 * cour2 + 2.0 * c2 - c3) * dxm;
    const double b1 = (6.0 * cour2 - 6.0 * c2) / (-1.0 * dxm);
    const double b2 = -b1;
    const double b3 = (2.0 * cour2 - 3.0 * c2) * (-1.0);
    const double b4 = (-1.0 + 4.0 * cour2 - 3.0 * c2) * (-1.0);
    const double dd1 = (6.0 - 12.0 * cour2) / (dxm * dxm);
    const double dd2 = -dd1;
    const double dd3 = (2.0 - 6.0 * cour2) / dxm;
    const double dd4 = (4.0 - 6.0 * cour2) / dxm;
    const double h1 = 12.0 / (dxm * dxm * dxm);
    const double h2 = -h1;
    const double h3 = 6.0 / (dxm * dxm);
    const double h4 = h3;
    const double alpha = (i == ncomp) ? 1.0 : DW_S(p.dx, i, j) / DW_S(p.dx, i - 1, j);
    const double qa = DW_S(p.oldQ, i - 1, j), qb = DW_S(p.oldQ, i, j);
    const double xa = DW_S(p.qpx, i - 1, j), xb = DW_S(p.qpx, i, j);
    const double qy = a1 * qa + a2 * qb + a3 * xa + a4 * xb;
    const double qxy = b1 * qa + b2 * qb + b3 * xa + b4 * xb;
    const double qxxy = dd1 * qa + dd2 * qb + dd3 * xa + dd4 * xb;
    const double qxxxy = h1 * qa + h2 * qb + h3 * xa + h4 * xb;
    const double dif = DW_S(p.diffusivity, i, j);
    FwdCoef c;
    c.ppi = -theta * dif * dtini / (dxm * dxm) * 2.0 / (alpha * (alpha + 1.0)) * alpha;
    c.qqi = 1.0 - c.ppi * (alpha + 1.0) / alpha;
    c.rri = c.ppi / alpha;
    c.ssi = qy + dtini * dif * (1.0 - theta) * qxxy;
    c.sxi = qxy + dtini * dif * (1.0 - theta) * qxxxy;
    return c;
}
template <class Scan> DW_HD inline void forward(Problem &p, int j)
{
    const int ncomp = DW_FRNW(j, 1);
    DW_L(p.eei)[0] = 1.0; DW_L(p.ffi)[0] = 0.0; DW_L(p.exi)[0] = 0.0; DW_L(p.fxi)[0] = 0.0;
    double allqlat = 0.0;
    for (int i = 2; i <= ncomp - 1; ++i) allqlat = allqlat + DW_G(p.lateralFlow, i, j) * DW_S(p.dx, i, j);
    for (int i = 2; i <= ncomp; ++i) {
        const FwdCoef c = forward_coef<Scan>(p, i, j, ncomp);
        const double ppi = c.ppi, qqi = c.qqi, rri = c.rri, ssi = c.ssi, sxi = c.sxi;
        DW_L(p.eei)[i - 1] = -1.0 * rri / (ppi * DW_L(p.eei)[i - 2] + qqi);
        DW_L(p.ffi)[i - 1] = (ssi - ppi * DW_L(p.ffi)[i - 2]) / (ppi * DW_L(p.eei)[i - 2] + qqi);
        DW_L(p.exi)[i - 1] = -1.0 * rri / (ppi * DW_L(p.exi)[i - 2] + qqi);
        DW_L(p.fxi)[i - 1] = (sxi - ppi * DW_L(p.fxi)[i - 2]) / (ppi * DW_L(p.exi)[i - 2] + qqi);
    }
    // (the reference also forms the coefficients of a ghost point behind the last node, :1239-1290, and never uses
    // them: qp(ncomp) = eei(ncomp) * oldQ(ncomp-1) + ffi(ncomp), :1305-1322)
    const double qp_ghost = DW_S(p.oldQ, ncomp - 1, j), qpx_ghost = 0.0;
    DW_S(p.qp, ncomp, j) = DW_L(p.eei)[ncomp - 1] * qp_ghost + DW_L(p.ffi)[ncomp - 1];
    DW_S(p.qpx, ncomp, j) = DW_L(p.exi)[ncomp - 1] * qpx_ghost + DW_L(p.fxi)[ncomp - 1];
    for (int i = ncomp - 1; i >= 1; --i) {
        DW_S(p.qp, i, j) = DW_L(p.eei)[i - 1] * DW_S(p.qp, i + 1, j) + DW_L(p.ffi)[i - 1];
        DW_S(p.qpx, i, j) = DW_L(p.exi)[i - 1] * DW_S(p.qpx, i + 1, j) + DW_L(p.fxi)[i - 1];
    }
    DW_S(p.qp, 1, j) = DW_S(p.newQ, 1, j);
    DW_S(p.qp, 1, j) = DW_S(p.qp, 1, j) + allqlat;
    for (int i = 1; i <= ncomp; ++i)
        if (fabs(DW_S(p.qp, i, j)) < p.q_llm) DW_S(p.qp, i, j) = p.q_llm;
    for (int i = 1; i <= ncomp; ++i) DW_S(p.newQ, i, j) = DW_S(p.qp, i, j);
}

// mesh_diffusive_backward (:1357-1553): water surface along reach j from its bottom node upwards.
// What a node contributes once its water surface is known -- area, perimeter, top width, roughness, and the celerity and
// diffusivity its reach averages -- depends on that node alone: NodePost.
struct NodePost {
    double co, celerity2, diffusivity2;
};
template <class Scan> DW_HD inline NodePost backward_node_post(Problem &p, int i, int j, const Scan &scan)
{
    const double *tb = scan.table(p, i, j);
    const double *elevT = tb + C_ELEV * kNel;
    const double xt = DW_S(p.newY, i, j);
    const double zz = DW_S(p.z, i, j);
    const double sq = (xt - zz) * (xt - zz);
    NodePost r;
    r.co = 1.0 * scan.apply(scan.bracket(elevT, true, zz, kNel, sq), tb + C_CONV * kNel, kNel, sq);
    const Bracket be = scan.bracket(elevT, false, 0.0, kNel, xt); // one search for the four columns at xt
    DW_G(p.newArea, i, j) = scan.apply(be, tb + C_AREA * kNel, kNel, xt);
    DW_G(p.pere, i, j) = scan.apply(be, tb + C_PERI * kNel, kNel, xt);
    DW_G(p.bo, i, j) = scan.apply(be, tb + C_TOPW * kNel, kNel, xt);
    DW_G(p.sk, i, j) = scan.apply(be, tb + C_SKK * kNel, kNel, xt);
    const double qpi = DW_S(p.qp, i, j);
    const double sfi = qpi * fabs(qpi) / (r.co * r.co);
    r.celerity2 = (double)(5.0f / 3.0f) * DW_POW(fabs(sfi), (double)0.3f) * DW_POW(fabs(qpi), (double)0.4f)
                  / DW_POW(DW_G(p.bo, i, j), (double)0.4f) / DW_POW(1. / (DW_G(p.sk, i, j) * 1.0), (double)0.6f);
    const double C_ulm = (i > 1) ? p.cfl * DW_S(p.dx, i - 1, j) / p.dtini_min : p.cfl * DW_S(p.dx, i, j) / p.dtini_min;
    if (r.celerity2 > C_ulm) r.celerity2 = C_ulm;
    r.diffusivity2 = fabs(qpi) / 2.0 / DW_G(p.bo, i, j) / fabs(sfi);
    return r;
}
template <class Scan> DW_HD inline void backward(Problem &p, int j, Scan &scan)
{
    const int ncomp = DW_FRNW(j, 1);
    scan.begin_node(p, ncomp, j);
    {
        const double *tb = scan.table(p, ncomp, j);
        const double yb = DW_S(p.newY, ncomp, j);
        const Bracket bb = scan.bracket(tb + C_ELEV * kNel, false, 0.0, kNel, yb);
        DW_G(p.newArea, ncomp, j) = scan.apply(bb, tb + C_AREA * kNel, kNel, yb);
        DW_G(p.bo, ncomp, j) = scan.apply(bb, tb + C_TOPW * kNel, kNel, yb);
    }
    for (int i = ncomp; i >= 1; --i) {
        scan.begin_node(p, i, j);               // node i (and i-1, which the depth solve reads) become resident
        if (p.counters) p.counters[1] += 1;
        const double *tb = scan.table(p, i, j);
        const NodePost np = backward_node_post(p, i, j, scan);
        DW_L(p.co)[i - 1] = np.co;
        DW_L(p.celerity2)[i - 1] = np.celerity2;
        DW_L(p.diffusivity2)[i - 1] = np.diffusivity2;
        if (i > 1) {
            const double zz = DW_S(p.z, i, j);
            const double Q_cur = DW_S(p.qp, i - 1, j), Q_ds = DW_S(p.qp, i, j);
            const double z_cur = DW_S(p.z, i - 1, j), z_ds = zz;
            double y_ds = DW_S(p.newY, i, j) - zz;
            y_ds = dmax(y_ds, (double)0.005f);
            const double y_cur = rtsafe(p, scan, scan.table(p, i - 1, j), tb, i - 1, j, Q_cur, Q_ds, z_cur, z_ds, y_ds);
            DW_S(p.newY, i - 1, j) = y_cur + DW_S(p.z, i - 1, j);
            if (DW_S(p.newY, i - 1, j) > 100000.0) DW_S(p.newY, i - 1, j) = 100000.0;
        }
    }
    double cs = 0.0, ds = 0.0;
    for (int i = 1; i <= ncomp; ++i) { cs = cs + DW_L(p.celerity2)[i - 1]; ds = ds + DW_L(p.diffusivity2)[i - 1]; }
    double cel = cs / ncomp;
    if (cel < p.C_llm) cel = p.C_llm;
    double dif = ds / ncomp;
    for (int i = 1; i <= ncomp; ++i) {
        DW_S(p.celerity, i, j) = cel;
        double d = dif;
        if (d > p.D_ulm) d = p.D_ulm;
        if (d < p.D_llm) d = p.D_llm;
        DW_S(p.diffusivity, i, j) = d;
    }
}

// calculateDT (:942-991)
DW_HD inline void calculate_dt(Problem &p, double initialTime, double time, double saveInterval, double tfin, double max_C_dx)
{
    p.dtini = p.cfl / max_C_dx;
    const int a = (int)floor((time - initialTime * 60.) / (saveInterval / 60.));
    const int b = (int)floor(((time - initialTime * 60.) + p.dtini / 60.) / (saveInterval / 60.));
    if (b > a) p.dtini = (a + 1) * (saveInterval) - (time - initialTime * 60.) * 60.;
    if (time + p.dtini / 60. > tfin * 60.) p.dtini = (tfin * 60. - time) * 60.;
}

// What diffnw does between the tables and its time loop (:488-607), in one thread: time axes, the initial water
// surface, the tributary hydrographs in the output arrays.
template <class Scan> DW_HD inline void solve_prologue(Problem &p, Scan &scan)
{
    const double TOL = (double)1e-8f;
    const double mindepth_nstab = (double)0.1f;
    const double t0 = p.timestep_ar[1], tfin = p.timestep_ar[2], saveInterval = p.timestep_ar[3];
    const double dt_ql = p.timestep_ar[4], dt_db = p.timestep_ar[6], dt_qtrib = p.timestep_ar[7];
    const int nts_ql = p.nts_ql, nts_qtrib = p.nts_qtrib, nts_db = p.nts_db, nlinks = p.nrch;
    // ---- time axes (:491-530)
    for (int n = 1; n <= nts_ql; ++n) p.tarr_ql[n] = t0 * 60.0 + dt_ql * (double)n / 60.0;
    p.tarr_ql[0] = t0 * 60;
    for (int n = 1; n <= nts_qtrib; ++n) p.tarr_qtrib[n - 1] = t0 * 60.0 + dt_qtrib * (double)(n - 1) / 60.0;
    for (int n = 1; n <= nts_db; ++n) p.tarr_db[n - 1] = t0 * 60.0 + dt_db * (double)(n - 1) / 60.0;
    // ---- initial water surface, downstream to upstream (:533-583)
    double t = t0 * 60.0;
    for (int jm = p.nmstem; jm >= 1; --jm) {
        const int j = p.mstem_frj[jm - 1], ncomp = DW_FRNW(j, 1);
        if (DW_FRNW(j, 2) < 0) {
            if (p.dsbc_option == 1) {
                for (int n = 1; n <= nts_db; ++n) p.varr_db[n - 1] = p.dbcd[n - 1] + DW_S(p.z, ncomp, j);
                t = t0 * 60.0;
                DW_S(p.oldY, ncomp, j) = intp_y(nts_db, p.tarr_db, p.varr_db, t);
                DW_S(p.newY, ncomp, j) = DW_S(p.oldY, ncomp, j);
                if ((DW_S(p.newY, ncomp, j) - DW_S(p.z, ncomp, j)) < mindepth_nstab)
                    DW_S(p.newY, ncomp, j) = mindepth_nstab + DW_S(p.z, ncomp, j);
            } else if (p.dsbc_option == 2) {
                DW_S(p.oldY, ncomp, j) = intp_tab(p, ncomp, j, C_UNIF, C_ELEV, DW_S(p.oldQ, ncomp, j));
                DW_S(p.newY, ncomp, j) = DW_S(p.oldY, ncomp, j);
            }
        } else {
            const int linknb = DW_FRNW(j, 2);
            DW_S(p.newY, ncomp, j) = DW_S(p.newY, 1, linknb);
        }
        const double wdepth = DW_S(p.newY, ncomp, j) - DW_S(p.z, ncomp, j);
        for (int i = 1; i <= ncomp - 1; ++i) DW_S(p.oldY, i, j) = wdepth + DW_S(p.z, i, j);
        backward(p, j, scan);
        for (int i = 1; i <= ncomp; ++i) {
            DW_S(p.oldY, i, j) = DW_S(p.newY, i, j);
            if (DW_S(p.oldY, i, j) < DW_S(p.oldY, ncomp, nlinks)) DW_S(p.oldY, i, j) = DW_S(p.oldY, ncomp, nlinks);
        }
    }
    // ---- tributary hydrographs into the output arrays (:590-607)
    int ts_ev = 1;
    while (t <= tfin * 60.0) {
        if (fmod((t - t0 * 60.) * 60., saveInterval) <= TOL || t == tfin * 60.) {
            for (int j = 1; j <= nlinks; ++j)
                if (!is_mainstem(p, j)) {
                    for (int n = 1; n <= nts_qtrib; ++n) p.varr_qtrib[n - 1] = p.qtrib[(n - 1) + (int64_t)(j - 1) * nts_qtrib];
                    const int nc = DW_FRNW(j, 1);
                    if (ts_ev <= p.ntss_ev) {
                        DW_EV(p.q_ev, ts_ev, nc, j) = intp_y(nts_qtrib, p.tarr_qtrib, p.varr_qtrib, t);
                        DW_EV(p.q_ev, ts_ev, 1, j) = DW_EV(p.q_ev, ts_ev, nc, j);
                    }
                }
            ts_ev = ts_ev + 1;
        }
        t = t + p.dtini / 60.;
    }
}

// Everything of diffnw after the tables exist (:488-870), in one thread.
template <class Scan> DW_HD inline void solve(Problem &p, double minDx, Scan &scan)
{
    const double TOL = (double)1e-8f;
    const double mindepth_nstab = (double)0.1f;
    const double t0 = p.timestep_ar[1], tfin = p.timestep_ar[2], saveInterval = p.timestep_ar[3];
    const double dtini_given = p.timestep_ar[0];
    const int nts_ql = p.nts_ql, nts_qtrib = p.nts_qtrib, nts_db = p.nts_db;
    solve_prologue(p, scan);
    // ---- the ordered time loop (:612-870)
    double maxCelDx = 1.0 / minDx;
    int ts_ev = 1;
    double t = t0 * 60.0;
    while (t < tfin * 60.) {
        if (p.counters) p.counters[0] += 1;
        // predictor: flow, upstream to downstream
        int ql_row = locate(p.tarr_ql, nts_ql + 1, t);
        if (ql_row == 0) ql_row = 1;
        if (ql_row == nts_ql + 1) ql_row = nts_ql;
        for (int jm = 1; jm <= p.nmstem; ++jm) {
            const int j = p.mstem_frj[jm - 1], ncomp = DW_FRNW(j, 1);
            if (jm == 1) calculate_dt(p, t0, t, saveInterval, tfin, maxCelDx);
            // lateral inflow at time t (:650-657): the time axis is shared, so the bracketing interval is found once
            // per sub-step (ql_row); varr_ql(n+1) = qlat_g(n, i, j), varr_ql(1) = qlat_g(1, i, j)
            for (int i = 1; i <= ncomp - 1; ++i) {
                const double *ql = p.qlat + (int64_t)nts_ql * ((i - 1) + (int64_t)p.mxncomp * (j - 1));
                const double y1 = ql_row == 1 ? ql[0] : ql[ql_row - 2], y2 = ql[ql_row - 1];
                DW_G(p.lateralFlow, i, j) = linterpol(p.tarr_ql[ql_row - 1], y1, p.tarr_ql[ql_row], y2, t);
            }
            if (DW_FRNW(j, 3) > 0) {
                DW_S(p.newQ, 1, j) = 0.0;
                for (int k = 1; k <= DW_FRNW(j, 3); ++k) {
                    const int usrchj = DW_FRNW(j, 3 + k);
                    double q_usrch;
                    if (is_mainstem(p, usrchj)) {
                        q_usrch = DW_S(p.newQ, DW_FRNW(usrchj, 1), usrchj);
                    } else {
                        const double tf0 = t + p.dtini / 60.;   // (the tributary's hydrograph column is read in place)
                        q_usrch = intp_y(nts_qtrib, p.tarr_qtrib, p.qtrib + (int64_t)(usrchj - 1) * nts_qtrib, tf0);
                    }
                    DW_S(p.newQ, 1, j) = DW_S(p.newQ, 1, j) + q_usrch;
                }
            } else {
                DW_S(p.newQ, 1, j) = 0.0;
            }
            DW_S(p.newQ, 1, j) = DW_S(p.newQ, 1, j) + DW_G(p.lateralFlow, 1, j) * DW_S(p.dx, 1, j);
            forward<Scan>(p, j);
        }
        // corrector: depth, downstream to upstream
        for (int jm = p.nmstem; jm >= 1; --jm) {
            const int j = p.mstem_frj[jm - 1], ncomp = DW_FRNW(j, 1);
            if (DW_FRNW(j, 2) >= 0) {
                const int linknb = DW_FRNW(j, 2);
                DW_S(p.newY, ncomp, j) = DW_S(p.newY, 1, linknb);
            } else if (p.dsbc_option == 1) {
                DW_S(p.newY, ncomp, j) = intp_y(nts_db, p.tarr_db, p.varr_db, t + p.dtini / 60.);
                if ((DW_S(p.newY, ncomp, j) - DW_S(p.z, ncomp, j)) < mindepth_nstab)
                    DW_S(p.newY, ncomp, j) = mindepth_nstab + DW_S(p.z, ncomp, j);
                DW_G(p.newArea, ncomp, j) = intp_tab(p, ncomp, j, C_ELEV, C_AREA, DW_S(p.newY, ncomp, j));
            } else if (p.dsbc_option == 2) {
                DW_S(p.newY, ncomp, j) = intp_tab(p, ncomp, j, C_UNIF, C_ELEV, fabs(DW_S(p.newQ, ncomp, j)));
                DW_G(p.newArea, ncomp, j) = intp_tab(p, ncomp, j, C_ELEV, C_AREA, DW_S(p.newY, ncomp, j));
            }
            backward(p, j, scan);
            if (jm == 1) {
                maxCelDx = 0.;
                for (int m = 1; m <= p.nmstem; ++m) {
                    const int jj = p.mstem_frj[m - 1];
                    for (int kkk = 1; kkk <= DW_FRNW(jj, 1) - 1; ++kkk)
                        maxCelDx = dmax(maxCelDx, DW_S(p.celerity, kkk, jj) / DW_S(p.dx, kkk, jj));
                }
            }
        }
        t = t + p.dtini / 60.;
        // results at the recording instants (:787-810)
        if (fmod((t - t0 * 60.) * 60., saveInterval) <= TOL || t == tfin * 60.) {
            for (int jm = 1; jm <= p.nmstem; ++jm) {
                const int j = p.mstem_frj[jm - 1], ncomp = DW_FRNW(j, 1);
                if (ts_ev + 1 <= p.ntss_ev) {
                    for (int i = 1; i <= ncomp; ++i) {
                        DW_EV(p.q_ev, ts_ev + 1, i, j) = DW_S(p.newQ, i, j);
                        DW_EV(p.elv_ev, ts_ev + 1, i, j) = DW_S(p.newY, i, j);
                        DW_EV(p.depth_ev, ts_ev + 1, i, j) = DW_EV(p.elv_ev, ts_ev + 1, i, j) - DW_S(p.z, i, j);
                    }
                    for (int k = 1; k <= DW_FRNW(j, 3); ++k) {
                        const int usrchj = DW_FRNW(j, 3 + k);
                        if (!is_mainstem(p, usrchj)) {
                            const double wdepth = DW_S(p.newY, 1, j) - DW_S(p.z, 1, j);
                            DW_EV(p.elv_ev, ts_ev + 1, DW_FRNW(usrchj, 1), usrchj) = DW_S(p.newY, 1, j);
                            DW_EV(p.depth_ev, ts_ev + 1, DW_FRNW(usrchj, 1), usrchj) = wdepth;
                        }
                    }
                }
            }
            ts_ev = ts_ev + 1;
        }
        // the initial state, once the first sub-step is known (:813-832; t0 in hours against t in minutes, as written)
        if (t == t0 + p.dtini / 60.) {
            for (int jm = 1; jm <= p.nmstem; ++jm) {
                const int j = p.mstem_frj[jm - 1], ncomp = DW_FRNW(j, 1);
                for (int i = 1; i <= ncomp; ++i) {
                    DW_EV(p.q_ev, 1, i, j) = DW_S(p.oldQ, i, j);
                    DW_EV(p.elv_ev, 1, i, j) = DW_S(p.oldY, i, j);
                    DW_EV(p.depth_ev, 1, i, j) = DW_EV(p.elv_ev, 1, i, j) - DW_S(p.z, i, j);
                }
                for (int k = 1; k <= DW_FRNW(j, 3); ++k) {
                    const int usrchj = DW_FRNW(j, 3 + k);
                    if (!is_mainstem(p, usrchj)) {
                        const double wdepth = DW_S(p.oldY, 1, j) - DW_S(p.z, 1, j);
                        DW_EV(p.elv_ev, 1, DW_FRNW(usrchj, 1), usrchj) = DW_S(p.oldY, 1, j);
                        DW_EV(p.depth_ev, 1, DW_FRNW(usrchj, 1), usrchj) = wdepth;
                    }
                }
            }
        }
        // old <- new (:835-841).  The reference copies the arrays and refills the new ones with -999; every entry of
        // newY / newQ that is read in a sub-step is written earlier in the same sub-step, so exchanging the roles of
        // the two buffers is equivalent (oldArea, newArea, pere are never read back).
        { double *sw = p.oldY; p.oldY = p.newY; p.newY = sw; }
        { double *sw = p.oldQ; p.oldQ = p.newQ; p.newQ = sw; }
    }
    (void)dtini_given;
}

// Results of one recording instant mapped from the refactored hydrofabric the domain was routed on to the original one
// (diffnw :849-920): every crosswalk row is a refactored segment (ri, rj) and the original links (oi, oj) it covers, each
// with the fraction of the link's length; flow and depth are interpolated linearly along the refactored segment, the
// water elevation is the depth over the original thalweg.  tq / te: the routed flows and elevations; used / flag:
// [mxncomp * nrch] scratch of this instant (length fraction of a link covered so far, partial covers seen).
DW_HD inline void crosswalk_instant(const Problem &p, int ts, const double *tq, const double *te, double *used, int32_t *flag)
{
    const double equiv_one = (double)0.99f;
    const int64_t nn = (int64_t)p.mxncomp * p.nrch;
    for (int64_t e = 0; e < nn; ++e) { used[e] = 0.0; flag[e] = 0; }
    for (int cwrow = 1; cwrow <= p.cwnrow; ++cwrow) {
        auto cw = [&](int c) { return p.crosswalk[(cwrow - 1) + (int64_t)(c - 1) * p.cwnrow]; };
        const int ri = (int)cw(1), rj = (int)cw(2), nlnk = (int)cw(3);
        const double rdx = DW_G(p.rdx_ar, ri, rj);
        const double slopeQ = (DW_EV(tq, ts, ri + 1, rj) - DW_EV(tq, ts, ri, rj)) / rdx;
        const double intcQ = DW_EV(tq, ts, ri, rj);
        const double slopeD = ((DW_EV(te, ts, ri + 1, rj) - DW_G(p.z, ri + 1, rj)) - (DW_EV(te, ts, ri, rj) - DW_G(p.z, ri, rj))) / rdx;
        const double intcD = DW_EV(te, ts, ri, rj) - DW_G(p.z, ri, rj);
        double dst_lnk = 0.0;
        for (int lnk = 1; lnk <= nlnk; ++lnk) {
            const int oi = (int)cw(4 + 3 * (lnk - 1)), oj = (int)cw(5 + 3 * (lnk - 1));
            const double lfrac = cw(6 + 3 * (lnk - 1));
            const double dst_top = dst_lnk;
            dst_lnk = dst_lnk + DW_G(p.dx_ar, oi, oj) * lfrac;
            const double dst_btm = dst_lnk;
            DW_G(used, oi, oj) = DW_G(used, oi, oj) + lfrac;
            if (DW_G(used, oi, oj) < equiv_one) DW_G(flag, oi, oj) = DW_G(flag, oi, oj) + 1;
            const double u = DW_G(used, oi, oj);
            const int fl = DW_G(flag, oi, oj);
            if (u >= equiv_one && fl == 0) {
                DW_EV(p.q_ev, ts, oi, oj) = intcQ + slopeQ * dst_top;
                DW_EV(p.q_ev, ts, oi + 1, oj) = intcQ + slopeQ * dst_btm;
                DW_EV(p.elv_ev, ts, oi, oj) = intcD + slopeD * dst_top + DW_G(p.z_thalweg, oi, oj);
                DW_EV(p.elv_ev, ts, oi + 1, oj) = intcD + slopeD * dst_btm + DW_G(p.z_thalweg, oi + 1, oj);
            } else if (u < equiv_one && fl == 1) {
                DW_EV(p.q_ev, ts, oi, oj) = intcQ + slopeQ * dst_top;
                DW_EV(p.elv_ev, ts, oi, oj) = intcD + slopeD * dst_top + DW_G(p.z_thalweg, oi, oj);
            } else if (u >= equiv_one && fl >= 1) {
                DW_EV(p.q_ev, ts, oi + 1, oj) = intcQ + slopeQ * dst_btm;
                DW_EV(p.elv_ev, ts, oi + 1, oj) = intcD + slopeD * dst_btm + DW_G(p.z_thalweg, oi + 1, oj);
                DW_G(flag, oi, oj) = 0;
            }
        }
    }
}

} // namespace trdw
