// TEST INFRASTRUCTURE ONLY -- host instantiation of the diffusive-wave solver (t-route_amd/csrc/diffusive_core.hpp,
// the same source the HIP kernel compiles) behind the argument list of the reference's c_diffnw
// (src/kernel/diffusive/pydiffusive.f90:8-55).  Built with libm and without FMA contraction; validated against the
// reference Fortran built from its own sources (oracle/_ref/libdiff_ref.so) by tests/test_diffusive.py.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../t-route_amd/csrc/diffusive_core.hpp"

extern "C" int dw_oracle_diffnw(
    const double *timestep_ar_g, const int *nts_ql_g, const int *nts_ub_g, const int *nts_db_g, const int *ntss_ev_g,
    const int *nts_qtrib_g, const int *nts_da_g, const int *mxncomp_g, const int *nrch_g, const double *z_ar_g,
    const double *bo_ar_g, const double *traps_ar_g, const double *tw_ar_g, const double *twcc_ar_g, const double *mann_ar_g,
    const double *manncc_ar_g, const double *so_ar_g, const double *dx_ar_g, const double *iniq, const int *frnw_col,
    const int *frnw_ar_g, const double *qlat_g, const double *ubcd_g, const double *dbcd_g, const double *qtrib_g,
    const int *paradim, const double *para_ar_g, const int *mxnbathy_g, const double *x_bathy_g, const double *z_bathy_g,
    const double *mann_bathy_g, const int *size_bathy_g, const double *usgs_da_g, const int *usgs_da_reach_g,
    const double *rdx_ar_g, const int *cwnrow_g, const int *cwncol_g, const double *crosswalk_g, const double *z_thalweg_g,
    double *q_ev_g, double *elv_ev_g, double *depth_ev_g)
{
    (void)nts_da_g; (void)so_ar_g; (void)ubcd_g; (void)paradim; (void)usgs_da_g; (void)usgs_da_reach_g;
    trdw::Problem p;
    memset(&p, 0, sizeof p);
    p.rdx_ar = rdx_ar_g; p.crosswalk = crosswalk_g; p.z_thalweg = z_thalweg_g; p.cwnrow = *cwnrow_g; p.cwncol = *cwncol_g;
    p.timestep_ar = timestep_ar_g;
    p.nts_ql = *nts_ql_g; p.nts_ub = *nts_ub_g; p.nts_db = *nts_db_g; p.ntss_ev = *ntss_ev_g; p.nts_qtrib = *nts_qtrib_g;
    p.nts_da = *nts_da_g; p.mxncomp = *mxncomp_g; p.nrch = *nrch_g;
    p.z_ar = z_ar_g; p.bo_ar = bo_ar_g; p.traps_ar = traps_ar_g; p.tw_ar = tw_ar_g; p.twcc_ar = twcc_ar_g;
    p.mann_ar = mann_ar_g; p.manncc_ar = manncc_ar_g; p.dx_ar = dx_ar_g; p.iniq = iniq;
    p.frnw_col = *frnw_col; p.frnw = frnw_ar_g; p.qlat = qlat_g; p.dbcd = dbcd_g; p.qtrib = qtrib_g; p.para_ar = para_ar_g;
    p.mxnbathy = *mxnbathy_g; p.x_bathy = x_bathy_g; p.z_bathy = z_bathy_g; p.mann_bathy = mann_bathy_g; p.size_bathy = size_bathy_g;
    p.q_ev = q_ev_g; p.elv_ev = elv_ev_g; p.depth_ev = depth_ev_g;
    const long long nout = (long long)p.ntss_ev * p.mxncomp * p.nrch;
    for (long long e = 0; e < nout; ++e) q_ev_g[e] = elv_ev_g[e] = depth_ev_g[e] = 0.0;
    double *w = (double *)calloc((size_t)trdw::work_doubles(p.mxncomp, p.nrch, p.nts_ql, p.nts_qtrib, p.nts_db, p.mxnbathy), sizeof(double));
    int32_t *frj = (int32_t *)calloc(2 * (size_t)p.nrch + 2, sizeof(int32_t));
    if (!w || !frj) return -2;
    trdw::bind_work(p, w);
    p.mstem_frj = frj;
    p.is_main = frj + p.nrch;
    const double minDx = trdw::setup_scalars(p);
    // tables of every mainstem node; the node's bed elevation becomes the lowest point of its section
    const bool natural = p.mxnbathy > 0;
    for (int m = 0; m < p.nmstem; ++m) {
        const int j = p.mstem_frj[m], ncomp = p.frnw[(j - 1) + 0];
        for (int k = 1; k <= ncomp; ++k) {
            if (natural) {
                trdw::nat_vertices(p, k, j);
                const trdw::NatSection s = trdw::nat_section(p, k, j);
                for (int l = 1; l <= trdw::kNel; ++l) trdw::nat_row(p, s, k, j, l);
                trdw::nat_smooth(p, k, j);
                p.z[(k - 1) + (long long)(j - 1) * p.mxncomp] = s.el_min;
            } else {
                trdw::Section s;
                trdw::make_section(p, k, j, s);
                for (int l = 1; l <= trdw::kNel; ++l) trdw::table_row(p, s, k, j, l);
                p.z[(k - 1) + (long long)(j - 1) * p.mxncomp] = s.el_min;
            }
        }
    }
    for (int m = 0; m < p.nmstem; ++m) {
        const int j = p.mstem_frj[m], ncomp = p.frnw[(j - 1) + 0];
        for (int k = 1; k <= ncomp; ++k)
            for (int l = trdw::kNel; l >= 1; --l) {
                if (natural) trdw::nat_row_finish(p, k, j, l); else trdw::table_row_finish(p, k, j, l);
            }
    }
    static long long counters[4];
    counters[0] = counters[1] = counters[2] = 0;
    p.counters = (int64_t *)counters;
    trdw::SerialScan scan;
    trdw::solve(p, minDx, scan);
    if (p.cwnrow > 0) { // results back onto the original hydrofabric (diffnw :849-920)
        const long long nn = (long long)p.mxncomp * p.nrch;
        double *tq = (double *)malloc((size_t)nout * sizeof(double)), *te = (double *)malloc((size_t)nout * sizeof(double));
        double *used = (double *)malloc((size_t)nn * sizeof(double));
        int32_t *flag = (int32_t *)malloc((size_t)nn * sizeof(int32_t));
        if (!tq || !te || !used || !flag) return -2;
        memcpy(tq, q_ev_g, (size_t)nout * sizeof(double));
        memcpy(te, elv_ev_g, (size_t)nout * sizeof(double));
        for (long long e = 0; e < nout; ++e) q_ev_g[e] = elv_ev_g[e] = 0.0;
        for (int ts = 1; ts <= p.ntss_ev; ++ts) trdw::crosswalk_instant(p, ts, tq, te, used, flag);
        free(tq); free(te); free(used); free(flag);
    }
    free(w);
    free(frj);
    if (getenv("DW_ORACLE_COUNTERS")) fprintf(stderr, "sub-steps %lld node sweeps %lld funcd %lld\n", counters[0], counters[1], counters[2]);
    return 0;
}
