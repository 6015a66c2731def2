/*
 * trdw.h -- C ABI of the MI355X diffusive-wave mainstem solver, exported by libtrmc.so
 * (t-route_amd/csrc/diffusive.hip + diffusive_core.hpp).
 *
 * Reference interface replaced (paths relative to the T-Route tree):
 *   src/kernel/diffusive/pydiffusive.f90:8-55          c_diffnw(...) bind(c) -- the symbol the Cython wrapper
 *   src/troute-routing/troute/routing/fast_reach/pydiffusive.h, fortran_wrappers.pxd   binds
 *   src/troute-routing/troute/routing/fast_reach/diffusive.pyx:8-127  (cdef diffnw -> c_diffnw)
 *   src/kernel/diffusive/diffusive.f90:75-940          diffnw, the routine behind it
 *
 * trdw_diffnw takes c_diffnw's arguments in c_diffnw's order, every one by reference, arrays column-major
 * (Fortran order) in double / int32 -- a Fortran or Cython caller can bind it in place of c_diffnw.  The only
 * additions are the int return value (0, or a negative status with the text in trdw_last_error()) and
 * trdw_select_device().  One call = one tailwater domain: the cross-section tables are built by one thread per
 * (node, water level); the ordered time loop runs in one workgroup per domain -- per-node and per-reach phases on 512
 * threads, the depth-solve chain from node to node on one wavefront.
 * Covered: synthetic (RouteLink) and natural (bathymetry, mxnbathy_g > 0) cross sections, both
 * downstream-boundary options, and the mapping of results from a refactored hydrofabric back to the original one
 * (cwnrow_g > 0, diffusive.f90:849-920).  No CPU fallback: without a HIP device the call fails with TRDW_ENODEVICE.
 */
#ifndef TRDW_H
#define TRDW_H

#ifdef __cplusplus
extern "C" {
#endif

typedef enum trdw_status {
    TRDW_OK = 0,
    TRDW_EINVAL = -1,
    TRDW_EUNSUPPORTED = -2,
    TRDW_ENODEVICE = -3,
    TRDW_EHIP = -4,
    TRDW_ENOMEM = -5
} trdw_status;

const char *trdw_last_error(void);
/* HIP device ordinal used by the calling thread's next trdw_diffnw calls (default 0). */
int trdw_select_device(int device);

int trdw_diffnw(const double *timestep_ar_g, const int *nts_ql_g, const int *nts_ub_g, const int *nts_db_g,
                const int *ntss_ev_g, const int *nts_qtrib_g, const int *nts_da_g, const int *mxncomp_g,
                const int *nrch_g, const double *z_ar_g, const double *bo_ar_g, const double *traps_ar_g,
                const double *tw_ar_g, const double *twcc_ar_g, const double *mann_ar_g, const double *manncc_ar_g,
                const double *so_ar_g, const double *dx_ar_g, const double *iniq, const int *frnw_col,
                const int *frnw_ar_g, const double *qlat_g, const double *ubcd_g, const double *dbcd_g,
                const double *qtrib_g, const int *paradim, const double *para_ar_g, const int *mxnbathy_g,
                const double *x_bathy_g, const double *z_bathy_g, const double *mann_bathy_g, const int *size_bathy_g,
                const double *usgs_da_g, const int *usgs_da_reach_g, const double *rdx_ar_g, const int *cwnrow_g,
                const int *cwncol_g, const double *crosswalk_g, const double *z_thalweg_g, double *q_ev_g,
                double *elv_ev_g, double *depth_ev_g);

/*
 * Several tailwater domains at once: trdw_args holds one c_diffnw argument list (same pointers, same order); block
 * b of ONE launch runs the time loop of domain b, so independent domains -- which the reference routes one after the
 * other (compute.py:1762, "TODO by-network parallel loop") -- advance side by side, one compute unit each.
 */
typedef struct trdw_args {
    const double *timestep_ar_g;
    const int *nts_ql_g, *nts_ub_g, *nts_db_g, *ntss_ev_g, *nts_qtrib_g, *nts_da_g, *mxncomp_g, *nrch_g;
    const double *z_ar_g, *bo_ar_g, *traps_ar_g, *tw_ar_g, *twcc_ar_g, *mann_ar_g, *manncc_ar_g, *so_ar_g, *dx_ar_g, *iniq;
    const int *frnw_col, *frnw_ar_g;
    const double *qlat_g, *ubcd_g, *dbcd_g, *qtrib_g;
    const int *paradim;
    const double *para_ar_g;
    const int *mxnbathy_g;
    const double *x_bathy_g, *z_bathy_g, *mann_bathy_g;
    const int *size_bathy_g;
    const double *usgs_da_g;
    const int *usgs_da_reach_g;
    const double *rdx_ar_g;
    const int *cwnrow_g, *cwncol_g;
    const double *crosswalk_g, *z_thalweg_g;
    double *q_ev_g, *elv_ev_g, *depth_ev_g;
} trdw_args;
int trdw_diffnw_batch(int ndomains, const trdw_args *args);

/* How the calls of THIS THREAD solve (the library reads no environment variable; troute_amd maps its TRDW_* test variables
 * onto this): solver 0 = the parallel time loop (one workgroup per domain: coefficients per node, recurrences per reach, the
 * depth chain on one wavefront), 1 = the whole loop in one wavefront (the first form, kept as a cross-check: same bits);
 * chain_global / window_rows: where the depth chain keeps its state (measurement knobs); phase_ticks: per-phase clock ticks
 * of the first domain printed on stderr (a diagnostic).  NULL or a zero-filled struct = the defaults. */
typedef struct trdw_options {
    int32_t struct_size;
    int32_t solver;
    int32_t chain_global;
    int32_t window_rows;
    int32_t phase_ticks;
} trdw_options;
int trdw_configure(const trdw_options *options);

/* Device time of the last trdw_diffnw / trdw_diffnw_batch call of this thread: tables_ms (cross-section tables), solve_ms (time loop). */
int trdw_last_timing(double *tables_ms, double *solve_ms);

#ifdef __cplusplus
}
#endif
#endif /* TRDW_H */
