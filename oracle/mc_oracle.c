/*
 * TEST INFRASTRUCTURE ONLY -- see mc_oracle_impl.inc for the contract and
 * the reference citations.  Builds libmc_oracle.so with the fp32 and fp64
 * instantiations of the restated Muskingum-Cunge segment step and network
 * loop.  Compile with:  gcc -O2 -ffp-contract=off -fPIC -shared (Makefile).
 */
#include <math.h>

/* analysis aid (not thread-safe): when set, the network loops count secant iterations per
 * segment-step into hist[0..1000] */
static long *mc_oracle_iter_hist = 0;
void mc_oracle_set_iter_hist(long *hist) { mc_oracle_iter_hist = hist; }
static unsigned char *mc_oracle_iter_map = 0; /* [nseg][nsteps] iterations per segment-step */
void mc_oracle_set_iter_map(unsigned char *m) { mc_oracle_iter_map = m; }

#define REAL float
#define SFX(x) x##_f32
#define POW powf
#define SQRT sqrtf
#define FABS fabsf
#define FMAX(a, b) ((a) > (b) ? (a) : (b)) /* Fortran MAX on non-NaN operands */
#define FMIN(a, b) ((a) < (b) ? (a) : (b))
#include "mc_oracle_impl.inc"
#undef REAL
#undef SFX
#undef POW
#undef SQRT
#undef FABS

/* fp32 with the bit-reproducible power the GPU uses (t-route_amd/csrc/det_pow.h):
 * same restated algorithm, only POW differs.  GPU fp32 results are compared
 * bit-for-bit with this instantiation; it is itself compared with the libm
 * instantiation above (which is pinned bit-exactly to the reference Fortran). */
#include "../t-route_amd/csrc/det_pow.h"
#define REAL float
#define SFX(x) x##_f32det
#define POW(x, y) trmc_det_powf((x), (y), trmc_pow_tab_init)
#define SQRT sqrtf
#define FABS fabsf
#include "mc_oracle_impl.inc"
#undef REAL
#undef SFX
#undef POW
#undef SQRT
#undef FABS

void mc_oracle_det_powf(long n, const float *x, const float *y, float *out)
{
    long i;
    for (i = 0; i < n; ++i) out[i] = trmc_det_powf(x[i], y[i], trmc_pow_tab_init);
}

#define REAL double
#define SFX(x) x##_f64
#define POW pow
#define SQRT sqrt
#define FABS fabs
#include "mc_oracle_impl.inc"

/* Batch driver for the WRF-Hydro original through oracle/wrf_bind.f90
 * (argument order of MUSKINGCUNGE.f90:8-12).  in[n][15] uses the same column
 * order as mc_oracle_segments_f32; out[n][3] = qdc velc depthc. */
typedef void (*wrf_fn)(const float *qup, const float *quc, const float *qdp, const float *ql,
                       const float *dt, const float *so, const float *dx, const float *n,
                       const float *cs, const float *bw, const float *tw, const float *twcc,
                       const float *ncc, const float *depthp, float *qdc, float *velc,
                       float *depthc);
void mc_oracle_wrf_segments_f32(wrf_fn f, long n, const float *in, float *out)
{
    long i;
    for (i = 0; i < n; ++i) {
        const float *a = in + 15 * i;
        float *o = out + 3 * i;
        o[0] = o[1] = o[2] = 0.0f;
        f(&a[1], &a[2], &a[3], &a[4], &a[0], &a[12], &a[5], &a[9], &a[11], &a[6], &a[7],
          &a[8], &a[10], &a[14], &o[0], &o[1], &o[2]);
    }
}

/* The restated fp32 segment step behind the reference's bind(c) signature (pyMCsingleSegStime_NoLoop.f90:8-21), so that
 * drivers written for the reference symbol (oracle/cpu_baseline.c) can run the port where oracle/_ref is absent. */
void mc_oracle_kernel_f32(const float *dt, const float *qup, const float *quc, const float *qdp, const float *ql,
                          const float *dx, const float *bw, const float *tw, const float *twcc, const float *n,
                          const float *ncc, const float *cs, const float *s0, const float *velp, const float *depthp,
                          float *qdc, float *velc, float *depthc, float *ck, float *cn, float *X)
{
    const float in[15] = {*dt, *qup, *quc, *qdp, *ql, *dx, *bw, *tw, *twcc, *n, *ncc, *cs, *s0, *velp, *depthp};
    float out[6];
    int iters;
    mc_oracle_segment_f32(in, *qdc, out, &iters);
    *qdc = out[0]; *velc = out[1]; *depthc = out[2]; *ck = out[3]; *cn = out[4]; *X = out[5];
}
