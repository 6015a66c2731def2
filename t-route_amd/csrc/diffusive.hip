// diffusive.hip -- the diffusive-wave mainstem solver on the device (include/trdw.h).
//
// Work decomposition (the algorithm is diffusive_core.hpp, shared with the host instantiation the tests compare
// against the reference Fortran):
//   k_dw_tables         one thread per (mainstem node, table level): hydraulic properties of the three
//                       sub-sections at that water elevation -- the data-parallel part (nodes x 501 levels)
//   k_dw_tables_finish  one thread per (node, level): dK/dA against the level below, uniform-flow column
//   k_dw_solve          ONE wavefront per tailwater domain.  The ordered time loop is a recurrence over
//                       (sub-step, reach, node); all 64 lanes execute it redundantly (uniform control flow,
//                       identical stores) and split the only wide operations inside it -- the linear scans of
//                       the 501-row tables -- with ballots and lane shuffles (WaveScan).
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

#include "../../include/trdw.h"
#include "diffusive_core.hpp"

namespace {

thread_local std::string g_dw_err;
thread_local int g_dw_device = 0;
thread_local double g_dw_tables_ms = 0.0, g_dw_solve_ms = 0.0;

int dw_fail(int code, const std::string &msg)
{
    g_dw_err = msg;
    return code;
}
#define DW_TRY(expr)                                                                                        \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess)                                                                               \
            return dw_fail(e_ == hipErrorOutOfMemory ? TRDW_ENOMEM : TRDW_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// Table searches of the time loop, for one wavefront.
//   * A search counts, with one ballot per position of each lane's run of entries, how many table entries lie
//     below the abscissa -- the tables are ascending, so that count IS the row Numerical Recipes' bisection
//     (locate) ends on and the first interval r_interpol's linear scan accepts; the loads of a search are
//     independent, where bisection or a scan would chain nine to five hundred dependent ones.
//   * The water-elevation columns of the node being solved and of the node above it -- the abscissae of nearly every
//     search of a node sweep -- are kept in LDS (2 x 4 KB); when the sweep moves one node up the upper column
//     becomes the current one and one new column is fetched.  (Staging whole 32 KB table blocks cost more than the
//     searches saved: 7.6 s against 5.8 s for 12 steps of the LowerColorado subset.)
struct WaveScan {
    double *lds;               // [2][kNel]
    const double *gcol[2];     // the global elevation column each LDS slot mirrors (nullptr = none)

    __device__ static double wmin(double v)
    {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v = trdw::dmin(v, __shfl_xor(v, d));
        return v;
    }
    __device__ void fetch(int slot, const double *col)
    {
        double *dst = lds + slot * trdw::kNel;
        const int lane = threadIdx.x & 63;
        double v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = lane + 64 * r < trdw::kNel ? col[lane + 64 * r] : 0.0;
#pragma unroll
        for (int r = 0; r < 8; ++r)
            if (lane + 64 * r < trdw::kNel) dst[lane + 64 * r] = v[r];
        gcol[slot] = col;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    __device__ void begin_node(const trdw::Problem &p, int i, int j)
    {
        const double *cur = trdw::node_block(p, i, j) + trdw::C_ELEV * trdw::kNel;
        int s = gcol[0] == cur ? 0 : (gcol[1] == cur ? 1 : -1);
        if (s < 0) { s = 0; fetch(0, cur); }
        if (i > 1) {
            const double *up = trdw::node_block(p, i - 1, j) + trdw::C_ELEV * trdw::kNel;
            if (gcol[s ^ 1] != up) fetch(s ^ 1, up);
        }
    }
    __device__ const double *table(const trdw::Problem &p, int i, int j) const { return trdw::node_block(p, i, j); }
    __device__ const double *fast(const double *col) const
    {
        return col == gcol[0] ? lds : (col == gcol[1] ? lds + trdw::kNel : col);
    }
    // number of entries with f(x[k]) < v (STRICT) or <= v
    template <bool STRICT, bool SQ> __device__ static int count_below(const double *x, double zz, int kk, double v)
    {
        const int lane = threadIdx.x & 63;
        const int per = (kk + 63) / 64, k0 = lane * per;
        int c = 0;
#pragma unroll 8
        for (int r = 0; r < per; ++r) {
            const int k = k0 + r;
            bool hit = false;
            if (k < kk) {
                const double e = SQ ? (x[k] - zz) * (x[k] - zz) : x[k];
                hit = STRICT ? e < v : e <= v;
            }
            c += __popcll(__ballot(hit));
        }
        return c;
    }
    // r_interpol's search on an ascending abscissa: extremes at the ends, first accepted interval
    // k = max(#{x < xrt} - 1, 0)
    __device__ trdw::Bracket bracket(const double *elev, bool squared, double zz, int kk, double xrt) const
    {
        const double *x = fast(elev);
        auto at = [&](int k) { return squared ? (x[k] - zz) * (x[k] - zz) : x[k]; };
        const double xmin = at(0), xmax = at(kk - 1);
        trdw::Bracket b;
        if (xrt <= xmax && xrt >= xmin) {
            const int c = squared ? count_below<true, true>(x, zz, kk, xrt) : count_below<true, false>(x, zz, kk, xrt);
            b.mode = 0;
            b.k = c > 0 ? c - 1 : 0;
            b.xk = at(b.k);
            b.xk1 = at(b.k + 1);
        } else if (xrt >= xmax) {
            b.mode = 1;
            b.k = kk - 2;
            b.xk = at(kk - 2);
            b.xk1 = at(kk - 1);
        } else {
            b.mode = 2;
            b.k = 0;
            b.xk = b.xk1 = 0.0;
        }
        return b;
    }
    __device__ double apply(const trdw::Bracket &b, const double *y, int kk, double xrt) const
    {
        if (b.mode <= 1) return (xrt - b.xk) / (b.xk1 - b.xk) * (y[b.k + 1] - y[b.k]) + y[b.k];
        const int lane = threadIdx.x & 63;
        double ym = INFINITY;
        for (int k = lane; k < kk; k += 64) ym = trdw::dmin(ym, y[k]);
        return wmin(ym);
    }
    // Numerical Recipes' locate on an ascending table: the bisection ends at jl = #{xx <= x}
    __device__ int locate_row(const double *xx_, int n, double x) const
    {
        const double *xx = fast(xx_);
        const int jl = count_below<false, false>(xx, 0.0, n, x);
        if (x == xx[0]) return 1;
        if (x == xx[n - 1]) return n - 1;
        return jl;
    }
};

// node list of the mainstem: node n -> (k, reach j), 1-based
__global__ void __launch_bounds__(256) k_dw_tables(trdw::Problem p, const int32_t *node_k, const int32_t *node_j, int nnodes)
{
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= (int64_t)nnodes * trdw::kNel) return;
    const int n = (int)(g / trdw::kNel), l = (int)(g % trdw::kNel) + 1;
    trdw::Section s;
    trdw::make_section(p, node_k[n], node_j[n], s);
    trdw::table_row(p, s, node_k[n], node_j[n], l);
}
__global__ void __launch_bounds__(256) k_dw_bed(trdw::Problem p, const int32_t *node_k, const int32_t *node_j, int nnodes)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= nnodes) return;
    trdw::Section s;
    trdw::make_section(p, node_k[n], node_j[n], s);
    p.z[(node_k[n] - 1) + (int64_t)(node_j[n] - 1) * p.mxncomp] = s.el_min;   // readXsection :2425
}
__global__ void __launch_bounds__(256) k_dw_tables_finish(trdw::Problem p, const int32_t *node_k, const int32_t *node_j, int nnodes)
{
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= (int64_t)nnodes * trdw::kNel) return;
    const int n = (int)(g / trdw::kNel), l = (int)(g % trdw::kNel) + 1;
    // dK/dA of level l reads the conveyance and area of level l-1, which this pass does not modify
    trdw::table_row_finish(p, node_k[n], node_j[n], l);
}
__global__ void __launch_bounds__(64) k_dw_setup(trdw::Problem p, double *min_dx)
{
    if (threadIdx.x == 0) *min_dx = trdw::setup_scalars(p);
}
__global__ void __launch_bounds__(64) k_dw_solve(trdw::Problem p, const double *min_dx, int64_t lds_state)
{
    // setup_scalars ran in its own launch; its scalar results are recomputed here (they live in the by-value
    // Problem), the arrays it filled are in the work space
    p.dtini = p.timestep_ar[0];
    p.dtini_min = p.dtini / p.timestep_ar[9];
    p.cfl = p.para_ar[0]; p.C_llm = p.para_ar[1]; p.D_llm = p.para_ar[2]; p.D_ulm = p.para_ar[3];
    p.q_llm = p.para_ar[7]; p.so_llm = p.para_ar[8]; p.theta = p.para_ar[9];
    p.dsbc_option = (int)p.para_ar[10];
    HIP_DYNAMIC_SHARED(double, s_tables)
    WaveScan scan;
    scan.lds = s_tables;
    scan.gcol[0] = scan.gcol[1] = nullptr;
    // The per-node state of the sweeps (ten arrays) and the per-reach scratch lines move into LDS when they fit:
    // a lone wavefront cannot hide the latency of the hundreds of dependent loads a sub-step makes, and from LDS
    // each costs tens of cycles instead of an L2 round trip.  (lds_state = number of doubles granted by the host.)
    if (lds_state > 0) {
        const int64_t nn = (int64_t)p.mxncomp * p.nrch;
        double *w = s_tables + 2 * trdw::kNel;
        double **grid[] = {&p.z, &p.dx, &p.celerity, &p.diffusivity, &p.qpx, &p.qp, &p.oldQ, &p.newQ, &p.oldY, &p.newY};
        for (int a = 0; a < 10; ++a) {
            double *src = *grid[a];
            for (int64_t e = threadIdx.x & 63; e < nn; e += 64) w[e] = src[e];
            *grid[a] = w;
            w += nn;
        }
        double **line[] = {&p.eei, &p.ffi, &p.exi, &p.fxi, &p.celerity2, &p.diffusivity2, &p.co};
        for (int a = 0; a < 7; ++a) { *line[a] = w; w += p.mxncomp; }
        double *tq = w;
        for (int e = threadIdx.x & 63; e < p.nts_qtrib; e += 64) tq[e] = 0.0;
        p.tarr_qtrib = tq;     // (filled by solve() itself)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    trdw::solve(p, *min_dx, scan);
}

} // namespace

extern "C" {

const char *trdw_last_error(void) { return g_dw_err.c_str(); }

int trdw_select_device(int device)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return dw_fail(TRDW_ENODEVICE, "no HIP device available; this library has no CPU fallback");
    if (device < 0 || device >= count) return dw_fail(TRDW_EINVAL, "device ordinal out of range");
    g_dw_device = device;
    return 0;
}

int trdw_last_timing(double *tables_ms, double *solve_ms)
{
    if (tables_ms) *tables_ms = g_dw_tables_ms;
    if (solve_ms) *solve_ms = g_dw_solve_ms;
    return 0;
}

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
                double *elv_ev_g, double *depth_ev_g)
{
    (void)so_ar_g; (void)ubcd_g; (void)x_bathy_g; (void)z_bathy_g; (void)mann_bathy_g; (void)size_bathy_g; (void)usgs_da_g;
    (void)usgs_da_reach_g; (void)rdx_ar_g; (void)cwncol_g; (void)crosswalk_g; (void)z_thalweg_g; (void)nts_ub_g; (void)nts_da_g;
    if (!timestep_ar_g || !nts_ql_g || !nts_db_g || !ntss_ev_g || !nts_qtrib_g || !mxncomp_g || !nrch_g || !frnw_col || !frnw_ar_g
        || !paradim || !para_ar_g || !mxnbathy_g || !cwnrow_g || !q_ev_g || !elv_ev_g || !depth_ev_g)
        return dw_fail(TRDW_EINVAL, "a required argument is NULL");
    if (*mxnbathy_g != 0) return dw_fail(TRDW_EUNSUPPORTED, "natural cross sections (mxnbathy_g > 0) are not covered");
    if (*cwnrow_g != 0) return dw_fail(TRDW_EUNSUPPORTED, "the refactored-hydrofabric crosswalk (cwnrow_g > 0) is not covered");
    if (*paradim < 11) return dw_fail(TRDW_EINVAL, "para_ar_g needs 11 entries");
    const int mx = *mxncomp_g, nr = *nrch_g, nql = *nts_ql_g, nqt = *nts_qtrib_g, ndb = *nts_db_g, nev = *ntss_ev_g, fc = *frnw_col;
    if (mx < 2 || nr < 1 || nql < 1 || nqt < 2 || ndb < 1 || nev < 1 || fc < 5) return dw_fail(TRDW_EINVAL, "bad dimensions");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return dw_fail(TRDW_ENODEVICE, "no HIP device available; this library has no CPU fallback");
    DW_TRY(hipSetDevice(g_dw_device < count ? g_dw_device : 0));

    // mainstem nodes (frnw flag 555 behind the upstream list, diffnw :383-394)
    std::vector<int32_t> node_k, node_j;
    for (int j = 1; j <= nr; ++j) {
        const int nus = frnw_ar_g[(j - 1) + (size_t)2 * nr];
        if (nus < 0 || 3 + nus + 1 > fc) return dw_fail(TRDW_EINVAL, "frnw_ar_g: upstream count does not fit frnw_col");
        const int ncomp = frnw_ar_g[(j - 1)];
        if (ncomp < 2 || ncomp > mx) return dw_fail(TRDW_EINVAL, "frnw_ar_g: node count of a reach outside [2, mxncomp_g]");
        if (frnw_ar_g[(j - 1) + (size_t)(3 + nus) * nr] == 555)
            for (int k = 1; k <= ncomp; ++k) {
                node_k.push_back(k);
                node_j.push_back(j);
            }
    }
    const int nnodes = (int)node_k.size();
    if (nnodes == 0) return dw_fail(TRDW_EINVAL, "no mainstem reach (flag 555) in frnw_ar_g");

    const size_t nn = (size_t)mx * nr;
    struct Dev {
        std::vector<void *> ptrs;
        ~Dev() { for (void *q : ptrs) (void)hipFree(q); }
        int up(const void *src, size_t bytes, void **out)
        {
            void *d = nullptr;
            if (hipMalloc(&d, bytes ? bytes : 8) != hipSuccess) return -1;
            ptrs.push_back(d);
            if (src && bytes && hipMemcpy(d, src, bytes, hipMemcpyHostToDevice) != hipSuccess) return -1;
            *out = d;
            return 0;
        }
    } dev;
    trdw::Problem p;
    std::memset(&p, 0, sizeof p);
    p.nts_ql = nql; p.nts_ub = *nts_ub_g; p.nts_db = ndb; p.ntss_ev = nev; p.nts_qtrib = nqt; p.nts_da = *nts_da_g;
    p.mxncomp = mx; p.nrch = nr; p.frnw_col = fc;
    void *d = nullptr;
#define DW_UP(field, src, count_, type)                                                                  \
    if (dev.up(src, (size_t)(count_) * sizeof(type), &d)) return dw_fail(TRDW_ENOMEM, "device allocation/copy failed: " #field); \
    p.field = (const type *)d;
    DW_UP(timestep_ar, timestep_ar_g, 10, double)
    DW_UP(z_ar, z_ar_g, nn, double)
    DW_UP(bo_ar, bo_ar_g, nn, double)
    DW_UP(traps_ar, traps_ar_g, nn, double)
    DW_UP(tw_ar, tw_ar_g, nn, double)
    DW_UP(twcc_ar, twcc_ar_g, nn, double)
    DW_UP(mann_ar, mann_ar_g, nn, double)
    DW_UP(manncc_ar, manncc_ar_g, nn, double)
    DW_UP(dx_ar, dx_ar_g, nn, double)
    DW_UP(iniq, iniq, nn, double)
    DW_UP(frnw, frnw_ar_g, (size_t)nr * fc, int32_t)
    DW_UP(qlat, qlat_g, (size_t)nql * nn, double)
    DW_UP(dbcd, dbcd_g, ndb, double)
    DW_UP(qtrib, qtrib_g, (size_t)nqt * nr, double)
    DW_UP(para_ar, para_ar_g, 11, double)
#undef DW_UP
    const size_t nout = (size_t)nev * nn;
    double *d_out = nullptr, *d_work = nullptr, *d_min = nullptr;
    int32_t *d_frj = nullptr, *d_nk = nullptr, *d_nj = nullptr;
    if (dev.up(nullptr, 3 * nout * sizeof(double), (void **)&d_out)) return dw_fail(TRDW_ENOMEM, "device allocation failed: outputs");
    const int64_t nwork = trdw::work_doubles(mx, nr, nql, nqt, ndb);
    if (dev.up(nullptr, (size_t)nwork * sizeof(double), (void **)&d_work)) return dw_fail(TRDW_ENOMEM, "device allocation failed: work space");
    if (dev.up(nullptr, sizeof(double), (void **)&d_min)) return dw_fail(TRDW_ENOMEM, "device allocation failed");
    if (dev.up(nullptr, (2 * (size_t)nr + 2) * sizeof(int32_t), (void **)&d_frj)) return dw_fail(TRDW_ENOMEM, "device allocation failed");
    if (dev.up(node_k.data(), (size_t)nnodes * sizeof(int32_t), (void **)&d_nk)) return dw_fail(TRDW_ENOMEM, "device allocation failed");
    if (dev.up(node_j.data(), (size_t)nnodes * sizeof(int32_t), (void **)&d_nj)) return dw_fail(TRDW_ENOMEM, "device allocation failed");
    DW_TRY(hipMemset(d_out, 0, 3 * nout * sizeof(double)));
    DW_TRY(hipMemset(d_work, 0, (size_t)nwork * sizeof(double)));
    p.q_ev = d_out; p.elv_ev = d_out + nout; p.depth_ev = d_out + 2 * nout;
    trdw::bind_work(p, d_work);
    p.mstem_frj = d_frj;
    p.is_main = d_frj + nr;

    hipEvent_t ev[3];
    for (auto &e : ev) DW_TRY(hipEventCreate(&e));
    DW_TRY(hipEventRecord(ev[0], 0));
    hipLaunchKernelGGL(k_dw_setup, dim3(1), dim3(64), 0, 0, p, d_min);
    // (setup_scalars fills p.nmstem in the kernel's copy of p: recompute it for the launches below)
    p.nmstem = 0;
    {
        std::vector<int32_t> frj;
        for (int j = 1; j <= nr; ++j) {
            const int nus = frnw_ar_g[(j - 1) + (size_t)2 * nr];
            if (frnw_ar_g[(j - 1) + (size_t)(3 + nus) * nr] == 555) frj.push_back(j);
        }
        p.nmstem = (int)frj.size();
    }
    p.so_llm = para_ar_g[8];
    const unsigned rows = (unsigned)(((int64_t)nnodes * trdw::kNel + 255) / 256);
    hipLaunchKernelGGL(k_dw_tables, dim3(rows), dim3(256), 0, 0, p, d_nk, d_nj, nnodes);
    hipLaunchKernelGGL(k_dw_bed, dim3((nnodes + 255) / 256), dim3(256), 0, 0, p, d_nk, d_nj, nnodes);
    hipLaunchKernelGGL(k_dw_tables_finish, dim3(rows), dim3(256), 0, 0, p, d_nk, d_nj, nnodes);
    DW_TRY(hipEventRecord(ev[1], 0));
    size_t lds_bytes = 2 * (size_t)trdw::kNel * sizeof(double);
    const int64_t state_doubles = 10 * (int64_t)nn + 7 * (int64_t)mx + nqt;
    int64_t lds_state = 0;
    if (lds_bytes + (size_t)state_doubles * sizeof(double) <= 160 * 1024 - 1024) { // the CU's 160 KB
        lds_state = state_doubles;
        lds_bytes += (size_t)state_doubles * sizeof(double);
        DW_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_dw_solve), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    }
    hipLaunchKernelGGL(k_dw_solve, dim3(1), dim3(64), lds_bytes, 0, p, d_min, lds_state);
    DW_TRY(hipEventRecord(ev[2], 0));
    DW_TRY(hipGetLastError());
    DW_TRY(hipDeviceSynchronize());
    float t01 = 0, t12 = 0;
    DW_TRY(hipEventElapsedTime(&t01, ev[0], ev[1]));
    DW_TRY(hipEventElapsedTime(&t12, ev[1], ev[2]));
    g_dw_tables_ms = t01;
    g_dw_solve_ms = t12;
    for (auto &e : ev) (void)hipEventDestroy(e);
    DW_TRY(hipMemcpy(q_ev_g, d_out, nout * sizeof(double), hipMemcpyDeviceToHost));
    DW_TRY(hipMemcpy(elv_ev_g, d_out + nout, nout * sizeof(double), hipMemcpyDeviceToHost));
    DW_TRY(hipMemcpy(depth_ev_g, d_out + 2 * nout, nout * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

} // extern "C"
