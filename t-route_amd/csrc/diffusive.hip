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
    __device__ static double *state(double *g) { return g; }
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

// the same searches for a run whose sweep state lives in LDS: a generic pointer into LDS is (aperture << 32 | offset),
// so the low word IS the LDS address, and accesses through it are ds_read / ds_write
struct WaveScanLds : WaveScan {
    typedef __attribute__((address_space(3))) double lds_double;
    __device__ static lds_double *state(double *g)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        return (lds_double *)(uint32_t)(uintptr_t)g;
#else
        return (lds_double *)(uintptr_t)g; // (host pass of the dual compilation; never executed)
#endif
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
    if (p.mxnbathy > 0) trdw::nat_row_finish(p, node_k[n], node_j[n], l);
    else trdw::table_row_finish(p, node_k[n], node_j[n], l);
}
// natural cross sections: vertex lists (one thread per node), table levels (one thread per (node, level)), then the
// two monotonicity passes over the levels of a node, which are sequential in the level (one thread per node)
__global__ void __launch_bounds__(256) k_dw_nat_vertices(trdw::Problem p, const int32_t *node_k, const int32_t *node_j, int nnodes)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n < nnodes) trdw::nat_vertices(p, node_k[n], node_j[n]);
}
__global__ void __launch_bounds__(256) k_dw_nat_rows(trdw::Problem p, const int32_t *node_k, const int32_t *node_j, int nnodes)
{
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= (int64_t)nnodes * trdw::kNel) return;
    const int n = (int)(g / trdw::kNel), l = (int)(g % trdw::kNel) + 1;
    trdw::nat_row(p, trdw::nat_section(p, node_k[n], node_j[n]), node_k[n], node_j[n], l);
}
__global__ void __launch_bounds__(64) k_dw_nat_smooth(trdw::Problem p, const int32_t *node_k, const int32_t *node_j, int nnodes)
{
    const int n = blockIdx.x * 64 + threadIdx.x;
    if (n >= nnodes) return;
    trdw::nat_smooth(p, node_k[n], node_j[n]);
    p.z[(node_k[n] - 1) + (int64_t)(node_j[n] - 1) * p.mxncomp] = trdw::nat_section(p, node_k[n], node_j[n]).el_min; // :2031
}
// one domain of a batch: its problem, where its minDx lives, how many doubles of LDS state it was granted
struct BatchItem {
    trdw::Problem p;
    double *min_dx;
    int64_t lds_state;
};
__global__ void __launch_bounds__(64) k_dw_setup(const BatchItem *items)
{
    if (threadIdx.x == 0) {
        trdw::Problem p = items[blockIdx.x].p;
        *items[blockIdx.x].min_dx = trdw::setup_scalars(p);
    }
}
// grid = domains: block b runs the whole time loop of domain b in its one wavefront
__global__ void __launch_bounds__(64) k_dw_solve(const BatchItem *items)
{
    trdw::Problem p = items[blockIdx.x].p;
    const int64_t lds_state = items[blockIdx.x].lds_state;
    // setup_scalars ran in its own launch; its scalar results are recomputed here (they live in the by-value
    // Problem), the arrays it filled are in the work space
    p.dtini = p.timestep_ar[0];
    p.dtini_min = p.dtini / p.timestep_ar[9];
    p.cfl = p.para_ar[0]; p.C_llm = p.para_ar[1]; p.D_llm = p.para_ar[2]; p.D_ulm = p.para_ar[3];
    p.q_llm = p.para_ar[7]; p.so_llm = p.para_ar[8]; p.theta = p.para_ar[9];
    p.dsbc_option = (int)p.para_ar[10];
    HIP_DYNAMIC_SHARED(double, s_tables)
    WaveScanLds scan;
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
        trdw::solve<WaveScanLds>(p, *items[blockIdx.x].min_dx, scan);
        return;
    }
    trdw::solve<WaveScan>(p, *items[blockIdx.x].min_dx, scan);
}

// host side of one domain: device copies of its inputs, its work space, its node list
struct Domain {
    std::vector<void *> ptrs;
    trdw::Problem p;
    double *d_out = nullptr, *d_min = nullptr;
    int32_t *d_nk = nullptr, *d_nj = nullptr;
    int nnodes = 0;
    size_t nout = 0;
    int64_t lds_state = 0;
    size_t lds_bytes = 0;
    double *q_ev = nullptr, *elv_ev = nullptr, *depth_ev = nullptr; // the caller's output arrays
    ~Domain() { for (void *q : ptrs) (void)hipFree(q); }
    int up(const void *src, size_t bytes, void **out, hipStream_t st)
    {
        void *d = nullptr;
        if (hipMalloc(&d, bytes ? bytes : 8) != hipSuccess) return -1;
        ptrs.push_back(d);
        if (src && bytes && hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, st) != hipSuccess) return -1;
        *out = d;
        return 0;
    }
};

// validate one domain's arguments, copy them to the device, queue its table kernels on `st`
int prepare(const trdw_args &a, Domain &dom, hipStream_t st)
{
    if (!a.timestep_ar_g || !a.nts_ql_g || !a.nts_db_g || !a.ntss_ev_g || !a.nts_qtrib_g || !a.mxncomp_g || !a.nrch_g || !a.frnw_col
        || !a.frnw_ar_g || !a.paradim || !a.para_ar_g || !a.mxnbathy_g || !a.cwnrow_g || !a.q_ev_g || !a.elv_ev_g || !a.depth_ev_g
        || !a.nts_ub_g || !a.nts_da_g)
        return dw_fail(TRDW_EINVAL, "a required argument is NULL");
    if (*a.mxnbathy_g < 0) return dw_fail(TRDW_EINVAL, "mxnbathy_g is negative");
    if (*a.mxnbathy_g > 0 && (!a.x_bathy_g || !a.z_bathy_g || !a.mann_bathy_g || !a.size_bathy_g))
        return dw_fail(TRDW_EINVAL, "bathymetry arrays are NULL although mxnbathy_g > 0");
    if (*a.cwnrow_g != 0) return dw_fail(TRDW_EUNSUPPORTED, "the refactored-hydrofabric crosswalk (cwnrow_g > 0) is not covered");
    if (*a.paradim < 11) return dw_fail(TRDW_EINVAL, "para_ar_g needs 11 entries");
    const int mx = *a.mxncomp_g, nr = *a.nrch_g, nql = *a.nts_ql_g, nqt = *a.nts_qtrib_g, ndb = *a.nts_db_g, nev = *a.ntss_ev_g,
              fc = *a.frnw_col;
    if (mx < 2 || nr < 1 || nql < 1 || nqt < 2 || ndb < 1 || nev < 1 || fc < 5) return dw_fail(TRDW_EINVAL, "bad dimensions");
    // mainstem nodes (frnw flag 555 behind the upstream list, diffnw :383-394)
    std::vector<int32_t> node_k, node_j;
    int nmstem = 0;
    for (int j = 1; j <= nr; ++j) {
        const int nus = a.frnw_ar_g[(j - 1) + (size_t)2 * nr];
        if (nus < 0 || 3 + nus + 1 > fc) return dw_fail(TRDW_EINVAL, "frnw_ar_g: upstream count does not fit frnw_col");
        const int ncomp = a.frnw_ar_g[(j - 1)];
        if (ncomp < 2 || ncomp > mx) return dw_fail(TRDW_EINVAL, "frnw_ar_g: node count of a reach outside [2, mxncomp_g]");
        if (a.frnw_ar_g[(j - 1) + (size_t)(3 + nus) * nr] == 555) {
            ++nmstem;
            for (int k = 1; k <= ncomp; ++k) {
                node_k.push_back(k);
                node_j.push_back(j);
            }
        }
    }
    dom.nnodes = (int)node_k.size();
    if (dom.nnodes == 0) return dw_fail(TRDW_EINVAL, "no mainstem reach (flag 555) in frnw_ar_g");
    const size_t nn = (size_t)mx * nr;
    trdw::Problem &p = dom.p;
    std::memset(&p, 0, sizeof p);
    p.nts_ql = nql; p.nts_ub = *a.nts_ub_g; p.nts_db = ndb; p.ntss_ev = nev; p.nts_qtrib = nqt; p.nts_da = *a.nts_da_g;
    p.mxncomp = mx; p.nrch = nr; p.frnw_col = fc;
    void *d = nullptr;
#define DW_UP(field, src, count_, type)                                                                                    \
    if (dom.up(src, (size_t)(count_) * sizeof(type), &d, st)) return dw_fail(TRDW_ENOMEM, "device allocation/copy failed: " #field); \
    p.field = (const type *)d;
    DW_UP(timestep_ar, a.timestep_ar_g, 10, double)
    DW_UP(z_ar, a.z_ar_g, nn, double)
    DW_UP(bo_ar, a.bo_ar_g, nn, double)
    DW_UP(traps_ar, a.traps_ar_g, nn, double)
    DW_UP(tw_ar, a.tw_ar_g, nn, double)
    DW_UP(twcc_ar, a.twcc_ar_g, nn, double)
    DW_UP(mann_ar, a.mann_ar_g, nn, double)
    DW_UP(manncc_ar, a.manncc_ar_g, nn, double)
    DW_UP(dx_ar, a.dx_ar_g, nn, double)
    DW_UP(iniq, a.iniq, nn, double)
    DW_UP(frnw, a.frnw_ar_g, (size_t)nr * fc, int32_t)
    DW_UP(qlat, a.qlat_g, (size_t)nql * nn, double)
    DW_UP(dbcd, a.dbcd_g, ndb, double)
    DW_UP(qtrib, a.qtrib_g, (size_t)nqt * nr, double)
    DW_UP(para_ar, a.para_ar_g, 11, double)
    p.mxnbathy = *a.mxnbathy_g;
    if (p.mxnbathy > 0) {
        for (size_t e = 0; e < nn; ++e)
            if (a.size_bathy_g[e] < 0 || a.size_bathy_g[e] > p.mxnbathy) return dw_fail(TRDW_EINVAL, "size_bathy_g entry outside [0, mxnbathy_g]");
        for (int n_ = 0; n_ < dom.nnodes; ++n_)
            if (a.size_bathy_g[(node_k[n_] - 1) + (size_t)(node_j[n_] - 1) * mx] < 2)
                return dw_fail(TRDW_EINVAL, "a mainstem node has fewer than two bathymetry stations");
        DW_UP(x_bathy, a.x_bathy_g, (size_t)p.mxnbathy * nn, double)
        DW_UP(z_bathy, a.z_bathy_g, (size_t)p.mxnbathy * nn, double)
        DW_UP(mann_bathy, a.mann_bathy_g, (size_t)p.mxnbathy * nn, double)
        DW_UP(size_bathy, a.size_bathy_g, nn, int32_t)
    }
#undef DW_UP
    dom.nout = (size_t)nev * nn;
    double *d_work = nullptr;
    int32_t *d_frj = nullptr;
    if (dom.up(nullptr, 3 * dom.nout * sizeof(double), (void **)&dom.d_out, st)) return dw_fail(TRDW_ENOMEM, "device allocation failed: outputs");
    const int64_t nwork = trdw::work_doubles(mx, nr, nql, nqt, ndb, p.mxnbathy);
    if (dom.up(nullptr, (size_t)nwork * sizeof(double), (void **)&d_work, st)) return dw_fail(TRDW_ENOMEM, "device allocation failed: work space");
    if (dom.up(nullptr, sizeof(double), (void **)&dom.d_min, st)) return dw_fail(TRDW_ENOMEM, "device allocation failed");
    if (dom.up(nullptr, (2 * (size_t)nr + 2) * sizeof(int32_t), (void **)&d_frj, st)) return dw_fail(TRDW_ENOMEM, "device allocation failed");
    if (dom.up(node_k.data(), (size_t)dom.nnodes * sizeof(int32_t), (void **)&dom.d_nk, st)) return dw_fail(TRDW_ENOMEM, "device allocation failed");
    if (dom.up(node_j.data(), (size_t)dom.nnodes * sizeof(int32_t), (void **)&dom.d_nj, st)) return dw_fail(TRDW_ENOMEM, "device allocation failed");
    DW_TRY(hipStreamSynchronize(st)); // node_k / node_j are about to go out of scope
    DW_TRY(hipMemsetAsync(dom.d_out, 0, 3 * dom.nout * sizeof(double), st));
    DW_TRY(hipMemsetAsync(d_work, 0, (size_t)nwork * sizeof(double), st));
    p.q_ev = dom.d_out; p.elv_ev = dom.d_out + dom.nout; p.depth_ev = dom.d_out + 2 * dom.nout;
    trdw::bind_work(p, d_work);
    p.mstem_frj = d_frj;
    p.is_main = d_frj + nr;
    p.nmstem = nmstem;
    p.so_llm = a.para_ar_g[8];
    dom.q_ev = a.q_ev_g; dom.elv_ev = a.elv_ev_g; dom.depth_ev = a.depth_ev_g;
    // LDS: two elevation columns, plus the sweep state when it fits the CU's 160 KB
    dom.lds_bytes = 2 * (size_t)trdw::kNel * sizeof(double);
    const int64_t state_doubles = 10 * (int64_t)nn + 7 * (int64_t)mx + nqt;
    if (dom.lds_bytes + (size_t)state_doubles * sizeof(double) <= 160 * 1024 - 1024) {
        dom.lds_state = state_doubles;
        dom.lds_bytes += (size_t)state_doubles * sizeof(double);
    }
    return 0;
}

// route `n` domains: tables per domain, then ONE launch whose blocks are the domains
int run_batch(const trdw_args *args, int n)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return dw_fail(TRDW_ENODEVICE, "no HIP device available; this library has no CPU fallback");
    DW_TRY(hipSetDevice(g_dw_device < count ? g_dw_device : 0));
    struct Run {
        hipStream_t st = nullptr;
        hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
        void *d_items = nullptr;
        std::vector<Domain> doms;
        ~Run()
        {
            if (st) (void)hipStreamSynchronize(st);
            doms.clear();
            if (d_items) (void)hipFree(d_items);
            for (auto &e : ev)
                if (e) (void)hipEventDestroy(e);
            if (st) (void)hipStreamDestroy(st);
        }
    } run;
    // a stream of its own, so that calls from several host threads also overlap on the device
    DW_TRY(hipStreamCreateWithFlags(&run.st, hipStreamNonBlocking));
    hipStream_t st = run.st;
    for (int k = 0; k < 3; ++k) DW_TRY(hipEventCreate(&run.ev[k]));
    run.doms.resize((size_t)n);
    DW_TRY(hipEventRecord(run.ev[0], st));
    std::vector<BatchItem> items((size_t)n);
    size_t lds_max = 0;
    for (int b = 0; b < n; ++b) {
        if (int rc = prepare(args[b], run.doms[b], st)) return rc;
        items[b].p = run.doms[b].p;
        items[b].min_dx = run.doms[b].d_min;
        items[b].lds_state = run.doms[b].lds_state;
        lds_max = lds_max > run.doms[b].lds_bytes ? lds_max : run.doms[b].lds_bytes;
    }
    DW_TRY(hipMalloc(&run.d_items, (size_t)n * sizeof(BatchItem)));
    DW_TRY(hipMemcpyAsync(run.d_items, items.data(), (size_t)n * sizeof(BatchItem), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_dw_setup, dim3(n), dim3(64), 0, st, (const BatchItem *)run.d_items);
    for (int b = 0; b < n; ++b) {
        Domain &dm = run.doms[b];
        const unsigned rows = (unsigned)(((int64_t)dm.nnodes * trdw::kNel + 255) / 256);
        if (dm.p.mxnbathy > 0) {
            hipLaunchKernelGGL(k_dw_nat_vertices, dim3((dm.nnodes + 255) / 256), dim3(256), 0, st, dm.p, dm.d_nk, dm.d_nj, dm.nnodes);
            hipLaunchKernelGGL(k_dw_nat_rows, dim3(rows), dim3(256), 0, st, dm.p, dm.d_nk, dm.d_nj, dm.nnodes);
            hipLaunchKernelGGL(k_dw_nat_smooth, dim3((dm.nnodes + 63) / 64), dim3(64), 0, st, dm.p, dm.d_nk, dm.d_nj, dm.nnodes);
        } else {
            hipLaunchKernelGGL(k_dw_tables, dim3(rows), dim3(256), 0, st, dm.p, dm.d_nk, dm.d_nj, dm.nnodes);
            hipLaunchKernelGGL(k_dw_bed, dim3((dm.nnodes + 255) / 256), dim3(256), 0, st, dm.p, dm.d_nk, dm.d_nj, dm.nnodes);
        }
        hipLaunchKernelGGL(k_dw_tables_finish, dim3(rows), dim3(256), 0, st, dm.p, dm.d_nk, dm.d_nj, dm.nnodes);
    }
    DW_TRY(hipEventRecord(run.ev[1], st));
    if (lds_max > 64 * 1024)
        DW_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_dw_solve), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
    hipLaunchKernelGGL(k_dw_solve, dim3(n), dim3(64), lds_max, st, (const BatchItem *)run.d_items);
    DW_TRY(hipEventRecord(run.ev[2], st));
    DW_TRY(hipGetLastError());
    for (int b = 0; b < n; ++b) {
        Domain &dm = run.doms[b];
        DW_TRY(hipMemcpyAsync(dm.q_ev, dm.d_out, dm.nout * sizeof(double), hipMemcpyDeviceToHost, st));
        DW_TRY(hipMemcpyAsync(dm.elv_ev, dm.d_out + dm.nout, dm.nout * sizeof(double), hipMemcpyDeviceToHost, st));
        DW_TRY(hipMemcpyAsync(dm.depth_ev, dm.d_out + 2 * dm.nout, dm.nout * sizeof(double), hipMemcpyDeviceToHost, st));
    }
    DW_TRY(hipStreamSynchronize(st));
    float t01 = 0, t12 = 0;
    DW_TRY(hipEventElapsedTime(&t01, run.ev[0], run.ev[1]));
    DW_TRY(hipEventElapsedTime(&t12, run.ev[1], run.ev[2]));
    g_dw_tables_ms = t01;
    g_dw_solve_ms = t12;
    return 0;
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
    const trdw_args a = {timestep_ar_g, nts_ql_g, nts_ub_g, nts_db_g, ntss_ev_g, nts_qtrib_g, nts_da_g, mxncomp_g, nrch_g,
                         z_ar_g, bo_ar_g, traps_ar_g, tw_ar_g, twcc_ar_g, mann_ar_g, manncc_ar_g, so_ar_g, dx_ar_g, iniq,
                         frnw_col, frnw_ar_g, qlat_g, ubcd_g, dbcd_g, qtrib_g, paradim, para_ar_g, mxnbathy_g, x_bathy_g,
                         z_bathy_g, mann_bathy_g, size_bathy_g, usgs_da_g, usgs_da_reach_g, rdx_ar_g, cwnrow_g, cwncol_g,
                         crosswalk_g, z_thalweg_g, q_ev_g, elv_ev_g, depth_ev_g};
    return run_batch(&a, 1);
}

int trdw_diffnw_batch(int ndomains, const trdw_args *args)
{
    if (ndomains < 1 || !args) return dw_fail(TRDW_EINVAL, "ndomains must be >= 1 and args non-NULL");
    return run_batch(args, ndomains);
}

} // extern "C"
