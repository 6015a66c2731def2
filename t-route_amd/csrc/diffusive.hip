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

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/trdw.h"
#include "diffusive_core.hpp"

namespace {

thread_local std::string g_dw_err;
thread_local int g_dw_device = 0;
thread_local double g_dw_tables_ms = 0.0, g_dw_solve_ms = 0.0;
thread_local trdw_options g_options = {(int32_t)sizeof(trdw_options), 0, 0, 0, 0};

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

// ------------------------------------------------------------------------------------------------------------------
// The time loop in parallel form (k_dw_solve_par): one workgroup of kParThreads threads per tailwater domain.
//
// Of a sub-step of diffnw only one thing is a recurrence from node to node: the depth solve of the corrector sweep, where
// node i needs the new depth of node i+1 (and, across a junction, the top node of the reach below).  Everything else
// depends on the old time level or on the node's own new state:
//   * predictor: the Crank-Nicolson coefficients of a node (forward_coef) -- one thread per node; the Thomas recurrences
//     and the back-substitution run along a REACH and never cross a junction (the upstream boundary of a reach enters its
//     first node only, forward()) -- one thread per reach; junction inflows -- one thread per reach;
//   * corrector, before the chain: the bracket of the depth solve and the table look-ups of its first three function
//     evaluations (depth_pre) -- one thread per node;
//   * the chain itself: one wavefront, from the domain's outlet upstream, everything it reads in LDS -- per node a record of
//     12 doubles from depth_pre and a window of kWinRows rows of the four table columns the iteration reads, centred on the
//     row the node's water surface was found in a sub-step ago (a water surface moves by a fraction of a table row per
//     sub-step); an abscissa outside the window takes the search over the whole column in global memory instead;
//   * corrector, after the chain: area, top width, roughness, celerity and diffusivity of a node (backward_node_post) and
//     the refresh of its window -- one thread per node; reach means -- one thread per reach.
// The arithmetic of every piece is the function the serial solver calls (diffusive_core.hpp), so the bits are the same.
constexpr int kParThreads = 512;
// table rows of a window (WR, a template parameter of the kernel: the host takes the widest of 7 / 6 / 5 whose chain state
// fits LDS); doubles of a window: the row elevations, then per column and interval {ordinate, quotient}
constexpr int win_doubles(int wr) { return wr + 3 * 2 * (wr - 1); }
constexpr int kRecDoubles = 12;                   // (the last one is the node's new water surface)
constexpr size_t par_lds_bytes(int nnodes, int wr)
{
    return (size_t)nnodes * ((kRecDoubles + win_doubles(wr)) * sizeof(double) + 2 * sizeof(int32_t));
}

// one thread, its own searches (bisection on an ascending column), tables in global memory
struct LaneScan {
    __device__ static double *state(double *g) { return g; }
    __device__ const double *table(const trdw::Problem &p, int i, int j) const { return trdw::node_block(p, i, j); }
    __device__ int locate_row(const double *xx, int n, double x) const { return trdw::locate(xx, n, x); }
    template <bool STRICT, bool SQ> __device__ static int count_below(const double *x, double zz, int kk, double v)
    {
        int lo = 0, hi = kk; // first index whose entry is not below v
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            const double e = SQ ? (x[mid] - zz) * (x[mid] - zz) : x[mid];
            if (STRICT ? e < v : e <= v) lo = mid + 1; else hi = mid;
        }
        return lo;
    }
    __device__ trdw::Bracket bracket(const double *x, bool squared, double zz, int kk, double xrt) const
    {
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
        double ym = y[0];
        for (int k = 1; k < kk; ++k) ym = trdw::dmin(ym, y[k]);
        return ym;
    }
};

struct ParItem {
    trdw::Problem p;
    double *min_dx;
    const int32_t *nk, *nj;      // node n -> (node i, reach j), mainstem reaches in routing order, nodes 1..ncomp
    const int32_t *rbase;        // [nmstem + 1] first node of the m-th mainstem reach
    const int32_t *m_of_reach;   // [nrch] reach j (1-based) -> m, -1 for tributaries
    double *scratch;             // [nnodes][12]: forward coefficients, recurrence lines, node contributions
    double *chain_state;         // records and windows of the chain in global memory, for a domain too long for LDS
    unsigned long long *phase_ticks; // developer aid (trdw_options.phase_ticks): [8] wall-clock ticks per phase, [8] window misses
    int nnodes;
};

// the chain's view of a node: record + window in LDS
struct ChainLds {
    double *rec;    // [nnodes][kRecDoubles]: 0 y_norm 1 x1 2 x2 3..5 sf 6 df_mid 7 slope 8 dxi 9 z 10 Q 11 -
    double *win;    // [nnodes][win_doubles(WR)]: elevations of the rows, then per column (conveyance, dK/dA, top width) and
                    // interval {ordinate at its lower row, (y2 - y1) / (x2 - x1)} -- linterpol's quotient, formed ahead
    int32_t *w0;    // [nnodes][2]: first table row of the window (0-based; -1: none); interval of the last look-up |
                    // (intervals narrower than linterpol's 1e-4 as a bit mask) << 8
    __device__ double &newY(int n) const { return rec[(size_t)n * kRecDoubles + 11]; }
    unsigned long long *misses; // developer aid: look-ups that left the window
};

// (re)load the window of node n so that interval k of its table sits in the middle
template <int WR> __device__ __forceinline__ void window_load(const ChainLds &L, int n, const double *tb, int k)
{
    int w0 = k - (WR - 1) / 2 + ((WR - 1) % 2 == 0 ? 1 : 0);
    w0 = w0 < 0 ? 0 : (w0 > trdw::kNel - WR ? trdw::kNel - WR : w0);
    double *w = L.win + (size_t)n * win_doubles(WR);
    const double *xe = tb + trdw::C_ELEV * trdw::kNel + w0;
    for (int r = 0; r < WR; ++r) w[r] = xe[r];
    int tiny = 0;
    for (int r = 0; r < WR - 1; ++r)
        if (fabs(xe[r + 1] - xe[r]) < (double)0.0001f) tiny |= 1 << r;
    const int cols[3] = {trdw::C_CONV, trdw::C_DKDA, trdw::C_TOPW};
    for (int c = 0; c < 3; ++c) {
        const double *y = tb + cols[c] * trdw::kNel + w0;
        double *o = w + WR + c * 2 * (WR - 1);
        for (int r = 0; r < WR - 1; ++r) {
            // linterpol: (y2 - y1) / (x2 - x1) * (x - x1) + y1, or the mean of the ordinates on a degenerate interval
            const bool deg = (tiny >> r) & 1;
            o[2 * r] = deg ? 0.5 * (y[r] + y[r + 1]) : y[r];
            o[2 * r + 1] = deg ? 0.0 : (y[r + 1] - y[r]) / (xe[r + 1] - xe[r]);
        }
    }
    L.w0[2 * n] = w0;
    int g = k - w0;
    g = g < 0 ? 0 : (g > WR - 2 ? WR - 2 : g);
    L.w0[2 * n + 1] = g | (tiny << 8);
}
// A node's window as the chain holds it in registers: the row elevations and the interval of the last look-up (the
// guess); the other intervals stay in LDS.  The chain fetches a node's window while it is still solving the node before,
// so a look-up that falls into the guessed interval -- nearly all do: the iterates of one node lie within centimetres of
// each other -- costs compares, selects and a multiply-add, no memory access.
template <int WR> struct WinRegs {
    int n, w0, g, tiny;
    double xe[WR];
    double xa, xb;           // the guessed interval [xa, xb)
    double yc, qc, yd, qd, yt, qt;
};
template <int WR> __device__ __forceinline__ void win_fetch_interval(const ChainLds &L, WinRegs<WR> &W, int g)
{
    const double *oc = L.win + (size_t)W.n * win_doubles(WR) + WR;
    W.g = g;
    W.xa = W.xe[0];
    W.xb = W.xe[1];
#pragma unroll
    for (int r = 1; r < WR - 1; ++r) {
        W.xa = g == r ? W.xe[r] : W.xa;
        W.xb = g == r ? W.xe[r + 1] : W.xb;
    }
    W.yc = oc[2 * g]; W.qc = oc[2 * g + 1];
    W.yd = oc[2 * (WR - 1) + 2 * g]; W.qd = oc[2 * (WR - 1) + 2 * g + 1];
    W.yt = oc[4 * (WR - 1) + 2 * g]; W.qt = oc[4 * (WR - 1) + 2 * g + 1];
}
template <int WR> __device__ __forceinline__ WinRegs<WR> win_fetch(const ChainLds &L, int n)
{
    WinRegs<WR> W;
    W.n = n;
    const double *w = L.win + (size_t)n * win_doubles(WR);
    W.w0 = L.w0[2 * n];
    const int meta = L.w0[2 * n + 1];
    W.tiny = meta >> 8;
#pragma unroll
    for (int r = 0; r < WR; ++r) W.xe[r] = w[r];
    win_fetch_interval<WR>(L, W, meta & 0xff);
    return W;
}
// conveyance (and optionally dK/dA, top width) of the window's node (i, j) at water elevation elv; an elevation outside the
// window takes the wavefront's search over the node's table in global memory
template <bool ALL, int WR>
__device__ __forceinline__ void chain_lookup(const trdw::Problem &p, const ChainLds &L, const WaveScan &ws, WinRegs<WR> &W, int i, int j, double elv,
                                             double &conv, double &dKdA, double &topw)
{
    if (DW_UNIFORM(W.w0 >= 0 && W.xe[0] < elv && elv < W.xe[WR - 1])) {
        if (DW_UNIFORM(!(W.xa <= elv && elv < W.xb))) { // the look-up moved to another interval of the window
            int c = 0;
#pragma unroll
            for (int r = 0; r < WR; ++r) c += W.xe[r] <= elv ? 1 : 0;
            win_fetch_interval<WR>(L, W, c - 1);
        }
        const double x1 = W.xa;
        const bool deg = (W.tiny >> W.g) & 1;
        conv = deg ? W.yc : W.qc * (elv - x1) + W.yc;
        if (ALL) {
            dKdA = deg ? W.yd : W.qd * (elv - x1) + W.yd;
            topw = deg ? W.yt : W.qt * (elv - x1) + W.yt;
        }
        return;
    }
    if (L.misses && (threadIdx.x & 63) == 0) ++*L.misses;
    const double *tb = trdw::node_block(p, i, j);
    const int irow = trdw::row_blk(ws, tb, trdw::C_ELEV, elv);
    conv = trdw::at_row(tb, trdw::C_ELEV, trdw::C_CONV, irow, elv);
    if (ALL) {
        dKdA = trdw::at_row(tb, trdw::C_ELEV, trdw::C_DKDA, irow, elv);
        topw = trdw::at_row(tb, trdw::C_ELEV, trdw::C_TOPW, irow, elv);
    }
}

template <int WR, bool IN_LDS> __global__ void __launch_bounds__(kParThreads) k_dw_solve_par(const ParItem *items)
{
    using namespace trdw;
    typedef LaneScan Scan; // (DW_S / DW_L: the sweep state stays in global memory; flat pointers)
    const ParItem &it = items[blockIdx.x];
    Problem p = it.p;
    p.dtini = p.timestep_ar[0];
    p.dtini_min = p.dtini / p.timestep_ar[9];
    p.cfl = p.para_ar[0]; p.C_llm = p.para_ar[1]; p.D_llm = p.para_ar[2]; p.D_ulm = p.para_ar[3];
    p.q_llm = p.para_ar[7]; p.so_llm = p.para_ar[8]; p.theta = p.para_ar[9];
    p.dsbc_option = (int)p.para_ar[10];
    const int tid = threadIdx.x, nnodes = it.nnodes, nmstem = p.nmstem;
    const int32_t *nk = it.nk, *nj = it.nj, *rbase = it.rbase, *m_of_reach = it.m_of_reach;
    double *scr = it.scratch;
    HIP_DYNAMIC_SHARED(double, s_mem)
    ChainLds L;
    L.rec = IN_LDS ? s_mem : it.chain_state;
    L.win = L.rec + (size_t)nnodes * kRecDoubles;
    L.w0 = (int32_t *)(L.win + (size_t)nnodes * win_doubles(WR));
    L.misses = it.phase_ticks ? it.phase_ticks + 8 : nullptr;
    __shared__ double s_red[kParThreads / 64];
    WaveScan ws;
    ws.lds = s_mem; // the serial prologue's two elevation columns, where the records and windows will be (host: >= 8 KB)
    ws.gcol[0] = ws.gcol[1] = nullptr;
    LaneScan ls;

    // ---- prologue: the serial code, one wavefront (a few hundred node sweeps, once)
    if (tid < 64) solve_prologue<WaveScan>(p, ws);
    ws.gcol[0] = ws.gcol[1] = nullptr; // the chain's fall-back searches read the columns where they are
    __threadfence_block();
    __syncthreads();

    const double TOL = (double)1e-8f;
    const double mindepth_nstab = (double)0.1f;
    const double t0 = p.timestep_ar[1], tfin = p.timestep_ar[2], saveInterval = p.timestep_ar[3];
    const int nts_ql = p.nts_ql, nts_qtrib = p.nts_qtrib, nts_db = p.nts_db;

    // windows around the initial water surface
    for (int n = tid; n < nnodes; n += kParThreads) {
        const int i = nk[n], j = nj[n];
        const double *tb = node_block(p, i, j);
        const int k = LaneScan::count_below<true, false>(tb + C_ELEV * kNel, 0.0, kNel, DW_S(p.oldY, i, j)) - 1;
        window_load<WR>(L, n, tb, k);
    }
    __syncthreads();

    double maxCelDx = 1.0 / *it.min_dx;
    int ts_ev = 1;
    double t = t0 * 60.0;
    unsigned long long *ticks = it.phase_ticks;
    unsigned long long tk = ticks ? wall_clock64() : 0;
#define DW_PHASE(k)                                        \
    if (ticks && tid == 0) {                               \
        const unsigned long long now_ = wall_clock64();    \
        ticks[k] += now_ - tk;                             \
        tk = now_;                                         \
    }
    while (t < tfin * 60.) {
        // ---- predictor ------------------------------------------------------------------------------------------
        int ql_row = locate(p.tarr_ql, nts_ql + 1, t);
        if (ql_row == 0) ql_row = 1;
        if (ql_row == nts_ql + 1) ql_row = nts_ql;
        calculate_dt(p, t0, t, saveInterval, tfin, maxCelDx);
        for (int n = tid; n < nnodes; n += kParThreads) { // lateral inflow and Crank-Nicolson coefficients of a node
            const int i = nk[n], j = nj[n], ncomp = DW_FRNW(j, 1);
            if (i <= ncomp - 1) {
                const double *ql = p.qlat + (int64_t)nts_ql * ((i - 1) + (int64_t)p.mxncomp * (j - 1));
                const double y1 = ql_row == 1 ? ql[0] : ql[ql_row - 2], y2 = ql[ql_row - 1];
                DW_G(p.lateralFlow, i, j) = linterpol(p.tarr_ql[ql_row - 1], y1, p.tarr_ql[ql_row], y2, t);
            }
            if (i >= 2) {
                const FwdCoef c = forward_coef<Scan>(p, i, j, ncomp);
                double *q = scr + (size_t)n * 12;
                q[0] = c.ppi; q[1] = c.qqi; q[2] = c.rri; q[3] = c.ssi; q[4] = c.sxi;
            }
        }
        __syncthreads();
        DW_PHASE(0)
        for (int m = tid; m < nmstem; m += kParThreads) { // the recurrences of a reach; new flows of its nodes 2..ncomp
            const int j = p.mstem_frj[m], ncomp = DW_FRNW(j, 1), n0 = rbase[m];
            double allqlat = 0.0;
            for (int i = 2; i <= ncomp - 1; ++i) allqlat = allqlat + DW_G(p.lateralFlow, i, j) * DW_S(p.dx, i, j);
            double e = 1.0, f = 0.0, ex = 0.0, fx = 0.0;
            { double *q = scr + (size_t)n0 * 12; q[5] = e; q[6] = f; q[7] = ex; q[8] = fx; }
            for (int i = 2; i <= ncomp; ++i) {
                double *q = scr + (size_t)(n0 + i - 1) * 12;
                const double ppi = q[0], qqi = q[1], rri = q[2], ssi = q[3], sxi = q[4];
                const double e1 = -1.0 * rri / (ppi * e + qqi);
                const double f1 = (ssi - ppi * f) / (ppi * e + qqi);
                const double ex1 = -1.0 * rri / (ppi * ex + qqi);
                const double fx1 = (sxi - ppi * fx) / (ppi * ex + qqi);
                e = e1; f = f1; ex = ex1; fx = fx1;
                q[5] = e; q[6] = f; q[7] = ex; q[8] = fx;
            }
            const double qp_ghost = DW_S(p.oldQ, ncomp - 1, j), qpx_ghost = 0.0;
            DW_S(p.qp, ncomp, j) = e * qp_ghost + f;
            DW_S(p.qpx, ncomp, j) = ex * qpx_ghost + fx;
            for (int i = ncomp - 1; i >= 1; --i) {
                const double *q = scr + (size_t)(n0 + i - 1) * 12;
                DW_S(p.qp, i, j) = q[5] * DW_S(p.qp, i + 1, j) + q[6];
                DW_S(p.qpx, i, j) = q[7] * DW_S(p.qpx, i + 1, j) + q[8];
            }
            for (int i = 2; i <= ncomp; ++i) {
                if (fabs(DW_S(p.qp, i, j)) < p.q_llm) DW_S(p.qp, i, j) = p.q_llm;
                DW_S(p.newQ, i, j) = DW_S(p.qp, i, j);
            }
            scr[(size_t)n0 * 12 + 9] = allqlat;
        }
        __syncthreads();
        DW_PHASE(1)
        for (int m = tid; m < nmstem; m += kParThreads) { // junction inflow -> first node of the reach
            const int j = p.mstem_frj[m], n0 = rbase[m];
            double q1 = 0.0;
            if (DW_FRNW(j, 3) > 0) {
                for (int k = 1; k <= DW_FRNW(j, 3); ++k) {
                    const int usrchj = DW_FRNW(j, 3 + k);
                    double q_usrch;
                    if (is_mainstem(p, usrchj)) {
                        q_usrch = DW_S(p.newQ, DW_FRNW(usrchj, 1), usrchj);
                    } else {
                        const double tf0 = t + p.dtini / 60.;
                        q_usrch = intp_y(nts_qtrib, p.tarr_qtrib, p.qtrib + (int64_t)(usrchj - 1) * nts_qtrib, tf0);
                    }
                    q1 = q1 + q_usrch;
                }
            }
            q1 = q1 + DW_G(p.lateralFlow, 1, j) * DW_S(p.dx, 1, j);
            double qp1 = q1;
            qp1 = qp1 + scr[(size_t)n0 * 12 + 9];
            if (fabs(qp1) < p.q_llm) qp1 = p.q_llm;
            DW_S(p.qp, 1, j) = qp1;
            DW_S(p.newQ, 1, j) = qp1;
        }
        __syncthreads();
        DW_PHASE(2)
        // ---- corrector ------------------------------------------------------------------------------------------
        for (int m = tid; m < nmstem; m += kParThreads) { // water surface at the domain's outlet(s)
            const int j = p.mstem_frj[m], ncomp = DW_FRNW(j, 1), n0 = rbase[m];
            if (DW_FRNW(j, 2) >= 0) continue;
            double y = DW_S(p.newY, ncomp, j);
            if (p.dsbc_option == 1) {
                y = intp_y(nts_db, p.tarr_db, p.varr_db, t + p.dtini / 60.);
                if ((y - DW_S(p.z, ncomp, j)) < mindepth_nstab) y = mindepth_nstab + DW_S(p.z, ncomp, j);
            } else if (p.dsbc_option == 2) {
                y = intp_tab(p, ncomp, j, C_UNIF, C_ELEV, fabs(DW_S(p.newQ, ncomp, j)));
            }
            L.newY(n0 + ncomp - 1) = y;
        }
        for (int n = tid; n < nnodes; n += kParThreads) { // what the depth solve of a node needs besides the node below
            const int i = nk[n], j = nj[n], ncomp = DW_FRNW(j, 1);
            double *r = L.rec + (size_t)n * kRecDoubles;
            const double z_cur = DW_S(p.z, i, j), Q_cur = DW_S(p.qp, i, j);
            r[9] = z_cur;
            r[10] = Q_cur;
            if (i <= ncomp - 1) {
                const DepthPre d = depth_pre(p, ls, node_block(p, i, j), i, j, Q_cur, z_cur);
                r[0] = d.y_norm; r[1] = d.x1; r[2] = d.x2; r[3] = d.sf[0]; r[4] = d.sf[1]; r[5] = d.sf[2];
                r[6] = d.df_mid; r[7] = d.slope; r[8] = d.dxi;
            }
        }
        __syncthreads();
        DW_PHASE(3)
        if (tid < 64) { // the chain: one wavefront, outlet first
            for (int jm = nmstem; jm >= 1; --jm) {
                const int j = p.mstem_frj[jm - 1], ncomp = DW_FRNW(j, 1), n0 = rbase[jm - 1];
                if (DW_FRNW(j, 2) >= 0) L.newY(n0 + ncomp - 1) = L.newY(rbase[m_of_reach[DW_FRNW(j, 2) - 1]]);
                WinRegs<WR> Wd = win_fetch<WR>(L, n0 + ncomp - 1);
                WinRegs<WR> Wc = win_fetch<WR>(L, n0 + ncomp - 2);
                double y_below = L.newY(n0 + ncomp - 1);
                for (int i = ncomp; i >= 2; --i) {
                    const int nd = n0 + i - 1, nc = nd - 1;
                    // the node after this one: its window and record are on their way while this node is solved
                    const int nn = nc >= 1 ? nc - 1 : 0;
                    const WinRegs<WR> Wn = win_fetch<WR>(L, nn);
                    const double *rd = L.rec + (size_t)nd * kRecDoubles, *rc = L.rec + (size_t)nc * kRecDoubles;
                    const double zz = rd[9], Q_ds = rd[10];
                    double y_ds = y_below - zz;
                    y_ds = dmax(y_ds, (double)0.005f);
                    const double elv_ds = y_ds + zz;
                    double conv_ds, u0, u1;
                    chain_lookup<false, WR>(p, L, ws, Wd, i, j, elv_ds, conv_ds, u0, u1);
                    const double sf_ds = fabs(Q_ds) * Q_ds / (conv_ds * conv_ds);
                    DepthPre d;
                    d.y_norm = rc[0]; d.x1 = rc[1]; d.x2 = rc[2]; d.sf[0] = rc[3]; d.sf[1] = rc[4]; d.sf[2] = rc[5];
                    d.df_mid = rc[6]; d.slope = rc[7]; d.dxi = rc[8]; d.z_cur = rc[9];
                    const double Q_cur = rc[10], z_cur = rc[9];
                    if (L.misses && tid == 0) ++L.misses[2];
                    const double y_cur = depth_solve(d, sf_ds, y_ds, [&](double yc) {
                        double conv, dKdA, topw;
                        if (L.misses && tid == 0) ++L.misses[1];
                        chain_lookup<true, WR>(p, L, ws, Wc, i - 1, j, yc + z_cur, conv, dKdA, topw);
                        return funcd_arith(Q_cur, sf_ds, conv, dKdA, topw, d.slope, d.dxi, yc, y_ds);
                    });
                    double ny = y_cur + z_cur;
                    if (ny > 100000.0) ny = 100000.0;
                    L.newY(nc) = ny;
                    y_below = ny;
                    Wd = Wc;
                    Wc = Wn;
                }
            }
        }
        __syncthreads();
        DW_PHASE(4)
        for (int n = tid; n < nnodes; n += kParThreads) { // the node's own new state, its window for the next sub-step
            const int i = nk[n], j = nj[n];
            const double xt = L.newY(n);
            DW_S(p.newY, i, j) = xt;
            const NodePost np = backward_node_post(p, i, j, ls);
            double *q = scr + (size_t)n * 12;
            q[10] = np.celerity2;
            q[11] = np.diffusivity2;
            const double *tb = node_block(p, i, j);
            const int k = LaneScan::count_below<true, false>(tb + C_ELEV * kNel, 0.0, kNel, xt) - 1; // interval of xt
            const int w0_old = L.w0[2 * n];
            if (w0_old < 0 || k < w0_old + 1 || k > w0_old + WR - 3) window_load<WR>(L, n, tb, k);
        }
        __syncthreads();
        DW_PHASE(5)
        double my_max = 0.;
        for (int m = tid; m < nmstem; m += kParThreads) { // reach means of celerity and diffusivity
            const int j = p.mstem_frj[m], ncomp = DW_FRNW(j, 1), n0 = rbase[m];
            double cs = 0.0, ds = 0.0;
            for (int i = 1; i <= ncomp; ++i) { cs = cs + scr[(size_t)(n0 + i - 1) * 12 + 10]; ds = ds + scr[(size_t)(n0 + i - 1) * 12 + 11]; }
            double cel = cs / ncomp;
            if (cel < p.C_llm) cel = p.C_llm;
            const double dif = ds / ncomp;
            for (int i = 1; i <= ncomp; ++i) {
                DW_S(p.celerity, i, j) = cel;
                double d = dif;
                if (d > p.D_ulm) d = p.D_ulm;
                if (d < p.D_llm) d = p.D_llm;
                DW_S(p.diffusivity, i, j) = d;
            }
            for (int kkk = 1; kkk <= ncomp - 1; ++kkk) my_max = dmax(my_max, cel / DW_S(p.dx, kkk, j));
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) my_max = dmax(my_max, __shfl_xor(my_max, d));
        if ((tid & 63) == 0) s_red[tid >> 6] = my_max;
        __syncthreads();
        maxCelDx = 0.;
        for (int w = 0; w < kParThreads / 64; ++w) maxCelDx = dmax(maxCelDx, s_red[w]);
        t = t + p.dtini / 60.;
        // ---- results at the recording instants (:787-810), the initial state after the first sub-step (:813-832)
        const bool rec_now = fmod((t - t0 * 60.) * 60., saveInterval) <= TOL || t == tfin * 60.;
        const bool first = t == t0 + p.dtini / 60.;
        if ((rec_now && ts_ev + 1 <= p.ntss_ev) || first) {
            for (int n = tid; n < nnodes; n += kParThreads) {
                const int i = nk[n], j = nj[n];
                if (rec_now && ts_ev + 1 <= p.ntss_ev) {
                    DW_EV(p.q_ev, ts_ev + 1, i, j) = DW_S(p.newQ, i, j);
                    DW_EV(p.elv_ev, ts_ev + 1, i, j) = DW_S(p.newY, i, j);
                    DW_EV(p.depth_ev, ts_ev + 1, i, j) = DW_EV(p.elv_ev, ts_ev + 1, i, j) - DW_S(p.z, i, j);
                }
                if (first) {
                    DW_EV(p.q_ev, 1, i, j) = DW_S(p.oldQ, i, j);
                    DW_EV(p.elv_ev, 1, i, j) = DW_S(p.oldY, i, j);
                    DW_EV(p.depth_ev, 1, i, j) = DW_EV(p.elv_ev, 1, i, j) - DW_S(p.z, i, j);
                }
            }
            for (int m = tid; m < nmstem; m += kParThreads) {
                const int j = p.mstem_frj[m];
                for (int k = 1; k <= DW_FRNW(j, 3); ++k) {
                    const int usrchj = DW_FRNW(j, 3 + k);
                    if (is_mainstem(p, usrchj)) continue;
                    if (rec_now && ts_ev + 1 <= p.ntss_ev) {
                        DW_EV(p.elv_ev, ts_ev + 1, DW_FRNW(usrchj, 1), usrchj) = DW_S(p.newY, 1, j);
                        DW_EV(p.depth_ev, ts_ev + 1, DW_FRNW(usrchj, 1), usrchj) = DW_S(p.newY, 1, j) - DW_S(p.z, 1, j);
                    }
                    if (first) {
                        DW_EV(p.elv_ev, 1, DW_FRNW(usrchj, 1), usrchj) = DW_S(p.oldY, 1, j);
                        DW_EV(p.depth_ev, 1, DW_FRNW(usrchj, 1), usrchj) = DW_S(p.oldY, 1, j) - DW_S(p.z, 1, j);
                    }
                }
            }
        }
        if (rec_now) ts_ev = ts_ev + 1;
        __syncthreads();
        { double *sw = p.oldY; p.oldY = p.newY; p.newY = sw; }
        { double *sw = p.oldQ; p.oldQ = p.newQ; p.newQ = sw; }
        DW_PHASE(6)
    }
#undef DW_PHASE
}

// results of the recording instants back onto the original hydrofabric: one thread per instant (the links of an
// instant are visited in crosswalk order, the fractions accumulate)
__global__ void __launch_bounds__(64) k_dw_crosswalk(trdw::Problem p, const double *tq, const double *te, double *used, int32_t *flag)
{
    const int ts = blockIdx.x * 64 + threadIdx.x + 1;
    if (ts > p.ntss_ev) return;
    const int64_t nn = (int64_t)p.mxncomp * p.nrch;
    trdw::crosswalk_instant(p, ts, tq, te, used + (int64_t)(ts - 1) * nn, flag + (int64_t)(ts - 1) * nn);
}

// host side of one domain: device copies of its inputs, its work space, its node list
struct Domain {
    std::vector<void *> ptrs;
    trdw::Problem p;
    double *d_out = nullptr, *d_min = nullptr;
    int32_t *d_nk = nullptr, *d_nj = nullptr, *d_rbase = nullptr, *d_mof = nullptr;
    double *d_scratch = nullptr, *d_chain = nullptr;
    int par_rows = 0;         // window rows of the parallel time loop (0: its chain state does not fit LDS, serial kernel)
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
    if (*a.cwnrow_g < 0) return dw_fail(TRDW_EINVAL, "cwnrow_g is negative");
    if (*a.cwnrow_g > 0 && (!a.cwncol_g || !a.crosswalk_g || !a.rdx_ar_g || !a.z_thalweg_g))
        return dw_fail(TRDW_EINVAL, "crosswalk arrays are NULL although cwnrow_g > 0");
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
    std::vector<int32_t> rbase, m_of((size_t)nr, -1);
    for (int n_ = 0; n_ < dom.nnodes; ++n_)
        if (node_k[n_] == 1) {
            m_of[(size_t)node_j[n_] - 1] = (int32_t)rbase.size();
            rbase.push_back(n_);
        }
    rbase.push_back(dom.nnodes);
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
    p.cwnrow = *a.cwnrow_g;
    p.cwncol = p.cwnrow > 0 ? *a.cwncol_g : 0;
    if (p.cwnrow > 0) { // results are mapped back from the refactored to the original hydrofabric (diffnw :849-920)
        if (p.cwncol < 6) return dw_fail(TRDW_EINVAL, "crosswalk_g needs at least 6 columns");
        for (int r = 0; r < p.cwnrow; ++r) {
            auto cw = [&](int c) { return a.crosswalk_g[r + (size_t)(c - 1) * p.cwnrow]; };
            const int ri = (int)cw(1), rj = (int)cw(2), nlnk = (int)cw(3);
            if (ri < 1 || ri + 1 > mx || rj < 1 || rj > nr || nlnk < 0 || 3 + 3 * nlnk > p.cwncol)
                return dw_fail(TRDW_EINVAL, "crosswalk_g: a row's refactored segment or link count is out of range");
            for (int l = 0; l < nlnk; ++l) {
                const int oi = (int)cw(4 + 3 * l), oj = (int)cw(5 + 3 * l);
                if (oi < 1 || oi + 1 > mx || oj < 1 || oj > nr) return dw_fail(TRDW_EINVAL, "crosswalk_g: an original link is out of range");
            }
        }
        DW_UP(rdx_ar, a.rdx_ar_g, nn, double)
        DW_UP(crosswalk, a.crosswalk_g, (size_t)p.cwnrow * p.cwncol, double)
        DW_UP(z_thalweg, a.z_thalweg_g, nn, double)
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
    if (dom.up(rbase.data(), rbase.size() * sizeof(int32_t), (void **)&dom.d_rbase, st)) return dw_fail(TRDW_ENOMEM, "device allocation failed");
    if (dom.up(m_of.data(), m_of.size() * sizeof(int32_t), (void **)&dom.d_mof, st)) return dw_fail(TRDW_ENOMEM, "device allocation failed");
    if (dom.up(nullptr, (size_t)dom.nnodes * 12 * sizeof(double), (void **)&dom.d_scratch, st)) return dw_fail(TRDW_ENOMEM, "device allocation failed");
    {
        // widest window whose chain state fits beside the kernel's static LDS (and the prologue's two columns)
        dom.par_rows = 0;
        for (int wr = 7; wr >= 5 && !dom.par_rows; --wr)
            if (par_lds_bytes(dom.nnodes, wr) + 1024 <= 160 * 1024) dom.par_rows = wr;
        // a longer mainstem keeps the same state in global memory (windows and records are fetched a node ahead)
        if (dom.up(nullptr, par_lds_bytes(dom.nnodes, 7), (void **)&dom.d_chain, st)) return dw_fail(TRDW_ENOMEM, "device allocation failed");
    }
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
        void *d_items = nullptr, *d_pitems = nullptr, *d_ticks = nullptr;
        std::vector<Domain> doms;
        ~Run()
        {
            if (st) (void)hipStreamSynchronize(st);
            doms.clear();
            if (d_items) (void)hipFree(d_items);
            if (d_pitems) (void)hipFree(d_pitems);
            if (d_ticks) (void)hipFree(d_ticks);
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
    // the parallel time loop when every domain's chain state fits LDS (TRDW_SOLVER=serial: the one-wavefront kernel)
    const trdw_options cfg = g_options; // (trdw_configure; the library reads no environment variable)
    bool par = cfg.solver != 1;
    int par_rows = 7, nn_max = 0;
    for (int b = 0; b < n; ++b) {
        par_rows = par_rows < run.doms[b].par_rows ? par_rows : run.doms[b].par_rows;
        nn_max = nn_max > run.doms[b].nnodes ? nn_max : run.doms[b].nnodes;
    }
    if (cfg.chain_global) par_rows = 0; // developer A/B: chain state in global memory
    if (cfg.window_rows >= 5 && cfg.window_rows <= par_rows) par_rows = cfg.window_rows; // developer A/B
    if (par) {
        std::vector<ParItem> pitems((size_t)n);
        for (int b = 0; b < n; ++b) {
            Domain &dm = run.doms[b];
            pitems[b].p = dm.p;
            pitems[b].min_dx = dm.d_min;
            pitems[b].nk = dm.d_nk;
            pitems[b].nj = dm.d_nj;
            pitems[b].rbase = dm.d_rbase;
            pitems[b].m_of_reach = dm.d_mof;
            pitems[b].scratch = dm.d_scratch;
            pitems[b].chain_state = dm.d_chain;
            pitems[b].nnodes = dm.nnodes;
            pitems[b].phase_ticks = nullptr;
        }
        if (cfg.phase_ticks) {
            DW_TRY(hipMalloc(&run.d_ticks, 16 * sizeof(unsigned long long)));
            DW_TRY(hipMemsetAsync(run.d_ticks, 0, 16 * sizeof(unsigned long long), st));
            pitems[0].phase_ticks = (unsigned long long *)run.d_ticks;
        }
        DW_TRY(hipMalloc(&run.d_pitems, (size_t)n * sizeof(ParItem)));
        DW_TRY(hipMemcpyAsync(run.d_pitems, pitems.data(), (size_t)n * sizeof(ParItem), hipMemcpyHostToDevice, st));
        DW_TRY(hipStreamSynchronize(st)); // pitems is about to go out of scope
        size_t par_lds = par_rows >= 5 ? par_lds_bytes(nn_max, par_rows) : 0;
        par_lds = par_lds < 2 * trdw::kNel * sizeof(double) ? 2 * trdw::kNel * sizeof(double) : par_lds; // (the prologue's two columns)
        auto launch = [&](auto kernel) -> hipError_t {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)par_lds);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(kernel, dim3(n), dim3(kParThreads), par_lds, st, (const ParItem *)run.d_pitems);
            return hipSuccess;
        };
        if (par_rows == 7) DW_TRY(launch(k_dw_solve_par<7, true>));
        else if (par_rows == 6) DW_TRY(launch(k_dw_solve_par<6, true>));
        else if (par_rows == 5) DW_TRY(launch(k_dw_solve_par<5, true>));
        else DW_TRY(launch(k_dw_solve_par<7, false>));
    } else {
        if (lds_max > 64 * 1024)
            DW_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_dw_solve), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
        hipLaunchKernelGGL(k_dw_solve, dim3(n), dim3(64), lds_max, st, (const BatchItem *)run.d_items);
    }
    for (int b = 0; b < n; ++b) {
        Domain &dm = run.doms[b];
        if (dm.p.cwnrow <= 0) continue;
        const size_t nn = (size_t)dm.p.mxncomp * dm.p.nrch, nev = (size_t)dm.p.ntss_ev;
        double *tq = nullptr, *used = nullptr;
        int32_t *flag = nullptr;
        if (dm.up(nullptr, 2 * dm.nout * sizeof(double), (void **)&tq, st) || dm.up(nullptr, nev * nn * sizeof(double), (void **)&used, st)
            || dm.up(nullptr, nev * nn * sizeof(int32_t), (void **)&flag, st))
            return dw_fail(TRDW_ENOMEM, "device allocation failed: crosswalk scratch");
        DW_TRY(hipMemcpyAsync(tq, dm.d_out, 2 * dm.nout * sizeof(double), hipMemcpyDeviceToDevice, st)); // q_ev, elv_ev
        DW_TRY(hipMemsetAsync(dm.d_out, 0, 2 * dm.nout * sizeof(double), st));
        hipLaunchKernelGGL(k_dw_crosswalk, dim3((unsigned)((nev + 63) / 64)), dim3(64), 0, st, dm.p, (const double *)tq,
                           (const double *)(tq + dm.nout), used, flag);
    }
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
    if (run.d_ticks) {
        unsigned long long tk[16];
        DW_TRY(hipMemcpy(tk, run.d_ticks, sizeof tk, hipMemcpyDeviceToHost));
        static const char *name[7] = {"coefficients", "recurrences", "junctions", "depth_pre", "chain", "node_post", "means+output"};
        for (int k = 0; k < 7; ++k) std::fprintf(stderr, "[trdw phases] %-14s %9.3f ms\n", name[k], tk[k] * 1e-5);
        std::fprintf(stderr, "[trdw phases] chain: %llu depth solves, %llu function evaluations inside them, %llu look-ups outside the window\n",
                     tk[10], tk[9], tk[8]);
    }
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

int trdw_configure(const trdw_options *options)
{
    trdw_options o = {(int32_t)sizeof(trdw_options), 0, 0, 0, 0};
    if (options) {
        if (options->struct_size < 8 || options->struct_size > (int32_t)sizeof(trdw_options)) {
            g_dw_err = "trdw_options.struct_size is not the size of a trdw_options this library knows";
            return TRDW_EINVAL;
        }
        std::memcpy(&o, options, (size_t)options->struct_size);
    }
    if (o.solver != 0 && o.solver != 1) {
        g_dw_err = "trdw_options.solver must be 0 (parallel) or 1 (serial)";
        return TRDW_EINVAL;
    }
    g_options = o;
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
