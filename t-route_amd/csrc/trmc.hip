// trmc.hip -- HIP kernels and C ABI of the MI355X Muskingum-Cunge engine (gfx950).
//
// Data layout in HBM (plan order = level-major, see topology.hpp):
//   params      8 SoA columns [nseg_pad] (dx bw tw twcc n ncc cs s0) + dt (scalar, or a
//               9th column when the caller's dt column is not uniform)
//   up_ptr/idx  CSR of upstream plan positions (int32)
//   level       int32 [nseg_pad]
//   qlat_tm     [nq][nseg_pad]           forcing, one coalesced row per forcing interval
//   q/v/d_tm    [nsteps+1][nseg_pad]     time-major state AND result: step t reads row
//               t-1 (own previous flow/depth, upstream previous flows) and, without the
//               short-timestep assumption, row t of upstream levels
//   out         [nseg][nsteps][3]        caller layout (mc_reach.pyx:807-813), produced
//               by an LDS-tiled transpose at the end of the route
// nseg_pad rounds nseg up to 64 so every time row starts on a 256-byte boundary.
//
// Launch structure
//   assume_short_ts = 1 : step t of every segment depends only on step t-1
//       (mc_reach.pyx:504-505, :135-136) -> one launch per timestep over all
//       routed segments.
//   assume_short_ts = 0 : (level l, step t) depends on (<l, t) and (l, t-1)
//       -> anti-diagonal wavefront d = l + t; the segments of all levels on
//       one diagonal form ONE contiguous slice of the plan order and run as
//       one launch; nlevels + nsteps - 1 launches in total.
// One thread per segment-step; all loads/stores are unit-stride except the
// upstream gather, which the intra-level ordering keeps near-monotone.
//
// No CPU fallback exists in this library.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/trmc.h"
#include "det_pow.h"
#include "det_pow64.h"
#include "mc_segment.hpp"
#include "levelpool.hpp"
#include "topology.hpp"
#include "internal.hpp"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

} // namespace
namespace trmc {
int fail_with(int code, const std::string &msg) { return fail(code, msg); }
} // namespace trmc
namespace {

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(e_ == hipErrorOutOfMemory ? TRMC_ENOMEM : TRMC_EHIP,                   \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                    \
    } while (0)

#include "dev_math.inc"
#include "kernels_levels.inc"
#include "kernels_flow.inc"
// ---------------------------------------------------------------- plan
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    bool borrowed = false; // the memory belongs to another plan (trmc_plan_clone): never freed, never regrown here
    void borrow(const DevBuf &o)
    {
        release();
        p = o.p;
        bytes = o.bytes;
        borrowed = true;
    }
    int ensure(size_t need, bool zero_new = false)
    {
        if (need <= bytes) return 0;
        if (borrowed) return fail(TRMC_ESTATE, "a buffer shared with the plan this one was cloned from would have to grow");
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        hipError_t e = hipMalloc(&p, need ? need : 1);
        if (e != hipSuccess) return fail(TRMC_ENOMEM, std::string("hipMalloc(") + std::to_string(need) + "): " + hipGetErrorString(e));
        bytes = need;
        if (zero_new && need) { // (granule planes: recycled memory must not hold a tag that could pass for a live one)
            // (hipMemset of device memory may return before it has run, on the null stream -- which the plans'
            // non-blocking streams do not wait for: without the synchronisation the fill can land AFTER the first
            // kernels of a window have written into the buffer; seen with two processes sharing one GPU)
            e = hipMemset(p, 0, need);
            if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
            if (e != hipSuccess) return fail(TRMC_EHIP, std::string("hipMemset: ") + hipGetErrorString(e));
        }
        return 0;
    }
    void release()
    {
        if (p && !borrowed) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        borrowed = false;
    }
};

} // namespace

struct RouteRun { // the routing window in progress (route_begin_t .. route_end_t)
    bool active = false;
    int32_t nsteps = 0, qts = 1, short_ts = 0;
    int32_t t_done = 0;           // launches ("diagonals") 1..t_done are queued: rows without lag are at step t_done,
                                  // lagged rows at step t_done - maxlag; the window ends at nsteps + maxlag
    int32_t boundary_through = 0; // boundary rows hold their hydrographs for steps 1..boundary_through
    int32_t tiles_done = 0, launches = 0;
    // wide levels routed K steps per launch with a skew of K steps per level (k_mc_tile); 0 = every level one step per launch
    int32_t wide = 0, wide_k = 0, wide_next = 0, wide_through = 0; // levels; K; tiles queued; last tile the tail waits for
    // second tier: the `mid` levels right below the wide ones, mid_k steps per launch under their own skew (k_mc_tile again),
    // queued on the plan's stream between the tail's launches
    int32_t mid = 0, mid_k = 0, mid_next = 0;
    // cluster tiles (k_mc_ctile): the rows below the wide levels K steps per launch too, cluster level c another tile behind;
    // `wide` is then the plan's cl_from_level (possibly 0: no slices at all), there is no one-step tail and no transposing pass
    bool cl = false;
    int32_t cl_next = 0;          // index of the next cluster tile to queue (they begin at tile `wide`)
    bool tail_active = false;     // the tail launches of this window go to the tail stream
    bool end_queued = false;      // route_end_queue has run for this window
    int32_t dec_stride = 0, dec_keep = 0; // the tiles of this window write the kept steps of their rows into the plan's `dec`
    int32_t dec_lo = 0, dec_hi = 0;       // ... the plan positions [dec_lo, dec_hi) that do
};

struct StreamRun;
struct trmc_plan {
    int device = 0;
    int precision = 32;
    size_t esz = 4;
    trmc::Topology topo;
    int64_t nseg = 0, nseg_pad = 0, nrouted = 0;
    bool dt_uniform = true;
    double dt = 0.0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t stream2 = nullptr;       // result transpose, overlapped with the step launches
    hipStream_t wstream = nullptr;       // the wide tiles (k_mc_tile), ordinary priority: the narrow tail of the level order runs one
                                         // step per launch on the plan's own high-priority stream BESIDE them -- the tail's 288
                                         // dependent launches are the latency-critical part, the tiles fill whatever they leave
    hipEvent_t ev_tail = nullptr;        // "every tile queued so far is complete" (tile stream -> plan stream)
    std::vector<hipEvent_t> wide_t0, wide_t1; // timing events around the wide launches of the window (trmc_stats.ms_wide)
    std::vector<hipEvent_t> tile_ev;     // "time tile b is complete" (main stream -> stream2)
    hipEvent_t ev_emit = nullptr;        // "all tiles emitted" (stream2 -> main stream)
    // static, plan order
    DevBuf params; // 9 columns x nseg_pad
    DevBuf up_ptr, up_idx, up2, level, row_of_pos, pos_of_row, it_prev, lag;
    DevBuf lagk, cblk_ptr;               // cluster order (topology.hpp): tiles every position runs behind level 0; the cluster blocks
    DevBuf it_sum;                       // per-position cost of the window (trmc_plan_collect_cost)
    bool collect_cost = false;
    bool hinted = false;                 // created with a cost hint: rows of a level are grouped by cost
    bool params_sane = false;            // see DevMathF::fast_ok
    int32_t cost_nsteps = -1;            // nsteps of the window it_sum was collected over
    int32_t maxlag = 0;                  // trmc_plan_set_lag: rows routed `maxlag` launches behind the others
    int64_t wide_safe_pos = -1;          // first plan position that is lagged or fed by a boundary row (-1: not looked for yet)
    // trmc_plan_chain_from: the next window's initial state has been set on the device from another plan's window
    bool chain_staged = false;
    hipEvent_t ev_chain[4] = {nullptr, nullptr, nullptr, nullptr}; // source's tiles / tail done; this plan's two copies done
    // as the SOURCE of a hand-over: "the receiver has read my planes" (its two copies), waited for at my next trmc_route_begin
    hipEvent_t ev_released[2] = {nullptr, nullptr};
    bool released_pending[2] = {false, false};
    std::vector<int32_t> lag_of_row;
    DevBuf gage_of_pos, da_mode, da_a, da_w, da_nudge; // nudging tables of the staged window
    DevBuf res_of_pos, res_par, res_inflow;             // level-pool reservoirs of the plan
    int64_t nres = 0;
    double res_dt = 0.0;
    int64_t ngage = 0;
    int64_t nraw = 0;                    // gages inside a reach whose successor reads the un-nudged flow (general mode)
    DevBuf raw_of_pos, da_raw;
    int32_t da_nsteps = -1;
    // per window
    DevBuf in_qlat, in_q0, in_bfvd, qlat_tm, tm, out, scratch, gathered;
    size_t gathered_bytes = 0;
    int64_t nq = 0;
    bool qlat_direct = false;   // qlat_tm was filled by trmc_upload_forcing_packed: no transpose at route time
    DevBuf qlat_alt;            // sequence mode: the STAGED forcing already transposed (trmc_stage_forcing, behind its copy) into
    bool qlat_alt_ready = false; // the buffer the window in progress does not read; the next window's set-up swaps the two
    bool have_boundary = true;  // boundary hydrographs present for the staged window
    int32_t staged_nsteps = -1; // nsteps the staged forcing was uploaded for
    int32_t routed_nsteps = -1; // nsteps of the last completed route
    // dataflow engine (k_mc_flow): the plan is in block order, `tm` holds the granule plane
    bool flow = false;
    uint32_t tag_base = 1;               // tag of step 0 of the current window (0 is never a live tag)
    int32_t tag_span = 0;                // tags the current window may use (nsteps + 1)
    DevBuf prio;                         // issue priority per wavefront
    DevBuf d_gran;                       // depth hand-over granules (k_mc_flow_lean)
    hipStream_t fstream = nullptr;       // second compute stream: consecutive time chunks of a resident window overlap
    hipEvent_t ev_chunk[2] = {nullptr, nullptr}; // the last launch queued on {stream, fstream}
    hipEvent_t ev_ctl = nullptr;         // ordering of the two compute streams against each other
    int flow_next = 0;                   // which of the two the next trmc_route_advance uses (only toggles in overlap mode)
    int flow_last = 0;                   // ... and which one the last launch went to
    DevBuf ticket_map;                   // [nblocks] ticket -> block of a general-mode plan with long main stems (topo.early_blocks)
    DevBuf d_state, ticket, rank, dbg;   // depth column; {block ticket, abort flag}; level rank of a position inside its block
    DevBuf cuq_ptr, cuq_blk, cuq_head, cu_index, cuq_perm; // blocks dealt to compute units by cost (flow_place_blocks); heads: one set per compute stream
    int32_t ncuq = 0;                    // number of queues (= compute units found), 0 = block tickets
    uint64_t watchdog_ticks = 3000000000ull; // 30 s of wall_clock64 (100 MHz): long enough for a device that is shared or profiled (TRMC_FLOW_WATCHDOG_MS)
    // trmc_plan_options, resolved at creation (a clone copies them)
    struct Opt {
        bool tol = false;                    // TRMC_ARITH_TOLERANCE
        int64_t wide_min_rows = 0;           // <= 0: no wide tier
        int32_t wide_levels = 16, wide_k = 16;
        int64_t mid_min_rows = 0;            // <= 0: no second tier
        int32_t mid_levels = 12, mid_k = 4;
        int32_t tile_perm_group = -1;        // -1: the default (see route_advance_t); 0: off; 1: on
        int32_t hot_rows = -1;               // -1 (default) or 1: with the partition; 0: off
        int32_t cluster_rows = 0;            // rows per cluster block of a short-timestep plan's deeper rows; 0: no cluster order
        int32_t cluster_late_lag = 0;        // tiles the rows fed by boundary rows run behind at least (cluster order)
        int32_t stream_split = 0;            // a stream's slices from this level on ride on the clusters' stream (0: all on the tile stream)
        int32_t hot_wave_rows = 0;           // rows of the hot list per wavefront (0: by the plan's size)
        bool velocity_on_demand = false;     // a stream without full_output forms a step's velocity only where it is handed on
        bool sequence = false;
        bool flow_overlap = false;
        int32_t flow_lean = 0;
        bool flow_debug = false;
    } opt;
    trmc_stats stats{};
    RouteRun run;
    // asynchronous fetch of what a throughput-mode caller consumes (trmc_fetch_begin / trmc_fetch_wait)
    // trmc_plan_set_output_stride: windows write every out_stride-th step of every row's (q, v, d) into `dec` as they go (the
    // tiled rows from k_mc_tile; the others are gathered from the time-major planes when the block is fetched)
    int32_t out_stride = 0;
    DevBuf dec;
    int32_t dec_stride_done = 0, dec_keep_done = 0, dec_nsteps_done = 0, dec_lo_done = 0, dec_hi_done = 0; // ... what the last window left there
    unsigned long long *stamps = nullptr; // trmc_plan_set_stamps: [nstamp_windows][4] in page-locked host memory (the caller's)
    int32_t nstamp_windows = 0;
    int64_t stamp_seq = -1;               // windows begun since the ring was set, minus one
    DevBuf fetch_hyd, fetch_q0, fetch_fvd;
    hipEvent_t ev_dec = nullptr;         // "the copy stream has read `out`" (a fetch of the decimated result): the next window's
    bool dec_pending = false;            // set-up goes behind it
    hipStream_t cstream = nullptr;       // copy stream: D2H of window k runs beside the kernels of window k + 1
    hipStream_t hstream = nullptr;       // ... and the one of the other direction: a staged forcing on its way to the device
    hipEvent_t ev_fetch_ready = nullptr, ev_fetch_done = nullptr;
    bool fetch_pending = false;
    // "the last gather queued on the plan's stream after a window has read the planes" -- what a set-up queued on ANOTHER stream
    // (TRMC_SETUP_ASIDE) waits for before it lets a new window's tiles overwrite them
    hipEvent_t ev_gather = nullptr;
    bool gather_pending = false;
    // trmc_plan_clone: a second set of WINDOW buffers on the static data (topology, parameters) of `parent`
    trmc_plan *parent = nullptr;
    int32_t clones = 0;                  // live clones of this plan
    bool zombie = false;                 // destroyed while clones were alive: freed with the last of them
    // trmc_stage_forcing: the next window's forcing is on its way to in_qlat on the copy stream
    hipEvent_t ev_forcing = nullptr;
    bool forcing_pending = false;
    bool state_missing = false;          // ... and there is no initial state yet: trmc_plan_chain_from must supply it
    bool q0_staged = false;              // in_q0 holds the initial state of the window that is staged (an upload's q0, or the last
                                         // window's final state gathered by an upload with q0 = NULL / trmc_stage_forcing): valid
                                         // until a window consumes it, whatever routed_nsteps says in the meantime
    struct StreamRun *seq = nullptr;     // trmc_stream_*: a stream of windows on a ring of day slots (stream.inc)
    DevBuf hot_list, hot_cnt;            // k_mc_tile's hot rows (StepArgs::hot_list): [3][hot_cap] positions, [3] lengths
    int32_t hot_cap = 0;
    bool nan_is_zero = false;            // uploads of forcing and state: NaN -> 0 on the device (trmc_plan_set_nan_is_zero)
    int64_t tile_seq = 0;                // tile launches of the wide tier so far, all windows: which of the three lists is read
    DevBuf cls_last;                     // the cost class every wide row showed at the end of its last tile (k_mc_tile's in-block partition)
    std::vector<DevBuf> rowsets;        // positions of registered row sets (trmc_rowset_create)
    std::vector<int64_t> rowset_n;
    std::vector<int32_t> rowset_lag;
    std::vector<int32_t> rowset_lagk;   // cluster order: the largest tile lag among the set's rows (trmc_stream_gather)
};

namespace {
#include "host_levels.inc"
#include "host_flow.inc"
#include "host_misc.inc"
} // namespace

// ---------------------------------------------------------------- C ABI
extern "C" {
#include "abi.inc"
#include "stream.inc"

} // extern "C"
