/*
 * trmc.h -- C ABI of the MI355X-native Muskingum-Cunge routing engine
 * (libtrmc.so, built from t-route_amd/csrc/ for gfx950).
 *
 * This is the drop-in boundary for ONE reference path:
 *
 *   troute.routing.compute.compute_nhd_routing_v02
 *     -> troute.routing.fast_reach.mc_reach.compute_network_structured
 *        -> reach.muskingcunge -> c_muskingcungenwm (Fortran)
 *
 * Reference interfaces replaced (paths relative to the T-Route tree):
 *   [R1] src/troute-routing/troute/routing/fast_reach/mc_reach.pyx:164-224
 *        compute_network_structured(...)  -- whole-network, all timesteps
 *   [R2] src/troute-routing/troute/routing/fast_reach/reach.pyx:66-103
 *        compute_reach_kernel(...) -> dict  -- one segment, one timestep
 *   [R3] src/kernel/muskingum/pyMCsingleSegStime_NoLoop.f90:8-21 and
 *        src/troute-routing/troute/routing/fast_reach/pyMCsingleSegStime_NoLoop.h:1-21
 *        c_muskingcungenwm(21 float*)  -- the Fortran bind(c) symbol
 *   [R4] src/troute-network/troute/network/musking/mc_reach_structs.h:8-17,
 *        reach_structs.h:11-25  -- AoS _MC_Segment/_MC_Reach/_Reach, replaced
 *        by the plan's SoA columns + CSR upstream lists
 *
 * Conventions
 *   - plain pointers and sizes only; every function returns 0 on success or a
 *     negative trmc_status and leaves a message in trmc_last_error()
 *     (thread-local).  The reference raises ValueError for shape errors
 *     (mc_reach.pyx:243-250); the Python shim maps TRMC_EINVAL to ValueError
 *     and everything else to RuntimeError.
 *   - "row" = position in the caller's segment table (the reference's
 *     data_idx order: ascending segment id, compute.py:1447-1461).
 *   - real arrays are float (precision 32) or double (precision 64), chosen at
 *     plan creation; the reference computes in float
 *     (src/kernel/muskingum/varPrecision.f90:5).
 *   - there is no CPU fallback: with no HIP device every entry point that
 *     computes fails with TRMC_ENODEVICE.
 */
#ifndef TRMC_H
#define TRMC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TRMC_ABI_VERSION 19

typedef enum trmc_status {
    TRMC_OK = 0,
    TRMC_EINVAL = -1,    /* bad argument / shape mismatch  -> ValueError     */
    TRMC_ECYCLE = -2,    /* upstream graph is not acyclic  -> ValueError     */
    TRMC_ENODEVICE = -3, /* no usable HIP device           -> RuntimeError   */
    TRMC_EHIP = -4,      /* HIP runtime error              -> RuntimeError   */
    TRMC_ENOMEM = -5,    /* allocation failed              -> MemoryError    */
    TRMC_ESTATE = -6     /* call order violated (e.g. download before route) */
} trmc_status;

/* Parameter column order of `params` (9 columns, row-major [nseg][9]): the
 * order column_mapper produces for the reference kernel, mc_reach.pyx:150-162 */
enum { TRMC_P_DT = 0, TRMC_P_DX, TRMC_P_BW, TRMC_P_TW, TRMC_P_TWCC, TRMC_P_N,
       TRMC_P_NCC, TRMC_P_CS, TRMC_P_S0, TRMC_NPARAM };

typedef struct trmc_plan trmc_plan;

/* Per-route timing and shape facts, filled by trmc_route_device().  Times are
 * HIP-event times on the plan's own stream. */
typedef struct trmc_stats {
    int64_t nseg;            /* rows in the plan (incl. boundary rows)        */
    int64_t nseg_routed;     /* rows that are computed                         */
    int32_t nlevels;         /* topological depth in segments                  */
    int32_t nsteps;          /* timesteps of the last route                    */
    int32_t assume_short_ts;
    int32_t main_launches;   /* launches of the segment-step kernel            */
    int64_t segment_steps;   /* nseg_routed * nsteps                           */
    double ms_prep;          /* forcing transpose + initial-state scatter      */
    double ms_main;          /* all launches of the segment-step kernel        */
    double ms_emit;          /* [t][seg] -> [row][t][q,v,d] transpose          */
    double ms_total;         /* prep + main + emit                             */
    /* short-timestep windows of a wide network: the leading `wide_levels` levels are routed `wide_k` timesteps per
     * launch by k_mc_tile (0: every level one step per launch) */
    int32_t wide_levels, wide_k;
    int32_t wide_launches;   /* launches of k_mc_tile (part of main_launches)  */
    int32_t mid_levels;      /* ... and the `mid_levels` levels right below them `mid_k` steps per launch (second tier; 0: none) */
    int64_t wide_segment_steps; /* segment-steps routed by the launches of both tiers */
    double ms_wide;          /* summed duration of the first tier's launches (they run beside the tail's, so this is not a share of ms_main) */
    int32_t mid_k, mid_launches;
    int32_t arithmetic;      /* TRMC_ARITH_EXACT / TRMC_ARITH_TOLERANCE of the plan */
    int32_t reserved0;
} trmc_stats;

const char *trmc_last_error(void);
int trmc_abi_version(void);
int trmc_device_count(int *count);

/*
 * Build a routing plan: topology flattened to segment levels, parameters laid
 * out as SoA columns in level-major order, everything resident in HBM.
 * Replaces the MC_Segment/MC_Reach object construction of [R1]
 * (mc_reach.pyx:283-378) and [R4].
 *
 *   nseg      rows
 *   up_ptr    [nseg+1] CSR offsets; up_idx[up_ptr[r]..up_ptr[r+1]) are the rows
 *             whose flow enters row r, IN THE ORDER THE REFERENCE SUMS THEM
 *             (mc_reach.pyx:499-502: upstream_connections[reach[0]] for the
 *             head of a reach, the previous segment inside a reach)
 *   params    [nseg][TRMC_NPARAM] float (always float: compute.py:549)
 *   boundary  [nseg] or NULL; 1 marks a row that is not routed but
 *             carries a prescribed hydrograph (the reference's
 *             upstream_results rows, mc_reach.pyx:451-469); 2 marks a
 *             ROUTED row that a short-timestep plan of the level engine
 *             keeps out of its leading levels -- the ones routed several
 *             timesteps per launch ahead of the window -- like the rows
 *             boundary rows feed (rows that will carry a lag,
 *             trmc_plan_set_lag); results do not depend on it
 *   precision 32 or 64
 *   device    HIP device ordinal
 */
int trmc_plan_create(int64_t nseg, const int64_t *up_ptr, const int64_t *up_idx,
                     const float *params, const uint8_t *boundary, int precision,
                     int device, trmc_plan **out);
/* The same with a per-row cost hint [nseg] (or NULL): the secant iterations each row needed at the end of an
 * earlier window (trmc_download_iterations).  Inside a level the order of the rows is free; with a hint rows of equal
 * cost sit together (wavefronts of one cost) and the costly blocks of a launch start first.  Results do not depend
 * on the hint.  No counterpart in the reference (its reach lists are traversed serially). */
int trmc_plan_create_hinted(int64_t nseg, const int64_t *up_ptr, const int64_t *up_idx,
                            const float *params, const uint8_t *boundary, const uint8_t *cost_hint,
                            int precision, int device, trmc_plan **out);
/* The same with flags: which engine runs the plan, and which timestep mode it is meant for.
 *   engine  TRMC_ENGINE_LEVELS   level engine: rows in level-major order, one launch per timestep (assume_short_ts) or per
 *                                wavefront diagonal (k_mc_step); the only engine of precision-64 plans
 *           TRMC_ENGINE_FLOW     dataflow engine (precision 32): rows in block order, one persistent launch per window,
 *                                rows exchange flows through tagged granules (k_mc_flow)
 *           TRMC_ENGINE_AUTO     flow, except for a precision-32 plan of 1 000 000 routed rows or more that is meant for
 *                                assume_short_ts: there every launch fills the device many times over and wavefronts
 *                                that are uniform in cost across a whole level outweigh the launch boundaries
 *   mode    TRMC_PLAN_SHORT_TS / TRMC_PLAN_FULL_TS: the assume_short_ts value the plan will be routed with, if the caller
 *           knows it (compute_nhd_routing_v02 does).  A plan routes correctly in either mode whatever it was built for;
 *           the flag chooses the row order that is fast for that mode (block order with cost tiers for the first, plain
 *           sub-tree blocks for the second and when unknown).
 * (troute_amd.plan maps the environment variable TRMC_ENGINE=levels|flow onto the engine flag of an "auto" plan: A/B runs.) */
enum { TRMC_ENGINE_AUTO = 0, TRMC_ENGINE_LEVELS = 1, TRMC_ENGINE_FLOW = 2, TRMC_ENGINE_MASK = 3,
       TRMC_PLAN_SHORT_TS = 4, TRMC_PLAN_FULL_TS = 8 };
int trmc_plan_create_ex(int64_t nseg, const int64_t *up_ptr, const int64_t *up_idx,
                        const float *params, const uint8_t *boundary, const uint8_t *cost_hint,
                        int precision, int device, int flags, trmc_plan **out);
/* The same with OPTIONS (since ABI 16): everything that used to be an environment variable read inside the library is a
 * field here, read ONCE when the plan is created (a clone inherits them).  A NULL pointer or a zero-filled struct means
 * the defaults; struct_size = sizeof(trmc_plan_options) lets the struct grow.  The library itself reads no environment
 * variable that decides how a plan routes (the Python host layer maps its documented TRMC_* test / measurement variables
 * onto this struct, troute_amd/plan.py).
 *   arithmetic     TRMC_ARITH_EXACT (0): every segment-step bit-identical to the reference Fortran (fp32: glibc's powf
 *                  restated, IEEE division and square root; fp64: glibc's pow restated).
 *                  TRMC_ARITH_TOLERANCE (1), precision 32 only: hardware log2 / exp2 power, reciprocal-multiply division,
 *                  hardware square root -- each within one unit in the last place, NOT bit-comparable.  Stated tolerance
 *                  against the reference Fortran (tests/test_gpu_tolerance.py asserts it, profiles/r05_tolerance_report.json
 *                  holds the measured distributions): a routed window with assume_short_ts: >= 99.9 % of all (row, step)
 *                  flows within rtol 1e-4 + atol 1e-6 m3/s, >= 99.99 % within rtol 3e-2 + atol 1e-4, every one within rtol
 *                  1e-1 + atol 1e-3; one segment-step on the reference's own kernel-vector population: 99 % of q / velocity /
 *                  depth within rtol 2e-5, 99.8 % within 1e-3, at most 0.1 % of the steps beyond 3e-2 -- the tail is the secant
 *                  iteration's 1 % exit test (MCsingleSegStime_f2py_NOLOOP.f90:83) amplifying a last-place difference where it
 *                  converges slowly, not the arithmetic's.  Without assume_short_ts the reference recurrence amplifies
 *                  differences from a cold start and no tolerance is claimed.
 *   wide_min_rows  rows a leading level must have to be routed wide_k steps per launch (k_mc_tile); 0 = default (288 per
 *                  compute unit; 256 in tolerance arithmetic), < 0 = never.  wide_levels: at most so many (0 = default 16).  wide_k: 0 = default 16.
 *   mid_min_rows   the same for a SECOND tier: the levels right below the wide ones, mid_k steps per launch under their own
 *                  skew, queued on the plan's stream between the tail's launches; 0 = default (OFF: measured slower on the
 *                  CONUS day at every setting tried), < 0 = never.  mid_levels (0 = 12), mid_k (0 = 4).
 *   tile_perm_group  the threads of every block of a wide tile take the block's rows by the cost class the rows showed in the
 *                  tile before (inside the launch; wavefronts of one class whatever the forcing does): 0 = default (ON, with or
 *                  without a cost hint), > 0 = on, < 0 = off.
 *   hot_rows       the few rows of a wide tile that showed three or more secant iterations (or went over bank) in the tile before
 *                  are routed by blocks of their own in the next one, so that they do not set the pace of the wavefront they
 *                  would otherwise sit in (1.5 % of the rows of an unordered CONUS plan are in half of its wavefronts); those
 *                  blocks are the first of the launch.  0 = default (on when the in-block partition is), > 0 = on, < 0 = off.
 *   cluster_rows   (since ABI 18) > 0: a plan created with TRMC_PLAN_SHORT_TS on the level engine lays the rows BELOW its wide
 *                  levels out in clusters -- connected pieces of the network of at most so many rows (at most 128 = one workgroup)
 *                  -- and routes them wide_k steps per launch as well (k_mc_ctile: flows inside a cluster through LDS, a cluster
 *                  level one tile behind the one that feeds it) instead of one launch per timestep: what a STREAM of windows
 *                  needs (trmc_stream_*; there wide_min_rows may be small -- more levels in slices cost nothing).  0 (default):
 *                  no cluster order -- a single window pays for the clusters' skew with some thirty small launches at its end.
 *                  Such a plan routes with assume_short_ts only.  Single windows whose boundary hydrographs arrive chunk by
 *                  chunk or whose rows carry a lag (trmc_plan_set_lag) keep the one-step launches.
 *   cluster_late_lag  cluster order: rows fed by boundary rows (and rows marked 2 in `boundary`) run at least so many tiles
 *                  behind the headwaters -- the trunk of a cut basin in a multi-GPU stream, whose inflows are exchanged once a
 *                  day (troute_amd.sequence.RouteStream).
 *   hot_wave_rows  rows of the hot list per WAVEFRONT of the blocks that route it (1..64; 0 = default 64).  An experiment's knob:
 *                  fewer rows per wavefront were meant to shorten the slowest wavefront's chain of wide_k dependent steps on a
 *                  device one launch does not fill (a rank of a multi-GPU job); measured slower at 32, 16 and 8.
 *   stream_split   s > 0: in a stream of windows the slices from level s on are launched on the clusters' stream instead of the
 *                  tile stream (an experiment: 3 % faster on CONUS on one GPU, twice as slow on a rank of eight; default 0).
 *   velocity_on_demand  != 0: in a stream of windows begun without full_output, a step's velocity is formed only where it is
 *                  handed on -- at the kept steps of output_stride, or nowhere when the products are hydrographs and final
 *                  states.  The velocity feeds nothing (MCsingleSegStime_f2py_NOLOOP.f90:163-169 computes it from the final
 *                  depth; the next step takes velp and does not read it), so every product keeps its bits; a tenth of a step's
 *                  instructions are not issued (CONUS: 12.3 instead of 13.5 ms per day).  Default 0: every step forms it.
 *   tail_sort      < 0: keep the per-level order below the tiled levels of a hinted short-timestep plan (default: by cost).
 *   stem_min_rows  general-mode dataflow plans: basins whose longest path has at least so many rows are laid out stem-last
 *                  (0 = default 1 024, < 0 = off).
 *   sequence_mode  != 0: the plan is one of several that take turns on the device (trmc_plan_clone, trmc_plan_chain_from):
 *                  a window's set-up is queued on the tile stream and its end with its last launch, so that another plan's
 *                  launches in the shared hardware queues do not hold it back.  Also trmc_plan_set_sequence_mode.
 *   flow_watchdog_ms  how long a poll of the dataflow engine may wait before the window is abandoned (0 = default 30 000).
 *   flow_overlap   != 0: consecutive time chunks of a dataflow window alternate between two compute streams.
 *   flow_lean      0 = automatic, > 0 always, < 0 never: the lean form of the dataflow kernel.
 *   flow_debug     != 0: the dataflow engine records when every block ran, and trmc_route_end prints a summary on stderr
 *                  (one line per block into the file named by TRMC_FLOW_DEBUG_FILE, if set) -- a diagnostic. */
enum { TRMC_ARITH_EXACT = 0, TRMC_ARITH_TOLERANCE = 1 };
typedef struct trmc_plan_options {
    int32_t struct_size;
    int32_t arithmetic;
    int64_t wide_min_rows;
    int32_t wide_levels, wide_k;
    int64_t mid_min_rows;
    int32_t mid_levels, mid_k;
    int32_t tile_perm_group;
    int32_t tail_sort;
    int32_t stem_min_rows;
    int32_t sequence_mode;
    int32_t flow_watchdog_ms;
    int32_t flow_overlap;
    int32_t flow_lean;
    int32_t flow_debug;
    int32_t hot_rows;
    int32_t cluster_rows;
    int32_t cluster_late_lag;
    int32_t stream_split;
    int32_t hot_wave_rows;
    int32_t velocity_on_demand;
    int32_t reserved[2];
} trmc_plan_options;
int trmc_plan_create_opt(int64_t nseg, const int64_t *up_ptr, const int64_t *up_idx,
                         const float *params, const uint8_t *boundary, const uint8_t *cost_hint,
                         int precision, int device, int flags, const trmc_plan_options *options, trmc_plan **out);
/* A STREAM OF WINDOWS on one plan (since ABI 18; csrc/stream.inc): the reference's run-set loop -- compute_nhd_routing_v02 per run
 * set, every call with that window's qlat_values (mc_reach.pyx:173,:723), new_q0 between them (AbstractNetwork.py:177-191; loop
 * nwm_routing/__main__.py:195-333) -- as ONE sequence of tile launches whose tile index runs on over the days.  assume_short_ts
 * only; the plan must be in cluster order (TRMC_PLAN_SHORT_TS on the level engine, cluster_rows >= 0), without reservoirs,
 * nudging tables or lagged rows.  Every launch routes every row through the next wide_k steps of ITS place in the stream (a row
 * `lag` tiles behind the headwaters works on an earlier tile, possibly of the day before): nsteps / wide_k launches per day, no
 * launches that only part of the network has work in, nothing between two days.  Results are the bits of the same days routed
 * one by one.
 *   trmc_stream_begin  after trmc_upload_forcing (the state q0 -- or NULL: what the plan's last window left -- and the shape of
 *                      the forcing: every day has its nq columns).  slots: days the ring holds (0 = as many as the rows' lag needs:
 *                      2 + ceil((lag_max + 1) / tiles_per_day)); full_output != 0: every day's out[nseg][nsteps][3] is assembled
 *                      (9.4 GB per slot for CONUS) and may be asked for; output_stride = n > 0: every n-th step of every row's
 *                      (q, v, d) is (what the reference's writers take, nwm_routing/output.py:209-216) -- otherwise a day's
 *                      products are its row-set hydrographs and its final state only and nothing else is written.
 *   trmc_stream_push   the next day: qlat [nseg][nq] on the host (page-locked memory for a copy that runs beside the launches; it
 *                      must stay unchanged until the day has begun on the device -- e.g. until the push after next returns);
 *                      boundary_q_dev: flows of the plan's boundary rows for the day [nboundary][nsteps] in DEVICE memory (NULL
 *                      without boundary rows); where the day's products go (any may be NULL): hyd_host [rows of rowset][nsteps],
 *                      q0_host [nseg][3], fvd_host [nseg][nsteps / output_stride][3] (or [nseg][nsteps][3] with full_output and no
 *                      stride) -- page-locked host arrays, filled when the day's LAST row has been routed through it, lag_max
 *                      launches into the days that follow.
 *   trmc_stream_flush  queues the launches that end every pushed day (nothing new starts); further days may be pushed afterwards.
 *   trmc_stream_wait   blocks until the products of `day` (0-based count of pushes) are on the host; TRMC_ESTATE if the day has
 *                      not been queued to its end yet (push on, or flush).
 *   trmc_stream_end    flush + wait for the device; the last day's final state stays staged (trmc_upload_forcing with q0 = NULL,
 *                      or another stream, continues from it).
 *   trmc_stream_info   facts: slots, tiles per day, the largest lag in tiles, wide levels, cluster levels, days pushed, days
 *                      queued to their end, tile launches so far (any pointer may be NULL). */
int trmc_stream_begin(trmc_plan *plan, int nsteps, int qts_subdivisions, int slots, int full_output, int output_stride);
int trmc_stream_push(trmc_plan *plan, const void *qlat, int64_t nq, const void *boundary_q_dev, int32_t rowset, void *hyd_host,
                     void *q0_host, void *fvd_host);
/* Multi-GPU streams (the trunk of a cut basin: rows fed by boundary rows, cluster_late_lag tiles behind).  trmc_stream_gather: the
 * flows of a row set over `day` [rows][nsteps] into DEVICE memory, queued on `stream` (NULL: the plan's) behind the launches
 * queued so far -- TRMC_ESTATE if the set's rows have not been queued through that day yet.  trmc_stream_boundary: the boundary
 * rows' flows of `day` (already pushed, with boundary_q_dev = NULL) from a block of hydrographs in DEVICE memory -- boundary row b
 * takes row index_dev[b] (NULL: row b) of q_dev, rows src_row_stride elements apart -- queued on `stream` (NULL: the copy stream);
 * the launches of the NEXT push go behind it, so the rows that read boundary rows must run at least that far behind
 * (trmc_plan_options.cluster_late_lag).  Calls for consecutive days in order. */
int trmc_stream_gather(trmc_plan *plan, int64_t day, int32_t rowset, void *dst_dev, void *stream);
int trmc_stream_boundary(trmc_plan *plan, int64_t day, const void *q_dev, int64_t src_row_stride, const int64_t *index_dev, void *stream);
/* The next ntiles launches without a new day (the rows still under way move on; at most up to the launch that ends the last day
 * pushed).  trmc_stream_flush = all of them.  A stream advanced this way takes a new day only once it has been flushed. */
int trmc_stream_advance(trmc_plan *plan, int ntiles);
int trmc_stream_flush(trmc_plan *plan);
int trmc_stream_wait(trmc_plan *plan, int64_t day);
int trmc_stream_info(const trmc_plan *plan, int32_t *slots, int32_t *tiles_per_day, int32_t *lag_max, int32_t *wide_levels,
                     int32_t *cluster_levels, int64_t *days_pushed, int64_t *days_complete, int64_t *launches);
/* Device time (HIP events, ms) from the first to the last of the nsteps / wide_k launches that pushing `day` queued on the stream
 * that carries the slices (the dominant kernel, k_mc_tile): / tiles_per_day = the mean duration of one launch beside whatever
 * else runs.  Waits for those launches.  The day must still be in the ring. */
int trmc_stream_day_ms(trmc_plan *plan, int64_t day, double *ms_out);
int trmc_stream_end(trmc_plan *plan);

/* Switch the sequence mode of a plan (see trmc_plan_options) between windows. */
int trmc_plan_set_sequence_mode(trmc_plan *plan, int on);
/* 1 if the plan computes in TRMC_ARITH_TOLERANCE, else 0. */
int trmc_plan_arithmetic(const trmc_plan *plan, int32_t *arithmetic);
/* 1 if the plan runs on the dataflow engine, 0 on the level engine. */
int trmc_plan_engine(const trmc_plan *plan, int32_t *is_flow);
void trmc_plan_destroy(trmc_plan *plan);

/* Host-only topology flattening (no device needed): the same routine the plan
 * uses.  level_of_row[nseg] (-1 for boundary rows), plan_pos_of_row[nseg],
 * *nlevels; any output pointer may be NULL.  TRMC_ECYCLE if the graph is cyclic. */
int trmc_topology_levels(int64_t nseg, const int64_t *up_ptr, const int64_t *up_idx,
                         const uint8_t *boundary, int32_t *level_of_row,
                         int64_t *plan_pos_of_row, int32_t *nlevels);

/* The same with the cost hint of trmc_plan_create_hinted (or NULL). */
int trmc_topology_levels_hinted(int64_t nseg, const int64_t *up_ptr, const int64_t *up_idx,
                                const uint8_t *boundary, const uint8_t *cost_hint, int32_t *level_of_row,
                                int64_t *plan_pos_of_row, int32_t *nlevels);

/* Host-only (since ABI 18): the CLUSTER ORDER a short-timestep plan of the level engine is laid out in (trmc_plan_options.cluster_rows;
 * csrc/topology.hpp).  The leading levels of at least wide_min_rows rows (at most wide_max_levels of them) stay level slices; the
 * rows below them form clusters of at most cluster_rows rows, packed into blocks of one cluster level each.  lag_of_row[nseg]:
 * tiles a row runs behind level 0 (-1 for boundary rows) -- a row only reads rows of its own block with its own lag, or rows
 * with a smaller one; block_of_row[nseg]: its cluster block, -1 in the slices.  Outputs may be NULL.
 * Reference analogue: build_subnetworks (nhd_network.py:691-771). */
int trmc_topology_clusters(int64_t nseg, const int64_t *up_ptr, const int64_t *up_idx, const uint8_t *boundary,
                           const uint8_t *cost_hint, int64_t wide_min_rows, int32_t wide_max_levels, int32_t cluster_rows,
                           int64_t *plan_pos_of_row, int32_t *lag_of_row, int32_t *block_of_row, int32_t *wide_levels,
                           int32_t *cluster_levels, int32_t *cluster_blocks);

/* Host-only: the BLOCK ORDER of the dataflow engine (fp32 plans; csrc/topology.hpp) -- routed rows in depth-first
 * post-order, stably sorted by a downstream-monotone cost tier (cost_tiers != 0), cut into blocks of *block_rows positions (one workgroup
 * each; boundary rows come first and belong to no block), rows of a block grouped by cost.  Every row's upstream rows
 * sit in its own block or an earlier one.  plan_pos_of_row[nseg]; rank_of_row[nseg] = dependency depth of the row
 * inside its block (the steps it trails by without assume_short_ts).  Outputs may be NULL.
 * Reference analogue: dfs_decomposition's reach order (nhd_network.py:503-557) and build_subnetworks (:691-771). */
int trmc_topology_blocks(int64_t nseg, const int64_t *up_ptr, const int64_t *up_idx,
                         const uint8_t *boundary, const uint8_t *cost_hint, int cost_tiers,
                         int64_t *plan_pos_of_row, int32_t *rank_of_row, int32_t *block_rows, int32_t *nblocks);

/* Host-only: the block order of a plan built for the GENERAL mode (TRMC_PLAN_FULL_TS on the dataflow engine) -- the plain
 * post-order (no cost tiers), except that a basin whose longest path has at least stem_min_rows rows (the library's
 * default: 1 024, TRMC_STEM_MIN_ROWS) is laid out as [its side tributaries, those joining at the TOP of that path first]
 * [the path, top to bottom], the path's run of positions beginning and ending on a block boundary (the gap filled with
 * whole small networks).  early_blocks[min(*nearly, early_cap)]: the blocks that hold such paths, ascending -- the general
 * mode's kernel starts them first, so that the dependent chain of mc_reach.pyx:499-505 (a row needs its upstream rows at the
 * SAME timestep) runs down behind the sweep over its basin instead of after it.  Outputs may be NULL. */
int trmc_topology_blocks_general(int64_t nseg, const int64_t *up_ptr, const int64_t *up_idx, const uint8_t *boundary,
                                 int32_t stem_min_rows, int64_t *plan_pos_of_row, int32_t *rank_of_row, int32_t *block_rows,
                                 int32_t *nblocks, int32_t *early_blocks, int32_t early_cap, int32_t *nearly);

/* Facts about the flattened topology (host side, no device work). */
int trmc_plan_info(const trmc_plan *plan, int64_t *nseg, int64_t *nseg_routed,
                   int32_t *nlevels, int32_t *precision, int32_t *device);
/* A plan in cluster order (trmc_plan_options.cluster_rows): lag_of_row[nseg] = tiles every row runs behind the headwaters (-1:
 * boundary row), the number of levels kept as slices, the number of cluster levels.  Pointers may be NULL. */
int trmc_plan_lags(const trmc_plan *plan, int32_t *lag_of_row, int32_t *wide_levels, int32_t *cluster_levels);
/* level_of_row[nseg] (-1 for boundary rows); plan_pos_of_row[nseg] = position in
 * the level-major device order.  Either pointer may be NULL. */
int trmc_plan_levels(const trmc_plan *plan, int32_t *level_of_row, int64_t *plan_pos_of_row);

/*
 * Stage one routing window's forcing into HBM (H2D).  Arrays are in row order,
 * element type per the plan's precision.
 *   qlat          [nseg][nq]   lateral inflow per forcing interval
 *   q0            [nseg][3]    qu0 qd0 h0  (mc_reach.pyx:361: initial q = column
 *                              0, initial depth = column 2); NULL = continue from the state
 *                              the previous window left in HBM (the reference's new_q0 warm
 *                              start between max_loop_size windows, AbstractNetwork.py:177-191,
 *                              without the host round trip)
 *   boundary_fvd  [nboundary][nsteps][3] q,v,d hydrographs of the boundary rows
 *                 in ascending row order, or NULL when the plan has none
 */
int trmc_upload_forcing(trmc_plan *plan, int nsteps, const void *qlat, int64_t nq,
                        const void *q0, const void *boundary_fvd);

/*
 * The same, with the lateral inflow taken straight from packed WRF-Hydro CHRTOUT columns:
 * the device decodes, joins on the feature axis and lays the forcing out -- replaces the host-side
 * get_ql_from_chrtout + DataFrame join (src/troute-network/troute/nhd_io.py:397-434, NHDNetwork.py:376-386).
 *   raw_a, raw_b  [nq][nfeat] int32, one row per forcing file, file order of the feature axis (qBucket and
 *                 qSfcLatRunoff; raw_b may be NULL: single variable q_lateral)
 *   pack_a/b      [6] doubles: scale_factor, add_offset, _FillValue, missing_value, valid_min, valid_max
 *                 (NaN for a fill value or bound the file does not define)
 *   feat_of_row   [nseg] position of each row's id on the feature axis, -1 = not in the files (inflow 0)
 * value = float32( unpack(a) + unpack(b) ), unpack(raw) = 0 where raw is masked (equal to a fill value or outside
 * the valid range, netCDF4's default auto-mask then .filled(0.0)), else raw * scale_factor + add_offset in double.
 */
int trmc_upload_forcing_packed(trmc_plan *plan, int nsteps, int64_t nq, int64_t nfeat, const int32_t *raw_a,
                               const int32_t *raw_b, const double *pack_a, const double *pack_b,
                               const int64_t *feat_of_row, const void *q0, const void *boundary_fvd);

/*
 * Supply the boundary rows' flow hydrographs from a DEVICE buffer q_dev[nboundary][nsteps] (ascending
 * boundary row order) after trmc_upload_forcing(..., boundary_fvd = NULL): the multi-GPU hand-off of
 * sub-basin outlet hydrographs to the trunk (reference: flowveldepth_interorder, compute.py:882-897)
 * without a host round trip.  Velocity/depth of boundary rows are not inputs of the routing.
 */
int trmc_set_boundary_flow_device(trmc_plan *plan, int nsteps, const void *q_dev);

/*
 * Level-pool reservoirs of the plan.  Reference: the RESERVOIR_LP branch of the
 * network loop, mc_reach.pyx:283-356 (setup) and :507-716 (wbody_type_code 1), around
 * LEVELPOOL_PHYSICS (src/kernel/reservoir/Level_Pool/module_levelpool.F:233-427).  A reservoir is one
 * row of the table (the reference's waterbody node); its inflow is the junction sum of its upstream
 * rows, its result row holds (outflow, 0, water elevation).
 *   res_rows [nres]     rows of the reservoirs
 *   par      [nres][9]  area max_depth orifice_area orifice_coefficient orifice_elevation
 *                       weir_coefficient weir_elevation weir_length dam_length  (plan precision)
 *   routing_period      seconds (the dt argument of compute_network_structured)
 * The initial outflow and water elevation of a reservoir row travel in q0[row] = (qd0, -, h0) of
 * trmc_upload_forcing.  nres = 0 removes the reservoirs.
 */
int trmc_set_reservoirs(trmc_plan *plan, int64_t nres, const int64_t *res_rows, const void *par,
                        double routing_period);
/* inflow_out[nres][nsteps]: the inflow every reservoir received at every step (the reservoir rows of the
 * reference's upstream_array, mc_reach.pyx:710).  D2H. */
int trmc_download_reservoir_inflow(trmc_plan *plan, void *inflow_out);

/*
 * Streamflow nudging at gages for the staged window.  Reference: simple_da
 * (src/troute-routing/troute/routing/fast_reach/simple_da.pyx:22-95) applied to the gage segment after
 * its reach has been routed for the timestep (mc_reach.pyx:761-796).  Which of simple_da's three
 * branches is taken at (gage, step) depends on the observation record only, so the caller resolves it
 * (troute_amd does, with the reference's float/double arithmetic incl. libc exp for the decay weight):
 *   mode[g][t-1] = 0  no observation, no last observation: the model value passes through
 *                  1  valid observation a[g][t-1]: flow := a, nudge := a - model
 *                  2  decay: nudge := (a - model) * w, flow := model + nudge   (a = last observation)
 * arrays [ngage][nsteps] (a, w in the plan's precision); gage_rows = rows of the gage segments.
 * Call after trmc_upload_forcing (which clears the tables); ngage = 0 switches nudging off.
 */
int trmc_set_nudging(trmc_plan *plan, int nsteps, int64_t ngage, const int64_t *gage_rows,
                     const uint8_t *mode, const void *a, const void *w);
/* Gages INSIDE a reach when the window is routed WITHOUT assume_short_ts (with it nothing is needed: a segment then reads
 * only stored flows of the step before).  successor_rows[g] = the row directly below gage g in its reach, or -1 when the
 * gage ends its reach.  The reference nudges after the whole reach has been routed (mc_reach.pyx:133-137, :761-796), so
 * that one row reads the gage segment's flow of the current step as it was BEFORE the nudge while every other reader gets
 * the nudged value.  Level engine only (TRMC_ESTATE on a dataflow plan); after trmc_set_nudging, per window. */
int trmc_set_nudging_successors(trmc_plan *plan, int64_t ngage, const int64_t *successor_rows);
/* nudge_out[ngage][nsteps]: the nudge applied at every gage and step (mc_reach.pyx:793). D2H. */
int trmc_download_nudge(trmc_plan *plan, void *nudge_out);

/*
 * Route nsteps timesteps on the device (asynchronous launches on the plan's
 * stream, then waits for completion).  Replaces the time x reach loop of [R1]
 * (mc_reach.pyx:492-505, :719-750) and the per-reach chain of
 * compute_reach_kernel (mc_reach.pyx:70-138).
 *   qts_subdivisions  timesteps per forcing interval (qlat column = (t-1)/qts)
 *   assume_short_ts   non-zero: quc := qup (mc_reach.pyx:504-505, :135-136)
 * Results stay in HBM until downloaded.
 */
int trmc_route_device(trmc_plan *plan, int nsteps, int qts_subdivisions, int assume_short_ts);

/*
 * The same window in parts, so that a caller can interleave other device work with it -- the multi-GPU
 * hand-off: sub-basin outlet hydrographs leave for the trunk's owner one time chunk at a time while the
 * later chunks are still being routed (reference: the ordered sub-network loop with its
 * flowveldepth_interorder hand-off, compute.py:553-1209, here pipelined in time instead of serialised).
 * Every call only queues work on the plan's stream (trmc_plan_stream) and returns; trmc_route_end waits.
 *   trmc_route_begin     as trmc_route_device up to the first timestep; boundary hydrographs may still be
 *                        missing (they then arrive through trmc_set_boundary_flow_range)
 *   trmc_route_advance   queue the launches (done, t_end]; launch d = timestep d (of the rows without lag)
 *   trmc_route_end       finish the result layout, wait, fill trmc_stats
 * trmc_route_device == begin + advance(nsteps) + end.
 */
int trmc_route_begin(trmc_plan *plan, int nsteps, int qts_subdivisions, int assume_short_ts);
int trmc_route_advance(trmc_plan *plan, int t_end);
int trmc_route_end(trmc_plan *plan);
/* The plan's HIP stream (hipStream_t as void*), for ordering foreign work (RCCL) against the plan's.  With
 * TRMC_FLOW_OVERLAP=1 in the environment (off by default: no gain measured), a window of the dataflow engine whose
 * launches are all resident alternates consecutive trmc_route_advance calls between two compute streams so that time
 * chunks overlap: the call then returns the stream the NEXT advance uses -- ask again before every advance;
 * trmc_gather_flow_range is queued behind the launches that hold its steps. */
int trmc_plan_stream(trmc_plan *plan, void **stream_out);
/* Register a set of rows once (their plan positions are kept in HBM); *id_out names it. */
int trmc_rowset_create(trmc_plan *plan, const int64_t *rows, int64_t nrows, int32_t *id_out);
/* Queue: dst_dev[i * dst_stride + (t - 1 - t_begin)] = flow of row i of the set at step t, t in (t_begin, t_end],
 * all of which must already be queued (trmc_route_advance) or routed.  dst_dev is device memory. */
int trmc_gather_flow_range(trmc_plan *plan, int32_t rowset, int t_begin, int t_end, void *dst_dev,
                           int64_t dst_stride);
/* Queue: boundary row b (ascending row order) takes flow q_dev[b * src_stride + (t - 1 - t_begin)] at the steps
 * (t_begin, t_end]; ranges must be supplied in order, t_begin = end of the previous one (0 first). */
int trmc_set_boundary_flow_range(trmc_plan *plan, int t_begin, int t_end, const void *q_dev, int64_t src_stride,
                                 void *stream /* hipStream_t to queue the copy on; NULL = the plan's */);
/* The same with a gather: boundary row b takes the source row src_index_dev[b] (device int64 [nboundary]) of q_dev --
 * e.g. its cut row's place in the block an all-gather delivered -- so that no separate gather kernel sits between the
 * collective and the boundary rows.  NULL = the identity. */
int trmc_set_boundary_flow_range_indexed(trmc_plan *plan, int t_begin, int t_end, const void *q_dev,
                                         int64_t src_stride, const int64_t *src_index_dev, void *stream);
/*
 * Time-skewed rows (assume_short_ts only).  lag_of_row[nseg] holds 0 or one common value L: launch d of the
 * window routes the rows without lag at step d and the lagged rows at step d - L, so a window takes
 * nsteps + L launches (trmc_route_advance counts launches).  Use: the trunk of a cut basin rides in the
 * launches of this GPU's sub-basins L steps behind them, which gives the cut-edge hydrographs of the other
 * GPUs L steps to arrive (boundary rows may feed lagged rows only) -- no launch of its own, no waiting.
 * With assume_short_ts a row at step t reads its upstream rows at step t-1 only, which the skew preserves.
 * NULL clears the lag.  Call before trmc_rowset_create.
 */
int trmc_plan_set_lag(trmc_plan *plan, const int32_t *lag_of_row);

/* Full result, row order: fvd_out[nseg][nsteps][3] = (q, vel, depth) per step,
 * i.e. flowveldepth[:, 1:, :] of [R1] (mc_reach.pyx:807-813).  D2H. */
int trmc_download_fvd(trmc_plan *plan, void *fvd_out);
/* Every `stride`-th step of it, decimated ON THE DEVICE: fvd_out[nseg][nsteps / stride][3], entry k = step stride (k + 1)
 * (1-based) -- the steps the reference's writers keep of a window (nwm_routing/output.py:209-216, :232-240: those whose end
 * falls on a multiple of dt * qts_subdivisions; stride = qts_subdivisions for hourly output of 5-minute steps).  A twelfth
 * of the bytes crosses the host link: a CONUS day 0.78 GB instead of 9.4.  Bit-identical to slicing the full array. */
int trmc_download_fvd_strided(trmc_plan *plan, int stride, void *fvd_out);
/* The same for the rows of a registered set (trmc_rowset_create), in the set's order: fvd_out[rows of the set][nsteps / stride][3]
 * (stride = 1: every step).  compute_nhd_routing_v02 hands its result back as one block per tailwater (compute.py:1738, the
 * per-network loop :1399-1738): with the set = the table's rows grouped by tailwater, each of those is a slice of fvd_out and
 * the host never permutes the block. */
int trmc_download_fvd_rowset(trmc_plan *plan, int stride, int32_t rowset, void *fvd_out);
/* on != 0: trmc_upload_forcing turns every NaN of qlat and q0 into 0 on the device, behind the copy.  The reference's reindexed
 * tables hold NaN on waterbody rows (compute.py:1466-1467), which the Muskingum-Cunge rows never read; the drop-in used to screen
 * the caller's frames for them on the host -- 20 ms of a CONUS call.  Default off: values go to the kernels as they are. */
int trmc_plan_set_nan_is_zero(trmc_plan *plan, int on);
/* Page-locked host memory for result arrays: a D2H copy into it runs at the speed of the link instead of through the
 * driver's staging buffers (the 9.4 GB flowveldepth array of a CONUS day: 0.2 s instead of 0.8 s).  The Python host side
 * keeps a small pool of these behind download_fvd(); a C caller may use them for any *_out argument.  Plain memory to the
 * host; release with trmc_host_free, not free(). */
int trmc_host_alloc(size_t bytes, void **ptr_out);
int trmc_host_free(void *ptr);
/* Final state in the reference's q0 layout, new_q0 = fvd[:, [-3,-3,-1]]
 * (AbstractNetwork.py:182-190): q0_out[nseg][3] = (q_T, q_T, depth_T).  D2H. */
int trmc_download_final_state(trmc_plan *plan, void *q0_out);
/* Cost collection for trmc_plan_create_hinted: with it enabled, every routing window also sums, per row,
 * min(secant iterations, 3) (+ 4 in a step that took the compound-channel branch) over its timesteps (2 more bytes read and written per segment-step); trmc_download_cost
 * returns the sums of the last window [nseg] and its length.  Off by default.  No counterpart in the reference. */
int trmc_plan_collect_cost(trmc_plan *plan, int enable);
int trmc_download_cost(trmc_plan *plan, uint16_t *cost_out, int32_t *nsteps_out);

/* Convergence diagnostic: iters_out[nseg] = secant iterations (all retries together, capped at 255) each row
 * spent on the LAST routed timestep (0 for boundary/reservoir rows and rows without flow; the reference
 * does not report its count, MCsingleSegStime_f2py_NOLOOP.f90:83-134).  Rows of slices too narrow for
 * the class partition keep 0.  D2H. */
int trmc_download_iterations(trmc_plan *plan, uint8_t *iters_out);
/* Flow hydrographs of selected rows (e.g. network outlets):
 * out[nrows][nsteps].  dst_is_device != 0: `out` is a device pointer on the
 * plan's device (used to hand outlet hydrographs to RCCL without a host trip). */
int trmc_gather_flow_rows(trmc_plan *plan, const int64_t *rows, int64_t nrows, void *out,
                          int dst_is_device);
/* With out == NULL and dst_is_device != 0 the gathered block stays in plan-owned HBM (results are
 * device-resident); this copies it to the host later: out[nrows][nsteps] of that gather. */
int trmc_download_gathered(trmc_plan *plan, void *out);

/* Throughput mode, copy hidden: what a caller consumes of a window -- the flow hydrographs of a registered row set (the
 * network outlets) and the final state (new_q0, AbstractNetwork.py:182-190) -- is gathered on the plan's stream into
 * plan-owned HBM and copied to the host on a SEPARATE copy stream, so the copy of window k runs beside the kernels of
 * window k + 1 (the planes the gathers read are only overwritten by kernels queued after them).  hyd_host
 * [rows of the set][nsteps] and q0_host [nseg][3] should be page-locked (trmc_host_alloc); either may be NULL.
 * trmc_fetch_wait returns when both arrays are complete; one fetch in flight per plan. */
/* Diagnosis: how many rows the tiles of this plan have routed from the hot list so far (trmc_plan_options.hot_rows; a running
 * total over all launches and windows, modulo 2^31; waits for the device). */
int trmc_plan_hot_rows(trmc_plan *plan, int64_t *rows_out);

/* Tell a plan which steps a caller will ask for with trmc_fetch_begin_fvd: windows begun afterwards write every stride-th step
 * of (q, v, d) into a block of their own AS THEY GO -- the rows of the tiled leading levels from the kernel that routes them,
 * the others gathered from the time-major planes when the block is fetched -- instead of reading the whole result once more
 * (9.4 GB of a CONUS day) to pick them out.  Results are the same; 0 switches it off.  A fetch with another stride, or of a
 * window that ran without tiles, decimates the result as before. */
int trmc_plan_set_output_stride(trmc_plan *plan, int32_t stride);

/* Diagnosis: a timeline of consecutive windows without a profiler attached (one that serialises what overlaps).  host_ring
 * [nwindows][4] uint64 in page-locked memory (trmc_host_alloc), or NULL to switch it off: window k since the call (level
 * engine, assume_short_ts, leading levels tiled) leaves in row k % nwindows the device's 100 MHz clock when its tiles
 * begin, when its last tile has ended, when its tail begins (the plan's stream is past the set-up and the hand-over) and
 * when its last step launch has ended -- four one-thread launches per window.  Clocks of plans on one device compare. */
int trmc_plan_set_stamps(trmc_plan *plan, void *host_ring, int32_t nwindows);

int trmc_fetch_begin(trmc_plan *plan, int32_t rowset, void *hyd_host, void *q0_host);
/* The same, and with it every `stride`-th step of (q, v, d) of EVERY row -- fvd_host [nseg][nsteps / stride][3], page-locked;
 * the steps stride, 2 stride, ... counted from 1, as trmc_download_fvd_strided -- decimated behind the window's last launch
 * and copied beside the next window: what the reference's writers take of a window at stream_output_internal_frequency
 * (nwm_routing/output.py:209-216).  The plan's next window starts its set-up behind the decimation (behind the copy itself
 * when stride == 1: the whole array then leaves straight from the result buffer).  fvd_host NULL: trmc_fetch_begin. */
int trmc_fetch_begin_fvd(trmc_plan *plan, int32_t rowset, void *hyd_host, void *q0_host, int stride, void *fvd_host);
int trmc_fetch_wait(trmc_plan *plan);

int trmc_get_stats(const trmc_plan *plan, trmc_stats *stats);

/* Convenience: upload + route + download in one call (what the drop-in
 * compute_network_structured shim uses). */
int trmc_route(trmc_plan *plan, int nsteps, int qts_subdivisions, int assume_short_ts,
               const void *qlat, int64_t nq, const void *q0, const void *boundary_fvd,
               void *fvd_out);

/*
 * Batch of independent single-segment steps on the device: the GPU
 * counterpart of calling [R3]/[R2] n times.
 *   in   [n][15]  dt qup quc qdp ql dx bw tw twcc n ncc cs s0 velp depthp
 *   out  [n][6]   qdc velc depthc ck cn X
 * precision 32: float arrays; 64: double arrays.  qdc is taken as 0 on entry,
 * as reach.pyx:55 passes it.
 */
int trmc_segments(int device, int precision, int64_t n, const void *in, void *out);
/* The same in a chosen arithmetic (TRMC_ARITH_EXACT / TRMC_ARITH_TOLERANCE, see trmc_plan_options), and -- iters_out [n]
 * or NULL -- the secant iterations every step took (f90:83-134, all retries together; 0 where nothing was routed). */
int trmc_segments_ex(int device, int precision, int arithmetic, int64_t n, const void *in, void *out, int32_t *iters_out);

/*
 * [R3] with its own signature: the reference's bind(c) entry point of one segment-step,
 * c_muskingcungenwm (src/kernel/muskingum/pyMCsingleSegStime_NoLoop.f90:8-21; C header
 * src/troute-routing/troute/routing/fast_reach/pyMCsingleSegStime_NoLoop.h:1-21; Cython declaration
 * fast_reach/fortran_wrappers.pxd:19-40; call site reach.pyx:37-94).  21 float pointers: dt qup quc qdp ql dx bw tw twcc
 * n ncc cs s0 velp depthp in, qdc velc depthc ck cn X out; no return value.  One step on the device per call (device
 * TRMC_DEVICE, default 0) -- the binding a maintainer swaps in for the Fortran symbol; batches go through trmc_segments.
 * It cannot signal either: on failure the six outputs are NaN and trmc_last_error() holds the reason.
 */
void trmc_muskingcungenwm(float *dt, float *qup, float *quc, float *qdp, float *ql, float *dx, float *bw, float *tw,
                          float *twcc, float *n, float *ncc, float *cs, float *s0, float *velp, float *depthp,
                          float *qdc, float *velc, float *depthc, float *ck, float *cn, float *X);

/*
 * Windows of ONE sequence on TWO plans of the same network (same inputs, same cost hint: the same order), taking turns:
 * the receiving plan's next window starts from the state the source plan's window leaves, handed over on the device --
 * the rows of the source's leading ("wide") levels behind its last tile, on the receiver's tile stream, the others behind
 * its tail, on the receiver's own stream -- so that the receiver's tiles can run while the source's tail is still
 * finishing its window (both plans in sequence mode: trmc_plan_options.sequence_mode).  The source's window must have been queued to its
 * end (trmc_route_advance to nsteps); the receiver must have its forcing staged; the next trmc_route_begin of the receiver
 * takes the handed-over state instead of the one staged with its forcing.  Level engine only.
 */
int trmc_plan_chain_from(trmc_plan *receiver, trmc_plan *source);

/*
 * A second set of WINDOW buffers on the static data of `plan`: the clone shares the original's topology, parameter and
 * constant columns in HBM (no second copy of them, no second flattening) and owns its own forcing, state planes, result
 * and streams -- so that consecutive windows of one sequence can take turns on the two (trmc_plan_chain_from), the next
 * day's forcing travelling to the idle one (trmc_stage_forcing) and its leading levels starting while the current day's
 * narrow levels finish.  The clone inherits the plan's options (trmc_plan_options) and the lag of its rows
 * (trmc_plan_set_lag: set it BEFORE cloning); reservoir / nudging tables, row sets and the cost collection are per clone
 * (set them on each).  Destroy order is free: the shared memory goes with the last user, the window buffers of a destroyed
 * original at once.
 */
int trmc_plan_clone(trmc_plan *plan, trmc_plan **out);

/*
 * Stage the NEXT window's forcing without waiting for anything: qlat [nseg][nq] (row order, the plan's precision; page-
 * locked host memory -- trmc_host_alloc -- for the copy to run asynchronously) is copied on the plan's copy stream, beside
 * whatever the device is routing; the next trmc_route_begin orders its set-up behind the copy.  The initial state is what
 * trmc_plan_chain_from hands over, else the state the plan's last window left (q0 = NULL of trmc_upload_forcing).  The
 * caller's counterpart in the reference: qlat_values arrives with every compute_network_structured call
 * (mc_reach.pyx:173,:723), the state through AbstractNetwork.new_q0 (AbstractNetwork.py:177-191).  A plan with
 * boundary rows gets their hydrographs of the staged window afterwards (trmc_set_boundary_flow_device before the
 * window begins, or trmc_set_boundary_flow_range while it runs).  The plan is idle (its last window ended), or routing a window that has been queued to its end -- the
 * staging area is only read by a window's set-up, so the copy goes behind that; the state then comes from
 * trmc_plan_chain_from, or, if nobody hands one over, from that very window once it has ended (trmc_route_end before the next
 * trmc_route_begin): one plan routing day after day with every day's forcing on its way a whole window ahead.
 */
int trmc_stage_forcing(trmc_plan *plan, int nsteps, const void *qlat, int64_t nq);

/*
 * Self-check, on the device, of the short exact forms the fp32 step takes under its range proofs (csrc/trmc.hip,
 * DevMathF: sqrt_r, k_of / quot, max_num) against the plain correctly rounded operations they stand for:
 *   what = 0   sqrt(x) for EVERY float x in [2**-60, 2**63], the range of the velocity's radicand under fast_ok
 *              (n and seed are ignored)
 *   what = 1   a / b and max(c, a / b) for n pseudo-random triples drawn from `seed`: a in [2**-10, 2**19] (dx), b in
 *              [2**-76, 2**62] (the celerity of an in-bank point), c in [2**-20, 2**40] (dt) -- significands uniform
 *              over all 2**23, exponents uniform over the ranges, both ends included; one triple in eight has a divisor
 *              whose significand is all ones, all ones but the last bit, zero or one (the hard cases of division by a
 *              refined reciprocal)
 * checked_out receives the number of values / triples compared, mismatches_out how many results differ in any bit
 * (0 is the claim).  No routing state is touched.
 */
int trmc_selfcheck_fast_arith(int device, int what, int64_t n, uint64_t seed, int64_t *checked_out, int64_t *mismatches_out);

/*
 * ---- communicator: the ranks of a multi-GPU job on one node (one process, or thread, per rank) ------------------
 * What ranks exchange on this path is what the reference hands from one sub-network order to the next as
 * flowveldepth_interorder (src/troute-routing/troute/routing/compute.py:882-897, consumed mc_reach.pyx:458-469):
 * hydrographs of the rows where the partition cuts a basin, and at the end the outlet hydrographs -- an all-gather of
 * equal-sized blocks.  Two transports behind one handle:
 *   trmc_comm_init      RCCL over xGMI (librccl.so is loaded when the first communicator is made, never by the
 *                       single-GPU path): rank 0 obtains an id with trmc_comm_unique_id and hands its
 *                       TRMC_COMM_ID_BYTES bytes to the other ranks by any means (a file, a socket); collectives on
 *                       device pointers are asynchronous on the caller's stream, like ncclAllGather.
 *   trmc_comm_init_shm  a POSIX shared-memory segment `name` ("/something", unique to the job) of `capacity_bytes`
 *                       (0: 64 MiB): blocks are staged through host memory, calls are synchronous.  For ranks that
 *                       share a device (RCCL refuses that) and, with device < 0, for hosts without a GPU, where only
 *                       the *_host collectives work.
 * Every rank makes the same sequence of collective calls.
 */
#define TRMC_COMM_ID_BYTES 128
typedef struct trmc_comm trmc_comm;
int trmc_comm_unique_id(void *id_out /* [TRMC_COMM_ID_BYTES] */);
int trmc_comm_init(int rank, int world, const void *id, int device, trmc_comm **out);
int trmc_comm_init_shm(int rank, int world, const char *name, int device, int64_t capacity_bytes, trmc_comm **out);
int trmc_comm_info(const trmc_comm *comm, int32_t *rank, int32_t *world, int32_t *is_rccl);
/* recv_dev[world][bytes] <- every rank's send_dev[bytes]; device pointers on the communicator's device; ordered on
 * `stream` (a hipStream_t; NULL = the null stream) */
int trmc_comm_all_gather(trmc_comm *comm, const void *send_dev, void *recv_dev, int64_t bytes, void *stream);
/* the same for host memory (small control data: a timing, a cost hint); returns when recv is complete */
int trmc_comm_all_gather_host(trmc_comm *comm, const void *send, void *recv, int64_t bytes);
int trmc_comm_barrier(trmc_comm *comm);
void trmc_comm_destroy(trmc_comm *comm);

/* Device buffers, streams and events for the hand-off between a plan's stream (trmc_plan_stream) and the collectives:
 * thin, typed-by-convention wrappers of hipMalloc / hipStream* / hipEvent* so that a host language needs no other GPU
 * library in the process.  `stream` / `event` are hipStream_t / hipEvent_t values. */
int trmc_dev_alloc(int device, int64_t bytes, void **ptr_out); /* zero-filled */
int trmc_dev_free(int device, void *ptr);
int trmc_dev_upload(int device, void *dst_dev, const void *src_host, int64_t bytes);
int trmc_dev_copy(int device, void *dst_dev, const void *src_dev, int64_t bytes, void *stream); /* device to device, on `stream` */
int trmc_dev_download(int device, void *dst_host, const void *src_dev, int64_t bytes, void *stream); /* waits for `stream` */
/* ... and without waiting: complete after trmc_stream_synchronize(stream); dst_host should be page-locked (trmc_host_alloc) */
int trmc_dev_download_async(int device, void *dst_host, const void *src_dev, int64_t bytes, void *stream);
/* dst_dev[i][row_bytes] <- src_dev[index_dev[i]][row_bytes], i < nrows (row_bytes a multiple of 4): picks the outlet
 * rows out of an all-gathered block */
int trmc_dev_gather_rows(int device, const void *src_dev, const int64_t *index_dev, int64_t nrows, int64_t row_bytes,
                         void *dst_dev, void *stream);
int trmc_stream_create(int device, void **stream_out);
/* ... with a priority: -1 low, 0 ordinary, +1 high (clamped to what the device offers).  A plan's own streams are: step
 * launches high, wide tiles ordinary, result transpose and result copies low; with one hardware queue per priority
 * (GPU_MAX_HW_QUEUES=1, which troute_amd sets) a caller's stream shares the queue of the plan stream of its priority, and
 * work in it that waits for an event holds back what was queued behind it in that queue. */
int trmc_stream_create_prio(int device, int priority, void **stream_out);
int trmc_stream_destroy(int device, void *stream);
int trmc_stream_synchronize(int device, void *stream);
int trmc_device_synchronize(int device);
int trmc_event_create(int device, void **event_out);
int trmc_event_destroy(int device, void *event);
int trmc_event_record(int device, void *event, void *stream);
int trmc_stream_wait_event(int device, void *stream, void *event);

#ifdef __cplusplus
}
#endif
#endif /* TRMC_H */
