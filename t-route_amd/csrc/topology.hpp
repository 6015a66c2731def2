// topology.hpp -- host-side flattening of a river network into segment levels.
//
// Input is the computational graph the reference builds out of reach lists and
// upstream_connections (mc_reach.pyx:283-289, :367-378): for every row the
// ordered list of rows whose flow enters it.  Output is the level-major device
// order the kernels use:
//   level(row) = 0 for headwaters, 1 + max(level(upstream rows)) otherwise;
//   boundary rows (prescribed hydrographs) get level -1 and constrain nothing;
//   rows are ordered by (level, depth-first rank from the outlets) so that each
//   level is one contiguous slice and rows that are neighbours inside a level
//   have neighbouring upstream rows in the slices before it (coalesced gathers).
// This replaces the reference's reach ordering contract ("every reach's
// upstream reaches precede it", nhd_network.py:503-557) by an explicit level
// structure; a reach of n segments simply occupies n consecutive levels.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace trmc {

struct Topology {
    int64_t nseg = 0;
    int64_t nboundary = 0;
    int32_t nlevels = 0;                 // number of routed levels (max level + 1)
    std::vector<int32_t> level_of_row;   // [nseg], -1 = boundary
    std::vector<int32_t> pos_of_row;     // [nseg] row -> plan position
    std::vector<int32_t> row_of_pos;     // [nseg] plan position -> row
    std::vector<int32_t> lvl_ptr;        // [nlevels+1] plan-position offsets of the routed levels
                                         // (positions [0, lvl_ptr[0]) hold the boundary rows)
    int32_t tail_from_level = 0;         // > 0: only the levels below it are contiguous slices; the rows of the deeper levels
                                         // -- positions [lvl_ptr[tail_from_level], lvl_ptr[nlevels]) -- are ordered by cost
                                         // across levels (build_topology, wide_min_rows): a short-timestep plan only
    std::vector<int32_t> up_ptr;         // [nseg+1] CSR over plan positions
    std::vector<int32_t> up_idx;         // upstream plan positions, reference summation order
    std::vector<int32_t> boundary_rows;  // ascending rows flagged as boundary
    // block order (build_topology with block_rows > 0), see below
    int32_t block_rows = 0;              // 0: level-major order; else rows per block of the dataflow engine
    int32_t nblocks = 0;                 // blocks of block_rows consecutive routed positions, from position nboundary
    std::vector<int32_t> rank_of_pos;    // [nseg] dependency depth of a position inside its block (topology.cpp)
    int32_t maxrank = 0;
    std::vector<uint8_t> cost_of_wave;   // [ceil(routed / 64)] the costliest row's hint (or drainage class) of every wavefront of the block order
    std::vector<uint8_t> prio_of_wave;   // [ceil(routed / 64)] issue priority 0..3 of every wavefront of the block order: by the
                                         // costliest row it holds (cost hint, else drainage size); costlier = higher
    std::vector<int32_t> early_blocks;   // block order with stem_min_rows > 0: the blocks that hold the long main stems, ascending
                                         // (the general mode starts them FIRST, see build_topology)
    // CLUSTER ORDER of the rows below the leading wide levels (build_topology with cluster_rows > 0; a short-timestep plan of
    // the level engine only): see below
    int32_t cl_rows = 0;                 // 0: none; else the most rows a cluster block holds
    int32_t cl_from_level = 0;           // W: the levels below it are level slices, the rows of the deeper ones are in clusters
    int32_t ncl = 0;                     // cluster levels: cluster level c runs W + c tiles behind level 0
    std::vector<int32_t> cblk_ptr;       // [ncblk + 1] plan positions: cluster block b holds [cblk_ptr[b], cblk_ptr[b + 1])
    std::vector<int32_t> cblk_of_cl;     // [ncl + 1] first block of every cluster level
    std::vector<int32_t> lagk_of_pos;    // [nseg] tiles a position runs behind level 0: its level (slices), W + c (clusters), 0 (boundary)
};

// Returns 0 on success; -1 bad argument, -2 cycle.  `err` receives a message.
// `cost_hint` (optional, [nseg]): rows of one level are grouped by descending hint (the secant iterations a row
// needed last time: waves then hold rows of one cost, the costly blocks of a launch start first); it only
// chooses among the orders that are valid anyway -- results do not depend on it.
//
// block_rows > 0 selects the BLOCK ORDER of the dataflow engine (k_mc_flow) instead of the level-major one: routed rows
// in depth-first post-order from the outlets (largest basin first, the larger tributary of a junction last), so that
// every row comes after all the rows draining into it and a run of consecutive positions is a handful of complete
// sub-trees plus the chain they hang off.  The order is cut into blocks of block_rows positions -- one workgroup of the
// engine each: a block only ever needs flows of its own rows and of EARLIER blocks.  Cost grouping happens twice: the
// whole order is stably sorted by a downstream-monotone cost tier (cheap rows first; still a valid order), so a block
// holds rows of one tier, and inside a block rows are grouped by descending cost (the hint, or without one the number
// of rows draining through, which decides how wet a channel is) so that a wavefront holds rows of one cost.
// cost_tiers = false keeps the plain post-order (blocks = whole sub-trees) and ignores the hint: the order for routing
// WITHOUT the short-timestep assumption, where rows of a wavefront trail each other by their dependency depth and what
// counts is that a block's rows are close in the network, not close in cost (k_mc_flow).  Reference analogue of the order: dfs_decomposition's
// "every reach's upstream reaches precede it" (nhd_network.py:503-557); of the blocks: build_subnetworks
// (nhd_network.py:691-771), here a few hundred rows instead of 10 000 and pipelined in time instead of by order.
// boundary[r]: 0 = routed, 1 = boundary row (prescribed hydrograph), 2 = routed but "late": see boundary_floor.
// boundary_floor (level order only): a routed row with a boundary row among its upstream rows, or marked late, gets at
// least this level.
// Boundary rows constrain nothing, so the rows they feed would otherwise be headwaters of the level order -- level 0, among
// the widest levels, the ones the level engine routes several timesteps per launch ahead of the window's progress
// (k_mc_tile); rows whose inflow arrives chunk by chunk during the window (the trunk of a cut basin, distributed.py) must
// stay out of those.  Levels between may be empty; results do not depend on it.
// wide_min_rows > 0 (level order, with a cost hint): the plan is only ever routed with assume_short_ts, where a row at step t
// reads flows of step t - 1 -- so the rows BELOW the leading wide levels (at most wide_max_levels levels of at least
// wide_min_rows rows: the ones the level engine routes several timesteps per launch) need not be grouped by level at all:
// one launch per timestep covers all of them whatever their order.  They are ordered by descending cost ACROSS levels
// (then by level, then as inside a level): a wavefront holds rows of one cost class also where the levels are a hundred
// rows wide and the per-level order mixed three classes in one wavefront (917 instructions per wavefront-step there
// against 788 in the wider levels).  Topology::tail_from_level says where that part begins.
// mid_min_rows / mid_max_levels: the level engine's SECOND tier -- up to mid_max_levels further levels of at least mid_min_rows
// rows right below the wide ones, routed a few timesteps per launch under a skew of their own -- stays in level slices too;
// the cost order begins below it.
// stem_min_rows > 0 (block order without cost tiers: plans built for the GENERAL mode, where a row needs its upstream rows
// at the SAME step and a chain of n rows cannot finish a window before n + nsteps dependent steps have run one after the
// other): a basin whose STEM -- the longest path into its outlet: from the outlet upstream, always into the tributary of
// the highest level -- has at least that many rows is laid out as  [side tributaries, the one joining at the TOP of the stem first] [the stem, top to
// bottom]  instead of the plain post-order (which comes to: side tributaries from the BOTTOM up, then the stem).  With the
// plain order the stem's first row waits for the tributary that is routed last; with this one every stem row's side
// tributary is routed before those of the rows below it, so a stem that STARTS with the window advances behind the sweep
// over its basin instead of after it.  Topology::early_blocks lists the blocks that hold such stems; k_mc_flow<false> hands
// them out first (they wait for their inflows in place; every other block still only needs blocks that were started before
// it or are among those few).
// cluster_rows > 0 (level order, a plan that is only ever routed with assume_short_ts): the rows BELOW the leading wide levels
// (the same rule as above picks those; none of them is wide: every routed row) are laid out in CLUSTERS -- connected pieces of
// the network of at most cluster_rows rows -- so that the level engine can route them several timesteps per launch as well
// (k_mc_ctile): the rows of a cluster advance together and hand their flows to each other through LDS, a cluster only reads
// rows OUTSIDE itself that run at least one tile ahead.  Cluster level of a row: W if nothing below the wide levels drains into
// it; else m = the highest cluster level among the rows draining into it, if the clusters of that level among them and the
// row itself fit into one cluster (they are merged); else m + 1 (a new cluster).  A chain of the network therefore climbs one
// cluster level per cluster_rows rows it collects, not one per row: the 3 538 levels of the synthetic CONUS network become
// 6 slices + 29 cluster levels of 128 rows.  Clusters of one level are packed into blocks of at most cluster_rows rows (by
// descending cost hint, so that a block holds clusters of one cost), blocks are ordered by cluster level: a launch takes a
// contiguous range of them.  cluster_late_lag > 0: a row fed by a boundary row (or marked late) runs at least so many tiles
// behind level 0 -- the trunk of a cut basin, whose inflows from other GPUs arrive a day or two after they were routed
// (troute_amd.sequence).  Results do not depend on it.  Reference analogue: build_subnetworks (nhd_network.py:691-771) --
// there sub-networks of 10 000 segments handed from one order to the next by the host, here of 128 rows pipelined in time.
int build_topology(int64_t nseg, const int64_t *up_ptr, const int64_t *up_idx,
                   const uint8_t *boundary, Topology &topo, std::string &err, const uint8_t *cost_hint = nullptr,
                   int32_t block_rows = 0, bool cost_tiers = true, int32_t boundary_floor = 0, int64_t wide_min_rows = 0,
                   int32_t wide_max_levels = 0, int32_t stem_min_rows = 0, int64_t mid_min_rows = 0, int32_t mid_max_levels = 0,
                   int32_t cluster_rows = 0, int32_t cluster_late_lag = 0);

} // namespace trmc
