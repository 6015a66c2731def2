// topology.cpp -- see topology.hpp.
#include "topology.hpp"

#include <algorithm>
#include <cstdio>
#include <string>
#include <cstdlib>
#include <limits>
#include <utility>

namespace trmc {

int build_topology(int64_t nseg, const int64_t *up_ptr, const int64_t *up_idx,
                   const uint8_t *boundary, Topology &t, std::string &err, const uint8_t *cost_hint, int32_t block_rows,
                   bool cost_tiers, int32_t boundary_floor, int64_t wide_min_rows, int32_t wide_max_levels, int32_t stem_min_rows,
                   int64_t mid_min_rows, int32_t mid_max_levels, int32_t cluster_rows, int32_t cluster_late_lag)
{
    if (nseg < 0 || nseg >= std::numeric_limits<int32_t>::max()) {
        err = "nseg out of range";
        return -1;
    }
    if (nseg > 0 && (!up_ptr || up_ptr[0] != 0)) {
        err = "up_ptr must start at 0";
        return -1;
    }
    const int64_t nnz = nseg ? up_ptr[nseg] : 0;
    for (int64_t r = 0; r < nseg; ++r)
        if (up_ptr[r + 1] < up_ptr[r]) {
            err = "up_ptr is not monotone at row " + std::to_string(r);
            return -1;
        }
    if (nnz >= std::numeric_limits<int32_t>::max()) {
        err = "too many upstream links";
        return -1;
    }
    for (int64_t k = 0; k < nnz; ++k)
        if (up_idx[k] < 0 || up_idx[k] >= nseg) {
            err = "up_idx[" + std::to_string(k) + "] = " + std::to_string(up_idx[k]) + " is not a row";
            return -1;
        }

    for (int64_t r = 0; boundary && r < nseg; ++r)
        if (boundary[r] > 2) {
            err = "boundary[" + std::to_string(r) + "] must be 0 (routed), 1 (boundary row) or 2 (routed, kept below the leading levels)";
            return -1;
        }
    t = Topology();
    t.nseg = nseg;
    auto is_b = [&](int64_t r) { return boundary && boundary[r] == 1; };
    auto is_late = [&](int64_t r) { return boundary && boundary[r] == 2; }; // a routed row that stays below the leading levels

    // downstream CSR over routed rows (edges u -> r for routed r)
    std::vector<int32_t> down_ptr(nseg + 1, 0), indeg(nseg, 0);
    for (int64_t r = 0; r < nseg; ++r) {
        if (is_b(r)) continue;
        for (int64_t k = up_ptr[r]; k < up_ptr[r + 1]; ++k) {
            const int64_t u = up_idx[k];
            if (u == r) {
                err = "row " + std::to_string(r) + " lists itself as upstream";
                return -2;
            }
            ++down_ptr[u + 1];
            if (!is_b(u)) ++indeg[r];
        }
    }
    for (int64_t r = 0; r < nseg; ++r) down_ptr[r + 1] += down_ptr[r];
    std::vector<int32_t> down_idx(down_ptr[nseg]);
    {
        std::vector<int32_t> fill(down_ptr.begin(), down_ptr.end() - 1);
        for (int64_t r = 0; r < nseg; ++r) {
            if (is_b(r)) continue;
            for (int64_t k = up_ptr[r]; k < up_ptr[r + 1]; ++k)
                down_idx[fill[up_idx[k]]++] = (int32_t)r;
        }
    }

    // levels: longest path from a headwater, Kahn order
    t.level_of_row.assign(nseg, -1);
    std::vector<int32_t> queue;
    queue.reserve(nseg);
    int64_t nrouted = 0;
    for (int64_t r = 0; r < nseg; ++r) {
        if (is_b(r)) {
            t.boundary_rows.push_back((int32_t)r);
            continue;
        }
        ++nrouted;
        int32_t lvl0 = 0;
        if (boundary_floor > 0 && block_rows == 0) {
            if (is_late(r)) lvl0 = boundary_floor;
            for (int64_t k = up_ptr[r]; k < up_ptr[r + 1]; ++k)
                if (is_b(up_idx[k])) lvl0 = boundary_floor;
        }
        if (lvl0 > 0) t.level_of_row[r] = lvl0; // (a floor under whatever its routed upstream rows will say)
        if (indeg[r] == 0) {
            t.level_of_row[r] = lvl0;
            queue.push_back((int32_t)r);
        }
    }
    t.nboundary = (int64_t)t.boundary_rows.size();
    int32_t maxlevel = -1;
    for (size_t head = 0; head < queue.size(); ++head) {
        const int32_t r = queue[head];
        const int32_t lr = t.level_of_row[r];
        maxlevel = std::max(maxlevel, lr);
        for (int32_t k = down_ptr[r]; k < down_ptr[r + 1]; ++k) {
            const int32_t dn = down_idx[k];
            t.level_of_row[dn] = std::max(t.level_of_row[dn], lr + 1);
            if (--indeg[dn] == 0) queue.push_back(dn);
        }
    }
    if ((int64_t)queue.size() != nrouted) {
        err = "upstream graph has a cycle (" + std::to_string(nrouted - (int64_t)queue.size())
              + " rows unreachable from headwaters)";
        return -2;
    }
    t.nlevels = maxlevel + 1;

    // upstream CSR over plan positions (boundary rows keep an empty list)
    auto csr_in_plan_order = [&]() {
        t.up_ptr.assign(nseg + 1, 0);
        for (int64_t p = 0; p < nseg; ++p) {
            const int32_t r = t.row_of_pos[p];
            t.up_ptr[p + 1] = t.up_ptr[p] + (is_b(r) ? 0 : (int32_t)(up_ptr[r + 1] - up_ptr[r]));
        }
        t.up_idx.resize(t.up_ptr[nseg]);
        for (int64_t p = 0; p < nseg; ++p) {
            const int32_t r = t.row_of_pos[p];
            if (is_b(r)) continue;
            int32_t w = t.up_ptr[p];
            for (int64_t k = up_ptr[r]; k < up_ptr[r + 1]; ++k) t.up_idx[w++] = t.pos_of_row[up_idx[k]];
        }
    };

    if (block_rows > 0) {
        // ---- block order of the dataflow engine (topology.hpp) ----------------------------------------------
        t.block_rows = block_rows;
        // rows draining through each row (itself included), in Kahn order: every upstream row is final before its
        // downstream rows are visited.  (On a graph with bifurcations the count is an over-estimate: it only ranks.)
        std::vector<int64_t> drain(nseg, 1);
        for (const int32_t r : queue)
            for (int32_t k = down_ptr[r]; k < down_ptr[r + 1]; ++k) drain[down_idx[k]] += drain[r];
        std::vector<int32_t> outlets;
        for (int64_t o = 0; o < nseg; ++o)
            if (!is_b(o) && down_ptr[o + 1] == down_ptr[o]) outlets.push_back((int32_t)o);
        std::stable_sort(outlets.begin(), outlets.end(), [&](int32_t a, int32_t b) { return drain[a] > drain[b]; });
        std::vector<int32_t> post; // routed rows, every row after all rows draining into it
        post.reserve(nrouted);
        std::vector<std::pair<int64_t, int64_t>> stem_runs; // [first, last) indices into `post` of the long main stems
        {
            std::vector<uint8_t> seen(nseg, 0);
            std::vector<int32_t> kids;               // scratch: routed upstream rows of the row being expanded
            std::vector<std::pair<int32_t, int32_t>> stack; // (row, state): state 0 = expand, 1 = emit
            // the routed upstream rows of r not visited yet, by descending size (marks them visited)
            auto kids_of = [&](int32_t r) {
                kids.clear();
                for (int64_t k = up_ptr[r]; k < up_ptr[r + 1]; ++k) {
                    const int32_t u = (int32_t)up_idx[k];
                    if (!is_b(u) && !seen[u]) {
                        seen[u] = 1;
                        kids.push_back(u);
                    }
                }
                std::stable_sort(kids.begin(), kids.end(), [&](int32_t a, int32_t b) { return drain[a] > drain[b]; });
            };
            // post-order of the sub-tree of `root` (marked visited by the caller)
            auto walk = [&](int32_t root) {
                stack.emplace_back(root, 0);
                while (!stack.empty()) {
                    auto [r, st] = stack.back();
                    stack.pop_back();
                    if (st == 1) {
                        post.push_back(r);
                        continue;
                    }
                    stack.emplace_back(r, 1);
                    // visited in ascending size (the largest tributary last, right before its junction): pushed in
                    // descending order of visit
                    kids_of(r);
                    for (const int32_t u : kids) stack.emplace_back(u, 0);
                }
            };
            std::vector<int32_t> stem, side;
            std::vector<int64_t> side_ptr;
            // A stem's run of positions begins and ends on a block boundary: the rows of a block advance together (a
            // wavefront waits when one of its lanes does), so a stem row must not share its block with rows whose inflows
            // are routed much later -- the end of the last side tributary in front of it, the start of the next basin behind
            // it.  The gap is filled with whole small networks from the end of the list (they need nobody and are through
            // at once).
            size_t small_end = outlets.size();
            auto pad_to_block = [&]() {
                int64_t pad = (block_rows - (int64_t)post.size() % block_rows) % block_rows;
                size_t j = small_end;
                for (int tries = 0; pad > 0 && j > 0 && tries < 65536; ++tries) {
                    const int32_t o2 = outlets[--j];
                    if (seen[o2]) {
                        if (j + 1 == small_end) small_end = j;
                        continue;
                    }
                    if (drain[o2] > pad) continue;
                    seen[o2] = 1;
                    const size_t before = post.size(); // (drain is an upper bound of what the walk emits: boundary rows, bifurcations)
                    walk(o2);
                    pad -= (int64_t)(post.size() - before);
                    if (j + 1 == small_end) small_end = j;
                }
            };
            for (const int32_t o : outlets) {
                if (seen[o]) continue;
                seen[o] = 1;
                if (stem_min_rows > 0 && !cost_tiers && t.level_of_row[o] + 1 >= stem_min_rows) {
                    // the stem: the LONGEST path into the outlet (always into the tributary of the highest level; the larger
                    // one among equals) -- the chain the window's last step has to come down -- and, per stem row, its
                    // other tributaries in the plain walk's order of visit (ascending size)
                    stem.clear();
                    side.clear();
                    side_ptr.assign(1, 0);
                    for (int32_t v = o;;) {
                        stem.push_back(v);
                        kids_of(v);
                        size_t up = 0;
                        for (size_t i = 1; i < kids.size(); ++i)
                            if (t.level_of_row[kids[i]] > t.level_of_row[kids[up]]) up = i;
                        for (size_t i = kids.size(); i-- > 0;)
                            if (i != up) side.push_back(kids[i]);
                        side_ptr.push_back((int64_t)side.size());
                        if (kids.empty()) break;
                        v = kids[up];
                    }
                    if ((int64_t)stem.size() >= stem_min_rows) {
                        for (size_t i = stem.size(); i-- > 0;) // from the top of the stem down
                            for (int64_t k = side_ptr[i]; k < side_ptr[i + 1]; ++k) walk(side[(size_t)k]);
                        pad_to_block();
                        stem_runs.emplace_back((int64_t)post.size(), (int64_t)(post.size() + stem.size()));
                        for (size_t i = stem.size(); i-- > 0;) post.push_back(stem[i]);
                        pad_to_block();
                        continue;
                    } else { // a short stem (levels lifted by a floor, not by rows): side tributaries from the bottom up
                        for (size_t i = 0; i < stem.size(); ++i)
                            for (int64_t k = side_ptr[i]; k < side_ptr[i + 1]; ++k) walk(side[(size_t)k]);
                    }
                    for (size_t i = stem.size(); i-- > 0;) post.push_back(stem[i]);
                    continue;
                }
                walk(o);
            }
        }
        if ((int64_t)post.size() != nrouted) { // cannot happen in a DAG
            err = "internal: depth-first walk missed rows";
            return -2;
        }
        {   // The block order's one promise: every row comes AFTER all the rows draining into it (a block only ever needs
            // its own rows and earlier blocks).  The plain post-order keeps it by construction; the stem layout marks stem
            // rows and side-tributary roots as seen before it emits them, and in a network with bifurcations a side
            // sub-tree can hold a row whose second upstream row is a stem row or a later side root -- emitted before it.
            // Checked here on every edge; a network the stem layout cannot order goes back to the plain post-order.
            std::vector<int32_t> at(nseg, -1);
            for (size_t i = 0; i < post.size(); ++i) at[post[i]] = (int32_t)i;
            bool ordered = true;
            for (size_t i = 0; ordered && i < post.size(); ++i) {
                const int32_t r = post[i];
                for (int64_t k = up_ptr[r]; k < up_ptr[r + 1]; ++k) {
                    const int64_t u = up_idx[k];
                    if (!is_b(u) && at[u] > (int32_t)i) {
                        ordered = false;
                        break;
                    }
                }
            }
            if (!ordered) {
                if (stem_min_rows > 0)
                    return build_topology(nseg, up_ptr, up_idx, boundary, t, err, cost_hint, block_rows, cost_tiers, boundary_floor,
                                          wide_min_rows, wide_max_levels, 0, mid_min_rows, mid_max_levels);
                err = "internal: the block order puts a row ahead of a row that drains into it";
                return -2;
            }
        }
        if (!cost_tiers) cost_hint = nullptr;
        // Cost tiers.  A block runs at the pace of its costliest wavefront, so blocks should hold rows of one cost.  Every
        // row gets a tier -- its hint quantised to at most 8 steps, or without a hint the size class of its drainage --
        // made monotone downstream (a row is at least as costly as anything draining into it: how wet a channel is only
        // grows downstream, and the closure repairs the exceptions), and the post-order is stably sorted by tier: cheap
        // rows first.  Every edge still points from an earlier to a later position, so the order stays a valid one.
        {
            std::vector<uint8_t> tier(nseg, 0);
            if (cost_hint) {
                int lo = 255, hi = 0;
                for (const int32_t r : post) {
                    lo = std::min(lo, (int)cost_hint[r]);
                    hi = std::max(hi, (int)cost_hint[r]);
                }
                const int span = hi - lo + 1, nt = std::min(8, span);
                for (const int32_t r : post) tier[r] = (uint8_t)(((int)cost_hint[r] - lo) * nt / span);
            } else {
                for (const int32_t r : post) {
                    int b = 0;
                    for (int64_t d = drain[r]; d >= 4 && b < 3; d >>= 2) ++b; // 1-3, 4-15, 16-63, 64 and more rows
                    tier[r] = (uint8_t)b;
                }
            }
            for (const int32_t r : queue) // Kahn order: a row's tier is final before it is pushed downstream
                for (int32_t k = down_ptr[r]; k < down_ptr[r + 1]; ++k)
                    tier[down_idx[k]] = std::max(tier[down_idx[k]], tier[r]);
            if (cost_tiers) std::stable_sort(post.begin(), post.end(), [&](int32_t a, int32_t b) { return tier[a] < tier[b]; });
        }
        t.pos_of_row.assign(nseg, -1);
        t.row_of_pos.assign(nseg, -1);
        for (int64_t b = 0; b < t.nboundary; ++b) {
            t.pos_of_row[t.boundary_rows[b]] = (int32_t)b;
            t.row_of_pos[b] = t.boundary_rows[b];
        }
        t.nblocks = (int32_t)((nrouted + block_rows - 1) / block_rows);
        // the blocks of the long stems, largest basin first, at most kEarlyBlocksMax of them in all: they hold their slots
        // from the first moment of a launch, and what they wait for needs slots too
        constexpr size_t kEarlyBlocksMax = 64;
        t.early_blocks.clear();
        for (const auto &run : stem_runs) {
            const int64_t b0 = run.first / block_rows, b1 = (run.second - 1) / block_rows;
            if (t.early_blocks.size() + (size_t)(b1 - b0 + 1) > kEarlyBlocksMax) break;
            for (int64_t b = b0; b <= b1; ++b)
                if (t.early_blocks.empty() || t.early_blocks.back() < (int32_t)b) t.early_blocks.push_back((int32_t)b);
        }
        t.rank_of_pos.assign(nseg, 0);
        std::vector<int32_t> idx;
        std::vector<int64_t> key;
        for (int32_t b = 0; b < t.nblocks; ++b) {
            const int64_t i0 = (int64_t)b * block_rows, i1 = std::min<int64_t>(nrouted, i0 + block_rows);
            const int32_t m = (int32_t)(i1 - i0);
            idx.resize(m);
            key.resize(m);
            for (int32_t i = 0; i < m; ++i) {
                const int32_t r = post[i0 + i];
                idx[i] = i;
                // (by size, not by the dependency depth inside the block -- lanes dealt out shallow rows first, so that a
                // wavefront's ranks span a quarter of the block's, were measured for the general mode: 36.4 ms against 31.3;
                // by size the rows of a chain sit in neighbouring lanes of ONE wavefront and hand over without a wait)
                key[i] = cost_hint ? (int64_t)cost_hint[r] : drain[r];
            }
            std::stable_sort(idx.begin(), idx.end(), [&](int32_t a, int32_t c) { return key[a] > key[c]; });
            for (int32_t i = 0; i < m; ++i) {
                const int32_t r = post[i0 + idx[i]];
                const int32_t p = (int32_t)(t.nboundary + i0 + i);
                t.pos_of_row[r] = p;
                t.row_of_pos[p] = r;
            }
        }
        csr_in_plan_order();
        // Issue priority of every wavefront.  A row stays in its wavefront for a whole window, so the window lasts as long
        // as its slowest wavefront needs -- one that holds rows of three secant iterations per step takes twice the
        // instructions of one that holds dry channels, and everything downstream of it waits.  The costlier a wavefront,
        // the higher its priority on the SIMD it shares (s_setprio): the slow ones run at the pace of a wavefront alone,
        // the cheap ones fill the issue slots they leave.
        {
            int64_t lo = std::numeric_limits<int64_t>::max();
            auto cost_of = [&](int32_t r) -> int64_t {
                if (cost_hint) return cost_hint[r];
                int64_t b = 0;
                for (int64_t d = drain[r]; d >= 4 && b < 3; d >>= 2) ++b;
                return b;
            };
            for (const int32_t r : post) lo = std::min(lo, cost_of(r));
            const int64_t nw = (nrouted + 63) / 64;
            t.prio_of_wave.assign((size_t)nw, 0);
            std::vector<int64_t> wcost((size_t)nw, lo);
            for (int64_t w = 0; w < nw; ++w)
                for (int64_t p = t.nboundary + w * 64; p < std::min<int64_t>(nseg, t.nboundary + (w + 1) * 64); ++p)
                    wcost[(size_t)w] = std::max(wcost[(size_t)w], cost_of(t.row_of_pos[p]));
            t.cost_of_wave.resize((size_t)nw);
            for (int64_t w = 0; w < nw; ++w) t.cost_of_wave[(size_t)w] = (uint8_t)std::min<int64_t>(255, wcost[(size_t)w]);
            if (nw > 0) {
                // By rank, not by value: a handful of rows with a hint far above the rest would compress everybody else into
                // priorities 0 and 1.  Equal costs get equal priorities.  (One priority per block -- its wavefronts exchange
                // flows every step -- measured worse on the ranks of an 8-way partition: 5.3 ms against 4.65 ms; priorities
                // linear in the cost value likewise.)
                std::vector<int64_t> sorted(wcost);
                std::sort(sorted.begin(), sorted.end());
                const int pc[3] = {60, 84, 94};
                const int64_t q1 = sorted[(size_t)((nw - 1) * pc[0] / 100)], q2 = sorted[(size_t)((nw - 1) * pc[1] / 100)],
                              q3 = sorted[(size_t)((nw - 1) * pc[2] / 100)];
                for (int64_t w = 0; w < nw; ++w) {
                    const int64_t c = wcost[(size_t)w];
                    t.prio_of_wave[(size_t)w] = (uint8_t)((c > q1) + (c > q2) + (c > q3));
                }
            }
        }
        // Rank of a position inside its block: 0 for a row none of whose upstream rows shares its block, else one more
        // than the highest rank among those that do (the depth of the dependence graph the block induces).  Without the
        // short-timestep assumption a row needs its upstream rows at the SAME step, and the lanes of a wavefront advance
        // together: lane i therefore works `rank` steps behind (k_mc_flow, round k = step t0 + k - rank), so that in
        // any round a lane only needs what lanes of its block produced in EARLIER rounds -- by induction over the rounds
        // no wavefront of the block ever waits for another one in a cycle.  (A rank per wavefront would not do: two
        // wavefronts may feed each other within one step through different rows.)  Rows of earlier blocks are waited for.
        {
            std::vector<int32_t> lanes;
            const int64_t B = block_rows;
            for (int64_t w0 = t.nboundary; w0 < nseg; w0 += B) {
                const int64_t w1 = std::min<int64_t>(nseg, w0 + B);
                lanes.clear();
                for (int64_t p = w0; p < w1; ++p) lanes.push_back((int32_t)p);
                // (levels order the rows of a block topologically)
                std::sort(lanes.begin(), lanes.end(), [&](int32_t a, int32_t b) {
                    return t.level_of_row[t.row_of_pos[a]] < t.level_of_row[t.row_of_pos[b]];
                });
                for (const int32_t p : lanes) {
                    int32_t rk = 0;
                    for (int32_t k = t.up_ptr[p]; k < t.up_ptr[p + 1]; ++k) {
                        const int32_t u = t.up_idx[k];
                        if (u >= w0 && u < w1) rk = std::max(rk, t.rank_of_pos[u] + 1);
                    }
                    t.rank_of_pos[p] = rk;
                    t.maxrank = std::max(t.maxrank, rk);
                }
            }
        }
        return 0;
    }

    // depth-first rank from the outlets, walking upstream in the reference's order
    std::vector<int32_t> rank_order; // routed rows in preorder
    rank_order.reserve(nrouted);
    {
        std::vector<uint8_t> seen(nseg, 0);
        std::vector<int32_t> stack;
        for (int64_t o = 0; o < nseg; ++o) {
            if (is_b(o) || seen[o]) continue;
            // outlet: routed row that feeds no routed row
            if (down_ptr[o + 1] != down_ptr[o]) continue;
            stack.push_back((int32_t)o);
            seen[o] = 1;
            while (!stack.empty()) {
                const int32_t r = stack.back();
                stack.pop_back();
                rank_order.push_back(r);
                // push in reverse so the first-listed upstream is visited first
                for (int64_t k = up_ptr[r + 1] - 1; k >= up_ptr[r]; --k) {
                    const int32_t u = (int32_t)up_idx[k];
                    if (!is_b(u) && !seen[u]) {
                        seen[u] = 1;
                        stack.push_back(u);
                    }
                }
            }
        }
    }
    if ((int64_t)rank_order.size() != nrouted) { // cannot happen in a DAG
        err = "internal: depth-first walk missed rows";
        return -2;
    }

    // stable counting sort of the preorder by level
    t.lvl_ptr.assign(t.nlevels + 1, 0);
    for (int32_t r : rank_order) ++t.lvl_ptr[t.level_of_row[r] + 1];
    t.lvl_ptr[0] = (int32_t)t.nboundary;
    for (int32_t l = 0; l < t.nlevels; ++l) t.lvl_ptr[l + 1] += t.lvl_ptr[l];
    t.pos_of_row.assign(nseg, -1);
    t.row_of_pos.assign(nseg, -1);
    for (int64_t b = 0; b < t.nboundary; ++b) {
        t.pos_of_row[t.boundary_rows[b]] = (int32_t)b;
        t.row_of_pos[b] = t.boundary_rows[b];
    }
    {
        std::vector<int32_t> fill(t.lvl_ptr.begin(), t.lvl_ptr.end() - 1);
        for (int32_t r : rank_order) {
            const int32_t p = fill[t.level_of_row[r]]++;
            t.pos_of_row[r] = p;
            t.row_of_pos[p] = r;
        }
    }
    // (With a cost hint: by descending hint first -- rows of equal cost together, the costly blocks of a launch
    // first -- then as below.  Dealing the cost groups out in block-sized chunks so that every compute unit holds a
    // mix of costly and cheap blocks was measured too: 82.1 us per CONUS launch against 80.5 us for plain descending.)
    // Inside a level, order the rows by the position of the row they flow into (top level first, so that position
    // is known; ties -- the upstream rows of one junction -- keep the preorder, i.e. the reference's listing order;
    // outlets last).  The step kernel gathers `q[upstream of s]` for 64 consecutive s per wave: with this order the
    // upstream positions of consecutive rows are consecutive too (1-3 per row), so the gather reads whole lines
    // instead of one 4-byte element per 64-byte sector.  Results do not depend on the order inside a level.
    {
        std::vector<int64_t> key;
        std::vector<int32_t> idx, rows;
        for (int32_t l = t.nlevels - (cost_hint ? 1 : 2); l >= 0; --l) {
            const int32_t p0 = t.lvl_ptr[l], p1 = t.lvl_ptr[l + 1], m = p1 - p0;
            if (m < 2) continue;
            key.resize(m);
            idx.resize(m);
            rows.assign(t.row_of_pos.begin() + p0, t.row_of_pos.begin() + p1);
            for (int32_t i = 0; i < m; ++i) {
                const int32_t r = rows[i];
                const int64_t dpos = down_ptr[r + 1] > down_ptr[r] ? (int64_t)t.pos_of_row[down_idx[down_ptr[r]]] : ((int64_t)1 << 40) - 1;
                key[i] = ((int64_t)(cost_hint ? 255 - cost_hint[r] : 0) << 40) | dpos;
                idx[i] = i;
            }
            std::stable_sort(idx.begin(), idx.end(), [&](int32_t a, int32_t b) { return key[a] < key[b]; });
            for (int32_t i = 0; i < m; ++i) {
                t.row_of_pos[p0 + i] = rows[idx[i]];
                t.pos_of_row[rows[idx[i]]] = p0 + i;
            }
        }
    }

    // the rows below the leading wide levels in clusters (topology.hpp, cluster_rows)
    if (cluster_rows > 0 && t.nlevels > 0) {
        int32_t W = 0;
        if (wide_min_rows > 0)
            while (W < t.nlevels && W < wide_max_levels && (int64_t)(t.lvl_ptr[W + 1] - t.lvl_ptr[W]) >= wide_min_rows) ++W;
        t.lagk_of_pos.assign(nseg, 0);
        if (W < t.nlevels) {
            const int32_t B = cluster_rows;
            // cluster level and cluster of every row below the slices, in Kahn order (`queue`: every row after the rows
            // draining into it)
            std::vector<int32_t> cl(nseg, -1), root(nseg), size(nseg, 1);
            for (int64_t r = 0; r < nseg; ++r) root[r] = (int32_t)r;
            auto find = [&](int32_t x) {
                while (root[x] != x) {
                    root[x] = root[root[x]];
                    x = root[x];
                }
                return x;
            };
            std::vector<int32_t> tops;
            int32_t maxcl = 0;
            for (const int32_t r : queue) {
                if (t.level_of_row[r] < W) continue;
                int32_t m = -1;
                for (int64_t k = up_ptr[r]; k < up_ptr[r + 1]; ++k) {
                    const int32_t u = (int32_t)up_idx[k];
                    if (!is_b(u) && t.level_of_row[u] >= W) m = std::max(m, cl[u]);
                }
                // (a row fed by a boundary row, or marked late: at least cluster_late_lag tiles behind level 0)
                int32_t floor_c = 0;
                if (cluster_late_lag > W) {
                    bool late = is_late(r);
                    for (int64_t k = up_ptr[r]; k < up_ptr[r + 1] && !late; ++k) late = is_b(up_idx[k]);
                    if (late) floor_c = cluster_late_lag - W;
                }
                if (m < floor_c) { // only slices (or boundary rows, or nothing) drain into it -- or rows that run further ahead
                    cl[r] = floor_c; //   than this one may: a cluster of its own
                    maxcl = std::max(maxcl, floor_c);
                    continue;
                }
                tops.clear();
                int64_t tot = 1;
                for (int64_t k = up_ptr[r]; k < up_ptr[r + 1]; ++k) {
                    const int32_t u = (int32_t)up_idx[k];
                    if (is_b(u) || t.level_of_row[u] < W || cl[u] != m) continue;
                    const int32_t ru = find(u);
                    if (std::find(tops.begin(), tops.end(), ru) == tops.end()) {
                        tops.push_back(ru);
                        tot += size[ru];
                    }
                }
                if (tot <= B) {
                    cl[r] = m;
                    const int32_t r0 = tops[0];
                    for (size_t i = 1; i < tops.size(); ++i) root[tops[i]] = r0;
                    root[r] = r0;
                    size[r0] = (int32_t)tot;
                } else {
                    cl[r] = m + 1;
                    maxcl = std::max(maxcl, m + 1);
                }
            }
            const int32_t C = maxcl + 1;
            // the members of every cluster in Kahn order; clusters by level
            std::vector<int32_t> cid(nseg, -1);
            std::vector<std::vector<int32_t>> members;
            std::vector<int32_t> cl_of, cost_of;
            for (const int32_t r : queue) {
                if (t.level_of_row[r] < W) continue;
                const int32_t rt = find(r);
                if (cid[rt] < 0) {
                    cid[rt] = (int32_t)members.size();
                    members.emplace_back();
                    cl_of.push_back(cl[r]);
                    cost_of.push_back(0);
                }
                const int32_t c = cid[rt];
                members[(size_t)c].push_back(r);
                if (cost_hint) cost_of[(size_t)c] = std::max(cost_of[(size_t)c], (int32_t)cost_hint[r]);
            }
            std::vector<std::vector<int32_t>> by_cl((size_t)C);
            for (int32_t c = 0; c < (int32_t)members.size(); ++c) by_cl[(size_t)cl_of[(size_t)c]].push_back(c);
            int32_t p = t.lvl_ptr[W];
            t.cblk_ptr.clear();
            t.cblk_of_cl.assign((size_t)C + 1, 0);
            std::vector<std::vector<int32_t>> bins; // clusters of every block of the level being packed
            std::vector<int32_t> fill;
            for (int32_t c = 0; c < C; ++c) {
                std::vector<int32_t> &cs = by_cl[(size_t)c];
                // by descending cost, then by descending size: the costly clusters open the blocks, the small cheap ones fill
                // what they leave (a block's threads take its rows by cost class anyway: k_mc_ctile)
                std::stable_sort(cs.begin(), cs.end(), [&](int32_t a, int32_t b) {
                    if (cost_of[(size_t)a] != cost_of[(size_t)b]) return cost_of[(size_t)a] > cost_of[(size_t)b];
                    return members[(size_t)a].size() > members[(size_t)b].size();
                });
                bins.clear();
                fill.clear();
                // best fit: the fullest block that still takes the cluster (by_free[f]: blocks with f rows free)
                std::vector<std::vector<int32_t>> by_free((size_t)B + 1);
                for (const int32_t k : cs) {
                    const int32_t n = (int32_t)members[(size_t)k].size();
                    int32_t b = -1;
                    for (int32_t f = n; f <= B && b < 0; ++f)
                        if (!by_free[(size_t)f].empty()) {
                            b = by_free[(size_t)f].back();
                            by_free[(size_t)f].pop_back();
                        }
                    if (b < 0) {
                        b = (int32_t)bins.size();
                        bins.emplace_back();
                        fill.push_back(0);
                    }
                    bins[(size_t)b].push_back(k);
                    fill[(size_t)b] += n;
                    by_free[(size_t)(B - fill[(size_t)b])].push_back(b);
                }
                t.cblk_of_cl[(size_t)c] = (int32_t)t.cblk_ptr.size();
                for (const auto &bin : bins) {
                    t.cblk_ptr.push_back(p);
                    for (const int32_t k : bin)
                        for (const int32_t r : members[(size_t)k]) {
                            t.row_of_pos[p] = r;
                            t.pos_of_row[r] = p;
                            t.lagk_of_pos[p] = W + c;
                            ++p;
                        }
                }
            }
            t.cblk_of_cl[(size_t)C] = (int32_t)t.cblk_ptr.size();
            t.cblk_ptr.push_back(p);
            if (p != t.lvl_ptr[t.nlevels]) {
                err = "internal: the cluster order lost rows";
                return -2;
            }
            t.ncl = C;
            t.tail_from_level = W; // (the deeper levels are not contiguous slices any more)
        }
        for (int32_t l = 0; l < W; ++l)
            for (int32_t q = t.lvl_ptr[l]; q < t.lvl_ptr[l + 1]; ++q) t.lagk_of_pos[q] = l;
        t.cl_rows = cluster_rows;
        t.cl_from_level = W;
    } else
    // the rows below the leading wide levels: by descending cost across levels (topology.hpp, wide_min_rows)
    if (cost_hint && wide_min_rows > 0 && wide_max_levels > 0) {
        int32_t W = 0;
        while (W < t.nlevels && W < wide_max_levels && (int64_t)(t.lvl_ptr[W + 1] - t.lvl_ptr[W]) >= wide_min_rows) ++W;
        if (W > 0 && mid_min_rows > 0) // the second tier (topology.hpp): level slices as well
            for (int32_t m = 0; W < t.nlevels && m < mid_max_levels && (int64_t)(t.lvl_ptr[W + 1] - t.lvl_ptr[W]) >= mid_min_rows; ++m) ++W;
        if (W > 0 && W < t.nlevels) {
            const int32_t p0 = t.lvl_ptr[W], p1 = t.lvl_ptr[t.nlevels];
            std::vector<int32_t> rows(t.row_of_pos.begin() + p0, t.row_of_pos.begin() + p1);
            std::stable_sort(rows.begin(), rows.end(), [&](int32_t a, int32_t b) { return cost_hint[a] > cost_hint[b]; });
            for (int32_t i = 0; i < p1 - p0; ++i) {
                t.row_of_pos[p0 + i] = rows[i];
                t.pos_of_row[rows[i]] = p0 + i;
            }
            t.tail_from_level = W;
        }
    }

    csr_in_plan_order();
    return 0;
}

} // namespace trmc
