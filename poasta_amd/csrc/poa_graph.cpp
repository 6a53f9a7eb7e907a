#include "poa_graph.hpp"

#include <algorithm>
#include <deque>
#include <limits>

#include "../../include/poasta_amd.h"

namespace poa_amd {

int build_flat_graph(uint32_t n, uint32_t start, uint32_t end, const uint8_t* symbol,
                     const uint32_t* succ_off, const uint32_t* succ, const uint32_t* pred_off,
                     const uint32_t* pred, FlatGraph& g, std::string& err) {
    if (n < 2 || !symbol || !succ_off || !pred_off || start >= n || end >= n || start == end) {
        err = "poa_graph_create: need n >= 2, distinct start/end < n and non-null arrays";
        return POA_ERR_INVALID_ARG;
    }
    const uint32_t n_edges = succ_off[n];
    if (pred_off[n] != n_edges || succ_off[0] != 0 || pred_off[0] != 0) {
        err = "poa_graph_create: succ/pred CSR disagree on the edge count";
        return POA_ERR_INVALID_ARG;
    }
    if (n_edges && (!succ || !pred)) { err = "poa_graph_create: null adjacency"; return POA_ERR_INVALID_ARG; }
    for (uint32_t v = 0; v < n; ++v) {
        if (succ_off[v + 1] < succ_off[v] || pred_off[v + 1] < pred_off[v]) {
            err = "poa_graph_create: CSR offsets not monotone";
            return POA_ERR_INVALID_ARG;
        }
    }
    for (uint32_t e = 0; e < n_edges; ++e) {
        if (succ[e] >= n || pred[e] >= n) { err = "poa_graph_create: adjacency index out of range"; return POA_ERR_INVALID_ARG; }
    }
    g.n = n; g.start = start; g.end = end; g.n_real = n - 2;
    g.symbol.assign(symbol, symbol + n);
    g.succ_off.assign(succ_off, succ_off + n + 1);
    g.pred_off.assign(pred_off, pred_off + n + 1);
    g.succ.assign(succ, succ + n_edges);
    g.pred.assign(pred, pred + n_edges);

    // the two CSRs must describe the same edge multiset
    {
        std::vector<uint32_t> indeg(n, 0);
        for (uint32_t e = 0; e < n_edges; ++e) indeg[g.succ[e]]++;
        for (uint32_t v = 0; v < n; ++v) {
            if (indeg[v] != g.pred_off[v + 1] - g.pred_off[v]) {
                err = "poa_graph_create: predecessor lists do not mirror successor lists";
                return POA_ERR_INVALID_ARG;
            }
        }
    }
    if (n == 2) {
        // empty POA graph (only the sentinels; src/graphs/poa.rs:100-112): the aligner never runs on it
        // (PoastaAligner::align shortcut, src/aligner/mod.rs:124-142) — keep a trivial row table.
        g.node_row.assign(2, 0);
        g.node_row[end] = 1;
        g.rows.assign(2, RowMeta{});
        g.rows[0].node = start; g.rows[0].sym = g.symbol[start]; g.rows[0].flags = ROW_START | ROW_OPENI_ALWAYS;
        g.rows[1].node = end; g.rows[1].sym = g.symbol[end]; g.rows[1].flags = ROW_END | ROW_OPENI_NEVER;
        g.start_row = 0; g.end_row = 1;
        g.pred_rows.clear();
        return POA_OK;
    }
    if (g.pred_off[start + 1] != g.pred_off[start]) { err = "start node has predecessors"; return POA_ERR_NOT_A_DAG; }
    if (g.succ_off[end + 1] != g.succ_off[end]) { err = "end node has successors"; return POA_ERR_NOT_A_DAG; }

    // chain-following topological order: ready stack, successors pushed in reverse iteration
    // order so the first successor is emitted right after its parent when it is ready.
    std::vector<uint32_t> remaining(n);
    for (uint32_t v = 0; v < n; ++v) remaining[v] = g.pred_off[v + 1] - g.pred_off[v];
    std::vector<uint32_t> order;
    order.reserve(n);
    std::vector<uint32_t> ready;
    ready.push_back(start);
    for (uint32_t v = 0; v < n; ++v)
        if (v != start && remaining[v] == 0) { err = "node without predecessors other than start"; return POA_ERR_NOT_A_DAG; }
    while (!ready.empty()) {
        uint32_t v = ready.back();
        ready.pop_back();
        order.push_back(v);
        for (uint32_t e = g.succ_off[v + 1]; e-- > g.succ_off[v];) {
            uint32_t s = g.succ[e];
            if (s == end) {  // keep the end node for last
                --remaining[s];
                continue;
            }
            if (--remaining[s] == 0) ready.push_back(s);
        }
    }
    if (order.size() != n - 1 || remaining[end] != 0) {
        err = "graph has a cycle, or nodes unreachable from start / not reaching end";
        return POA_ERR_NOT_A_DAG;
    }
    order.push_back(end);
    for (uint32_t v = 0; v < n; ++v)
        if (v != end && g.succ_off[v + 1] == g.succ_off[v]) { err = "node without successors other than end"; return POA_ERR_NOT_A_DAG; }

    g.node_row.assign(n, 0);
    for (uint32_t r = 0; r < n; ++r) g.node_row[order[r]] = r;
    g.start_row = g.node_row[start];
    g.end_row = g.node_row[end];
    g.rows.assign(n, RowMeta{});
    g.pred_rows.clear();
    g.pred_rows.reserve(n_edges);
    g.max_indegree = 0;
    for (uint32_t r = 0; r < n; ++r) {
        uint32_t v = order[r];
        RowMeta& m = g.rows[r];
        m.node = v;
        m.sym = g.symbol[v];
        m.pred_begin = (uint32_t)g.pred_rows.size();
        m.pred_count = g.pred_off[v + 1] - g.pred_off[v];
        g.max_indegree = std::max(g.max_indegree, m.pred_count);
        for (uint32_t e = g.pred_off[v]; e < g.pred_off[v + 1]; ++e) {
            uint32_t pr = g.node_row[g.pred[e]];
            if (pr >= r) { err = "internal: predecessor not before successor"; return POA_ERR_NOT_A_DAG; }
            g.pred_rows.push_back(pr);
        }
        m.flags = 0;
        for (uint32_t pe = 0; pe < m.pred_count; ++pe)
            if (r - g.pred_rows[m.pred_begin + pe] > ROW_NEAR) m.flags |= ROW_FAR_PRED;
        if (m.pred_count == 1 && g.pred_rows[m.pred_begin] + 1 == r) m.flags |= ROW_CHAIN;
        if (v == start) m.flags |= ROW_START;
        if (v == end) m.flags |= ROW_END;
        bool has_end_child = false, has_real = false, multi = false;
        uint8_t cs = 0;
        for (uint32_t e = g.succ_off[v]; e < g.succ_off[v + 1]; ++e) {
            uint32_t c = g.succ[e];
            if (c == end) { has_end_child = true; continue; }
            if (!has_real) { cs = g.symbol[c]; has_real = true; }
            else if (g.symbol[c] != cs) multi = true;
        }
        m.child_sym = cs;
        if (has_end_child || multi) m.flags |= ROW_OPENI_ALWAYS;
        else if (!has_real) m.flags |= ROW_OPENI_NEVER;
        {
            auto idx = [](uint8_t c) -> uint8_t { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 8; };
            const uint8_t ci = (m.flags & ROW_OPENI_ALWAYS) ? 4 : idx(m.child_sym);
            m.sym_idx = (uint8_t)(idx(m.sym) | (ci << 4));
        }
    }
    // sibling rows: a non-chain row whose predecessor SET equals the previous row's
    for (uint32_t r = 1; r < n; ++r) {
        RowMeta& m = g.rows[r];
        const RowMeta& p = g.rows[r - 1];
        if ((m.flags | p.flags) & (ROW_CHAIN | ROW_END | ROW_START)) continue;
        if (m.pred_count == 0 || m.pred_count != p.pred_count) continue;
        std::vector<uint32_t> a(g.pred_rows.begin() + m.pred_begin, g.pred_rows.begin() + m.pred_begin + m.pred_count);
        std::vector<uint32_t> b(g.pred_rows.begin() + p.pred_begin, g.pred_rows.begin() + p.pred_begin + p.pred_count);
        std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
        if (a == b) m.flags |= ROW_SAME_PREDS;
    }
    // D rows that must stay in memory for the compact plane layout: a row whose D some successor reads back
    // (forward pass: non-adjacent predecessor; traceback: any predecessor of a non-chain row), plus the end row.
    for (uint32_t r = 0; r < n; ++r) {
        const RowMeta& m = g.rows[r];
        if (m.flags & ROW_END) g.rows[r].flags |= ROW_STORE_D;
        if (m.flags & ROW_CHAIN) continue;
        for (uint32_t pe = 0; pe < m.pred_count; ++pe) g.rows[g.pred_rows[m.pred_begin + pe]].flags |= ROW_STORE_D;
    }
    // compact layout: the kept D rows are stored densely, row r at slot d_slot[r]
    // (a graph that keeps most of its D rows anyway — bubble-rich: every row has several predecessors — stores them by row:
    // d_slot stays empty, the kernels skip the lookup)
    {
        uint32_t kept = 0;
        for (uint32_t r = 0; r < n; ++r) kept += (g.rows[r].flags & ROW_STORE_D) ? 1u : 0u;
        g.d_slot.clear(); g.pred_dslot.clear();
        g.n_store_d = n;
        if (2ull * kept <= n) {
            g.d_slot.assign(n, 0xFFFFFFFFu);
            g.n_store_d = 0;
            for (uint32_t r = 0; r < n; ++r)
                if (g.rows[r].flags & ROW_STORE_D) g.d_slot[r] = g.n_store_d++;
            g.pred_dslot.assign(g.pred_rows.size(), 0xFFFFFFFFu);
            for (size_t k = 0; k < g.pred_rows.size(); ++k) g.pred_dslot[k] = g.d_slot[g.pred_rows[k]];
        }
    }
    // shortest start -> end path, in real nodes (bounds the optimal score from above: see poa_batch_run_ex)
    {
        std::vector<uint32_t> dist(n, 0xFFFFFFFFu);
        dist[g.start_row] = 0;
        for (uint32_t r = 0; r < n; ++r) {
            const RowMeta& m = g.rows[r];
            for (uint32_t pe = 0; pe < m.pred_count; ++pe) {
                const uint32_t d = dist[g.pred_rows[m.pred_begin + pe]];
                if (d != 0xFFFFFFFFu && d + 1 < dist[r]) dist[r] = d + 1;
            }
        }
        g.min_path_nodes = (dist[g.end_row] == 0xFFFFFFFFu || dist[g.end_row] == 0) ? 0 : dist[g.end_row] - 1;
        // depth potential (see FlatGraph::row_depth): d(v) = 1 + min over predecessors; the end row consumes no query
        // symbol, so it sits at the depth of its shallowest predecessor
        g.row_depth.assign(n, 0);
        for (uint32_t r = 0; r < n; ++r) g.row_depth[r] = dist[r] == 0xFFFFFFFFu ? 0u : dist[r];
        if (dist[g.end_row] != 0xFFFFFFFFu && dist[g.end_row] > 0) g.row_depth[g.end_row] = dist[g.end_row] - 1;
        g.pred_k.assign(g.pred_rows.size(), 0);
        for (uint32_t r = 0; r < n; ++r) {
            const RowMeta& m = g.rows[r];
            if (dist[r] == 0xFFFFFFFFu) continue;
            for (uint32_t pe = 0; pe < m.pred_count; ++pe) {
                const uint32_t pr = g.pred_rows[m.pred_begin + pe];
                if (dist[pr] == 0xFFFFFFFFu) continue;  // an unreachable predecessor holds INF everywhere
                g.pred_k[m.pred_begin + pe] = g.row_depth[pr] + ((m.flags & ROW_END) ? 0u : 1u) - g.row_depth[r];
            }
        }
    }
    // shortest path to the end row, in edges (what the reference's dist_to_end BFS finds, gap_affine.rs:91-119)
    g.sp_to_end.assign(n, 0xFFFFFFFFu);
    g.sp_to_end[g.end_row] = 0;
    for (uint32_t r = n; r-- > 0;) {
        const uint32_t d = g.sp_to_end[r];
        if (d == 0xFFFFFFFFu) continue;
        const RowMeta& m = g.rows[r];
        for (uint32_t pe = 0; pe < m.pred_count; ++pe) {
            uint32_t& pd = g.sp_to_end[g.pred_rows[m.pred_begin + pe]];
            if (d + 1 < pd) pd = d + 1;
        }
    }
    return POA_OK;
}


// ---------------------------------------------------------------------------------------------
// The reference's superbubble preprocessing, on node ids, results stored per row.
int build_bubble_index(FlatGraph& g, std::string& err) {
    if (g.bubbles_built) return POA_OK;
    const uint32_t n = g.n;
    auto succ_of = [&](uint32_t v) { return std::make_pair(g.succ_off[v], g.succ_off[v + 1]); };
    // successors as rows
    g.succ_row_off.assign(n + 1, 0);
    g.succ_rows.clear();
    for (uint32_t r = 0; r < n; ++r) {
        const uint32_t v = g.rows[r].node;
        for (uint32_t e = g.succ_off[v]; e < g.succ_off[v + 1]; ++e) g.succ_rows.push_back(g.node_row[g.succ[e]]);
        g.succ_row_off[r + 1] = (uint32_t)g.succ_rows.size();
    }
    g.dist_min.assign(n, 0); g.dist_max.assign(n, 0); g.is_exit.assign(n, 0);
    g.exit_idx.assign(n, 0xFFFFFFFFu); g.n_exit = 0;
    g.nbm_off.assign(n + 1, 0); g.nbm.clear();
    if (n == 2) { g.bubbles_built = true; return POA_OK; }

    // rev_postorder_nodes (tools.rs:5-37): DFS from start following successors in iteration order
    std::vector<uint32_t> inv_rpo;
    {
        std::vector<std::pair<uint32_t, uint32_t>> stack;  // (node, next successor edge)
        std::vector<uint8_t> visited(n, 0);
        stack.push_back({g.start, g.succ_off[g.start]});
        while (!stack.empty()) {
            auto& top = stack.back();
            bool pushed = false;
            while (top.second < g.succ_off[top.first + 1]) {
                const uint32_t child = g.succ[top.second++];
                if (!visited[child]) { visited[child] = 1; stack.push_back({child, g.succ_off[child]}); pushed = true; break; }
            }
            if (!pushed) { inv_rpo.push_back(stack.back().first); stack.pop_back(); }
        }
        std::reverse(inv_rpo.begin(), inv_rpo.end());
    }
    if (inv_rpo.size() != n) { err = "bubble index: node unreachable from start"; return POA_ERR_NOT_A_DAG; }
    std::vector<int64_t> rpo(n, 0);
    for (uint32_t i = 0; i < n; ++i) rpo[inv_rpo[i]] = i;
    // finder.rs:37-57
    std::vector<int64_t> out_parent(n, -1), out_child(n, std::numeric_limits<int64_t>::max());
    for (uint32_t v = 0; v < n; ++v) {
        bool any = false; int64_t mn = -1;
        for (uint32_t e = g.pred_off[v]; e < g.pred_off[v + 1]; ++e) { const int64_t r = rpo[g.pred[e]]; if (!any || r < mn) mn = r; any = true; }
        out_parent[v] = any ? mn : -1;
        any = false; int64_t mx = 0;
        for (uint32_t e = g.succ_off[v]; e < g.succ_off[v + 1]; ++e) { const int64_t r = rpo[g.succ[e]]; if (!any || r > mx) mx = r; any = true; }
        out_child[v] = any ? mx : std::numeric_limits<int64_t>::max();
    }
    // finder.rs:115-178 run to exhaustion
    std::vector<uint8_t> is_entrance(n, 0), is_exit_node(n, 0);
    {
        std::vector<int64_t> opm(n, 0);
        std::vector<uint8_t> has(n, 0);
        bool bad = false;
        auto get = [&](uint32_t k) -> int64_t { if (!has[k]) bad = true; return opm[k]; };
        auto set = [&](uint32_t k, int64_t v) { opm[k] = v; has[k] = 1; };
        std::vector<uint32_t> stack;
        bool have = false; uint32_t cand = 0;
        auto pop_cand = [&]() { if (stack.empty()) have = false; else { cand = stack.back(); stack.pop_back(); have = true; } };
        for (uint32_t curr = n; curr-- > 0;) {
            const uint32_t nn = inv_rpo[curr];
            const int64_t furthest = out_child[nn];
            if (furthest == (int64_t)curr + 1) {
                if (have) stack.push_back(cand);
                cand = inv_rpo[curr + 1]; have = true;
            } else {
                while (have) {
                    const uint32_t c0 = cand;
                    if (furthest <= rpo[c0]) break;
                    pop_cand();
                    if (have) set(cand, std::min(get(c0), get(cand)));
                }
            }
            if (have) {
                const uint32_t c0 = cand;
                if ((uint64_t)get(c0) == (uint64_t)curr) {
                    is_entrance[nn] = 1; is_exit_node[c0] = 1;
                    pop_cand();
                    if (have) set(cand, std::min(get(c0), get(cand)));
                }
            }
            set(nn, out_parent[nn]);
            if (have) set(cand, std::min(get(nn), get(cand)));
        }
        if (bad) { err = "bubble index: superbubble finder read an unset entry (the reference would panic)"; return POA_ERR_UNSUPPORTED; }
    }
    // index.rs:73-131: backward BFS from end with the per-path bubble stack
    std::vector<std::vector<FlatGraph::NodeBubble>> nbm(n);  // exit stored as NODE id here
    std::vector<uint32_t> dmin(n, 0), dmax(n, 0);
    {
        using BStack = std::vector<std::pair<uint32_t, uint32_t>>;  // (dist, exit node)
        struct Item { uint32_t node, dist; BStack bs; };
        std::deque<Item> queue;
        BStack init;
        if (is_exit_node[g.end]) init.push_back({0u, g.end});
        queue.push_back({g.end, 0u, init});
        std::vector<uint8_t> visited(n, 0);
        visited[g.end] = 1;
        while (!queue.empty()) {
            Item it = std::move(queue.front());
            queue.pop_front();
            for (auto& b : it.bs) nbm[it.node].push_back({b.second, it.dist - b.first, 0});
            dmin[it.node] = it.dist;
            for (uint32_t e = g.pred_off[it.node]; e < g.pred_off[it.node + 1]; ++e) {
                const uint32_t p = g.pred[e];
                if (visited[p]) continue;
                const uint32_t nd = it.dist + 1;
                BStack nb = it.bs;
                if (is_entrance[p]) {
                    if (nb.empty()) { err = "bubble index: empty bubble stack (the reference would panic)"; return POA_ERR_UNSUPPORTED; }
                    auto top = nb.back(); nb.pop_back();
                    nbm[p].push_back({top.second, nd - top.first, 0});
                }
                if (is_exit_node[p]) nb.push_back({nd, p});
                visited[p] = 1;
                queue.push_back({p, nd, std::move(nb)});
            }
        }
    }
    // index.rs:135-148: longest distance in post-order
    for (uint32_t i = n; i-- > 0;) {
        const uint32_t v = inv_rpo[i];
        uint32_t mx = 0;
        for (uint32_t e = g.succ_off[v]; e < g.succ_off[v + 1]; ++e) mx = std::max(mx, dmax[g.succ[e]] + 1);
        dmax[v] = mx;
        for (auto& b : nbm[v]) b.max_dist = mx - dmax[b.exit_row];
    }
    (void)succ_of;
    for (uint32_t r = 0; r < n; ++r) {
        const uint32_t v = g.rows[r].node;
        g.dist_min[r] = dmin[v]; g.dist_max[r] = dmax[v]; g.is_exit[r] = is_exit_node[v];
        g.nbm_off[r] = (uint32_t)g.nbm.size();
        for (auto& b : nbm[v]) g.nbm.push_back({g.node_row[b.exit_row], b.min_dist, b.max_dist});
    }
    g.nbm_off[n] = (uint32_t)g.nbm.size();
    g.exit_idx.assign(n, 0xFFFFFFFFu);
    g.n_exit = 0;
    for (uint32_t r = 0; r < n; ++r) if (g.is_exit[r]) g.exit_idx[r] = g.n_exit++;
    // the per-row records of the replay's step
    g.row_rec.clear();
    if (n < 0xFFFFu) {
        g.row_rec.resize(n);
        for (uint32_t r = 0; r < n; ++r) {
            FlatGraph::RowRec rr{};
            rr.c0 = rr.c1 = rr.e0 = rr.e1 = 0xFFFFu; rr.x0 = rr.x1 = 0xFFFFu;
            const uint32_t s0 = g.succ_row_off[r], ns = g.succ_row_off[r + 1] - s0;
            uint8_t fl = r == g.end_row ? FlatGraph::RR_END : 0;
            rr.sym = g.rows[r].sym;
            if (ns >= 1) { rr.c0 = (uint16_t)g.succ_rows[s0]; rr.sym0 = g.rows[g.succ_rows[s0]].sym; }
            if (ns >= 2) { rr.c1 = (uint16_t)g.succ_rows[s0 + 1]; rr.sym1 = g.rows[g.succ_rows[s0 + 1]].sym; }
            if ((ns == 1 || ns == 2) && g.succ_rows[s0] != g.end_row && (ns == 1 || g.succ_rows[s0 + 1] != g.end_row)) fl |= FlatGraph::RR_SUCC_OK;
            if (ns == 2) fl |= FlatGraph::RR_HAS_C1;
            bool pok = true; uint32_t nb = 0;
            for (uint32_t k = g.nbm_off[r]; k < g.nbm_off[r + 1]; ++k) {
                const FlatGraph::NodeBubble& b = g.nbm[k];
                if (b.exit_row == r) continue;                  // reached.rs:56-58: the bubble the row exits is never tested
                const uint32_t md = g.dist_min[b.exit_row] ? g.dist_min[b.exit_row] - 1 : 0;
                if (nb == 2 || b.max_dist > 255 || b.min_dist > b.max_dist || b.max_dist - b.min_dist > 2 || md > 0xFFFFu) { pok = false; break; }
                if (nb == 0) { rr.e0 = (uint16_t)b.exit_row; rr.x0 = (uint16_t)g.exit_idx[b.exit_row]; rr.d0min = (uint8_t)b.min_dist; rr.d0max = (uint8_t)b.max_dist; rr.mde0 = (uint16_t)md; }
                else { rr.e1 = (uint16_t)b.exit_row; rr.x1 = (uint16_t)g.exit_idx[b.exit_row]; rr.d1min = (uint8_t)b.min_dist; rr.d1max = (uint8_t)b.max_dist; rr.mde1 = (uint16_t)md; }
                nb += 1;
            }
            if (pok) fl |= FlatGraph::RR_PROBE_OK;
            const uint32_t dn = g.dist_min[r] ? g.dist_min[r] - 1 : 0, dx = g.dist_max[r] ? g.dist_max[r] - 1 : 0;
            rr.dmin = (uint16_t)dn; rr.dmax = (uint16_t)dx;   // (< 65535: at most n - 1)
            rr.flags = fl;
            g.row_rec[r] = rr;
        }
    }
    g.bubbles_built = true;
    return POA_OK;
}

}  // namespace poa_amd
