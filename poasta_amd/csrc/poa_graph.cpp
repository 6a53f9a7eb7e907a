#include "poa_graph.hpp"

#include <algorithm>

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
    }
    return POA_OK;
}

}  // namespace poa_amd
