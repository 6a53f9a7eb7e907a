// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
//
// CPU restatement of the graph surface the reference aligner reads
// (`AlignableRefGraph`, /root/reference/src/graphs/mod.rs:23-53) together with
// the two graph types that implement it in the reference:
//   * POAGraph        /root/reference/src/graphs/poa.rs:85-471
//   * MockGraph       /root/reference/src/graphs/mock.rs:14-90  (test fixture)
//
// The reference stores both in petgraph 0.6.5 (Cargo.lock:528), whose source is
// NOT under /root/reference.  What is restated here is petgraph's published
// adjacency behaviour: every node keeps an outgoing and
// an incoming intrusive edge list, `add_edge` PREPENDS to both, so
// `neighbors()` / `neighbors_directed(Incoming)` yield the most recently added
// edge first; `remove_edge` unlinks an edge and keeps the order of the rest.
// The reference tests that pin this are src/graphs/tools.rs:53-69 and
// src/bubbles/index.rs:231-318 (both transcribed in tests/test_oracle_kat.py).
#pragma once
#include <algorithm>
#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace poa_oracle {

constexpr uint32_t NONE32 = 0xFFFFFFFFu;

struct AlignedPair {
    uint32_t rpos;  // node index or NONE32   (alignment.rs:4-13)
    uint32_t qpos;  // query position or NONE32
};

// Graph with petgraph-like adjacency iteration order.
struct Graph {
    // symbol per node.  0 == "no symbol" (MockGraph: never equal to anything).
    std::vector<uint8_t> symbol;
    // adjacency in ITERATION order (index 0 == newest edge == first yielded).
    std::vector<std::vector<uint32_t>> succ, pred;
    // POAGraph only: nodes aligned to each other (poa.rs:44).
    std::vector<std::vector<uint32_t>> aligned_nodes;
    uint32_t start = 0, end = 0;
    // POAGraph::is_symbol_equal: the end node equals every symbol (poa.rs:463-465).
    bool end_matches_all = true;
    // POAGraph bookkeeping
    bool is_poa = false;
    size_t n_sequences = 0;
    std::vector<uint32_t> seq_start_nodes;  // Sequence(name, start_node), poa.rs:21
    std::vector<std::string> seq_names;
    std::vector<uint32_t> topo;  // topological_sorted (node ids in order)
    // POAEdgeData::sequence_ids (poa.rs:58-62), keyed by (source << 32 | target); start/end edges carry none (poa.rs:73-78)
    std::map<uint64_t, std::vector<uint32_t>> edge_seq_ids;

    size_t node_count_with_start_and_end() const { return symbol.size(); }

    uint32_t add_node(uint8_t sym) {
        symbol.push_back(sym);
        succ.emplace_back();
        pred.emplace_back();
        aligned_nodes.emplace_back();
        return (uint32_t)symbol.size() - 1;
    }
    bool has_edge(uint32_t s, uint32_t t) const {
        for (uint32_t v : succ[s]) if (v == t) return true;
        return false;
    }
    // petgraph add_edge: prepend to both lists.
    void raw_add_edge(uint32_t s, uint32_t t) {
        succ[s].insert(succ[s].begin(), t);
        pred[t].insert(pred[t].begin(), s);
    }
    // POAGraph::add_edge (poa.rs:118-134): an existing edge keeps its position and gets the sequence id appended.
    void poa_add_edge(uint32_t s, uint32_t t, uint32_t sequence_id) {
        if (!has_edge(s, t)) raw_add_edge(s, t);
        edge_seq_ids[((uint64_t)s << 32) | t].push_back(sequence_id);
    }
    void remove_edge(uint32_t s, uint32_t t) {
        edge_seq_ids.erase(((uint64_t)s << 32) | t);
        auto& a = succ[s];
        a.erase(std::find(a.begin(), a.end(), t));
        auto& b = pred[t];
        b.erase(std::find(b.begin(), b.end(), s));
    }

    bool is_symbol_equal(uint32_t node, uint8_t sym) const {
        if (end_matches_all && node == end) return true;
        return symbol[node] != 0 && symbol[node] == sym;
    }

    // ---- POAGraph (poa.rs:100-112) ------------------------------------
    static Graph new_poa() {
        Graph g;
        g.is_poa = true;
        g.start = g.add_node('#');
        g.end = g.add_node('$');
        return g;
    }
    size_t node_count() const { return symbol.size() - (is_poa ? 2 : 0); }

    // poa.rs:136-169
    bool add_nodes_for_sequence(const uint8_t* seq, size_t start_pos, size_t end_pos,
                                uint32_t& first, uint32_t& last) {
        if (start_pos == end_pos) return false;
        bool have_first = false, have_prev = false;
        uint32_t prev = 0;
        for (size_t pos = start_pos; pos < end_pos; ++pos) {
            uint32_t curr = add_node(seq[pos]);
            if (!have_first) { first = curr; have_first = true; }
            if (have_prev) poa_add_edge(prev, curr, (uint32_t)n_sequences);
            prev = curr; have_prev = true;
        }
        last = prev;
        return true;
    }

    // poa.rs:171-321.  alignment == nullptr  <=>  alignment_opt == None.
    // Returns 0 on success, 1 == PoastaError::InvalidAlignment.
    int add_alignment(const std::string& name, const uint8_t* seq, size_t len,
                      const std::vector<AlignedPair>* alignment) {
        if (!alignment) {
            if (len == 0) {
                seq_start_nodes.push_back(start); seq_names.push_back(name); n_sequences++;
                post_process();
                return 0;
            }
            uint32_t f = 0, l = 0;
            add_nodes_for_sequence(seq, 0, len, f, l);
            seq_start_nodes.push_back(f); seq_names.push_back(name); n_sequences++;
            post_process();
            return 0;
        }
        std::vector<size_t> valid_ix;
        for (const auto& e : *alignment)
            if (e.qpos != NONE32 && e.qpos < len) valid_ix.push_back(e.qpos);
        if (valid_ix.empty()) {
            if (len == 0) {
                seq_start_nodes.push_back(start); seq_names.push_back(name); n_sequences++;
                post_process();
                return 0;
            }
            return 1;
        }
        size_t first = valid_ix.front(), last = valid_ix.back();
        uint32_t b1 = 0, b2 = 0, e1 = 0, e2 = 0;
        bool have_begin = add_nodes_for_sequence(seq, 0, first, b1, b2);
        bool have_prev = have_begin;
        uint32_t prev = b2;
        uint32_t begin_first = b1;
        bool have_end = add_nodes_for_sequence(seq, last + 1, len, e1, e2);

        for (const auto& ap : *alignment) {
            if (ap.qpos == NONE32) continue;
            size_t q = ap.qpos;
            uint8_t qsym = seq[q];
            uint32_t curr = NONE32;
            if (ap.rpos != NONE32) {
                uint32_t r = ap.rpos;
                if (symbol[r] == qsym) {
                    curr = r;
                } else {
                    for (uint32_t other : aligned_nodes[r])
                        if (symbol[other] == qsym) { curr = other; break; }
                    if (curr == NONE32) {
                        uint32_t nn = add_node(qsym);
                        curr = nn;
                        std::vector<uint32_t> others = aligned_nodes[r];
                        for (uint32_t o : others) {
                            aligned_nodes[o].push_back(nn);
                            aligned_nodes[nn].push_back(o);
                        }
                        aligned_nodes[r].push_back(nn);
                        aligned_nodes[nn].push_back(r);
                    }
                }
            } else {
                curr = add_node(qsym);
            }
            if (!have_begin) { begin_first = curr; have_begin = true; }
            if (have_prev) poa_add_edge(prev, curr, (uint32_t)n_sequences);
            prev = curr; have_prev = true;
        }
        if (have_end) poa_add_edge(prev, e1, (uint32_t)n_sequences);
        seq_start_nodes.push_back(begin_first); seq_names.push_back(name); n_sequences++;
        post_process();
        return 0;
    }

    // poa.rs:323-363
    void post_process() {
        while (!succ[start].empty()) remove_edge(start, succ[start][0]);
        while (!pred[end].empty()) remove_edge(pred[end][0], end);
        uint32_t n = (uint32_t)symbol.size();
        for (uint32_t v = 0; v < n; ++v)
            if (v != start && v != end && pred[v].empty()) raw_add_edge(start, v);
        for (uint32_t v = 0; v < n; ++v)
            if (v != end && v != start && succ[v].empty()) raw_add_edge(v, end);
        compute_topo();
    }

    // Stand-in for petgraph::algo::toposort (poa.rs:360, mock.rs:79): DFS finish
    // order seeded from the highest node index downwards.  Only the table layout
    // of the visited-score storage depends on it (gap_affine.rs:463,:470) —
    // results do not — so any valid topological order is acceptable here.
    void compute_topo() {
        uint32_t n = (uint32_t)symbol.size();
        std::vector<uint8_t> state(n, 0);
        std::vector<uint32_t> finished;
        finished.reserve(n);
        std::vector<std::pair<uint32_t, size_t>> stack;
        for (uint32_t s = n; s-- > 0;) {
            if (state[s]) continue;
            // roots only: petgraph starts its DFS from nodes without incoming edges
            if (!pred[s].empty()) continue;
            stack.push_back({s, 0}); state[s] = 1;
            while (!stack.empty()) {
                auto& top = stack.back();
                if (top.second < succ[top.first].size()) {
                    uint32_t c = succ[top.first][top.second++];
                    if (!state[c]) { state[c] = 1; stack.push_back({c, 0}); }
                } else {
                    finished.push_back(top.first);
                    stack.pop_back();
                }
            }
        }
        if (finished.size() != n) throw std::runtime_error("graph has a cycle or unreachable cycle component");
        topo.assign(finished.rbegin(), finished.rend());
    }
    // get_node_ordering (poa.rs:365-372, :468-470): rank per node index.
    std::vector<uint32_t> node_ranks() const {
        std::vector<uint32_t> ranks(symbol.size(), 0);
        for (size_t r = 0; r < topo.size(); ++r) ranks[topo[r]] = (uint32_t)r;
        return ranks;
    }

    // ---- generic construction from CSR in trait-iteration order ----------
    static Graph from_csr(uint32_t n, uint32_t start, uint32_t end, const uint8_t* sym,
                          const uint32_t* succ_off, const uint32_t* succ_ix,
                          const uint32_t* pred_off, const uint32_t* pred_ix,
                          bool end_matches_all) {
        Graph g;
        g.symbol.assign(sym, sym + n);
        g.succ.resize(n); g.pred.resize(n); g.aligned_nodes.resize(n);
        for (uint32_t v = 0; v < n; ++v) {
            g.succ[v].assign(succ_ix + succ_off[v], succ_ix + succ_off[v + 1]);
            g.pred[v].assign(pred_ix + pred_off[v], pred_ix + pred_off[v + 1]);
        }
        g.start = start; g.end = end; g.end_matches_all = end_matches_all;
        g.compute_topo();
        return g;
    }

    // MockGraph (mock.rs:14-90): start = first node, end = last node, no symbols
    // unless a symbol overlay is given (dfa.rs:276-348 SymbolMockGraph: plain
    // equality, the end node is NOT special).
    static Graph new_mock(uint32_t n_nodes) {
        Graph g;
        g.end_matches_all = false;
        for (uint32_t i = 0; i < n_nodes; ++i) g.add_node(0);
        g.start = 0; g.end = n_nodes ? n_nodes - 1 : 0;
        return g;
    }
};

// mock.rs:92-125
inline Graph create_test_graph1() {
    Graph g = Graph::new_mock(9);
    const int edges[][2] = {{1,2},{2,3},{3,4},{4,5},{5,6},{3,7},{7,8},{8,9}};
    for (auto& e : edges) g.raw_add_edge(e[0] - 1, e[1] - 1);
    uint32_t end_node = g.add_node(0);
    for (uint32_t n = 0; n < g.symbol.size(); ++n)
        if (n != end_node && g.succ[n].empty()) g.raw_add_edge(n, end_node);
    g.start = 0; g.end = end_node;
    g.compute_topo();
    return g;
}

// mock.rs:127-165
inline Graph create_test_graph2() {
    Graph g = Graph::new_mock(15);
    const int edges[][2] = {{1,2},{1,3},{2,3},{3,4},{3,5},{3,11},{4,8},{5,6},{5,9},{6,7},{6,10},
                            {7,8},{8,13},{8,15},{9,10},{10,7},{11,12},{12,8},{13,14},{13,15},{14,15}};
    for (auto& e : edges) g.raw_add_edge(e[0] - 1, e[1] - 1);
    g.start = 0; g.end = 14;
    g.compute_topo();
    return g;
}

}  // namespace poa_oracle
