// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
//
// CPU restatement of the two host-side functions through which the reference's only
// alignment-dependent fixtures (tests/*.truth.fa, tests/io_fasta.rs) are produced:
//   poa_graph_to_fasta          /root/reference/src/io/fasta.rs:69-156  (+ fasta_aln_for_seq :19-67)
//   load_graph_from_fasta_msa   /root/reference/src/io/graph.rs:36-103
// Pinned by tests/io_fasta.rs:4-34 ("---AC" / "ACGT--", the empty sequence) and src/io/fasta.rs:165-209 ("ACG" / "A-G");
// transcribed in tests/test_oracle_msa.py.
#pragma once
#include <cstdint>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "graph.hpp"

namespace poa_oracle {

// fasta.rs:19-67.  NB the reference's `node_col.saturating_sub(1) - last_col` with `last_col` starting at 0: a row whose first
// node sits in column c > 0 gets c-1 leading gaps, one short (tests/io_fasta.rs:19 asserts exactly that: "---AC" vs "ACGT--").
inline std::string fasta_aln_for_seq(const Graph& g, const std::map<uint32_t, size_t>& node_to_column, uint32_t seq_id,
                                     uint32_t start_node) {
    std::string seq;
    bool have = true;
    uint32_t n = start_node;
    size_t last_col = 0;
    while (have) {
        auto it = node_to_column.find(n);
        if (it == node_to_column.end()) return std::string();  // empty sequence: start node is not in the column map
        size_t node_col = it->second;
        size_t gap_length = (node_col ? node_col - 1 : 0) - last_col;  // wraps like usize would panic: never negative here
        seq.append(gap_length, '-');
        seq.push_back((char)g.symbol[n]);
        have = false;
        uint32_t next = 0;
        for (uint32_t t : g.succ[n]) {  // graph.graph.edges(n): every out-edge, the LAST match wins (no break, :47-56)
            auto e = g.edge_seq_ids.find(((uint64_t)n << 32) | t);
            if (e == g.edge_seq_ids.end()) continue;
            if (std::find(e->second.begin(), e->second.end(), seq_id) != e->second.end()) { next = t; have = true; }
        }
        n = next;
        last_col = node_col;
    }
    if (!node_to_column.empty()) {
        size_t max_col = 0;
        for (auto& kv : node_to_column) max_col = std::max(max_col, kv.second);
        seq.append(max_col - last_col, '-');
    }
    return seq;
}

// fasta.rs:69-156
inline std::string poa_graph_to_fasta(const Graph& g) {
    std::map<uint32_t, size_t> node_to_column;
    struct Frame { uint32_t node; std::vector<uint32_t> succ; };
    std::vector<Frame> stack;
    stack.push_back({g.start, g.succ[g.start]});
    std::set<uint32_t> visited;
    std::vector<uint32_t> rev_postorder;
    while (!stack.empty()) {
        // next_valid_child: pop from the BACK of the successor vector (:83-93)
        bool found = false;
        uint32_t child = 0;
        {
            auto& it = stack.back().succ;
            while (!it.empty()) {
                uint32_t s = it.back(); it.pop_back();
                if (!visited.count(s)) { child = s; found = true; break; }
            }
        }
        if (found) {
            visited.insert(child);
            std::vector<uint32_t> successors = g.succ[child];
            for (uint32_t aln : g.aligned_nodes[child])
                if (!visited.count(aln)) {
                    visited.insert(aln);
                    successors.insert(successors.end(), g.succ[aln].begin(), g.succ[aln].end());
                }
            stack.push_back({child, std::move(successors)});
        } else {
            rev_postorder.push_back(stack.back().node);
            stack.pop_back();
        }
    }
    std::reverse(rev_postorder.begin(), rev_postorder.end());
    size_t curr_col = 0;
    for (uint32_t n : rev_postorder) {
        if (n == g.start || n == g.end) continue;
        if (!node_to_column.count(n)) {
            node_to_column[n] = curr_col;
            for (uint32_t a : g.aligned_nodes[n]) node_to_column[a] = curr_col;  // HashMap::insert overwrites
            curr_col += 1;
        }
    }
    std::string out;
    for (uint32_t seq_id = 0; seq_id < g.n_sequences; ++seq_id) {
        out += ">" + g.seq_names[seq_id] + "\n";
        std::string row = fasta_aln_for_seq(g, node_to_column, seq_id, g.seq_start_nodes[seq_id]);
        // noodles fasta::Writer wraps sequence lines at 80 columns; an empty sequence writes no sequence line
        for (size_t p = 0; p < row.size(); p += 80) out += row.substr(p, 80) + "\n";
    }
    return out;
}

// graph.rs:36-103: one node per distinct symbol per column, '-' skipped, edges in row order (weight 2), then post_process.
inline Graph load_graph_from_fasta_msa(const std::vector<std::string>& names, const std::vector<std::string>& rows) {
    Graph g = Graph::new_poa();
    std::vector<std::vector<uint32_t>> nodes_per_col;
    for (uint32_t seq_id = 0; seq_id < rows.size(); ++seq_id) {
        const std::string& chars = rows[seq_id];
        if (chars.size() > nodes_per_col.size()) nodes_per_col.resize(chars.size());
        bool have_prev = false;
        uint32_t prev = 0;
        for (size_t col = 0; col < chars.size(); ++col) {
            uint8_t c = (uint8_t)chars[col];
            if (c == '-') continue;
            uint32_t node = NONE32;
            for (uint32_t v : nodes_per_col[col]) if (g.symbol[v] == c) { node = v; break; }
            if (node == NONE32) {
                node = g.add_node(c);
                for (uint32_t other : nodes_per_col[col]) {
                    g.aligned_nodes[other].push_back(node);
                    g.aligned_nodes[node].push_back(other);
                }
                nodes_per_col[col].push_back(node);
            }
            if (have_prev) {
                g.poa_add_edge(prev, node, seq_id);
            } else {
                g.seq_names.push_back(names[seq_id]);
                g.seq_start_nodes.push_back(node);
                g.n_sequences++;
            }
            prev = node; have_prev = true;
        }
    }
    g.post_process();
    return g;
}

}  // namespace poa_oracle
