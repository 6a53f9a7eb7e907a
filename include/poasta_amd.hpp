// poasta_amd.hpp — C++17 host-side mirror of the reference's aligner interface over the C ABI
// (include/poasta_amd.h).  Header-only.  The reference is Rust (no toolchain in this image), so this
// is the compiled-language host layer: same names, argument meaning and error behaviour as
//   poasta::aligner::{PoastaAligner, AlignedPair, AstarResult}      /root/reference/src/aligner/mod.rs:40-146, astar.rs:81-90
//   poasta::aligner::scoring::{GapAffine, AlignmentType}            src/aligner/scoring/gap_affine.rs:20-30, scoring/mod.rs:50-62
//   poasta::aligner::config::{AffineMinGapCost, AffineDijkstra}     src/aligner/config.rs:49,:104
//   poasta::graphs::poa::POAGraph (construction, graph update)      src/graphs/poa.rs:85-363, :384-471
//   poasta::io::{load_graph_from_fasta_msa, poa_graph_to_fasta}     src/io/graph.rs:36-103, src/io/fasta.rs:19-156
//   poasta::io::{load_graph_from_gfa, GraphSegments}                src/io/graph.rs:103-227, src/io/gfa.rs:233-360
//   poasta::io::gaf::{alignment_to_gaf, GAFRecord, NodeSegmentResolver}   src/io/gaf.rs:11-304
// Where the reference panics, this layer throws poasta::PoastaError.  All alignment work happens in the
// gfx950 library; nothing here computes an alignment on the CPU.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cctype>
#include <fstream>
#include <map>
#include <set>
#include <memory>
#include <optional>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "poasta_amd.h"

namespace poasta {

struct PoastaError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// =====================================================================================================
namespace graphs { using NodeIndex = uint32_t; }

namespace aligner {
struct AlignedPair {  // alignment.rs:4-38
    std::optional<graphs::NodeIndex> rpos;
    std::optional<size_t> qpos;
    bool is_aligned() const { return rpos && qpos; }
    bool is_indel() const { return !is_aligned(); }
    bool is_deletion() const { return !rpos && qpos; }   // sic: the reference's naming (alignment.rs:30-32)
    bool is_insertion() const { return rpos && !qpos; }  // sic (alignment.rs:34-36)
};
using Alignment = std::vector<AlignedPair>;
}  // namespace aligner

namespace graphs {

// POAGraph with petgraph's adjacency semantics: add_edge prepends, so successors()/predecessors() yield the
// most recently added edge first.  Node 0 = start '#', node 1 = end '$'.
class POAGraph {
public:
    POAGraph() { add_node('#'); add_node('$'); }

    NodeIndex start_node() const { return 0; }
    NodeIndex end_node() const { return 1; }
    size_t node_count() const { return symbol_.size() - 2; }
    size_t node_count_with_start_and_end() const { return symbol_.size(); }
    bool is_empty() const { return node_count() == 0; }
    const std::vector<NodeIndex>& successors(NodeIndex v) const { return succ_[v]; }
    const std::vector<NodeIndex>& predecessors(NodeIndex v) const { return pred_[v]; }
    uint8_t get_symbol(NodeIndex v) const { return symbol_[v]; }
    bool is_symbol_equal(NodeIndex v, uint8_t s) const { return v == end_node() || symbol_[v] == s; }  // poa.rs:463-465

    struct POAEdgeData { size_t weight = 0; std::vector<size_t> sequence_ids; };   // poa.rs:58-79
    struct Sequence { std::string name; NodeIndex start_node; };                    // poa.rs:21-37
    std::vector<Sequence> sequences;
    const std::vector<NodeIndex>& get_aligned_nodes(NodeIndex v) const { return aligned_nodes_[v]; }
    const POAEdgeData* edge_data(NodeIndex s, NodeIndex t) const {
        auto it = edges_.find(((uint64_t)s << 32) | t);
        return it == edges_.end() ? nullptr : &it->second;
    }

    NodeIndex add_node(uint8_t sym) {
        symbol_.push_back(sym); succ_.emplace_back(); pred_.emplace_back(); aligned_nodes_.emplace_back();
        flat_dirty_ = true;
        return (NodeIndex)symbol_.size() - 1;
    }
    // POAGraph::add_edge (poa.rs:118-134): an existing edge keeps its position, gets the sequence id and the weight added
    void add_edge(NodeIndex s, NodeIndex t, size_t sequence_id = 0, size_t weight = 0) {
        POAEdgeData& d = edges_[((uint64_t)s << 32) | t];
        d.sequence_ids.push_back(sequence_id);
        d.weight += weight;
        if (std::find(succ_[s].begin(), succ_[s].end(), t) != succ_[s].end()) return;
        succ_[s].insert(succ_[s].begin(), t);
        pred_[t].insert(pred_[t].begin(), s);
        flat_dirty_ = true;
    }
    // poa.rs:136-169; returns (first, last)
    std::optional<std::pair<NodeIndex, NodeIndex>> add_nodes_for_sequence(const std::string& seq, size_t start, size_t end,
                                                                        const std::vector<size_t>* weights = nullptr) {
        if (start == end) return std::nullopt;
        NodeIndex first = 0, prev = 0;
        for (size_t pos = start; pos < end; ++pos) {
            NodeIndex cur = add_node((uint8_t)seq[pos]);
            if (pos == start) first = cur;
            else add_edge(prev, cur, sequences.size(), weights ? (*weights)[pos - 1] + (*weights)[pos] : 0);
            prev = cur;
        }
        return std::make_pair(first, prev);
    }
    // add_alignment_with_weights (poa.rs:171-321): the graph update of `poasta align` — consumes the Alignment verbatim
    void add_alignment_with_weights(const std::string& name, const std::string& seq, const aligner::Alignment* alignment,
                                    const std::vector<size_t>& weights) {
        if (seq.size() != weights.size()) throw PoastaError("WeightsUnequalSize");
        if (!alignment) {
            if (seq.empty()) sequences.push_back(Sequence{name, start_node()});
            else sequences.push_back(Sequence{name, add_nodes_for_sequence(seq, 0, seq.size(), &weights)->first});
            post_process();
            return;
        }
        std::vector<size_t> valid_ix;
        for (const auto& e : *alignment) if (e.qpos && *e.qpos < seq.size()) valid_ix.push_back(*e.qpos);
        if (valid_ix.empty()) {
            if (seq.empty()) { sequences.push_back(Sequence{name, start_node()}); post_process(); return; }
            throw PoastaError("InvalidAlignment");
        }
        const size_t first = valid_ix.front(), last = valid_ix.back();
        auto nodes_unaligned_begin = add_nodes_for_sequence(seq, 0, first, &weights);
        std::optional<NodeIndex> prev;
        if (nodes_unaligned_begin) prev = nodes_unaligned_begin->second;
        const auto nodes_unaligned_end = add_nodes_for_sequence(seq, last + 1, seq.size(), &weights);
        for (const auto& ap : *alignment) {
            if (!ap.qpos) continue;
            const size_t q = *ap.qpos;
            const uint8_t qsymbol = (uint8_t)seq[q];
            NodeIndex curr;
            if (ap.rpos) {
                const NodeIndex r = *ap.rpos;
                if (symbol_[r] == qsymbol) curr = r;
                else {
                    // aligned to a node with another symbol: reuse a node of that column with this symbol, else make one
                    std::optional<NodeIndex> found;
                    for (NodeIndex other : aligned_nodes_[r]) if (symbol_[other] == qsymbol) { found = other; break; }
                    if (found) curr = *found;
                    else {
                        const NodeIndex nn = add_node(qsymbol);
                        const std::vector<NodeIndex> others = aligned_nodes_[r];
                        for (NodeIndex o : others) { aligned_nodes_[o].push_back(nn); aligned_nodes_[nn].push_back(o); }
                        aligned_nodes_[r].push_back(nn);
                        aligned_nodes_[nn].push_back(r);
                        curr = nn;
                    }
                }
            } else {
                curr = add_node(qsymbol);  // an insertion
            }
            if (!nodes_unaligned_begin) nodes_unaligned_begin = std::make_pair(curr, curr);
            if (prev) add_edge(*prev, curr, sequences.size(), weights[q - 1] + weights[q]);
            prev = curr;
        }
        if (nodes_unaligned_end) add_edge(*prev, nodes_unaligned_end->first, sequences.size(), weights[last] + weights[last + 1]);
        sequences.push_back(Sequence{name, nodes_unaligned_begin->first});
        post_process();
    }
    // poa.rs:323-363
    void post_process() {
        const NodeIndex s = start_node(), e = end_node();
        for (NodeIndex v : std::vector<NodeIndex>(succ_[s])) { pred_[v].erase(std::find(pred_[v].begin(), pred_[v].end(), s)); edges_.erase(((uint64_t)s << 32) | v); }
        succ_[s].clear();
        for (NodeIndex v : std::vector<NodeIndex>(pred_[e])) { succ_[v].erase(std::find(succ_[v].begin(), succ_[v].end(), e)); edges_.erase(((uint64_t)v << 32) | e); }
        pred_[e].clear();
        const NodeIndex n = (NodeIndex)symbol_.size();
        for (NodeIndex v = 0; v < n; ++v)
            if (v != s && v != e && pred_[v].empty()) { succ_[s].insert(succ_[s].begin(), v); pred_[v].insert(pred_[v].begin(), s); }
        for (NodeIndex v = 0; v < n; ++v)
            if (v != s && v != e && succ_[v].empty()) { succ_[v].insert(succ_[v].begin(), e); pred_[e].insert(pred_[e].begin(), v); }
        flat_dirty_ = true;
    }

    // The flattened AlignableRefGraph handed to the device library (built lazily, by iterating the trait; after a graph update the
    // same handle is refreshed in place: poa_graph_update).
    struct Flat {
        poa_graph_t* handle = nullptr;
        ~Flat() { if (handle) poa_graph_destroy(handle); }
    };
    const poa_graph_t* device_graph() const {
        if (!flat_ || flat_dirty_) {
            const uint32_t n = (uint32_t)symbol_.size();
            std::vector<uint32_t> so(n + 1, 0), po(n + 1, 0), s, p;
            for (uint32_t v = 0; v < n; ++v) {
                for (NodeIndex t : succ_[v]) s.push_back(t);
                for (NodeIndex t : pred_[v]) p.push_back(t);
                so[v + 1] = (uint32_t)s.size(); po[v + 1] = (uint32_t)p.size();
            }
            if (flat_) {
                const int rc = poa_graph_update(flat_->handle, n, start_node(), end_node(), symbol_.data(), so.data(), s.data(), po.data(), p.data());
                if (rc != POA_OK) throw PoastaError(std::string("poa_graph_update: ") + poa_last_error());
            } else {
                auto f = std::make_shared<Flat>();
                const int rc = poa_graph_create(n, start_node(), end_node(), symbol_.data(), so.data(), s.data(), po.data(), p.data(), &f->handle);
                if (rc != POA_OK) throw PoastaError(std::string("poa_graph_create: ") + poa_last_error());
                flat_ = f;
            }
            flat_dirty_ = false;
        }
        return flat_->handle;
    }

private:
    std::vector<uint8_t> symbol_;
    std::vector<std::vector<NodeIndex>> succ_, pred_, aligned_nodes_;
    std::map<uint64_t, POAEdgeData> edges_;   // start / end edges carry no data (poa.rs:73-78)
    mutable std::shared_ptr<Flat> flat_;      // the library's handle: refreshed from the trait view after a graph update (O(N + E), host)
    mutable bool flat_dirty_ = false;
public:
    // used by io::load_graph_from_fasta_msa (it wires aligned_nodes itself, graph.rs:71-85)
    void link_aligned_nodes(NodeIndex a, NodeIndex b) { aligned_nodes_[a].push_back(b); aligned_nodes_[b].push_back(a); }
};

}  // namespace graphs

// =====================================================================================================
namespace aligner {

// GapAffine::new(cost_mismatch, cost_gap_extend, cost_gap_open) — NB the argument order (gap_affine.rs:27)
struct GapAffine {
    uint8_t cost_mismatch, cost_gap_extend, cost_gap_open;
    GapAffine(uint8_t mismatch, uint8_t gap_extend, uint8_t gap_open)
        : cost_mismatch(mismatch), cost_gap_extend(gap_extend), cost_gap_open(gap_open) {}
    uint8_t mismatch() const { return cost_mismatch; }
    uint8_t gap_open() const { return cost_gap_open; }
    uint8_t gap_extend() const { return cost_gap_extend; }
};

// GapAffine2Piece::new(cost_mismatch, cost_gap_extend1, cost_gap_open1, cost_gap_extend2, cost_gap_open2) — the reference's
// argument order; it panics unless extend1 >= extend2 (gap_affine_2piece.rs:28-33)
struct GapAffine2Piece {
    uint8_t cost_mismatch, cost_gap_extend1, cost_gap_open1, cost_gap_extend2, cost_gap_open2;
    GapAffine2Piece(uint8_t mismatch, uint8_t gap_extend1, uint8_t gap_open1, uint8_t gap_extend2, uint8_t gap_open2)
        : cost_mismatch(mismatch), cost_gap_extend1(gap_extend1), cost_gap_open1(gap_open1), cost_gap_extend2(gap_extend2), cost_gap_open2(gap_open2) {
        if (gap_extend1 < gap_extend2) throw PoastaError("gap_extend1 must be greater than or equal to gap_extend2 for two-piece model");
    }
    uint8_t mismatch() const { return cost_mismatch; }
    uint8_t gap_open() const { return cost_gap_open1; }
    uint8_t gap_extend() const { return cost_gap_extend1; }
    uint8_t gap_open2() const { return cost_gap_open2; }
    uint8_t gap_extend2() const { return cost_gap_extend2; }
};

// std::ops::Bound<usize> and AlignmentType (scoring/mod.rs:50-62).  `AlignmentType::Global` or
// `AlignmentType::EndsFree(qry_free_begin, qry_free_end, graph_free_begin, graph_free_end)`; an ends-free result is
// defined by the reference's search, so the library replays that search for every query (exact mode implied).
struct Bound {
    uint32_t kind = POA_BOUND_UNBOUNDED;
    uint32_t value = 0;
    static Bound Unbounded() { return Bound{}; }
    static Bound Included(size_t n) { return Bound{POA_BOUND_INCLUDED, (uint32_t)n}; }
    static Bound Excluded(size_t n) { return Bound{POA_BOUND_EXCLUDED, (uint32_t)n}; }
};
struct AlignmentType {
    bool ends_free = false;
    Bound qry_free_begin, qry_free_end, graph_free_begin, graph_free_end;
    static const AlignmentType Global;
    static AlignmentType EndsFree(Bound qb = Bound{}, Bound qe = Bound{}, Bound gb = Bound{}, Bound ge = Bound{}) {
        return AlignmentType{true, qb, qe, gb, ge};
    }
};
inline const AlignmentType AlignmentType::Global{};

struct AffineMinGapCost { GapAffine costs; static constexpr uint32_t heuristic = POA_HEURISTIC_MINGAP; static constexpr bool two_piece = false; explicit AffineMinGapCost(GapAffine c) : costs(c) {} };
struct AffineDijkstra { GapAffine costs; static constexpr uint32_t heuristic = POA_HEURISTIC_DIJKSTRA; static constexpr bool two_piece = false; explicit AffineDijkstra(GapAffine c) : costs(c) {} };
// config.rs:160-272: what `poasta align -g o1,o2 -e e1,e2` constructs.  Mode::Dense (Global only) is the optimum of the model's
// alignment graph; Mode::Exact / Hybrid replay the reference's own five-state search (poa_align_batch_2piece_ex), whose
// score may exceed that optimum — the reference's result, tie-breaks included.
struct Affine2PieceMinGapCost { GapAffine2Piece costs; static constexpr uint32_t heuristic = POA_HEURISTIC_MINGAP; static constexpr bool two_piece = true; explicit Affine2PieceMinGapCost(GapAffine2Piece c) : costs(c) {} };
struct Affine2PieceDijkstra { GapAffine2Piece costs; static constexpr uint32_t heuristic = POA_HEURISTIC_DIJKSTRA; static constexpr bool two_piece = true; explicit Affine2PieceDijkstra(GapAffine2Piece c) : costs(c) {} };

struct AstarResult {  // astar.rs:81-90 (+ the exactness certificate of the dense pass)
    uint32_t score = 0;
    Alignment alignment;
    size_t num_queued = 0, num_visited = 0, num_pruned = 0;  // no meaning for a dense pass: 0
    uint32_t flags = 0;  // POA_FLAG_*; 0 == bit-identical to the reference guaranteed
};

enum class Mode : uint32_t { Dense = POA_MODE_DENSE, Exact = POA_MODE_EXACT, Hybrid = POA_MODE_HYBRID };

template <typename Config>
class PoastaAligner {
public:
    // queue_entries_per_cell: workspace of the replayed search per table cell (0: the engine's default; raise it when a result
    // comes back with POA_FLAG_EXACT_OVERFLOW)
    PoastaAligner(Config config, AlignmentType aln_type, int device = 0, Mode mode = Mode::Dense, float queue_entries_per_cell = 0.f)
        : config_(config), aln_type_(aln_type), device_(device), mode_(mode), queue_entries_per_cell_(queue_entries_per_cell) {}

    // mod.rs:114-145
    AstarResult align(const graphs::POAGraph& g, const std::string& seq) const { return align_batch(g, {seq}, true).at(0); }
    // mod.rs:69-79 — the bubble index only steers the reference's search; the library builds what it needs itself
    AstarResult align_with_existing_bubbles(const graphs::POAGraph& g, const std::string& seq) const { return align(g, seq); }
    // mod.rs:81-90
    AstarResult align_no_pruning(const graphs::POAGraph& g, const std::string& seq) const { return align_batch(g, {seq}, false).at(0); }

    // the data-parallel shape of `lasagna align` (src/bin/lasagna.rs:246-268)
    std::vector<AstarResult> align_batch(const graphs::POAGraph& g, const std::vector<std::string>& seqs, bool pruning = true,
                                         poa_stats_t* stats = nullptr) const {
        const uint32_t n = (uint32_t)seqs.size();
        std::vector<uint64_t> qoff(n + 1, 0);
        std::string qseq;
        for (uint32_t i = 0; i < n; ++i) { qseq += seqs[i]; qoff[i + 1] = qseq.size(); }
        const uint64_t cap = qseq.size() + (uint64_t)n * g.node_count_with_start_and_end() + 1;
        std::vector<uint32_t> score(n), flags(n);
        std::vector<uint64_t> pair_off(n + 1, 0);
        std::vector<poa_aln_pair_t> pairs(cap);
        poa_config_t cfg{};
        cfg.mode = (uint32_t)mode_; cfg.heuristic = Config::heuristic; cfg.pruning = pruning ? 1u : 0u;
        cfg.queue_entries_per_cell = queue_entries_per_cell_;
        if (aln_type_.ends_free) {
            cfg.span = POA_SPAN_ENDS_FREE;
            cfg.qry_free_begin = poa_bound_t{aln_type_.qry_free_begin.kind, aln_type_.qry_free_begin.value};
            cfg.qry_free_end = poa_bound_t{aln_type_.qry_free_end.kind, aln_type_.qry_free_end.value};
            cfg.graph_free_begin = poa_bound_t{aln_type_.graph_free_begin.kind, aln_type_.graph_free_begin.value};
            cfg.graph_free_end = poa_bound_t{aln_type_.graph_free_end.kind, aln_type_.graph_free_end.value};
        }
        std::vector<uint32_t> counters;
        int rc;
        if constexpr (Config::two_piece) {
            poa_costs2_t c{};
            c.mismatch = config_.costs.mismatch(); c.gap_open1 = config_.costs.gap_open(); c.gap_extend1 = config_.costs.gap_extend();
            c.gap_open2 = config_.costs.gap_open2(); c.gap_extend2 = config_.costs.gap_extend2();
            if (mode_ != Mode::Dense || aln_type_.ends_free) { cfg.mode = POA_MODE_EXACT; counters.resize(4 * (size_t)n); }
            rc = poa_align_batch_2piece_ex(g.device_graph(), &c, &cfg, n, (const uint8_t*)qseq.data(), qoff.data(), score.data(), pairs.data(),
                                           pair_off.data(), cap, flags.data(), stats, counters.empty() ? nullptr : counters.data(), device_);
        } else {
            const poa_costs_t c{config_.costs.mismatch(), config_.costs.gap_open(), config_.costs.gap_extend(), 0};
            rc = poa_align_batch_ex(g.device_graph(), &c, &cfg, n, (const uint8_t*)qseq.data(), qoff.data(), score.data(),
                                    pairs.data(), pair_off.data(), cap, flags.data(), stats, device_);
        }
        if (rc != POA_OK) throw PoastaError(std::string("poa_align_batch: ") + poa_last_error());
        std::vector<AstarResult> out(n);
        for (uint32_t i = 0; i < n; ++i) {
            out[i].score = score[i]; out[i].flags = flags[i];
            if (!counters.empty()) { out[i].num_queued = counters[4 * (size_t)i]; out[i].num_visited = counters[4 * (size_t)i + 1]; out[i].num_pruned = counters[4 * (size_t)i + 2]; }
            for (uint64_t k = pair_off[i]; k < pair_off[i + 1]; ++k) {
                AlignedPair ap;
                if (pairs[k].rpos != POA_NONE) ap.rpos = pairs[k].rpos;
                if (pairs[k].qpos != POA_NONE) ap.qpos = pairs[k].qpos;
                out[i].alignment.push_back(ap);
            }
            if (flags[i] & POA_FLAG_REF_PANIC) out[i].num_pruned = 0;  // the reference would have panicked here; the caller may check flags
        }
        return out;
    }

private:
    Config config_;
    AlignmentType aln_type_;
    int device_;
    Mode mode_;
    float queue_entries_per_cell_;
};

}  // namespace aligner

// =====================================================================================================
namespace io {

struct GraphSegments {  // src/io/graph.rs:113-121
    std::vector<std::string> names;
    std::vector<graphs::NodeIndex> start_nodes, end_nodes;
    std::vector<size_t> segment_lengths;
};
struct POAGraphFromGFA { graphs::POAGraph graph; GraphSegments graph_segments; };

inline std::vector<std::string> split_tabs(const std::string& s, size_t max_parts) {
    std::vector<std::string> out;
    size_t b = 0;
    while (out.size() + 1 < max_parts) {
        const size_t t = s.find('\t', b);
        if (t == std::string::npos) break;
        out.push_back(s.substr(b, t - b));
        b = t + 1;
    }
    out.push_back(s.substr(b));
    return out;
}
inline std::string trim(const std::string& s) {
    const size_t a = s.find_first_not_of(" \t\r\n");
    if (a == std::string::npos) return "";
    return s.substr(a, s.find_last_not_of(" \t\r\n") - a + 1);
}

// load_graph_from_gfa (src/io/graph.rs:125-227): every S line becomes a chain of 1-bp nodes (sequence upper-cased,
// gfa.rs:255), L lines connect segment end -> segment start (forward strands only), links seen before both
// segments are added afterwards; post_process at the end.
inline POAGraphFromGFA load_graph_from_gfa(std::istream& in) {
    POAGraphFromGFA out;
    std::unordered_map<std::string, size_t> name_to_ix;
    std::vector<std::pair<std::string, std::string>> later;
    std::string line;
    while (std::getline(in, line)) {
        const std::string t = trim(line);
        if (t.empty()) continue;
        if (t[0] == 'S') {
            auto parts = split_tabs(t, 4);
            if (parts.size() < 3 || parts[0] != "S") continue;  // "Failed to parse line"
            if (parts[2] == "*") continue;                       // no sequence: omitted
            std::string seq = parts[2];
            for (auto& ch : seq) ch = (char)std::toupper((unsigned char)ch);
            auto se = out.graph.add_nodes_for_sequence(seq, 0, seq.size());
            if (!se) throw PoastaError("GraphError: empty segment sequence");
            name_to_ix[parts[1]] = out.graph_segments.names.size();
            out.graph_segments.names.push_back(parts[1]);
            out.graph_segments.start_nodes.push_back(se->first);
            out.graph_segments.end_nodes.push_back(se->second);
            out.graph_segments.segment_lengths.push_back(seq.size());
        } else if (t[0] == 'L') {
            auto parts = split_tabs(t, 6);
            if (parts.size() < 6 || parts[0] != "L") continue;
            if ((parts[2] != "+" && parts[2] != "-") || (parts[4] != "+" && parts[4] != "-")) continue;
            if (parts[2] == "-" || parts[4] == "-") throw PoastaError("GraphError: links using the reverse strand are not supported");
            auto a = name_to_ix.find(parts[1]), b = name_to_ix.find(parts[3]);
            if (a != name_to_ix.end() && b != name_to_ix.end())
                out.graph.add_edge(out.graph_segments.end_nodes[a->second], out.graph_segments.start_nodes[b->second]);
            else
                later.push_back({parts[1], parts[3]});
        }
    }
    for (auto& l : later) {
        auto a = name_to_ix.find(l.first), b = name_to_ix.find(l.second);
        if (a != name_to_ix.end() && b != name_to_ix.end())
            out.graph.add_edge(out.graph_segments.end_nodes[a->second], out.graph_segments.start_nodes[b->second]);
    }
    out.graph.post_process();
    return out;
}
inline POAGraphFromGFA load_graph_from_gfa(const std::string& path) {
    std::ifstream f(path);
    if (!f) throw PoastaError("could not open " + path);
    return load_graph_from_gfa(f);
}


// load_graph_from_fasta_msa (src/io/graph.rs:36-103): per column one node per distinct symbol, '-' skipped, each row threads
// an edge (weight 2) from its previous node; post_process at the end.  records = (name, gapped row).
inline graphs::POAGraph load_graph_from_fasta_msa(const std::vector<std::pair<std::string, std::string>>& records) {
    graphs::POAGraph graph;
    std::vector<std::vector<graphs::NodeIndex>> nodes_per_col;
    for (size_t seq_id = 0; seq_id < records.size(); ++seq_id) {
        const std::string& chars = records[seq_id].second;
        if (chars.size() > nodes_per_col.size()) nodes_per_col.resize(chars.size());
        std::optional<graphs::NodeIndex> prev;
        for (size_t col = 0; col < chars.size(); ++col) {
            const uint8_t c = (uint8_t)chars[col];
            if (c == '-') continue;
            std::optional<graphs::NodeIndex> node;
            for (graphs::NodeIndex v : nodes_per_col[col]) if (graph.get_symbol(v) == c) { node = v; break; }
            if (!node) {
                node = graph.add_node(c);
                for (graphs::NodeIndex other : nodes_per_col[col]) graph.link_aligned_nodes(other, *node);
                nodes_per_col[col].push_back(*node);
            }
            if (prev) graph.add_edge(*prev, *node, seq_id, 2);
            else graph.sequences.push_back(graphs::POAGraph::Sequence{records[seq_id].first, *node});
            prev = node;
        }
    }
    graph.post_process();
    return graph;
}

// poa_graph_to_fasta (src/io/fasta.rs:69-156) with fasta_aln_for_seq (:19-67): columns by a depth-first post-order over the
// graph in which aligned nodes share a column; each sequence is then walked along the edges that carry its id.  Kept with the
// reference's leading-gap arithmetic (`node_col.saturating_sub(1) - last_col`, last_col starting at 0: tests/io_fasta.rs:19
// asserts "---AC" beside "ACGT--").
inline std::string poa_graph_to_fasta(const graphs::POAGraph& graph) {
    using graphs::NodeIndex;
    std::map<NodeIndex, size_t> node_to_column;
    struct Frame { NodeIndex node; std::vector<NodeIndex> succ; };
    std::vector<Frame> stack;
    stack.push_back(Frame{graph.start_node(), graph.successors(graph.start_node())});
    std::set<NodeIndex> visited;
    std::vector<NodeIndex> rev_postorder;
    while (!stack.empty()) {
        std::optional<NodeIndex> child;
        auto& it = stack.back().succ;
        while (!it.empty()) { const NodeIndex s = it.back(); it.pop_back(); if (!visited.count(s)) { child = s; break; } }
        if (child) {
            visited.insert(*child);
            std::vector<NodeIndex> successors = graph.successors(*child);
            for (NodeIndex aln : graph.get_aligned_nodes(*child))
                if (!visited.count(aln)) {
                    visited.insert(aln);
                    const auto& more = graph.successors(aln);
                    successors.insert(successors.end(), more.begin(), more.end());
                }
            stack.push_back(Frame{*child, std::move(successors)});
        } else {
            rev_postorder.push_back(stack.back().node);
            stack.pop_back();
        }
    }
    std::reverse(rev_postorder.begin(), rev_postorder.end());
    size_t curr_col = 0, max_col = 0;
    for (NodeIndex n : rev_postorder) {
        if (n == graph.start_node() || n == graph.end_node()) continue;
        if (!node_to_column.count(n)) {
            node_to_column[n] = curr_col;
            for (NodeIndex a : graph.get_aligned_nodes(n)) node_to_column[a] = curr_col;
            max_col = curr_col;
            curr_col += 1;
        }
    }
    std::string out;
    for (size_t seq_id = 0; seq_id < graph.sequences.size(); ++seq_id) {
        std::string row;
        std::optional<NodeIndex> curr = graph.sequences[seq_id].start_node;
        size_t last_col = 0;
        bool empty_seq = false;
        while (curr) {
            auto col = node_to_column.find(*curr);
            if (col == node_to_column.end()) { empty_seq = true; break; }  // an empty sequence starts at the start node
            const size_t node_col = col->second;
            row.append((node_col ? node_col - 1 : 0) - last_col, '-');
            row.push_back((char)graph.get_symbol(*curr));
            std::optional<NodeIndex> next;
            for (NodeIndex t : graph.successors(*curr)) {   // the LAST out-edge carrying this id wins (no break, fasta.rs:47-56)
                const auto* d = graph.edge_data(*curr, t);
                if (d && std::binary_search(d->sequence_ids.begin(), d->sequence_ids.end(), seq_id)) next = t;
            }
            curr = next;
            last_col = node_col;
        }
        if (empty_seq) row.clear();
        else if (!node_to_column.empty()) row.append(max_col - last_col, '-');
        out += ">" + graph.sequences[seq_id].name + "\n";
        for (size_t p2 = 0; p2 < row.size(); p2 += 80) out += row.substr(p2, 80) + "\n";   // noodles' writer wraps at 80
    }
    return out;
}

// NodeSegmentResolver (src/io/gaf.rs:11-55): (segment index, position in segment) of a node; the reference
// walks every segment per call, here the map is built once (segments are chains of consecutively created nodes).
class NodeSegmentResolver {
public:
    NodeSegmentResolver(const graphs::POAGraph& g, const GraphSegments& segs) : map_(g.node_count_with_start_and_end(), {-1, 0}) {
        for (size_t s = 0; s < segs.names.size(); ++s) {
            graphs::NodeIndex cur = segs.start_nodes[s];
            size_t pos = 0;
            for (;;) {
                if (map_[cur].first < 0) map_[cur] = {(int64_t)s, pos};
                if (cur == segs.end_nodes[s]) break;
                const auto& su = g.successors(cur);
                if (su.empty()) break;
                cur = su.front();
                pos += 1;
            }
        }
    }
    std::optional<std::pair<size_t, size_t>> resolve(graphs::NodeIndex node) const {
        if (node >= map_.size() || map_[node].first < 0) return std::nullopt;
        return std::make_pair((size_t)map_[node].first, map_[node].second);
    }
private:
    std::vector<std::pair<int64_t, size_t>> map_;
};

struct GAFRecord {  // src/io/gaf.rs:58-72, Display :120-147
    std::string query_name; size_t query_length = 0, query_start = 0, query_end = 0; char strand = '+';
    std::string graph_path; size_t path_length = 0, path_aln_start = 0, path_aln_end = 0, num_matches = 0, aln_block_len = 0,
        mapping_quality = 60;
    std::vector<std::string> additional_fields;  // already formatted "tag:type:value"
    std::string to_string() const {
        std::ostringstream o;
        o << query_name << '\t' << query_length << '\t' << query_start << '\t' << query_end << '\t' << strand << '\t' << graph_path
          << '\t' << path_length << '\t' << path_aln_start << '\t' << path_aln_end << '\t' << num_matches << '\t' << aln_block_len
          << '\t' << mapping_quality << '\t';
        for (size_t i = 0; i < additional_fields.size(); ++i) o << (i ? "\t" : "") << additional_fields[i];
        return o.str();
    }
};

// alignment_to_gaf (src/io/gaf.rs:152-304), including its quirks: leading (node, None) pairs bump query_start,
// leading (None, qpos) pairs are skipped, one trailing I or D run is dropped from the CIGAR.
inline std::optional<GAFRecord> alignment_to_gaf(const graphs::POAGraph& graph, const GraphSegments& segs, const std::string& seq_name,
                                                 const std::string& sequence, const aligner::Alignment& alignment,
                                                 const NodeSegmentResolver& resolver) {
    if (alignment.empty()) return std::nullopt;
    size_t query_start = 0, path_aln_start = 0, last_match_segment_ix = 0, last_match_segment_pos = 0, num_matches = 0;
    std::vector<size_t> path_segments;
    std::string ops;
    bool at_start = true;
    for (const auto& ap : alignment) {
        if (at_start) {
            if (ap.is_insertion()) { query_start += 1; }
            else if (ap.is_aligned()) {
                auto r = resolver.resolve(*ap.rpos);
                if (!r) throw PoastaError("node not found in any segment");
                path_aln_start = r->second;
                path_segments.push_back(r->first);
                const bool eq = graph.is_symbol_equal(*ap.rpos, (uint8_t)sequence[*ap.qpos]);
                num_matches += eq; ops.push_back(eq ? '=' : 'X');
                at_start = false;
                last_match_segment_ix = path_segments.size() - 1; last_match_segment_pos = r->second;
            }
        } else if (ap.rpos && ap.qpos) {
            auto r = resolver.resolve(*ap.rpos);
            if (!r) throw PoastaError("node not found in any segment");
            if (path_segments.empty() || path_segments.back() != r->first) path_segments.push_back(r->first);
            const bool eq = graph.is_symbol_equal(*ap.rpos, (uint8_t)sequence[*ap.qpos]);
            num_matches += eq; ops.push_back(eq ? '=' : 'X');
            last_match_segment_ix = path_segments.size() - 1; last_match_segment_pos = r->second;
        } else if (ap.rpos) {
            auto r = resolver.resolve(*ap.rpos);
            if (!r) throw PoastaError("node not found in any segment");
            if (path_segments.empty() || path_segments.back() != r->first) path_segments.push_back(r->first);
            ops.push_back('D');
        } else {
            ops.push_back('I');
        }
    }
    if (path_segments.empty()) throw PoastaError("alignment without an aligned pair (the reference would panic)");
    GAFRecord rec;
    size_t path_length = 0;
    for (size_t i = 0; i <= last_match_segment_ix; ++i) { rec.graph_path += ">" + segs.names[path_segments[i]]; path_length += segs.segment_lengths[path_segments[i]]; }
    const size_t path_aln_end = path_length - segs.segment_lengths[path_segments[last_match_segment_ix]] + last_match_segment_pos;
    size_t query_end = 0;
    for (auto it = alignment.rbegin(); it != alignment.rend(); ++it) if (it->is_aligned()) { query_end = *it->qpos; break; }
    std::vector<std::pair<char, size_t>> rle;
    for (char c : ops) { if (!rle.empty() && rle.back().first == c) rle.back().second++; else rle.push_back({c, 1}); }
    if (!rle.empty() && (rle.back().first == 'I' || rle.back().first == 'D')) rle.pop_back();
    size_t block = 0;
    std::string cigar;
    for (auto& r : rle) { block += r.second; cigar += std::to_string(r.second) + r.first; }
    rec.query_name = seq_name; rec.query_length = sequence.size(); rec.query_start = query_start; rec.query_end = query_end;
    rec.path_length = path_length; rec.path_aln_start = path_aln_start; rec.path_aln_end = path_aln_end;
    rec.num_matches = num_matches; rec.aln_block_len = block;
    rec.additional_fields.push_back("cg:Z:" + cigar);
    return rec;
}

}  // namespace io
}  // namespace poasta
