// poasta_amd.hpp — C++17 host-side mirror of the reference's aligner interface over the C ABI
// (include/poasta_amd.h).  Header-only.  The reference is Rust (no toolchain in this image), so this
// is the compiled-language host layer: same names, argument meaning and error behaviour as
//   poasta::aligner::{PoastaAligner, AlignedPair, AstarResult}      /root/reference/src/aligner/mod.rs:40-146, astar.rs:81-90
//   poasta::aligner::scoring::{GapAffine, AlignmentType}            src/aligner/scoring/gap_affine.rs:20-30, scoring/mod.rs:50-62
//   poasta::aligner::config::{AffineMinGapCost, AffineDijkstra}     src/aligner/config.rs:49,:104
//   poasta::graphs::poa::POAGraph (the parts a batch driver needs)  src/graphs/poa.rs:85-134, :323-363, :384-471
//   poasta::io::{load_graph_from_gfa, GraphSegments}                src/io/graph.rs:103-227, src/io/gfa.rs:233-360
//   poasta::io::gaf::{alignment_to_gaf, GAFRecord, NodeSegmentResolver}   src/io/gaf.rs:11-304
// Where the reference panics, this layer throws poasta::PoastaError.  All alignment work happens in the
// gfx950 library; nothing here computes an alignment on the CPU.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cctype>
#include <fstream>
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
namespace graphs {

using NodeIndex = uint32_t;

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

    NodeIndex add_node(uint8_t sym) {
        symbol_.push_back(sym); succ_.emplace_back(); pred_.emplace_back();
        flat_.reset();
        return (NodeIndex)symbol_.size() - 1;
    }
    // POAGraph::add_edge (poa.rs:118-134): an existing edge keeps its position
    void add_edge(NodeIndex s, NodeIndex t) {
        if (std::find(succ_[s].begin(), succ_[s].end(), t) != succ_[s].end()) return;
        succ_[s].insert(succ_[s].begin(), t);
        pred_[t].insert(pred_[t].begin(), s);
        flat_.reset();
    }
    // poa.rs:136-169; returns (first, last)
    std::optional<std::pair<NodeIndex, NodeIndex>> add_nodes_for_sequence(const std::string& seq, size_t start, size_t end) {
        if (start == end) return std::nullopt;
        NodeIndex first = 0, prev = 0;
        for (size_t pos = start; pos < end; ++pos) {
            NodeIndex cur = add_node((uint8_t)seq[pos]);
            if (pos == start) first = cur; else add_edge(prev, cur);
            prev = cur;
        }
        return std::make_pair(first, prev);
    }
    // poa.rs:323-363
    void post_process() {
        const NodeIndex s = start_node(), e = end_node();
        for (NodeIndex v : std::vector<NodeIndex>(succ_[s])) pred_[v].erase(std::find(pred_[v].begin(), pred_[v].end(), s));
        succ_[s].clear();
        for (NodeIndex v : std::vector<NodeIndex>(pred_[e])) succ_[v].erase(std::find(succ_[v].begin(), succ_[v].end(), e));
        pred_[e].clear();
        const NodeIndex n = (NodeIndex)symbol_.size();
        for (NodeIndex v = 0; v < n; ++v)
            if (v != s && v != e && pred_[v].empty()) { succ_[s].insert(succ_[s].begin(), v); pred_[v].insert(pred_[v].begin(), s); }
        for (NodeIndex v = 0; v < n; ++v)
            if (v != s && v != e && succ_[v].empty()) { succ_[v].insert(succ_[v].begin(), e); pred_[e].insert(pred_[e].begin(), v); }
        flat_.reset();
    }

    // The flattened AlignableRefGraph handed to the device library (built lazily, by iterating the trait).
    struct Flat {
        poa_graph_t* handle = nullptr;
        ~Flat() { if (handle) poa_graph_destroy(handle); }
    };
    const poa_graph_t* device_graph() const {
        if (!flat_) {
            const uint32_t n = (uint32_t)symbol_.size();
            std::vector<uint32_t> so(n + 1, 0), po(n + 1, 0), s, p;
            for (uint32_t v = 0; v < n; ++v) {
                for (NodeIndex t : succ_[v]) s.push_back(t);
                for (NodeIndex t : pred_[v]) p.push_back(t);
                so[v + 1] = (uint32_t)s.size(); po[v + 1] = (uint32_t)p.size();
            }
            auto f = std::make_shared<Flat>();
            const int rc = poa_graph_create(n, start_node(), end_node(), symbol_.data(), so.data(), s.data(), po.data(), p.data(), &f->handle);
            if (rc != POA_OK) throw PoastaError(std::string("poa_graph_create: ") + poa_last_error());
            flat_ = f;
        }
        return flat_->handle;
    }

private:
    std::vector<uint8_t> symbol_;
    std::vector<std::vector<NodeIndex>> succ_, pred_;
    mutable std::shared_ptr<Flat> flat_;
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

struct AffineMinGapCost { GapAffine costs; static constexpr uint32_t heuristic = POA_HEURISTIC_MINGAP; explicit AffineMinGapCost(GapAffine c) : costs(c) {} };
struct AffineDijkstra { GapAffine costs; static constexpr uint32_t heuristic = POA_HEURISTIC_DIJKSTRA; explicit AffineDijkstra(GapAffine c) : costs(c) {} };

struct AlignedPair {  // alignment.rs:4-38
    std::optional<graphs::NodeIndex> rpos;
    std::optional<size_t> qpos;
    bool is_aligned() const { return rpos && qpos; }
    bool is_indel() const { return !is_aligned(); }
    bool is_deletion() const { return !rpos && qpos; }   // sic: the reference's naming (alignment.rs:30-32)
    bool is_insertion() const { return rpos && !qpos; }  // sic (alignment.rs:34-36)
};
using Alignment = std::vector<AlignedPair>;

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
    PoastaAligner(Config config, AlignmentType aln_type, int device = 0, Mode mode = Mode::Dense)
        : config_(config), aln_type_(aln_type), device_(device), mode_(mode) {}

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
        const poa_costs_t c{config_.costs.mismatch(), config_.costs.gap_open(), config_.costs.gap_extend(), 0};
        poa_config_t cfg{};
        cfg.mode = (uint32_t)mode_; cfg.heuristic = Config::heuristic; cfg.pruning = pruning ? 1u : 0u;
        if (aln_type_.ends_free) {
            cfg.span = POA_SPAN_ENDS_FREE;
            cfg.qry_free_begin = poa_bound_t{aln_type_.qry_free_begin.kind, aln_type_.qry_free_begin.value};
            cfg.qry_free_end = poa_bound_t{aln_type_.qry_free_end.kind, aln_type_.qry_free_end.value};
            cfg.graph_free_begin = poa_bound_t{aln_type_.graph_free_begin.kind, aln_type_.graph_free_begin.value};
            cfg.graph_free_end = poa_bound_t{aln_type_.graph_free_end.kind, aln_type_.graph_free_end.value};
        }
        const int rc = poa_align_batch_ex(g.device_graph(), &c, &cfg, n, (const uint8_t*)qseq.data(), qoff.data(), score.data(),
                                          pairs.data(), pair_off.data(), cap, flags.data(), stats, device_);
        if (rc != POA_OK) throw PoastaError(std::string("poa_align_batch: ") + poa_last_error());
        std::vector<AstarResult> out(n);
        for (uint32_t i = 0; i < n; ++i) {
            out[i].score = score[i]; out[i].flags = flags[i];
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
