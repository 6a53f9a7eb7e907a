// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
//
// Literal CPU restatement of the reference's gap-affine aligner hot path
// (SURVEY.md §8(a), appendix A).  Every function cites the reference lines it
// follows; all paths are relative to /root/reference/.
//
//   Score                      src/aligner/scoring/mod.rs:64-180
//   GapAffine::gap_cost        src/aligner/scoring/gap_affine.rs:68-80
//   LayeredQueue               src/aligner/queue.rs:19-71
//   AffineQueueLayer           src/aligner/scoring/gap_affine.rs:929-992
//   MinimumGapCostAffine::h    src/aligner/heuristic.rs:68-103   (Dijkstra :37-47)
//   BlockedVisitedStorage      src/aligner/scoring/gap_affine.rs:442-657
//   AffineAstarData            src/aligner/scoring/gap_affine.rs:702-921
//   ReachedBubbleExitsMatch    src/bubbles/reached.rs:13-255
//   DepthFirstGreedyAlignment  src/aligner/dfa.rs:86-251
//   astar_alignment            src/aligner/astar.rs:108-226
//   PoastaAligner::align       src/aligner/mod.rs:69-145
//   two-piece affine model     src/aligner/scoring/gap_affine_2piece.rs:19-133 (costs), :292-516 (edges), :639-794,
//                              :944-1043 (backtrace), :1049-1115 (queue layer) — selected by Costs::two_piece
//
// Parity pin: the reference cannot be compiled in this container (Rust, no
// toolchain), so this restatement is pinned by the reference's own known-answer
// tests (tests/test_oracle_kat.py lists each with its file:line).  Alignment
// OUTPUT (traceback) is not asserted by any reference test: "parity unpinned"
// for traceback, it rests on this restatement being literal.
#pragma once
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <deque>
#include <set>
#include <stdexcept>
#include <vector>

#include "bubbles.hpp"
#include "graph.hpp"

namespace poa_oracle {

// Score: u32 with 0xFFFFFFFF == Unvisited (NonMaxU32 niche, scoring/mod.rs:64-70).
// Ordering: every Score < Unvisited (mod.rs:78-91) == plain u32 ordering.
using Score = uint32_t;
constexpr Score UNVISITED = 0xFFFFFFFFu;

struct RefPanic : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// Score + x  (mod.rs:93-113): panics on Unvisited; NonMaxU32::new(MAX).unwrap() panics.
inline Score score_add(Score s, uint64_t rhs) {
    if (s == UNVISITED) throw RefPanic("Can't add to Score::Unvisited!");
    uint32_t r = s + (uint32_t)rhs;  // release build: wrapping u32 add
    if (r == UNVISITED) throw RefPanic("NonMaxU32::new(u32::MAX).unwrap()");
    return r;
}
// Score - x  (mod.rs:144-152): unchecked u32 subtraction in release builds.
inline Score score_sub(Score s, uint32_t rhs) {
    if (s == UNVISITED) throw RefPanic("Can't subtract from Score::Unvisited!");
    uint32_t r = s - rhs;
    if (r == UNVISITED) throw RefPanic("NonMaxU32::new(u32::MAX).unwrap()");
    return r;
}

enum AlignState : uint8_t { ST_M = 0, ST_D = 1, ST_I = 2, ST_D2 = 3, ST_I2 = 4 };  // aln_graph.rs:8-14

struct Costs {  // GapAffine, gap_affine.rs:20-30.  NB ctor order in the reference: (mismatch, extend, open)
    uint8_t mismatch, gap_open, gap_extend;
    // GapAffine2Piece (gap_affine_2piece.rs:19-33; ctor order (mismatch, extend1, open1, extend2, open2)): gap_open /
    // gap_extend are the first piece, these the second
    uint8_t gap_open2 = 0, gap_extend2 = 0;
    bool two_piece = false;
    // gap_affine.rs:68-80  /  gap_affine_2piece.rs:99-127
    size_t gap_cost(AlignState st, size_t length) const {
        if (length == 0) return 0;
        if (!two_piece) {
            size_t open = (st == ST_M) ? gap_open : 0;
            return open + length * (size_t)gap_extend;
        }
        const size_t cost1 = (size_t)gap_open + length * (size_t)gap_extend;
        const size_t cost2 = (size_t)gap_open2 + length * (size_t)gap_extend2;
        if (st == ST_I || st == ST_D) return cost1;
        if (st == ST_I2 || st == ST_D2) return cost2;
        return std::min(cost1, cost2);
    }
    // gap_affine_2piece.rs:36-66
    size_t breakpoint() const {
        if (gap_extend == gap_extend2) return gap_open <= gap_open2 ? (size_t)-1 : 0;
        const size_t den = (size_t)(gap_extend - gap_extend2);
        if (gap_open2 >= gap_open) return (size_t)(gap_open2 - gap_open) / den;
        return ((size_t)(gap_open - gap_open2) + den - 1) / den;
    }
};

enum Heuristic : int { H_DIJKSTRA = 0, H_MINGAP = 1 };

struct AlnNode { uint32_t node; uint32_t offset; };

// ---------------------------------------------------------------------------
// Visited-score table.  The reference keeps 8x8 tiles in per-node-block hash
// maps (gap_affine.rs:442-548); only the VALUES matter for results, so tiles are
// kept in a flat pool behind a dense tile index (faster than the reference's
// FxHashMap — a generous CPU baseline).
struct VisitedStorage {
    static constexpr uint32_t B = 8;
    struct Tile { Score m[B][B], i[B][B], d[B][B], i2[B][B], d2[B][B]; };
    std::vector<uint32_t> node_ranks;
    uint32_t n_oblocks = 0;
    std::vector<int32_t> tile_ix;
    std::vector<Tile> tiles;

    void init(const Graph& g, const std::vector<uint32_t>& ranks, size_t seq_len) {
        node_ranks = ranks;
        uint32_t n_nblocks = (uint32_t)(g.node_count_with_start_and_end() / B) + 1;
        n_oblocks = (uint32_t)((seq_len + 1) / B) + 1;
        tile_ix.assign((size_t)n_nblocks * n_oblocks, -1);
        tiles.clear();
    }
    inline Score* cell(const AlnNode& a, AlignState st, bool create) {
        uint32_t rank = node_ranks[a.node];
        // (the reference's per-block hash maps take any offset; this flat index does not: an ends-free search that is never
        // allowed to end keeps opening insertions past the query — there the reference itself only stops on u32 overflow)
        if (a.offset / B >= n_oblocks) throw RefPanic("offset beyond the visited table: the search cannot end");
        size_t ti = (size_t)(rank / B) * n_oblocks + a.offset / B;
        int32_t t = tile_ix[ti];
        if (t < 0) {
            if (!create) return nullptr;
            t = (int32_t)tiles.size();
            tiles.emplace_back();
            std::memset(&tiles.back(), 0xFF, sizeof(Tile));
            tile_ix[ti] = t;
        }
        Tile& tl = tiles[t];
        uint32_t r = rank & (B - 1), c = a.offset & (B - 1);
        switch (st) {
            case ST_M: return &tl.m[r][c];
            case ST_I: return &tl.i[r][c];
            case ST_D: return &tl.d[r][c];
            case ST_I2: return &tl.i2[r][c];
            default: return &tl.d2[r][c];
        }
    }
    // gap_affine.rs:483-500
    inline Score get_score(const AlnNode& a, AlignState st) const {
        Score* p = const_cast<VisitedStorage*>(this)->cell(a, st, false);
        return p ? *p : UNVISITED;
    }
    // gap_affine.rs:503-517
    inline void set_score(const AlnNode& a, AlignState st, Score s) { *cell(a, st, true) = s; }
    // gap_affine.rs:520-548 — strict '<' only
    inline bool update_score_if_lower(const AlnNode& a, AlignState st, Score s) {
        Score* p = cell(a, st, true);
        if (s < *p) { *p = s; return true; }
        return false;
    }
};

// ---------------------------------------------------------------------------
// queue.rs:19-71 + gap_affine.rs:929-1013
struct QueuedItem { Score score; AlnNode node; AlignState state; };

struct QueueLayer {
    std::vector<std::pair<Score, AlnNode>> m, i, d, i2, d2;
    bool empty() const { return m.empty() && d.empty() && i.empty() && d2.empty() && i2.empty(); }
    void queue(const QueuedItem& it) {
        auto& v = it.state == ST_M ? m : it.state == ST_I ? i : it.state == ST_D ? d : it.state == ST_I2 ? i2 : d2;
        v.push_back({it.score, it.node});
    }
    // gap_affine.rs:954-966: M stack, else D stack, else I stack — LIFO each;
    // gap_affine_2piece.rs:1072-1097: M, D1, D2, I1, I2 (the second-piece stacks stay empty in the one-piece model)
    bool pop(QueuedItem& out) {
        auto take = [&](std::vector<std::pair<Score, AlnNode>>& v, AlignState st) {
            if (v.empty()) return false;
            out = {v.back().first, v.back().second, st}; v.pop_back(); return true;
        };
        return take(m, ST_M) || take(d, ST_D) || take(d2, ST_D2) || take(i, ST_I) || take(i2, ST_I2);
    }
};

struct LayeredQueue {
    std::deque<QueueLayer> layers;
    size_t layer_min = 0;
    // queue.rs:31-54
    void queue(const QueuedItem& item, size_t priority) {
        if (layers.empty()) {
            layers.emplace_back();
            layer_min = priority;
        } else {
            size_t layer_max = layer_min + layers.size();
            if (priority < layer_min) {
                size_t diff = layer_min - priority;
                for (size_t k = 0; k < diff; ++k) layers.emplace_front();
                layer_min = priority;
            } else if (priority >= layer_max) {
                layers.resize(priority - layer_min + 1);
            }
        }
        layers[priority - layer_min].queue(item);
    }
    // queue.rs:56-70
    bool pop(QueuedItem& out) {
        if (layers.empty()) return false;
        bool got = layers[0].pop(out);
        while (!layers.empty()) {
            if (layers[0].empty()) { layers.pop_front(); layer_min += 1; }
            else break;
        }
        return got;
    }
};

// ---------------------------------------------------------------------------
struct AstarResult {  // astar.rs:81-90
    Score score = 0;
    std::vector<AlignedPair> alignment;
    size_t num_queued = 0, num_visited = 0, num_pruned = 0;
};

// AlignmentType, scoring/mod.rs:50-62 (std::ops::Bound<usize> per end)
enum BoundKind : uint32_t { BOUND_UNBOUNDED = 0, BOUND_INCLUDED = 1, BOUND_EXCLUDED = 2 };
struct Bound { uint32_t kind = BOUND_UNBOUNDED; uint64_t v = 0; };
struct AlnType {
    bool ends_free = false;  // false: Global
    Bound qry_free_begin, qry_free_end, graph_free_begin, graph_free_end;
};

class Aligner {
public:
    const Graph& g;
    const BubbleIndex& bubbles;
    Costs costs;
    Heuristic heuristic;
    bool enable_pruning;
    AlnType aln_type;  // Global unless set
    std::vector<uint32_t> ranks;

    // per-alignment state (AffineAstarData, gap_affine.rs:702-712)
    const uint8_t* seq = nullptr;
    size_t seq_len = 0;
    VisitedStorage visited;
    std::vector<std::set<uint32_t>> bubbles_reached_m;
    std::vector<uint32_t> touched_exits;

    Aligner(const Graph& graph, const BubbleIndex& bi, Costs c, Heuristic h, bool prune)
        : g(graph), bubbles(bi), costs(c), heuristic(h), enable_pruning(prune), ranks(graph.node_ranks()) {
        bubbles_reached_m.resize(g.node_count_with_start_and_end());
    }

    // heuristic.rs:70-102 (MinimumGapCostAffine) / :41-46 (Dijkstra)
    size_t h(const AlnNode& a, AlignState st) const {
        if (heuristic == H_DIJKSTRA) return 0;
        size_t mn = bubbles.get_min_dist_to_end(a.node); mn = mn ? mn - 1 : 0;
        size_t mx = bubbles.get_max_dist_to_end(a.node); mx = mx ? mx - 1 : 0;
        size_t tmin = (size_t)a.offset + mn, tmax = (size_t)a.offset + mx;
        size_t gap;
        if (tmin > seq_len) {
            gap = tmin - seq_len;
            if (st != ST_D) st = ST_M;
        } else if (tmax < seq_len) {
            gap = seq_len - tmax;
            if (st != ST_I) st = ST_M;
        } else {
            gap = 0;
        }
        return costs.gap_cost(st, gap);
    }

    // gap_affine.rs:767-777
    void mark_reached(const AlnNode& a, AlignState st) {
        if (st == ST_M && bubbles.is_exit(a.node)) {
            auto& s = bubbles_reached_m[a.node];
            if (s.empty()) touched_exits.push_back(a.node);
            s.insert(a.offset);
        }
    }

    // reached.rs:191-255
    bool can_improve_at_offset(uint32_t bubble_node, uint32_t offset_to_check, Score score,
                               const uint32_t* left, const uint32_t* right, size_t min_dist_to_end) const {
        bool have = false; Score implicit = 0;
        if (left && right) {
            Score ls = visited.get_score({bubble_node, *left}, ST_M);
            Score rs = visited.get_score({bubble_node, *right}, ST_M);
            uint32_t gl = offset_to_check - *left, gr = *right - offset_to_check;
            Score from_left = score_add(ls, costs.gap_cost(ST_M, gl));
            Score from_right = score_add(rs, costs.gap_cost(ST_M, gr));
            implicit = ((size_t)gr > min_dist_to_end) ? from_left : std::min(from_left, from_right);
            have = true;
        } else if (!left && right) {
            Score rs = visited.get_score({bubble_node, *right}, ST_M);
            uint32_t gr = *right - offset_to_check;
            Score from_right = score_add(rs, costs.gap_cost(ST_M, gr));
            if ((size_t)gr > min_dist_to_end) have = false;
            else { implicit = from_right; have = true; }
        } else if (left && !right) {
            Score ls = visited.get_score({bubble_node, *left}, ST_M);
            uint32_t gl = offset_to_check - *left;
            implicit = score_add(ls, costs.gap_cost(ST_M, gl));
            have = true;
        }
        return have ? (score < implicit) : true;
    }

    // reached.rs:38-189
    bool can_improve_bubble(const NodeBubbleMap& bubble, const AlnNode& a, AlignState st, Score current) const {
        const std::set<uint32_t>& reached = bubbles_reached_m[bubble.bubble_exit];
        if (reached.empty()) return true;
        if (a.node == bubble.bubble_exit) return true;
        uint32_t tmin = a.offset + (uint32_t)bubble.min_dist_to_exit;
        uint32_t tmax = a.offset + (uint32_t)bubble.max_dist_to_exit;
        size_t mde = bubbles.get_min_dist_to_end(bubble.bubble_exit);
        mde = mde ? mde - 1 : 0;
        if ((size_t)tmax > seq_len) return true;

        // prev_reached = reached.range(..tmin).next_back()
        const uint32_t* prev = nullptr;
        {
            auto it = reached.lower_bound(tmin);
            if (it != reached.begin()) { --it; prev = &*it; }
        }
        bool have_last = false; uint32_t last_offset = 0;
        if (tmin > tmax) throw RefPanic("BTreeSet::range start > end");
        for (auto it = reached.lower_bound(tmin); it != reached.end() && *it <= tmax; ++it) {
            const uint32_t* next = &*it;
            uint32_t offset1 = prev ? std::max(tmin, *prev + 1u) : tmin;
            // reached.rs:84-101 (a state inside a gap does not pay the open cost again; the second piece: gap_open2)
            if (st == ST_D || st == ST_D2) {
                Score c = visited.get_score({bubble.bubble_exit, *next}, ST_M);
                if (score_add(c, st == ST_D ? costs.gap_open : costs.gap_open2) > current) return true;
            }
            // reached.rs:104-124
            if (prev && (st == ST_I || st == ST_I2)) {
                Score c = visited.get_score({bubble.bubble_exit, *prev}, ST_M);
                if (score_add(c, st == ST_I ? costs.gap_open : costs.gap_open2) > current) return true;
            }
            if (can_improve_at_offset(bubble.bubble_exit, offset1, current, prev, next, mde)) return true;
            uint32_t offset2 = std::min(tmax, std::max(tmin, *next - 1u));  // wrapping u32 sub
            if (offset2 != offset1) {
                if (can_improve_at_offset(bubble.bubble_exit, offset2, current, prev, next, mde)) return true;
            }
            prev = next;
            last_offset = offset2; have_last = true;
        }
        const uint32_t* next = nullptr;
        {
            uint32_t from = (tmax == 0xFFFFFFFFu) ? tmax : tmax + 1u;  // saturating_add
            auto it = reached.lower_bound(from);
            if (it != reached.end()) next = &*it;
        }
        if (!have_last && can_improve_at_offset(bubble.bubble_exit, tmin, current, prev, next, mde)) return true;
        if ((!have_last || last_offset < tmax) &&
            can_improve_at_offset(bubble.bubble_exit, tmax, current, prev, next, mde)) return true;
        if (prev && (st == ST_I || st == ST_I2)) {   // reached.rs:165-186
            Score c = visited.get_score({bubble.bubble_exit, *prev}, ST_M);
            if (score_add(c, st == ST_I ? costs.gap_open : costs.gap_open2) > current) return true;
        }
        return false;
    }

    // gap_affine.rs:780-792
    int forced_prune = -1;  // test hook == DummyVisited::prune_next (dfa.rs:351-378); -1 = real rule
    bool prune(Score score, const AlnNode& a, AlignState st) const {
        if (forced_prune >= 0) return forced_prune != 0;
        if (!bubbles.node_is_part_of_bubble(a.node)) return false;
        for (const auto& b : bubbles.node_bubble_map[a.node])
            if (!can_improve_bubble(b, a, st, score)) return true;
        return false;
    }

    // dist_to_end, gap_affine.rs:91-119: BFS over successors, nodes at distance >= max are not expanded
    bool dist_to_end(uint32_t from, size_t max, size_t& out) const {
        std::vector<std::pair<uint32_t, size_t>> queue;
        std::vector<uint8_t> seen(g.node_count_with_start_and_end(), 0);
        queue.push_back({from, 0});
        seen[from] = 1;
        for (size_t head = 0; head < queue.size(); ++head) {
            const auto [n, dist] = queue[head];
            if (n == g.end) { out = dist; return true; }
            if (dist >= max) continue;
            for (uint32_t sn : g.succ[n])
                if (!seen[sn]) { seen[sn] = 1; queue.push_back({sn, dist + 1}); }
        }
        return false;
    }

    // gap_affine.rs:136-183
    std::vector<AlnNode> initial_states() const {
        std::vector<AlnNode> init;
        if (!aln_type.ends_free) { init.push_back({g.start, 0}); return init; }
        if (aln_type.graph_free_begin.kind == BOUND_UNBOUNDED) {
            // every real node at offset 0, pushed in REVERSE index order ("queue processes in LIFO order")
            const uint32_t n = (uint32_t)g.node_count_with_start_and_end();
            for (uint32_t v = 0; v < n; ++v)
                if (v != g.start && v != g.end) init.push_back({v, 0});
            std::reverse(init.begin(), init.end());
        } else {
            init.push_back({g.start, 0});
        }
        if (init.empty()) init.push_back({g.start, 0});
        return init;
    }

    // gap_affine.rs:185-248
    bool is_end(const AlnNode& a, AlignState st) const {
        if (!aln_type.ends_free) return st == ST_M && a.node == g.end && (size_t)a.offset == seq_len;
        bool q_ok;
        const Bound& qe = aln_type.qry_free_end;
        if (qe.kind == BOUND_UNBOUNDED) q_ok = costs.two_piece ? ((size_t)a.offset >= seq_len || seq_len == 0)   // gap_affine_2piece.rs:246-250
                                                               : (a.offset > 0 || seq_len == 0);          // sic: ANY consumed prefix may end
        else if (qe.kind == BOUND_INCLUDED) q_ok = seq_len - (size_t)a.offset <= qe.v;
        else q_ok = seq_len - (size_t)a.offset < qe.v;
        bool g_ok;
        const Bound& ge = aln_type.graph_free_end;
        size_t d = 0;
        if (ge.kind == BOUND_UNBOUNDED) g_ok = true;
        else if (ge.kind == BOUND_INCLUDED) g_ok = dist_to_end(a.node, ge.v, d) && d <= ge.v;
        else g_ok = dist_to_end(a.node, ge.v ? ge.v - 1 : 0, d) && d < ge.v;
        return st == ST_M && q_ok && g_ok;
    }

    struct Ctx {
        LayeredQueue queue;
        AstarResult result;
    };

    void queue_state(Ctx& c, const AlnNode& succ, AlignState st, Score new_score) {
        // astar.rs:178-182 closure + gap_affine.rs:1005-1012
        size_t hh = h(succ, st);
        c.result.num_queued += 1;
        c.queue.queue({new_score, succ, st}, (size_t)new_score + hh);
    }

    // gap_affine.rs:346-367
    void expand_ref_graph_end(Ctx& c, const AlnNode& parent, Score score) {
        AlnNode ins{parent.node, parent.offset + 1};
        Score ns = score_add(score_add(score, costs.gap_open), costs.gap_extend);
        if (visited.update_score_if_lower(ins, ST_I, ns)) queue_state(c, ins, ST_I, ns);
    }
    // gap_affine.rs:369-391
    void expand_query_end(Ctx& c, const AlnNode& parent, uint32_t child, Score score) {
        AlnNode del{child, parent.offset};
        Score ns = score_add(score_add(score, costs.gap_open), costs.gap_extend);
        if (visited.update_score_if_lower(del, ST_D, ns)) queue_state(c, del, ST_D, ns);
    }
    // gap_affine.rs:393-430
    void expand_mismatch(Ctx& c, const AlnNode& parent, const AlnNode& child, Score score) {
        Score nm = score_add(score, costs.mismatch);
        if (visited.update_score_if_lower(child, ST_M, nm)) queue_state(c, child, ST_M, nm);
        AlnNode ins{parent.node, parent.offset + 1};
        Score ni = score_add(score_add(score, costs.gap_open), costs.gap_extend);
        if (visited.update_score_if_lower(ins, ST_I, ni)) queue_state(c, ins, ST_I, ni);
        AlnNode del{child.node, parent.offset};
        Score nd = score_add(score_add(score, costs.gap_open), costs.gap_extend);
        if (visited.update_score_if_lower(del, ST_D, nd)) queue_state(c, del, ST_D, nd);
    }
    // gap_affine.rs:307-341 (the Match arm :265-306 is unreachable from astar_alignment);
    // gap_affine_2piece.rs:346-431 for the two-piece model: a gap opens in the first piece (open1 + extend1), every further
    // step either stays in its piece or moves from the first to the second (cost extend2; open2 is never charged)
    void expand_all(Ctx& c, Score score, const AlnNode& node, AlignState st) {
        if (st == ST_M) throw RefPanic("expand_all(Match) is unreachable from astar_alignment");
        if (visited.update_score_if_lower(node, ST_M, score)) queue_state(c, node, ST_M, score);
        if (st == ST_I || st == ST_I2) {
            AlnNode ins{node.node, node.offset + 1};
            if (st == ST_I) {
                Score ns = score_add(score, costs.gap_extend);
                if ((size_t)node.offset < seq_len && visited.update_score_if_lower(ins, ST_I, ns))
                    queue_state(c, ins, ST_I, ns);
            }
            if (costs.two_piece) {
                Score ns2 = score_add(score, costs.gap_extend2);
                if ((size_t)node.offset < seq_len && visited.update_score_if_lower(ins, ST_I2, ns2))
                    queue_state(c, ins, ST_I2, ns2);
            }
        } else {
            for (uint32_t s : g.succ[node.node]) {
                AlnNode del{s, node.offset};
                if (st == ST_D) {
                    Score ns = score_add(score, costs.gap_extend);
                    if (visited.update_score_if_lower(del, ST_D, ns)) queue_state(c, del, ST_D, ns);
                }
                if (costs.two_piece) {
                    Score ns2 = score_add(score, costs.gap_extend2);
                    if (visited.update_score_if_lower(del, ST_D2, ns2)) queue_state(c, del, ST_D2, ns2);
                }
            }
        }
    }

    // dfa.rs:86-251
    struct DFA {
        enum Kind { NONE, REF_GRAPH_END, QUERY_END, MISMATCH };
        struct Event { Kind kind; AlnNode parent; AlnNode child; };
        struct StackNode { AlnNode node; size_t it; };
        Aligner& A;
        Score score;
        size_t num_visited = 0, num_pruned = 0;
        std::vector<StackNode> stack;
        DFA(Aligner& a, Score s, const AlnNode& start) : A(a), score(s) { stack.push_back({start, 0}); }

        // dfa.rs:138-208
        Event extend() {
            const Graph& g = A.g;
            if (stack.size() == 1 && A.seq_len != 0) {
                AlnNode initial = stack[0].node;
                if (initial.offset == 0) {
                    if (g.is_symbol_equal(initial.node, A.seq[0])) {
                        AlnNode match{initial.node, initial.offset + 1};
                        if (A.visited.update_score_if_lower(match, ST_M, score)) {
                            stack[0] = {match, 0};
                            A.mark_reached(match, ST_M);  // dfa_match
                            num_visited += 1;
                            if ((size_t)match.offset == A.seq_len)
                                return {REF_GRAPH_END, initial, match};
                        }
                    }
                }
            }
            while (!stack.empty()) {
                // next_valid_successor, dfa.rs:210-250
                StackNode& parent = stack.back();
                const auto& children = g.succ[parent.node.node];
                bool descended = false;
                while (parent.it < children.size()) {
                    uint32_t child = children[parent.it++];
                    if (child == g.end) {
                        AlnNode term{child, parent.node.offset};
                        A.visited.update_score_if_lower(term, ST_M, score);
                        return {REF_GRAPH_END, parent.node, term};
                    }
                    if ((size_t)parent.node.offset >= A.seq_len)
                        return {QUERY_END, parent.node, AlnNode{child, 0}};
                    AlnNode cn{child, parent.node.offset + 1};
                    if (g.is_symbol_equal(child, A.seq[cn.offset - 1])) {
                        if (A.visited.update_score_if_lower(cn, ST_M, score)) {
                            // Successor::Match — dfa.rs:182-194
                            if (A.prune(score, cn, ST_M)) { num_pruned += 1; descended = true; break; }
                            A.mark_reached(cn, ST_M);
                            num_visited += 1;
                            stack.push_back({cn, 0});  // invalidates `parent`
                            descended = true;
                            break;
                        }
                    } else {
                        return {MISMATCH, parent.node, cn};
                    }
                }
                if (!descended) stack.pop_back();  // SuccessorsExhausted
            }
            return {NONE, {0, 0}, {0, 0}};
        }
    };

    // gap_affine.rs:550-657.  Returns false for None.
    bool get_backtrace(const AlnNode& a, AlignState st, AlnNode& out, AlignState& out_st) const {
        if (costs.two_piece) return get_backtrace2(a, st, out, out_st);
        Score cs = visited.get_score(a, st);
        if (cs == UNVISITED) return false;
        if (st == ST_M) {
            if (a.offset > 0) {
                bool match_or_end = g.is_symbol_equal(a.node, seq[a.offset - 1]) || a.node == g.end;
                uint32_t po = (a.node == g.end) ? a.offset : a.offset - 1;
                for (uint32_t p : g.pred[a.node]) {
                    Score ps = visited.get_score({p, po}, ST_M);
                    if ((match_or_end && ps == cs) ||
                        (!match_or_end && ps == score_sub(cs, costs.mismatch))) {
                        out = {p, po}; out_st = ST_M; return true;
                    }
                }
            }
            if (visited.get_score(a, ST_D) == cs) { out = a; out_st = ST_D; return true; }
            if (visited.get_score(a, ST_I) == cs) { out = a; out_st = ST_I; return true; }
        } else if (st == ST_D) {
            for (uint32_t p : g.pred[a.node]) {
                Score ps = visited.get_score({p, a.offset}, ST_M);
                if (ps == score_sub(score_sub(cs, costs.gap_open), costs.gap_extend)) {
                    out = {p, a.offset}; out_st = ST_M; return true;
                }
            }
            for (uint32_t p : g.pred[a.node]) {
                Score ps = visited.get_score({p, a.offset}, ST_D);
                if (ps == score_sub(cs, costs.gap_extend)) { out = {p, a.offset}; out_st = ST_D; return true; }
            }
        } else {
            if (a.offset > 0) {
                AlnNode pr{a.node, a.offset - 1};
                if (visited.get_score(pr, ST_M) == score_sub(score_sub(cs, costs.gap_open), costs.gap_extend)) {
                    out = pr; out_st = ST_M; return true;
                }
                if (visited.get_score(pr, ST_I) == score_sub(cs, costs.gap_extend)) {
                    out = pr; out_st = ST_M;  // sic: gap_affine.rs:649 returns Match
                    return true;
                }
            }
        }
        return false;
    }

    // gap_affine_2piece.rs:639-794
    bool get_backtrace2(const AlnNode& a, AlignState st, AlnNode& out, AlignState& out_st) const {
        Score cs = visited.get_score(a, st);
        if (cs == UNVISITED) return false;
        const uint32_t o1 = costs.gap_open, e1 = costs.gap_extend, e2 = costs.gap_extend2;
        if (st == ST_M) {
            if (a.offset > 0) {
                bool match_or_end = g.is_symbol_equal(a.node, seq[a.offset - 1]) || a.node == g.end;
                uint32_t po = (a.node == g.end) ? a.offset : a.offset - 1;
                for (uint32_t p : g.pred[a.node]) {
                    Score ps = visited.get_score({p, po}, ST_M);
                    if ((match_or_end && ps == cs) || (!match_or_end && ps == score_sub(cs, costs.mismatch))) {
                        out = {p, po}; out_st = ST_M; return true;
                    }
                }
            }
            for (AlignState gs : {ST_D, ST_D2, ST_I, ST_I2})
                if (visited.get_score(a, gs) == cs) { out = a; out_st = gs; return true; }
        } else if (st == ST_D) {
            for (uint32_t p : g.pred[a.node])
                if (visited.get_score({p, a.offset}, ST_M) == score_sub(score_sub(cs, o1), e1)) { out = {p, a.offset}; out_st = ST_M; return true; }
            for (uint32_t p : g.pred[a.node])
                if (visited.get_score({p, a.offset}, ST_D) == score_sub(cs, e1)) { out = {p, a.offset}; out_st = ST_D; return true; }
        } else if (st == ST_D2) {
            for (uint32_t p : g.pred[a.node])
                if (visited.get_score({p, a.offset}, ST_D) == score_sub(cs, e2)) { out = {p, a.offset}; out_st = ST_D; return true; }
            for (uint32_t p : g.pred[a.node])
                if (visited.get_score({p, a.offset}, ST_D2) == score_sub(cs, e2)) { out = {p, a.offset}; out_st = ST_D2; return true; }
        } else if (st == ST_I) {
            if (a.offset > 0) {
                AlnNode pr{a.node, a.offset - 1};
                if (visited.get_score(pr, ST_M) == score_sub(score_sub(cs, o1), e1)) { out = pr; out_st = ST_M; return true; }
                if (visited.get_score(pr, ST_I) == score_sub(cs, e1)) { out = pr; out_st = ST_I; return true; }
            }
        } else {
            if (a.offset > 0) {
                AlnNode pr{a.node, a.offset - 1};
                if (visited.get_score(pr, ST_I) == score_sub(cs, e2)) { out = pr; out_st = ST_I; return true; }
                if (visited.get_score(pr, ST_I2) == score_sub(cs, e2)) { out = pr; out_st = ST_I2; return true; }
            }
        }
        return false;
    }

    // gap_affine.rs:804-915  /  gap_affine_2piece.rs:944-1043
    std::vector<AlignedPair> backtrace(const AlnNode& end_cell) const {
        std::vector<AlignedPair> aln;
        if (seq_len == 0) return aln;
        if (seq_len == 1 && end_cell.offset == 1) {
            if (g.is_symbol_equal(end_cell.node, seq[0])) {
                aln.push_back({end_cell.node, 0});
                return aln;
            }
        }
        AlnNode curr; AlignState cst;
        bool ok = get_backtrace(end_cell, ST_M, curr, cst) ||
                  get_backtrace(end_cell, ST_I, curr, cst) ||
                  (costs.two_piece && get_backtrace(end_cell, ST_I2, curr, cst)) ||
                  get_backtrace(end_cell, ST_D, curr, cst) ||
                  (costs.two_piece && get_backtrace(end_cell, ST_D2, curr, cst));
        if (!ok) {
            if (seq_len <= 3 && !costs.two_piece) {  // (the two-piece file has this fallback commented out, :1025-1036)
                for (size_t i = 0; i < seq_len; ++i) aln.push_back({end_cell.node, (uint32_t)i});
                return aln;
            }
            throw RefPanic("No backtrace for alignment end state?");
        }
        AlnNode bt; AlignState bst;
        while (get_backtrace(curr, cst, bt, bst)) {
            if (cst == ST_M && bst != ST_M) { curr = bt; cst = bst; continue; }
            if (cst == ST_M) aln.push_back({curr.node, curr.offset - 1});
            else if (cst == ST_I || cst == ST_I2) aln.push_back({NONE32, curr.offset - 1});
            else aln.push_back({curr.node, NONE32});
            if (bt.node == g.start) break;
            curr = bt; cst = bst;
        }
        std::reverse(aln.begin(), aln.end());
        return aln;
    }

    // astar.rs:108-226 (Global)
    AstarResult astar_alignment(const uint8_t* s, size_t len) {
        seq = s; seq_len = len;
        visited.init(g, ranks, len);
        for (uint32_t v : touched_exits) bubbles_reached_m[v].clear();
        touched_exits.clear();

        Ctx c;
        for (const AlnNode& init : initial_states()) {  // astar.rs:133-139
            c.queue.queue({0, init, ST_M}, 0 + h(init, ST_M));
            visited.set_score(init, ST_M, 0);
            c.result.num_queued += 1;
        }

        Score end_score; AlnNode end_node;
        for (;;) {
            QueuedItem it;
            if (!c.queue.pop(it)) throw RefPanic("Could not align sequence! Empty queue before reaching end!");
            if (it.score > visited.get_score(it.node, it.state)) continue;
            if (is_end(it.node, it.state)) {
                c.result.num_visited += 1;
                end_score = it.score; end_node = it.node;
                break;
            }
            if (enable_pruning && prune(it.score, it.node, it.state)) { c.result.num_pruned += 1; continue; }
            mark_reached(it.node, it.state);
            c.result.num_visited += 1;
            if (it.state == ST_M) {
                DFA dfa(*this, it.score, it.node);
                bool done = false;
                for (;;) {
                    DFA::Event ev = dfa.extend();
                    if (ev.kind == DFA::NONE) break;
                    if (ev.kind == DFA::REF_GRAPH_END) {
                        if (is_end(ev.child, ST_M)) { end_score = it.score; end_node = ev.child; done = true; break; }
                        expand_ref_graph_end(c, ev.parent, it.score);
                    } else if (ev.kind == DFA::QUERY_END) {
                        expand_query_end(c, ev.parent, ev.child.node, it.score);
                    } else {
                        expand_mismatch(c, ev.parent, ev.child, it.score);
                    }
                }
                if (done) break;  // NB: `break 'main` skips `num_visited += dfa.get_num_visited()`
                c.result.num_visited += dfa.num_visited;
            } else {
                expand_all(c, it.score, it.node, it.state);
            }
        }
        c.result.score = end_score;
        c.result.alignment = backtrace(end_node);
        return std::move(c.result);
    }

    // PoastaAligner::align, mod.rs:114-145 (empty-graph shortcut :124-142)
    AstarResult align(const uint8_t* s, size_t len) {
        if (g.is_poa && g.node_count() == 0) {
            AstarResult r;
            r.score = len == 0 ? 0 : (Score)(len * 4);
            return r;
        }
        return astar_alignment(s, len);
    }
};

}  // namespace poa_oracle
