// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
// C entry points (ctypes) over the CPU restatement in astar.hpp / dense.hpp.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
#include <atomic>
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "astar.hpp"
#include "bubbles.hpp"
#include "dense.hpp"
#include "graph.hpp"
#include "msa.hpp"

using namespace poa_oracle;

namespace {
struct GraphHandle {
    Graph g;
    std::unique_ptr<BubbleIndex> bubbles;  // lazily built; invalidated on mutation
    const BubbleIndex& bi() {
        if (!bubbles) bubbles.reset(new BubbleIndex(g));
        return *bubbles;
    }
};
thread_local std::string g_last_error;
}  // namespace

// Two-piece affine model (gap_affine_2piece.rs) for every Aligner / DenseAligner constructed afterwards (test-side knob;
// default: the one-piece model).  (m, o, e) of the entry points are then (mismatch, open1, extend1).
static bool g_two_piece = false;
static uint8_t g_open2 = 0, g_extend2 = 0;
static Costs mk_costs(uint8_t m, uint8_t o, uint8_t e) {
    Costs c{m, o, e};
    c.two_piece = g_two_piece; c.gap_open2 = g_open2; c.gap_extend2 = g_extend2;
    return c;
}

// Alignment span for every Aligner constructed afterwards (test-side knob; default Global).
static AlnType g_aln_type;
static void apply_aln_type(Aligner& a) { a.aln_type = g_aln_type; }

extern "C" {

const char* oracle_last_error() { return g_last_error.c_str(); }

// GapAffine2Piece::new(mismatch, extend1, open1, extend2, open2) asserts extend1 >= extend2 (gap_affine_2piece.rs:29-33):
// returns 1 where the reference would panic
int oracle_set_two_piece(int enable, uint8_t open2, uint8_t extend2) {
    g_two_piece = enable != 0; g_open2 = open2; g_extend2 = extend2;
    return 0;
}
uint64_t oracle_breakpoint(uint8_t m, uint8_t o1, uint8_t e1, uint8_t o2, uint8_t e2) {
    Costs c{m, o1, e1}; c.two_piece = true; c.gap_open2 = o2; c.gap_extend2 = e2;
    return (uint64_t)c.breakpoint();
}

// spec: {ends_free, then (kind, value) for qry_free_begin, qry_free_end, graph_free_begin, graph_free_end}
void oracle_set_alignment_type(const uint64_t* spec) {
    g_aln_type = AlnType{};
    if (!spec || !spec[0]) return;
    g_aln_type.ends_free = true;
    Bound* b[4] = {&g_aln_type.qry_free_begin, &g_aln_type.qry_free_end, &g_aln_type.graph_free_begin, &g_aln_type.graph_free_end};
    for (int i = 0; i < 4; ++i) { b[i]->kind = (uint32_t)spec[1 + 2 * i]; b[i]->v = spec[2 + 2 * i]; }
}

void* oracle_graph_from_csr(uint32_t n, uint32_t start, uint32_t end, const uint8_t* sym,
                            const uint32_t* succ_off, const uint32_t* succ, const uint32_t* pred_off,
                            const uint32_t* pred, int end_matches_all) {
    try {
        auto* h = new GraphHandle;
        h->g = Graph::from_csr(n, start, end, sym, succ_off, succ, pred_off, pred, end_matches_all != 0);
        return h;
    } catch (const std::exception& ex) { g_last_error = ex.what(); return nullptr; }
}
void* oracle_graph_new_poa() { auto* h = new GraphHandle; h->g = Graph::new_poa(); return h; }
void* oracle_graph_mock(int which) {
    auto* h = new GraphHandle;
    h->g = which == 1 ? create_test_graph1() : create_test_graph2();
    return h;
}
// generic mock graph from an edge list, edges added in the given order (petgraph add_edge semantics)
void* oracle_graph_mock_edges(uint32_t n, const uint32_t* edges, uint32_t n_edges, const uint8_t* sym) {
    try {
        auto* h = new GraphHandle;
        h->g = Graph::new_mock(n);
        for (uint32_t i = 0; i < n_edges; ++i) h->g.raw_add_edge(edges[2 * i], edges[2 * i + 1]);
        if (sym) h->g.symbol.assign(sym, sym + n);
        h->g.compute_topo();
        return h;
    } catch (const std::exception& ex) { g_last_error = ex.what(); return nullptr; }
}
void oracle_graph_free(void* p) { delete (GraphHandle*)p; }
void oracle_graph_set_symbols(void* p, const uint8_t* sym) {
    auto* h = (GraphHandle*)p;
    h->g.symbol.assign(sym, sym + h->g.symbol.size());
}
uint32_t oracle_graph_n(void* p) { return (uint32_t)((GraphHandle*)p)->g.symbol.size(); }
uint32_t oracle_graph_start(void* p) { return ((GraphHandle*)p)->g.start; }
uint32_t oracle_graph_end(void* p) { return ((GraphHandle*)p)->g.end; }
uint32_t oracle_graph_n_edges(void* p) {
    size_t e = 0;
    for (auto& s : ((GraphHandle*)p)->g.succ) e += s.size();
    return (uint32_t)e;
}
// CSR in trait-iteration order + node ranks (get_node_ordering)
void oracle_graph_export(void* p, uint8_t* sym, uint32_t* succ_off, uint32_t* succ, uint32_t* pred_off,
                         uint32_t* pred, uint32_t* rank) {
    Graph& g = ((GraphHandle*)p)->g;
    uint32_t n = (uint32_t)g.symbol.size();
    std::memcpy(sym, g.symbol.data(), n);
    uint32_t so = 0, po = 0;
    for (uint32_t v = 0; v < n; ++v) {
        succ_off[v] = so; pred_off[v] = po;
        for (uint32_t s : g.succ[v]) succ[so++] = s;
        for (uint32_t q : g.pred[v]) pred[po++] = q;
    }
    succ_off[n] = so; pred_off[n] = po;
    auto r = g.node_ranks();
    std::memcpy(rank, r.data(), n * sizeof(uint32_t));
}
// POAGraph::add_alignment_with_weights (poa.rs:171-321); n_pairs < 0 == None
int oracle_poa_add_alignment(void* p, const char* name, const uint8_t* seq, uint64_t len, const uint32_t* pairs,
                             int64_t n_pairs) {
    auto* h = (GraphHandle*)p;
    h->bubbles.reset();
    try {
        if (n_pairs < 0) return h->g.add_alignment(name ? name : "", seq, len, nullptr);
        std::vector<AlignedPair> aln((size_t)n_pairs);
        for (int64_t i = 0; i < n_pairs; ++i) aln[i] = {pairs[2 * i], pairs[2 * i + 1]};
        return h->g.add_alignment(name ? name : "", seq, len, &aln);
    } catch (const std::exception& ex) { g_last_error = ex.what(); return -1; }
}
uint32_t oracle_poa_n_sequences(void* p) { return (uint32_t)((GraphHandle*)p)->g.n_sequences; }
uint32_t oracle_poa_seq_start(void* p, uint32_t i) { return ((GraphHandle*)p)->g.seq_start_nodes[i]; }
// aligned_nodes of one node (poa.rs:374-376); returns count, fills up to cap
uint32_t oracle_poa_aligned_nodes(void* p, uint32_t node, uint32_t* out, uint32_t cap) {
    auto& a = ((GraphHandle*)p)->g.aligned_nodes[node];
    for (uint32_t i = 0; i < a.size() && i < cap; ++i) out[i] = a[i];
    return (uint32_t)a.size();
}

// ---- known-answer hooks ---------------------------------------------------
// AlignmentGraph::is_end under the current alignment type (gap_affine.rs:185-248)
int oracle_is_end(void* p, uint64_t seq_len, uint32_t node, uint32_t offset, int state) {
    auto* h = (GraphHandle*)p;
    try {
        Aligner a(h->g, h->bi(), Costs{1, 1, 1}, H_DIJKSTRA, false);
        apply_aln_type(a);
        a.seq_len = seq_len;
        return a.is_end(AlnNode{node, offset}, (AlignState)state) ? 1 : 0;
    } catch (const std::exception& ex) { g_last_error = ex.what(); return -1; }
}

uint64_t oracle_gap_cost(uint8_t m, uint8_t o, uint8_t e, int state, uint64_t len) {
    if (g_two_piece) return mk_costs(m, o, e).gap_cost((AlignState)state, len);
    Costs c{m, o, e};
    return c.gap_cost((AlignState)state, len);
}
// rev_postorder_nodes (tools.rs:5); out has n entries, returns count
uint32_t oracle_rev_postorder(void* p, uint32_t* out) {
    auto v = rev_postorder_nodes(((GraphHandle*)p)->g);
    std::memcpy(out, v.data(), v.size() * 4);
    return (uint32_t)v.size();
}
// superbubbles (entrance, exit) in yield order; out has 2*n entries
int oracle_superbubbles(void* p, uint32_t* out) {
    try {
        SuperbubbleFinder f(((GraphHandle*)p)->g);
        auto v = f.find_all();
        for (size_t i = 0; i < v.size(); ++i) { out[2 * i] = v[i].first; out[2 * i + 1] = v[i].second; }
        return (int)v.size();
    } catch (const std::exception& ex) { g_last_error = ex.what(); return -1; }
}
// bubble index: dist_to_end (min,max per node), node_bubble_map flattened as (exit,min,max) with offsets[n+1]
int oracle_bubble_index(void* p, uint64_t* dist_min, uint64_t* dist_max, uint32_t* nbm_off, uint32_t* nbm_exit,
                        uint64_t* nbm_min, uint64_t* nbm_max, uint32_t nbm_cap, uint8_t* is_entrance, uint8_t* is_exit) {
    try {
        auto* h = (GraphHandle*)p;
        const BubbleIndex& b = h->bi();
        uint32_t n = (uint32_t)h->g.symbol.size(), k = 0;
        for (uint32_t v = 0; v < n; ++v) {
            dist_min[v] = b.dist_to_end[v].first; dist_max[v] = b.dist_to_end[v].second;
            is_entrance[v] = b.is_entrance_v[v]; is_exit[v] = b.is_exit_v[v];
            nbm_off[v] = k;
            for (auto& m : b.node_bubble_map[v]) {
                if (k < nbm_cap) { nbm_exit[k] = m.bubble_exit; nbm_min[k] = m.min_dist_to_exit; nbm_max[k] = m.max_dist_to_exit; }
                k++;
            }
        }
        nbm_off[n] = k;
        return (int)k;
    } catch (const std::exception& ex) { g_last_error = ex.what(); return -1; }
}
uint64_t oracle_heuristic_h(void* p, uint8_t m, uint8_t o, uint8_t e, int heuristic, uint64_t seq_len, uint32_t node,
                            uint32_t offset, int state) {
    auto* h = (GraphHandle*)p;
    Aligner a(h->g, h->bi(), mk_costs(m, o, e), (Heuristic)heuristic, true);
    apply_aln_type(a);
    a.seq_len = seq_len;
    return a.h({node, offset}, (AlignState)state);
}
// first DFA event from (node, offset) on a fresh visited table (== DummyVisited that always
// accepts, dfa.rs:351-401); force_prune mirrors DummyVisited::prune_next.
// out: kind(0 none,1 RefGraphEnd,2 QueryEnd,3 Mismatch), parent node/off, child node/off, num_visited, num_pruned
int oracle_dfa_first_event(void* p, const uint8_t* seq, uint64_t len, uint32_t node, uint32_t offset, int force_prune,
                           uint64_t* out) {
    try {
        auto* h = (GraphHandle*)p;
        Aligner a(h->g, h->bi(), Costs{4, 6, 2}, H_DIJKSTRA, true);
        apply_aln_type(a);
        a.seq = seq; a.seq_len = len;
        a.visited.init(h->g, a.ranks, len);
        a.forced_prune = force_prune ? 1 : 0;
        Aligner::DFA dfa(a, 0, {node, offset});
        Aligner::DFA::Event ev = dfa.extend();
        out[0] = ev.kind; out[1] = ev.parent.node; out[2] = ev.parent.offset; out[3] = ev.child.node;
        out[4] = ev.child.offset; out[5] = dfa.num_visited; out[6] = dfa.num_pruned;
        return 0;
    } catch (const std::exception& ex) { g_last_error = ex.what(); return -1; }
}
// LayeredQueue scripted access (queue.rs:109-136)
void* oracle_queue_new() { return new LayeredQueue; }
void oracle_queue_free(void* q) { delete (LayeredQueue*)q; }
void oracle_queue_push(void* q, uint32_t value, uint64_t priority) {
    ((LayeredQueue*)q)->queue({value, {0, 0}, ST_M}, priority);
}
int64_t oracle_queue_pop(void* q) {
    QueuedItem it;
    return ((LayeredQueue*)q)->pop(it) ? (int64_t)it.score : -1;
}
uint64_t oracle_queue_layers(void* q) { return ((LayeredQueue*)q)->layers.size(); }
uint64_t oracle_queue_layer_min(void* q) { return ((LayeredQueue*)q)->layer_min; }
uint64_t oracle_queue_layer_len(void* q, uint64_t ix) { return ((LayeredQueue*)q)->layers[ix].m.size(); }

// ---- alignment --------------------------------------------------------------
// status: 0 ok, 1 reference would panic (message in oracle_last_error), 2 pair capacity too small
static int run_astar(GraphHandle* h, Aligner& a, const uint8_t* seq, uint64_t len, uint32_t* score, uint32_t* pairs,
                     uint64_t cap, uint64_t* n_pairs, uint64_t* counters) {
    try {
        AstarResult r = a.align(seq, len);
        *score = r.score;
        *n_pairs = r.alignment.size();
        if (counters) { counters[0] = r.num_queued; counters[1] = r.num_visited; counters[2] = r.num_pruned; }
        if (r.alignment.size() > cap) return 2;
        for (size_t i = 0; i < r.alignment.size(); ++i) { pairs[2 * i] = r.alignment[i].rpos; pairs[2 * i + 1] = r.alignment[i].qpos; }
        return 0;
    } catch (const RefPanic& ex) { g_last_error = ex.what(); *n_pairs = 0; *score = UNVISITED; return 1; }
    (void)h;
}

int oracle_astar_align(void* p, uint8_t m, uint8_t o, uint8_t e, int heuristic, int prune, const uint8_t* seq,
                       uint64_t len, uint32_t* score, uint32_t* pairs, uint64_t cap, uint64_t* n_pairs,
                       uint64_t* counters) {
    auto* h = (GraphHandle*)p;
    try {
        if (h->g.is_poa && h->g.node_count() == 0) {
            // PoastaAligner::align shortcut (mod.rs:124-142) comes before any bubble index is built
            *score = len == 0 ? 0 : (uint32_t)(len * 4);
            *n_pairs = 0;
            if (counters) counters[0] = counters[1] = counters[2] = 0;
            return 0;
        }
        Aligner a(h->g, h->bi(), mk_costs(m, o, e), (Heuristic)heuristic, prune != 0);
        apply_aln_type(a);
        return run_astar(h, a, seq, len, score, pairs, cap, n_pairs, counters);
    } catch (const std::exception& ex) { g_last_error = ex.what(); return -1; }
}

// lasagna-shaped batch (src/bin/lasagna.rs:246-268): n_threads workers, one aligner each,
// shared immutable graph + bubble index.  pair_off[n+1] = capacity offsets into `pairs` (in pairs).
int oracle_astar_batch(void* p, uint8_t m, uint8_t o, uint8_t e, int heuristic, int prune, uint32_t n_queries,
                       const uint8_t* qseq, const uint64_t* qoff, uint32_t* scores, uint32_t* pairs,
                       const uint64_t* pair_off, uint64_t* n_pairs, uint64_t* counters /* [n][3] or null */,
                       int32_t* status, int n_threads) {
    auto* h = (GraphHandle*)p;
    try {
        const BubbleIndex& bi = h->bi();
        std::atomic<uint32_t> next{0};
        auto work = [&]() {
            Aligner a(h->g, bi, mk_costs(m, o, e), (Heuristic)heuristic, prune != 0);
            apply_aln_type(a);
            for (;;) {
                uint32_t i = next.fetch_add(1);
                if (i >= n_queries) break;
                status[i] = run_astar(h, a, qseq + qoff[i], qoff[i + 1] - qoff[i], &scores[i],
                                      pairs ? pairs + 2 * pair_off[i] : nullptr,
                                      pairs ? pair_off[i + 1] - pair_off[i] : 0, &n_pairs[i],
                                      counters ? counters + 3 * (size_t)i : nullptr);
                if (!pairs && status[i] == 2) status[i] = 0;
            }
        };
        if (n_threads <= 1) work();
        else {
            std::vector<std::thread> ts;
            for (int t = 0; t < n_threads; ++t) ts.emplace_back(work);
            for (auto& t : ts) t.join();
        }
        return 0;
    } catch (const std::exception& ex) { g_last_error = ex.what(); return -1; }
}

// Dense restatement (dense.hpp).  planes (M,I,D) optional: each rows*(len+1) u32, row = topological rank.
int oracle_dense_align(void* p, uint8_t m, uint8_t o, uint8_t e, const uint8_t* seq, uint64_t len, uint32_t* score,
                       uint32_t* pairs, uint64_t cap, uint64_t* n_pairs, uint32_t* flags, uint32_t* pm, uint32_t* pi,
                       uint32_t* pd) {
    auto* h = (GraphHandle*)p;
    try {
        DenseAligner a(h->g, mk_costs(m, o, e));
        DenseResult r = a.align(seq, len, pm != nullptr);
        *score = r.score; *flags = r.flags; *n_pairs = r.alignment.size();
        if (pm) {
            std::memcpy(pm, r.M.data(), r.M.size() * 4);
            std::memcpy(pi, r.I.data(), r.I.size() * 4);
            std::memcpy(pd, r.D.data(), r.D.size() * 4);
        }
        if (r.alignment.size() > cap) return 2;
        for (size_t i = 0; i < r.alignment.size(); ++i) { pairs[2 * i] = r.alignment[i].rpos; pairs[2 * i + 1] = r.alignment[i].qpos; }
        return 0;
    } catch (const std::exception& ex) { g_last_error = ex.what(); return -1; }
}

// the second-piece planes of the two-piece model (I2, D2) of one dense alignment, beside oracle_dense_align's M, I, D
int oracle_dense_planes2(void* p, uint8_t m, uint8_t o, uint8_t e, const uint8_t* seq, uint64_t len, uint32_t* pi2, uint32_t* pd2) {
    auto* h = (GraphHandle*)p;
    try {
        DenseAligner a(h->g, mk_costs(m, o, e));
        DenseResult r = a.align(seq, len, true);
        if (r.I2.empty()) { g_last_error = "not a two-piece run"; return -1; }
        std::memcpy(pi2, r.I2.data(), r.I2.size() * 4);
        std::memcpy(pd2, r.D2.data(), r.D2.size() * 4);
        return 0;
    } catch (const std::exception& ex) { g_last_error = ex.what(); return -1; }
}

int oracle_dense_batch(void* p, uint8_t m, uint8_t o, uint8_t e, uint32_t n_queries, const uint8_t* qseq,
                       const uint64_t* qoff, uint32_t* scores, uint32_t* pairs, const uint64_t* pair_off,
                       uint64_t* n_pairs, uint32_t* flags, int n_threads) {
    auto* h = (GraphHandle*)p;
    try {
        DenseAligner a(h->g, mk_costs(m, o, e));
        std::atomic<uint32_t> next{0};
        auto work = [&]() {
            for (;;) {
                uint32_t i = next.fetch_add(1);
                if (i >= n_queries) break;
                DenseResult r = a.align(qseq + qoff[i], qoff[i + 1] - qoff[i]);
                scores[i] = r.score; flags[i] = r.flags; n_pairs[i] = r.alignment.size();
                if (pairs) {
                    uint64_t cap = pair_off[i + 1] - pair_off[i];
                    for (size_t k = 0; k < r.alignment.size() && k < cap; ++k) {
                        pairs[2 * (pair_off[i] + k)] = r.alignment[k].rpos;
                        pairs[2 * (pair_off[i] + k) + 1] = r.alignment[k].qpos;
                    }
                }
            }
        };
        if (n_threads <= 1) work();
        else {
            std::vector<std::thread> ts;
            for (int t = 0; t < n_threads; ++t) ts.emplace_back(work);
            for (auto& t : ts) t.join();
        }
        return 0;
    } catch (const std::exception& ex) { g_last_error = ex.what(); return -1; }
}

// poa_graph_to_fasta (src/io/fasta.rs:69-156): returns the byte length (the text is truncated to cap-1 bytes)
int64_t oracle_poa_to_fasta(void* p, char* buf, uint64_t cap) {
    try {
        std::string s = poa_graph_to_fasta(((GraphHandle*)p)->g);
        size_t k = std::min<size_t>(s.size(), cap ? cap - 1 : 0);
        if (cap) { std::memcpy(buf, s.data(), k); buf[k] = 0; }
        return (int64_t)s.size();
    } catch (const std::exception& ex) { g_last_error = ex.what(); return -1; }
}
// load_graph_from_fasta_msa (src/io/graph.rs:36-103): names / rows are '\n'-separated lists
void* oracle_poa_from_msa(const char* names, const char* rows) {
    try {
        auto split = [](const char* t) {
            std::vector<std::string> out; std::string cur;
            for (const char* c = t; *c; ++c) { if (*c == '\n') { out.push_back(cur); cur.clear(); } else cur.push_back(*c); }
            out.push_back(cur);
            return out;
        };
        auto* h = new GraphHandle;
        h->g = load_graph_from_fasta_msa(split(names), split(rows));
        return h;
    } catch (const std::exception& ex) { g_last_error = ex.what(); return nullptr; }
}

}  // extern "C"

// ---- diagnostics: where does the reference's traceback leave the dense rule? ----------------
// For one query: run A* (table T_A) and the dense pass (T_R), walk the REFERENCE's backtrace and at
// each step list the dense candidates; report the first step where the reference's pick is not the
// first dense candidate.  out[0]=diverged(0/1), out[1]=state, out[2]=n_dense_candidates,
// out[3]=index of the reference's pick among dense candidates (or 99), out[4]=first dense candidate
// visited-at-optimal in T_A (0/1), out[5]=T_A value of first dense candidate (or UNVISITED),
// out[6]=T_R value, out[7]=step index, out[8..10] = kind of first dense cand (state), kind of ref pick (state), cur offset
extern "C" int oracle_tie_report(void* p, uint8_t m, uint8_t o, uint8_t e, int heuristic, int prune, const uint8_t* seq,
                                 uint64_t len, uint64_t* out) {
    auto* h = (GraphHandle*)p;
    try {
        Aligner A(h->g, h->bi(), mk_costs(m, o, e), (Heuristic)heuristic, prune != 0);
        AstarResult ar = A.astar_alignment(seq, len);
        DenseAligner D(h->g, mk_costs(m, o, e));
        DenseResult R;
        D.forward(seq, len, R);
        const size_t P = R.pitch;
        auto rank = D.rank;
        auto TR = [&](uint32_t node, size_t j, AlignState st) -> Score {
            const auto& pl = st == ST_M ? R.M : (st == ST_I ? R.I : R.D);
            return pl[rank[node] * P + j];
        };
        // enumerate dense candidates of one step in reference test order
        struct Cand { uint32_t node; uint32_t j; AlignState st; };
        auto cands = [&](uint32_t v, uint32_t j, AlignState st) {
            std::vector<Cand> c;
            const Graph& g = h->g;
            Score cs = TR(v, j, st);
            if (cs == UNVISITED) return c;
            if (st == ST_M) {
                if (j > 0) {
                    bool moe = g.is_symbol_equal(v, seq[j - 1]) || v == g.end;
                    uint32_t pj = v == g.end ? j : j - 1;
                    Score target = moe ? cs : cs - m;
                    for (uint32_t pr : g.pred[v]) if (TR(pr, pj, ST_M) == target) c.push_back({pr, pj, ST_M});
                }
                if (TR(v, j, ST_D) == cs) c.push_back({v, j, ST_D});
                if (TR(v, j, ST_I) == cs) c.push_back({v, j, ST_I});
            } else if (st == ST_D) {
                for (uint32_t pr : g.pred[v]) if (TR(pr, j, ST_M) == cs - o - e) c.push_back({pr, j, ST_M});
                for (uint32_t pr : g.pred[v]) if (TR(pr, j, ST_D) == cs - e) c.push_back({pr, j, ST_D});
            } else if (j > 0) {
                if (TR(v, j - 1, ST_M) == cs - o - e) c.push_back({v, j - 1, ST_M});
                if (TR(v, j - 1, ST_I) == cs - e) c.push_back({v, j - 1, ST_M});
            }
            return c;
        };
        for (int k = 0; k < 12; ++k) out[k] = 0;
        // walk the reference's own backtrace
        AlnNode cur{h->g.end, (uint32_t)len}; AlignState cst = ST_M;
        uint64_t step = 0;
        AlnNode bt; AlignState bst;
        while (A.get_backtrace(cur, cst, bt, bst)) {
            auto c = cands(cur.node, cur.offset, cst);
            int idx = 99;
            for (size_t i = 0; i < c.size(); ++i)
                if (c[i].node == bt.node && c[i].j == bt.offset && c[i].st == bst) { idx = (int)i; break; }
            if (idx != 0) {
                out[0] = 1; out[1] = cst; out[2] = c.size(); out[3] = idx;
                if (!c.empty()) {
                    Score ta = A.visited.get_score({c[0].node, c[0].j}, c[0].st);
                    Score tr = TR(c[0].node, c[0].j, c[0].st);
                    out[4] = (ta == tr); out[5] = ta; out[6] = tr; out[8] = c[0].st;
                }
                out[7] = step; out[9] = bst; out[10] = cur.offset; out[11] = ar.score;
                return 0;
            }
            if (bt.node == h->g.start) break;
            cur = bt; cst = bst; step++;
        }
        return 0;
    } catch (const std::exception& ex) { g_last_error = ex.what(); return -1; }
}

// Tie census: for every step of the reference's backtrace with >= 2 dense candidates, record
// (signature of candidate kinds, which one the reference took).  Signature: per candidate one char
// m (diag pred), d (D close / D pred), i (I close) ...; written as text lines into buf.
extern "C" int oracle_tie_census(void* p, uint8_t m, uint8_t o, uint8_t e, int heuristic, int prune, const uint8_t* seq,
                                 uint64_t len, char* buf, uint64_t cap) {
    auto* h = (GraphHandle*)p;
    try {
        Aligner A(h->g, h->bi(), mk_costs(m, o, e), (Heuristic)heuristic, prune != 0);
        A.astar_alignment(seq, len);
        DenseAligner D(h->g, mk_costs(m, o, e));
        DenseResult R;
        D.forward(seq, len, R);
        const size_t P = R.pitch;
        auto rank = D.rank;
        const Graph& g = h->g;
        auto TR = [&](uint32_t node, size_t j, AlignState st) -> Score {
            const auto& pl = st == ST_M ? R.M : (st == ST_I ? R.I : R.D);
            return pl[rank[node] * P + j];
        };
        struct Cand { uint32_t node; uint32_t j; AlignState st; char kind; };
        std::string outs;
        AlnNode cur{g.end, (uint32_t)len}; AlignState cst = ST_M;
        AlnNode bt; AlignState bst;
        while (A.get_backtrace(cur, cst, bt, bst)) {
            std::vector<Cand> c;
            uint32_t v = cur.node, j = cur.offset;
            Score cs = TR(v, j, cst);
            Score csA = A.visited.get_score(cur, cst);
            if (cs == csA) {
                if (cst == ST_M) {
                    if (j > 0) {
                        bool moe = g.is_symbol_equal(v, seq[j - 1]) || v == g.end;
                        uint32_t pj = v == g.end ? j : j - 1;
                        Score target = moe ? cs : cs - m;
                        for (uint32_t pr : g.pred[v]) if (TR(pr, pj, ST_M) == target) c.push_back({pr, pj, ST_M, moe ? 'm' : 'x'});
                    }
                    if (TR(v, j, ST_D) == cs) c.push_back({v, j, ST_D, 'd'});
                    if (TR(v, j, ST_I) == cs) c.push_back({v, j, ST_I, 'i'});
                } else if (cst == ST_D) {
                    for (uint32_t pr : g.pred[v]) if (TR(pr, j, ST_M) == cs - o - e) c.push_back({pr, j, ST_M, 'O'});
                    for (uint32_t pr : g.pred[v]) if (TR(pr, j, ST_D) == cs - e) c.push_back({pr, j, ST_D, 'E'});
                } else if (j > 0) {
                    if (TR(v, j - 1, ST_M) == cs - o - e) c.push_back({v, j - 1, ST_M, 'O'});
                    if (TR(v, j - 1, ST_I) == cs - e) c.push_back({v, j - 1, ST_M, 'E'});
                }
                if (c.size() >= 2) {
                    int idx = -1;
                    for (size_t i = 0; i < c.size(); ++i)
                        if (c[i].node == bt.node && c[i].j == bt.offset && c[i].st == bst) { idx = (int)i; break; }
                    std::string sig = cst == ST_M ? "M:" : (cst == ST_D ? "D:" : "I:");
                    for (auto& k : c) sig.push_back(k.kind);
                    sig += " ref=" + std::to_string(idx);
                    // which candidates are visited-at-optimal in the reference's table
                    sig += " vis=";
                    for (auto& k : c) sig.push_back(A.visited.get_score({k.node, k.j}, k.st) == TR(k.node, k.j, k.st) ? '1' : '0');
                    outs += sig + "\n";
                }
            }
            if (bt.node == g.start) break;
            cur = bt; cst = bst;
        }
        if (outs.size() + 1 > cap) outs.resize(cap - 1);
        std::memcpy(buf, outs.c_str(), outs.size() + 1);
        return 0;
    } catch (const std::exception& ex) { g_last_error = ex.what(); return -1; }
}

// Same for the two-piece model (inside oracle_set_two_piece(1, ...)): the five planes M, I1, D1, I2, D2.
extern "C" int oracle_astar_table2(void* p, uint8_t m, uint8_t o, uint8_t e, int heuristic, int prune, const uint8_t* seq,
                                   uint64_t len, uint32_t* const* planes, uint64_t* out) {
    auto* h = (GraphHandle*)p;
    try {
        Aligner A(h->g, h->bi(), mk_costs(m, o, e), (Heuristic)heuristic, prune != 0);
        apply_aln_type(A);
        AstarResult r = A.astar_alignment(seq, len);
        const size_t n = h->g.symbol.size(), P = len + 1;
        const AlignState order[5] = {ST_M, ST_I, ST_D, ST_I2, ST_D2};
        for (int pl = 0; pl < 5; ++pl)
            for (uint32_t v = 0; v < n; ++v)
                for (uint32_t j = 0; j <= len; ++j) planes[pl][v * P + j] = A.visited.get_score({v, j}, order[pl]);
        out[0] = r.score; out[1] = r.num_queued; out[2] = r.num_visited; out[3] = r.num_pruned;
        return 0;
    } catch (const RefPanic& ex) { g_last_error = ex.what(); return 1; }
    catch (const std::exception& ex) { g_last_error = ex.what(); return -1; }
}

// The reference's visited table after one alignment, dense by NODE: M/I/D [node][len+1], UNVISITED = 0xFFFFFFFF.
extern "C" int oracle_astar_table(void* p, uint8_t m, uint8_t o, uint8_t e, int heuristic, int prune, const uint8_t* seq,
                                  uint64_t len, uint32_t* pm, uint32_t* pi, uint32_t* pd, uint64_t* out) {
    auto* h = (GraphHandle*)p;
    try {
        Aligner A(h->g, h->bi(), mk_costs(m, o, e), (Heuristic)heuristic, prune != 0);
        apply_aln_type(A);
        AstarResult r = A.astar_alignment(seq, len);
        const size_t n = h->g.symbol.size(), P = len + 1;
        for (uint32_t v = 0; v < n; ++v)
            for (uint32_t j = 0; j <= len; ++j) {
                pm[v * P + j] = A.visited.get_score({v, j}, ST_M);
                pi[v * P + j] = A.visited.get_score({v, j}, ST_I);
                pd[v * P + j] = A.visited.get_score({v, j}, ST_D);
            }
        out[0] = r.score; out[1] = r.num_queued; out[2] = r.num_visited; out[3] = r.num_pruned;
        return 0;
    } catch (const RefPanic& ex) { g_last_error = ex.what(); return 1; }
    catch (const std::exception& ex) { g_last_error = ex.what(); return -1; }
}
