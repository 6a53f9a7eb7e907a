// Exact-replay mode: the reference's A* search itself, restated over flat arrays so that it can run
// one query per GPU thread (and, compiled for the host, be unit-tested against the oracle).
// Product code — it shares no source with the test-side CPU restatement.
//
// Why it exists: the reference's traceback reads the table ITS SEARCH filled; among co-optimal
// alignments it returns the one whose cells it happened to visit (pruning hides the others).  The
// dense pass cannot know that; replaying the search — same pop order, same greedy extension, same
// pruning — can.  This file follows, statement by statement (paths relative to /root/reference/):
//   astar_alignment               src/aligner/astar.rs:108-226
//   DepthFirstGreedyAlignment     src/aligner/dfa.rs:138-250
//   expand_* / AffineAstarData    src/aligner/scoring/gap_affine.rs:307-430, :753-802
//   LayeredQueue/AffineQueueLayer src/aligner/queue.rs:31-70, gap_affine.rs:945-966, :1005-1012
//   MinimumGapCostAffine::h       src/aligner/heuristic.rs:70-102
//   ReachedBubbleExitsMatch       src/bubbles/reached.rs:38-255
// The search leaves its visited scores in the same u32 plane layout the dense kernels use, so the
// ordinary traceback kernel (the reference's backtrace rule) finishes the job.
#pragma once
#include <stdint.h>

#include "poa_graph.hpp"

#if defined(__HIPCC__)
#define POA_HD __host__ __device__
#else
#define POA_HD
#endif

namespace poa_amd {

enum : uint32_t { EX_ST_M = 0, EX_ST_D = 1, EX_ST_I = 2 };
enum : uint32_t { EX_H_DIJKSTRA = 0, EX_H_MINGAP = 1 };
enum : uint32_t {
    EX_OK = 0,
    EX_PANIC = 1,          // the reference would panic (empty queue, add to Unvisited, u32::MAX score)
    EX_POOL_FULL = 2,      // workspace exhausted (queue pool / priority range / DFA stack)
};
constexpr uint32_t EX_INF = 0xFFFFFFFFu;
constexpr uint32_t EX_NIL = 0xFFFFFFFFu;

struct ExactGraph {  // row-indexed, read-only, shared by all queries
    uint32_t n_rows, start_row, end_row;
    const uint8_t* sym;        // [n_rows] node symbol
    const uint32_t* succ_off;  // [n_rows+1]
    const uint32_t* succ;      // successor rows in trait iteration order
    const uint32_t* dist_min;
    const uint32_t* dist_max;
    const uint32_t* exit_idx;  // [n_rows] index among the bubble-exit rows, EX_NIL for the others
    uint32_t n_exit;
    const uint32_t* nbm_off;   // [n_rows+1]
    const FlatGraph::NodeBubble* nbm;
    // ends-free spans only (null for Global)
    const uint32_t* node_row;  // [n_rows] node index -> row (initial states are pushed by node index)
    const uint32_t* sp_to_end; // [n_rows] edges on the shortest path to the end row (0xFFFFFFFF: none)
};

struct ExQEntry { uint32_t score, row, offset, next; };
struct ExStackEntry { uint32_t row, offset, it; };

struct ExactWork {  // per query
    uint32_t* T;            // visited table, tiled (ex_cell_index), INF-initialised; states: EX_ST_M / EX_ST_D / EX_ST_I
    uint32_t n_rows, pitch;
    uint64_t* reached;      // [n_exit * wpn] bitset of offsets reached in Match state, one row per bubble exit
    uint64_t* rsum;         // [n_exit * swpn] summary: bit w set iff word w of that exit's bitset is non-zero
    uint32_t wpn, swpn;
    uint32_t* head;         // [3 * n_prio] LIFO heads per (priority, state): M, D, I
    uint32_t n_prio;
    ExQEntry* pool; uint32_t pool_cap;
    ExStackEntry* stack; uint32_t stack_cap;
};

struct ExactResult {
    uint32_t status;  // EX_*
    uint32_t score;
    uint32_t num_queued, num_visited, num_pruned;
    uint32_t end_row, end_off;  // the end cell the backtrace starts from (astar.rs:141, :159)
};

enum : uint32_t { EX_BOUND_UNBOUNDED = 0, EX_BOUND_INCLUDED = 1, EX_BOUND_EXCLUDED = 2 };
struct ExactCosts {
    uint32_t x, o, e; uint32_t heuristic; uint32_t prune;
    // AlignmentType (scoring/mod.rs:50-62); ends_free == 0: Global
    uint32_t ends_free;
    uint32_t qfe_kind, qfe_val;   // qry_free_end
    uint32_t gfb_kind;            // graph_free_begin (only Unbounded-or-not matters, gap_affine.rs:150-167)
    uint32_t gfe_kind, gfe_val;   // graph_free_end
};

class ExactSearch {
public:
    const ExactGraph& G;
    ExactWork& W;
    const uint8_t* seq;
    uint32_t L;
    ExactCosts C;
    uint32_t err = EX_OK;
    // queue state (queue.rs:19-22)
    uint32_t layer_min = 0, n_layers = 0, pool_top = 0;
    uint32_t num_queued = 0, num_visited = 0, num_pruned = 0;

    POA_HD ExactSearch(const ExactGraph& g, ExactWork& w, const uint8_t* s, uint32_t len, ExactCosts c)
        : G(g), W(w), seq(s), L(len), C(c) {}

    // ---- Score arithmetic (scoring/mod.rs:93-152) -------------------------------------------
    // (32-bit throughout: the engine refuses graphs / queries whose priorities could reach 2^26, so no sum here can wrap)
    POA_HD uint32_t score_add(uint32_t s, uint32_t rhs) {
        if (s == EX_INF) { err = EX_PANIC; return 0; }
        const uint32_t r = s + rhs;
        if (r == EX_INF) { err = EX_PANIC; return 0; }
        return r;
    }
    POA_HD uint32_t gap_cost(uint32_t st, uint32_t length) const {  // gap_affine.rs:68-80
        if (length == 0) return 0;
        return (st == EX_ST_M ? C.o : 0u) + length * C.e;
    }
    POA_HD bool is_symbol_equal(uint32_t row, uint8_t c) const {  // graphs/poa.rs:463-465
        return row == G.end_row || G.sym[row] == c;
    }

    // ---- visited table (gap_affine.rs:483-548) -----------------------------------------------
    POA_HD uint32_t* cell(uint32_t row, uint32_t off, uint32_t st) const {
        return W.T + ex_cell_index(row, off, st, W.n_rows, W.pitch);
    }
    // Offsets beyond the row: the reference's table is a hash of tiles and takes any offset — an ends-free search that is
    // not allowed to stop at the query end opens an insertion at offset len + 1 (expand_ref_graph_end has no bound,
    // gap_affine.rs:346-368).  The flat planes have `pitch` columns: beyond them a cell reads as unvisited and a write
    // is a workspace overflow (the query keeps its flag, nothing is written out of bounds).
    POA_HD uint32_t get_score(uint32_t row, uint32_t off, uint32_t st) const { return off < W.pitch ? *cell(row, off, st) : EX_INF; }
    POA_HD bool update_if_lower(uint32_t row, uint32_t off, uint32_t st, uint32_t s) {
        if (off >= W.pitch) { err = EX_POOL_FULL; return false; }
        uint32_t* p = cell(row, off, st);
        if (s < *p) { *p = s; return true; }
        return false;
    }

    // ---- reached sets: BTreeSet<offset> per exit node as a two-level bitset (gap_affine.rs:711,:767-773) ---
    POA_HD void mark_reached(uint32_t row, uint32_t off, uint32_t st) {
        if (st != EX_ST_M) return;
        const uint32_t x = G.exit_idx[row];
        if (x == EX_NIL) return;
        const uint32_t wi = off >> 6;
        if (wi >= W.wpn) { err = EX_POOL_FULL; return; }
        uint64_t* w = W.reached + (uint64_t)x * W.wpn + wi;
        const uint64_t old = *w;
        *w = old | (1ull << (off & 63));
        if (old == 0) W.rsum[(uint64_t)x * W.swpn + (wi >> 6)] |= 1ull << (wi & 63);
    }
    POA_HD bool reached_any(uint32_t row) const {  // !reached_offsets.is_empty()
        const uint64_t* s = W.rsum + (uint64_t)G.exit_idx[row] * W.swpn;
        for (uint32_t i = 0; i < W.swpn; ++i) if (s[i]) return true;
        return false;
    }
    // largest reached offset < t, or EX_NIL  (row must be an exit row)
    POA_HD uint32_t reached_before(uint32_t row, uint32_t t) const {
        if (t == 0) return EX_NIL;
        const uint32_t x = G.exit_idx[row];
        const uint64_t* b = W.reached + (uint64_t)x * W.wpn;
        const uint64_t* s = W.rsum + (uint64_t)x * W.swpn;
        uint32_t last = t - 1;
        if ((last >> 6) >= W.wpn) last = W.wpn * 64 - 1;
        const uint32_t wi = last >> 6;
        const uint64_t w = b[wi] & (~0ull >> (63 - (last & 63)));
        if (w) return wi * 64 + 63 - (uint32_t)clz64(w);
        if (wi == 0) return EX_NIL;
        // non-empty words below wi, through the summary
        const uint32_t lw = wi - 1;
        int32_t si = (int32_t)(lw >> 6);
        uint64_t sw = s[si] & (~0ull >> (63 - (lw & 63)));
        for (;;) {
            if (sw) {
                const uint32_t fw = (uint32_t)si * 64 + 63 - (uint32_t)clz64(sw);
                return fw * 64 + 63 - (uint32_t)clz64(b[fw]);
            }
            if (--si < 0) return EX_NIL;
            sw = s[si];
        }
    }
    // smallest reached offset >= t, or EX_NIL  (row must be an exit row)
    POA_HD uint32_t reached_from(uint32_t row, uint32_t t) const {
        const uint32_t wi = t >> 6;
        if (wi >= W.wpn) return EX_NIL;
        const uint32_t x = G.exit_idx[row];
        const uint64_t* b = W.reached + (uint64_t)x * W.wpn;
        const uint64_t* s = W.rsum + (uint64_t)x * W.swpn;
        const uint64_t w = b[wi] & (~0ull << (t & 63));
        if (w) return wi * 64 + (uint32_t)ctz64(w);
        const uint32_t nw = wi + 1;
        if (nw >= W.wpn) return EX_NIL;
        uint32_t si = nw >> 6;
        uint64_t sw = s[si] & (~0ull << (nw & 63));
        for (;;) {
            if (sw) {
                const uint32_t fw = si * 64 + (uint32_t)ctz64(sw);
                return fw * 64 + (uint32_t)ctz64(b[fw]);
            }
            if (++si >= W.swpn) return EX_NIL;
            sw = s[si];
        }
    }
    static POA_HD int clz64(uint64_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
        return __clzll((long long)v);
#else
        return __builtin_clzll(v);
#endif
    }
    static POA_HD int ctz64(uint64_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
        return __ffsll((unsigned long long)v) - 1;
#else
        return __builtin_ctzll(v);
#endif
    }

    // ---- heuristic (heuristic.rs:70-102 / :41-46) --------------------------------------------
    POA_HD uint32_t h(uint32_t row, uint32_t off, uint32_t st) const {
        if (C.heuristic == EX_H_DIJKSTRA) return 0;
        uint32_t mn = G.dist_min[row]; mn = mn ? mn - 1 : 0;
        uint32_t mx = G.dist_max[row]; mx = mx ? mx - 1 : 0;
        const uint32_t tmin = off + mn, tmax = off + mx;
        uint32_t gap;
        if (tmin > L) { gap = tmin - L; if (st != EX_ST_D) st = EX_ST_M; }
        else if (tmax < L) { gap = L - tmax; if (st != EX_ST_I) st = EX_ST_M; }
        else gap = 0;
        return gap_cost(st, gap);
    }

    // ---- pruning (reached.rs:38-255, gap_affine.rs:780-792) ------------------------------------
    POA_HD bool can_improve_at_offset(uint32_t exit_row, uint32_t to_check, uint32_t score, uint32_t left, uint32_t right,
                                      uint32_t min_dist_to_end) {
        bool have = false;
        uint32_t implicit = 0;
        if (left != EX_NIL && right != EX_NIL) {
            const uint32_t ls = get_score(exit_row, left, EX_ST_M), rs = get_score(exit_row, right, EX_ST_M);
            const uint32_t gl = to_check - left, gr = right - to_check;
            const uint32_t from_left = score_add(ls, gap_cost(EX_ST_M, gl));
            const uint32_t from_right = score_add(rs, gap_cost(EX_ST_M, gr));
            implicit = (gr > min_dist_to_end) ? from_left : (from_left < from_right ? from_left : from_right);
            have = true;
        } else if (left == EX_NIL && right != EX_NIL) {
            const uint32_t rs = get_score(exit_row, right, EX_ST_M);
            const uint32_t gr = right - to_check;
            const uint32_t from_right = score_add(rs, gap_cost(EX_ST_M, gr));
            if (gr > min_dist_to_end) have = false;
            else { implicit = from_right; have = true; }
        } else if (left != EX_NIL) {
            const uint32_t ls = get_score(exit_row, left, EX_ST_M);
            implicit = score_add(ls, gap_cost(EX_ST_M, to_check - left));
            have = true;
        }
        return have ? (score < implicit) : true;
    }

    POA_HD bool can_improve_bubble(const FlatGraph::NodeBubble& b, uint32_t row, uint32_t off, uint32_t st, uint32_t current) {
        const uint32_t ex = b.exit_row;
        if (!reached_any(ex)) return true;
        if (row == ex) return true;
        const uint32_t tmin = off + b.min_dist, tmax = off + b.max_dist;
        uint32_t mde = G.dist_min[ex]; mde = mde ? mde - 1 : 0;
        if (tmax > L) return true;
        uint32_t prev = reached_before(ex, tmin);
        bool have_last = false;
        uint32_t last_offset = 0;
        if (tmin > tmax) { err = EX_PANIC; return true; }
        for (uint32_t next = reached_from(ex, tmin); next != EX_NIL && next <= tmax; next = reached_from(ex, next + 1)) {
            uint32_t offset1 = tmin;
            if (prev != EX_NIL) { const uint32_t pv = prev + 1u; offset1 = tmin > pv ? tmin : pv; }
            if (st == EX_ST_D) {
                const uint32_t cst = get_score(ex, next, EX_ST_M);
                if (score_add(cst, C.o) > current) return true;
            }
            if (prev != EX_NIL && st == EX_ST_I) {
                const uint32_t cst = get_score(ex, prev, EX_ST_M);
                if (score_add(cst, C.o) > current) return true;
            }
            if (can_improve_at_offset(ex, offset1, current, prev, next, mde)) return true;
            const uint32_t nm1 = next - 1u;  // wrapping u32 subtraction, as in a release build
            const uint32_t mx = tmin > nm1 ? tmin : nm1;
            const uint32_t offset2 = tmax < mx ? tmax : mx;
            if (offset2 != offset1) {
                if (can_improve_at_offset(ex, offset2, current, prev, next, mde)) return true;
            }
            prev = next;
            last_offset = offset2; have_last = true;
        }
        const uint32_t from = (tmax == 0xFFFFFFFFu) ? tmax : tmax + 1u;
        const uint32_t nxt = reached_from(ex, from);
        if (!have_last && can_improve_at_offset(ex, tmin, current, prev, nxt, mde)) return true;
        if ((!have_last || last_offset < tmax) && can_improve_at_offset(ex, tmax, current, prev, nxt, mde)) return true;
        if (prev != EX_NIL && st == EX_ST_I) {
            const uint32_t cst = get_score(ex, prev, EX_ST_M);
            if (score_add(cst, C.o) > current) return true;
        }
        return false;
    }

    POA_HD bool prune(uint32_t score, uint32_t row, uint32_t off, uint32_t st) {
        const uint32_t b0 = G.nbm_off[row], b1 = G.nbm_off[row + 1];
        if (b0 == b1) return false;
        for (uint32_t k = b0; k < b1; ++k)
            if (!can_improve_bubble(G.nbm[k], row, off, st, score)) return true;
        return false;
    }

    // ---- bucket queue (queue.rs:31-70; gap_affine.rs:945-966) ----------------------------------
    POA_HD void queue_state(uint32_t row, uint32_t off, uint32_t st, uint32_t new_score) {
        const uint32_t pr64 = new_score + h(row, off, st);
        num_queued += 1;
        if (pr64 >= W.n_prio || pool_top >= W.pool_cap) { err = EX_POOL_FULL; return; }
        const uint32_t prio = pr64;
        if (n_layers == 0) { n_layers = 1; layer_min = prio; }
        else if (prio < layer_min) { n_layers += layer_min - prio; layer_min = prio; }
        else if (prio >= layer_min + n_layers) { n_layers = prio - layer_min + 1; }
        const uint32_t e = pool_top++;
        uint32_t* hd = &W.head[3 * (uint64_t)prio + st];
        W.pool[e] = ExQEntry{new_score, row, off, *hd};
        *hd = e;
    }
    POA_HD bool layer_empty(uint32_t prio) const {
        const uint32_t* hd = &W.head[3 * (uint64_t)prio];
        return hd[0] == EX_NIL && hd[1] == EX_NIL && hd[2] == EX_NIL;
    }
    POA_HD bool pop_state(uint32_t& score, uint32_t& row, uint32_t& off, uint32_t& st) {
        if (n_layers == 0) return false;
        uint32_t* hd = &W.head[3 * (uint64_t)layer_min];
        bool got = false;
        for (uint32_t s = 0; s < 3; ++s) {  // Match stack, else Deletion, else Insertion (state codes 0,1,2)
            if (hd[s] != EX_NIL) {
                const ExQEntry e = W.pool[hd[s]];
                hd[s] = e.next;
                score = e.score; row = e.row; off = e.offset; st = s;
                got = true;
                break;
            }
        }
        while (n_layers != 0) {
            if (layer_empty(layer_min)) { layer_min += 1; n_layers -= 1; }
            else break;
        }
        return got;
    }

    POA_HD bool is_end(uint32_t row, uint32_t off, uint32_t st) const {  // gap_affine.rs:185-248
        if (!C.ends_free) return st == EX_ST_M && row == G.end_row && off == L;
        bool q_ok;
        if (C.qfe_kind == EX_BOUND_UNBOUNDED) q_ok = off > 0 || L == 0;  // sic: any consumed prefix may end
        else if (C.qfe_kind == EX_BOUND_INCLUDED) q_ok = L - off <= C.qfe_val;
        else q_ok = L - off < C.qfe_val;
        // dist_to_end(node, max) (gap_affine.rs:91-119) finds the end iff the shortest path has <= max edges
        const uint32_t d = G.sp_to_end[row];
        bool g_ok;
        if (C.gfe_kind == EX_BOUND_UNBOUNDED) g_ok = true;
        else if (C.gfe_kind == EX_BOUND_INCLUDED) g_ok = d != EX_INF && d <= C.gfe_val;
        else g_ok = d != EX_INF && d <= (C.gfe_val ? C.gfe_val - 1 : 0u) && d < C.gfe_val;
        return st == EX_ST_M && q_ok && g_ok;
    }

    // ---- expansions (gap_affine.rs:307-430) ----------------------------------------------------
    POA_HD void expand_ref_graph_end(uint32_t prow, uint32_t poff, uint32_t score) {
        const uint32_t ns = score_add(score_add(score, C.o), C.e);
        if (err) return;
        if (update_if_lower(prow, poff + 1, EX_ST_I, ns)) queue_state(prow, poff + 1, EX_ST_I, ns);
    }
    POA_HD void expand_query_end(uint32_t poff, uint32_t child, uint32_t score) {
        const uint32_t ns = score_add(score_add(score, C.o), C.e);
        if (err) return;
        if (update_if_lower(child, poff, EX_ST_D, ns)) queue_state(child, poff, EX_ST_D, ns);
    }
    POA_HD void expand_mismatch(uint32_t prow, uint32_t poff, uint32_t crow, uint32_t coff, uint32_t score) {
        const uint32_t nm = score_add(score, C.x);
        if (err) return;
        if (update_if_lower(crow, coff, EX_ST_M, nm)) queue_state(crow, coff, EX_ST_M, nm);
        const uint32_t ng = score_add(score_add(score, C.o), C.e);
        if (err) return;
        if (update_if_lower(prow, poff + 1, EX_ST_I, ng)) queue_state(prow, poff + 1, EX_ST_I, ng);
        if (update_if_lower(crow, poff, EX_ST_D, ng)) queue_state(crow, poff, EX_ST_D, ng);
    }
    POA_HD void expand_all(uint32_t score, uint32_t row, uint32_t off, uint32_t st) {
        if (update_if_lower(row, off, EX_ST_M, score)) queue_state(row, off, EX_ST_M, score);
        if (st == EX_ST_I) {
            const uint32_t ns = score_add(score, C.e);
            if (err) return;
            if (off < L && update_if_lower(row, off + 1, EX_ST_I, ns)) queue_state(row, off + 1, EX_ST_I, ns);
        } else {
            for (uint32_t e = G.succ_off[row]; e < G.succ_off[row + 1]; ++e) {
                const uint32_t ns = score_add(score, C.e);
                if (err) return;
                if (update_if_lower(G.succ[e], off, EX_ST_D, ns)) queue_state(G.succ[e], off, EX_ST_D, ns);
            }
        }
    }

    // ---- depth-first greedy extension (dfa.rs:138-250) -----------------------------------------
    enum : uint32_t { EV_NONE = 0, EV_REF_GRAPH_END = 1, EV_QUERY_END = 2, EV_MISMATCH = 3 };
    struct Event { uint32_t kind, prow, poff, crow, coff; };
    uint32_t sp = 0, dfa_visited = 0;
    uint32_t dfa_score = 0;

    POA_HD Event dfa_extend() {
        if (sp == 1 && L != 0) {
            const ExStackEntry init = W.stack[0];
            if (init.offset == 0 && is_symbol_equal(init.row, seq[0])) {
                if (update_if_lower(init.row, 1, EX_ST_M, dfa_score)) {
                    W.stack[0] = ExStackEntry{init.row, 1, G.succ_off[init.row]};
                    mark_reached(init.row, 1, EX_ST_M);
                    dfa_visited += 1;
                    if (1 == L) return Event{EV_REF_GRAPH_END, init.row, 0, init.row, 1};
                }
            }
        }
        while (sp != 0) {
            ExStackEntry& parent = W.stack[sp - 1];
            const uint32_t cend = G.succ_off[parent.row + 1];
            bool again = false;
            while (parent.it < cend) {
                const uint32_t child = G.succ[parent.it++];
                if (child == G.end_row) {
                    update_if_lower(child, parent.offset, EX_ST_M, dfa_score);
                    return Event{EV_REF_GRAPH_END, parent.row, parent.offset, child, parent.offset};
                }
                if (parent.offset >= L) return Event{EV_QUERY_END, parent.row, parent.offset, child, 0};
                const uint32_t coff = parent.offset + 1;
                if (is_symbol_equal(child, seq[coff - 1])) {
                    if (update_if_lower(child, coff, EX_ST_M, dfa_score)) {
                        if (prune(dfa_score, child, coff, EX_ST_M)) { num_pruned_dfa += 1; again = true; break; }
                        if (err) return Event{EV_NONE, 0, 0, 0, 0};
                        mark_reached(child, coff, EX_ST_M);
                        dfa_visited += 1;
                        if (sp >= W.stack_cap) { err = EX_POOL_FULL; return Event{EV_NONE, 0, 0, 0, 0}; }
                        W.stack[sp++] = ExStackEntry{child, coff, G.succ_off[child]};
                        again = true;
                        break;
                    }
                } else {
                    return Event{EV_MISMATCH, parent.row, parent.offset, child, coff};
                }
            }
            if (err) return Event{EV_NONE, 0, 0, 0, 0};
            if (!again) sp -= 1;
        }
        return Event{EV_NONE, 0, 0, 0, 0};
    }
    uint32_t num_pruned_dfa = 0;  // DFA-internal count (dfa.rs:103); NOT added to AstarResult::num_pruned

    // ---- main loop (astar.rs:124-226) -----------------------------------------------------------
    POA_HD ExactResult run() {
        ExactResult R{EX_OK, EX_INF, 0, 0, 0, G.end_row, L};
        if (C.ends_free && C.gfb_kind == EX_BOUND_UNBOUNDED && G.n_rows > 2) {
            // gap_affine.rs:150-163: every real node at offset 0, pushed in reverse node-index order
            for (uint32_t v = G.n_rows; v-- > 0;) {
                const uint32_t r = G.node_row[v];
                if (r == G.start_row || r == G.end_row) continue;
                queue_state(r, 0, EX_ST_M, 0);
                *cell(r, 0, EX_ST_M) = 0;
            }
        } else {
            queue_state(G.start_row, 0, EX_ST_M, 0);
            *cell(G.start_row, 0, EX_ST_M) = 0;  // visited_data.set_score
        }
        uint32_t end_score = EX_INF;
        bool found = false;
        while (!found && !err) {
            uint32_t score, row, off, st;
            if (!pop_state(score, row, off, st)) { err = EX_PANIC; break; }  // "Could not align sequence!"
            if (score > get_score(row, off, st)) continue;
            if (is_end(row, off, st)) { num_visited += 1; end_score = score; found = true; R.end_row = row; R.end_off = off; break; }
            if (C.prune && prune(score, row, off, st)) { num_pruned += 1; continue; }
            if (err) break;
            mark_reached(row, off, st);
            num_visited += 1;
            if (st == EX_ST_M) {
                sp = 0; dfa_visited = 0; dfa_score = score;
                W.stack[sp++] = ExStackEntry{row, off, G.succ_off[row]};
                for (;;) {
                    const Event ev = dfa_extend();
                    if (err || ev.kind == EV_NONE) break;
                    if (ev.kind == EV_REF_GRAPH_END) {
                        if (is_end(ev.crow, ev.coff, EX_ST_M)) { end_score = score; found = true; R.end_row = ev.crow; R.end_off = ev.coff; break; }
                        expand_ref_graph_end(ev.prow, ev.poff, score);
                    } else if (ev.kind == EV_QUERY_END) {
                        expand_query_end(ev.poff, ev.crow, score);
                    } else {
                        expand_mismatch(ev.prow, ev.poff, ev.crow, ev.coff, score);
                    }
                    if (err) break;
                }
                if (found) break;  // `break 'main` skips the dfa counter update (astar.rs:172,:205)
                num_visited += dfa_visited;
            } else {
                expand_all(score, row, off, st);
            }
        }
        R.status = err ? err : (found ? EX_OK : EX_PANIC);
        R.score = end_score;
        R.num_queued = num_queued; R.num_visited = num_visited; R.num_pruned = num_pruned;
        return R;
    }
};

}  // namespace poa_amd
