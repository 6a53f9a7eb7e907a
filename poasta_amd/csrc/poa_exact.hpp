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
// always inlined on the device: a call would force the search object (graph / workspace pointers, queue state) out of
// registers into scratch memory
#define POA_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define POA_HD
#endif

// analysis hooks (scripts/trace): what a pop reads and writes; no-ops in the product
#if !defined(EX_TRACE_CELL)
#define EX_TRACE_CELL(row, off, st, wr) ((void)0)
#define EX_TRACE_REACH(row, lo, hi, wr) ((void)0)
#endif

namespace poa_amd {

enum : uint32_t { EX_ST_M = 0, EX_ST_D = 1, EX_ST_I = 2, EX_ST_D2 = 3, EX_ST_I2 = 4 };   // aln_graph.rs:8-14 (the last two: two-piece model)
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
    const FlatGraph::RowRec* rec = nullptr;   // [n_rows] per-row records of the lean step (null: graph too large for them)
};

struct ExQEntry { uint32_t score, row, offset, next; };
struct ExStackEntry { uint32_t row, offset, it; };

// Bucket queue of the wave-per-query search (poa_wsearch.hpp): per (priority, state) a LIFO stack kept as a chain of
// 64-slot chunks (slot 0: {previous chunk}, slots 1..63: entries), so that the top entries of a stack are contiguous and a
// wave reads one per lane.  The (top chunk, entries in it) descriptors of the live priorities sit in a ring of `bq_win`
// priorities (LDS on the device); a drained stack resets its descriptor, which frees the ring slot for priority + bq_win.
struct BqDesc { uint32_t top, n_top; };  // in the ring: one word, top << 6 | n_top (0xFFFFFFFF: empty stack); chunk index < 2^26
constexpr uint32_t BQ_EMPTY = 0xFFFFFFFFu;
struct ExU4 { uint32_t x, y, z, w; };
constexpr uint32_t BQ_CHUNK = 64;

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
    // bucket queue (null bq_desc: the linked-list queue above)
    uint32_t* bq_desc = nullptr; uint32_t bq_win = 0;
    ExU4* bq_chunks = nullptr; uint32_t bq_chunk_cap = 0;
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
    // GapAffine2Piece (gap_affine_2piece.rs:19-33): o / e above are the first piece, these the second (EX_AS_TWO_PIECE only)
    uint32_t o2 = 0, e2 = 0;
};

// ---- speculative ("log") mode of the search object: the step of the wave kernel poa_psearch.hpp ----------------------------
// The top entries of the current stack are expanded by different lanes AT ONCE.  Each lane runs the ordinary code of this
// file on its entry — and on the entries that expansion pushes in front of the next entry of the stack (same bucket, state
// of equal or higher pop priority: they would be popped before it) — but reads the table as it was when the step began,
// through an overlay of its own writes, and writes nothing: cell writes, reached marks and queue pushes go to a per-lane
// log.  Afterwards the step finds the first lane that read something an earlier lane of the step logged a write to
// (cell by cell, mark by mark), commits the logs of the lanes before it in lane order — exactly the writes and pushes the
// reference's sequential loop makes — and leaves the rest of the stack for the next step.
constexpr uint32_t SP_KW = 8;    // cell writes a lane may log per step
constexpr uint32_t SP_KM = 6;    // reached marks
constexpr uint32_t SP_KP = 8;    // queue pushes (those that do not come back within the lane's own group)
constexpr uint32_t SP_KPD = 4;   // pending entries of the lane's own group
constexpr uint32_t SP_KRC = 24;  // cells whose value a lane's decisions depended on (kept as 16-bit hashes of their index:
                                 // a chance match only ends a step early)
constexpr uint32_t SP_KRM = 4;   // exits whose reached set they depended on (one range each)
constexpr uint32_t SP_KDS = 4;   // depth of the greedy extension's stack (parents with successors left)
enum : uint32_t {
    SPF_COMPLEX = 1,    // the lane met something the log mode does not do (log / box overflow, an error, a greedy extension
                        // that needs its stack): it is cut off and its entry takes the sequential code
    SPF_FOUND = 2,      // the search ends in this lane
    SPF_LEFTOVER = 4,   // entries of the lane's group are still pending: they go to the queue and the step ends behind this lane
};
struct SpecLane {
    uint32_t n_w = 0, n_m = 0, n_p = 0, n_pd = 0, flags = 0;
    uint32_t dq = 0, dv = 0, dp = 0, n_ent = 0;   // num_queued / num_visited / num_pruned of this lane's step; entries it processed
    uint32_t cur_f = 0, root_st = 0;              // the stack the step pops from
    uint32_t flat = 0, min_child = 0xFFFFFFFFu;   // flat schedule: every push is logged; the lowest (priority << 2 | state) among them
    // logs: element k of this lane at base[k * stride]
    uint32_t* w_idx = nullptr; uint32_t* w_val = nullptr; uint32_t* m_x = nullptr; uint32_t* m_off = nullptr;
    ExU4* p = nullptr;                            // {score, row, offset, priority << 2 | state}
    ExStackEntry* dstack = nullptr;               // the lane's own stack of the greedy extension
    uint32_t stride = 1;
    uint32_t pd_key[SP_KPD] = {}, pd_score[SP_KPD] = {}, pd_row[SP_KPD] = {}, pd_off[SP_KPD] = {};   // in push order
    // what the lane read: cells (index into the table) and ranges [lo, hi] of the reached set of exit x — element k at base[k * stride]
    uint32_t n_rc = 0, n_rm = 0;
    uint16_t* rc = nullptr; uint32_t* rm_x = nullptr; uint32_t* rm_lo = nullptr; uint32_t* rm_hi = nullptr;
    // which cells / words of the reached sets the lane has logged writes to, hashed into 32 bits: a read that misses
    // them (nearly every read) skips the overlay
    uint32_t wmask = 0, mmask = 0;
};

// Address-space tags of the two tables a kernel may stage in LDS (the graph arrays; the queue's descriptor ring).  A pointer
// that may be global or LDS compiles to FLAT loads, which wait for every outstanding vector-memory AND LDS operation
// (they count on both counters): one such load in the middle of a batch of table reads serialises the batch.  With the
// tag the access is a ds_read / ds_write and overlaps with the global loads in flight.
enum : int { EX_AS_GRAPH_LDS = 1, EX_AS_RING_LDS = 2, EX_AS_READSET_LDS = 4, EX_AS_REC_LDS = 8,
             EX_AS_NO_SPEC = 16, EX_AS_TWO_PIECE = 32 };   // (32: the two-piece affine model, gap_affine_2piece.rs — five states, five plain planes, the generic code only)  (16: this instantiation never runs in log mode — the kernels of rounds 1 and 2: the log-mode branches compile away)   // (4: logs and read sets of the log mode: SpecLane::w_*, m_*, rc, rm_*)
#if defined(__HIP_DEVICE_COMPILE__)
template <class T> __device__ inline __attribute__((always_inline)) T ex_lds_load(const T* p) {
    return *(const __attribute__((address_space(3))) T*)p;
}
template <class T> __device__ inline __attribute__((always_inline)) void ex_lds_store(T* p, T v) {
    *(__attribute__((address_space(3))) T*)p = v;
}
#endif

template <int AS = 0>
class ExactSearchT {
public:
    template <class T> POA_HD T gld(const T* p) const {
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr ((AS & EX_AS_GRAPH_LDS) != 0) return ex_lds_load(p);
#endif
        return *p;
    }
    POA_HD uint32_t rld(const uint32_t* p) const {
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr ((AS & EX_AS_RING_LDS) != 0) return ex_lds_load(p);
#endif
        return *p;
    }
    POA_HD void rst(uint32_t* p, uint32_t v) const {
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr ((AS & EX_AS_RING_LDS) != 0) { ex_lds_store(p, v); return; }
#endif
        *p = v;
    }
    template <class T> POA_HD T sld(const T* p) const {
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr ((AS & EX_AS_READSET_LDS) != 0) return ex_lds_load(p);
#endif
        return *p;
    }
    template <class T> POA_HD void sst(T* p, T v) const {
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr ((AS & EX_AS_READSET_LDS) != 0) { ex_lds_store(p, v); return; }
#endif
        *p = v;
    }
    static POA_HD uint16_t sp_hash16(uint32_t ix) { return (uint16_t)((ix * 0x9E3779B1u) >> 16); }
    static POA_HD uint32_t sp_bit(uint32_t key) { return 1u << ((key * 0x9E3779B1u) >> 27); }
    const ExactGraph& G;
    ExactWork& W;
    const uint8_t* seq;
    uint32_t L;
    ExactCosts C;
    uint32_t err = EX_OK;
    // queue state (queue.rs:19-22)
    uint32_t layer_min = 0, n_layers = 0, pool_top = 0, pool_free = EX_NIL;
    uint32_t num_queued = 0, num_visited = 0, num_pruned = 0;

    POA_HD ExactSearchT(const ExactGraph& g, ExactWork& w, const uint8_t* s, uint32_t len, ExactCosts c)
        : G(g), W(w), seq(s), L(len), C(c) {}

    // a search object is reused for the next query of its wave / lane group
    POA_HD void begin_query(const uint8_t* s, uint32_t len) {
        seq = s; L = len; err = EX_OK;
        layer_min = 0; n_layers = 0; pool_top = 0; pool_free = EX_NIL; num_queued = num_visited = num_pruned = 0;
        bq_live = 0; bq_chunk_top = 0; bq_hi = 0; bq_free = EX_NIL;
        sp = 0; dfa_visited = 0; dfa_score = 0; n_fast = 0; num_pruned_dfa = 0;
    }
    // ---- log mode (see SpecLane) ---------------------------------------------------------------
    bool spec = false;
    SpecLane sl;
    static constexpr bool TP = (AS & EX_AS_TWO_PIECE) != 0;
    static constexpr uint32_t NST = TP ? 5u : 3u;   // states (planes of the table, stacks per priority)
    // the one-round-trip path of the wave kernels reads the per-row records when the launch staged them (EX_AS_REC_LDS | EX_AS_NO_SPEC)
    POA_HD bool use_rec() const {
        if constexpr ((AS & EX_AS_NO_SPEC) != 0 && (AS & EX_AS_REC_LDS) != 0) return G.rec != nullptr; else return false;
    }
    POA_HD bool in_spec() const { if constexpr ((AS & EX_AS_NO_SPEC) != 0) return false; else return spec; }
#if defined(POA_PS_PROF_FINE)
    unsigned long long pf[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pf_t = 0;
#endif
#if defined(POA_PS_PROF_FINE) && defined(__HIP_DEVICE_COMPILE__)
#define PF_START() do { pf_t = clock64(); } while (0)
#define PF_TICK(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long n_ = clock64(); pf[k] += n_ - pf_t; pf_t = n_; } while (0)
#else
#define PF_START() ((void)0)
#define PF_TICK(k) ((void)0)
#endif

#if defined(POA_EXACT_DIAG)
    uint32_t why_complex[12] = {};
    uint32_t pd_hist[2][8] = {};   // probe neighbours: distance below / above t (1, 2, 3-4 -> 4, 5-8 -> 5, > 8 -> 6, none -> 7)
    void sp_flag(uint32_t f, uint32_t why = 0) { sl.flags |= f; why_complex[why] += 1; }
#else
    POA_HD void sp_flag(uint32_t f, uint32_t = 0) { sl.flags |= f; }
#endif
    // a cell / a range of a reached set whose content the lane's decisions depend on
    POA_HD void note_cell(uint32_t ix) {
        if (sl.n_rc >= SP_KRC) { sp_flag(SPF_COMPLEX, 1); return; }
        sst(&sl.rc[sl.n_rc * sl.stride], sp_hash16(ix));
        sl.n_rc += 1;
    }
    POA_HD void note_marks(uint32_t x, uint32_t lo, uint32_t hi) {
        // (one range per exit: the hull of what was looked at)
        for (uint32_t k = 0; k < sl.n_rm; ++k)
            if (sld(&sl.rm_x[k * sl.stride]) == x) {
                if (lo < sld(&sl.rm_lo[k * sl.stride])) sst(&sl.rm_lo[k * sl.stride], lo);
                if (hi > sld(&sl.rm_hi[k * sl.stride])) sst(&sl.rm_hi[k * sl.stride], hi);
                return;
            }
        if (sl.n_rm >= SP_KRM) { sp_flag(SPF_COMPLEX, 2); return; }
        sst(&sl.rm_x[sl.n_rm * sl.stride], x); sst(&sl.rm_lo[sl.n_rm * sl.stride], lo); sst(&sl.rm_hi[sl.n_rm * sl.stride], hi);
        sl.n_rm += 1;
    }
    // the lane's own logged value of cell `ix`, else v
    POA_HD uint32_t ovl(uint32_t ix, uint32_t v) const {
        if (!(sl.wmask & sp_bit(ix))) return v;
        for (uint32_t k = 0; k < sl.n_w; ++k) if (sld(&sl.w_idx[k * sl.stride]) == ix) v = sld(&sl.w_val[k * sl.stride]);
        return v;
    }
    POA_HD uint32_t cix(uint32_t row, uint32_t off, uint32_t st) const {
        if constexpr (TP) {
            // [row][offset][state], the five states of a cell side by side in the order of the two-piece planes (M, I1, D1, I2, D2;
            // poa2_traceback_kernel reads it through ex2_cell_index): this search lives on the latency of dependent loads, and
            // the cells of an expansion — (v, j) and (v, j + 1), or the same offset one row on — share their cache lines
            const uint32_t pl = st == EX_ST_M ? 0u : st == EX_ST_I ? 1u : st == EX_ST_D ? 2u : st == EX_ST_I2 ? 3u : 4u;
            return (row * W.pitch + off) * 5u + pl;
        } else return ex_cell_index32(row, off, st, W.n_rows, W.pitch);
    }
    // visited score of a cell inside the table
    POA_HD uint32_t ld(uint32_t row, uint32_t off, uint32_t st) {
        const uint32_t ix = cix(row, off, st);
        uint32_t v = W.T[ix];
        if (in_spec()) { note_cell(ix); if (sl.n_w) v = ovl(ix, v); }
        return v;
    }
    POA_HD void wr(uint32_t row, uint32_t off, uint32_t st, uint32_t val) {
        const uint32_t ix = cix(row, off, st);
        EX_TRACE_CELL(row, off, st, 1);
        if (!in_spec()) { W.T[ix] = val; return; }
        // (append only — a later entry for the same cell wins, in the overlay and at the commit — so that the log can be cut back)
        if (sl.n_w >= SP_KW) { sp_flag(SPF_COMPLEX, 3); return; }
        sst(&sl.w_idx[sl.n_w * sl.stride], ix); sst(&sl.w_val[sl.n_w * sl.stride], val);
        sl.wmask |= sp_bit(ix);
        sl.n_w += 1;
    }
    // words of the reached sets, with the lane's own logged marks
    // (log mode: the sets do not change while the lanes of a step read them — the last word and the last summary word stay
    // in registers: the generic pruning test asks for the same word again and again, reached.rs:67-170)
    mutable uint32_t rc_x = EX_NIL, rc_wi = 0, rs_x = EX_NIL, rs_si = 0;
    mutable uint64_t rc_w = 0, rs_w = 0;
    POA_HD uint64_t rword(uint32_t x, uint32_t wi) const {
        uint64_t w;
        if (in_spec() && rc_x == x && rc_wi == wi) w = rc_w;
        else { w = W.reached[(uint64_t)x * W.wpn + wi]; if (in_spec()) { rc_x = x; rc_wi = wi; rc_w = w; } }
        if (in_spec() && (sl.mmask & sp_bit(x))) for (uint32_t k = 0; k < sl.n_m; ++k) {
            const uint32_t o = sld(&sl.m_off[k * sl.stride]);
            if (sld(&sl.m_x[k * sl.stride]) == x && (o >> 6) == wi) w |= 1ull << (o & 63);
        }
        return w;
    }
    POA_HD uint64_t rsw(uint32_t x, uint32_t si) const {
        uint64_t w;
        if (in_spec() && rs_x == x && rs_si == si) w = rs_w;
        else { w = W.rsum[(uint64_t)x * W.swpn + si]; if (in_spec()) { rs_x = x; rs_si = si; rs_w = w; } }
        if (in_spec() && (sl.mmask & sp_bit(x))) for (uint32_t k = 0; k < sl.n_m; ++k) {
            const uint32_t o = sld(&sl.m_off[k * sl.stride]);
            if (sld(&sl.m_x[k * sl.stride]) == x && (o >> 12) == si) w |= 1ull << ((o >> 6) & 63);
        }
        return w;
    }

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
        if constexpr (TP) {   // gap_affine_2piece.rs:99-127 (sic: a state inside a gap is charged its open cost again)
            const uint32_t c1 = C.o + length * C.e, c2 = C.o2 + length * C.e2;
            if (st == EX_ST_I || st == EX_ST_D) return c1;
            if (st == EX_ST_I2 || st == EX_ST_D2) return c2;
            return c1 < c2 ? c1 : c2;
        }
        return (st == EX_ST_M ? C.o : 0u) + length * C.e;
    }
    POA_HD bool is_symbol_equal(uint32_t row, uint8_t c) const {  // graphs/poa.rs:463-465
        return row == G.end_row || gld(&G.sym[row]) == c;
    }

    // ---- visited table (gap_affine.rs:483-548) -----------------------------------------------
    POA_HD uint32_t* cell(uint32_t row, uint32_t off, uint32_t st) const {
        // (a query's table has 3 * n_rows * pitch < 2^32 elements: the engine plans the workspace in 32-bit element counts per query)
        return W.T + cix(row, off, st);
    }
    // Offsets beyond the row: the reference's table is a hash of tiles and takes any offset — an ends-free search that is
    // not allowed to stop at the query end opens an insertion at offset len + 1 (expand_ref_graph_end has no bound,
    // gap_affine.rs:346-368).  The flat planes have `pitch` columns: beyond them a cell reads as unvisited and a write
    // is a workspace overflow (the query keeps its flag, nothing is written out of bounds).
    POA_HD uint32_t get_score(uint32_t row, uint32_t off, uint32_t st) { EX_TRACE_CELL(row, off, st, 0); return off < W.pitch ? ld(row, off, st) : EX_INF; }
    POA_HD bool update_if_lower(uint32_t row, uint32_t off, uint32_t st, uint32_t s) {
        if (off >= W.pitch) { err = EX_POOL_FULL; return false; }
        EX_TRACE_CELL(row, off, st, 0);
        if (s < ld(row, off, st)) { wr(row, off, st, s); return true; }
        return false;
    }

    // ---- reached sets: BTreeSet<offset> per exit node as a two-level bitset (gap_affine.rs:711,:767-773) ---
    POA_HD void mark_word(uint32_t x, uint32_t off) {   // the write itself (direct mode; commit of a logged mark)
        if (!mark_on) return;   // (wave kernel: sections every lane executes alike — one lane sets the mark)
        const uint32_t wi = off >> 6;
        uint64_t* w = W.reached + (uint64_t)x * W.wpn + wi;
        uint64_t* sm = W.rsum + (uint64_t)x * W.swpn + (wi >> 6);
#if defined(__HIP_DEVICE_COMPILE__)
        // two ORs nobody waits for: the next reader of these words is this wave, behind them in its own memory stream
        __hip_atomic_fetch_or(w, 1ull << (off & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_fetch_or(sm, 1ull << (wi & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#else
        *w |= 1ull << (off & 63);
        *sm |= 1ull << (wi & 63);
#endif
    }
    POA_HD void mark_reached(uint32_t row, uint32_t off, uint32_t st) {
        if (st != EX_ST_M) return;
        const uint32_t x = gld(&G.exit_idx[row]);
        if (x == EX_NIL) return;
        if ((off >> 6) >= W.wpn) { err = EX_POOL_FULL; return; }
        EX_TRACE_REACH(row, off, off, 1);
        if (!in_spec()) { mark_word(x, off); return; }
        if (sl.mmask & sp_bit(x)) for (uint32_t k = 0; k < sl.n_m; ++k) if (sld(&sl.m_x[k * sl.stride]) == x && sld(&sl.m_off[k * sl.stride]) == off) return;
        if (sl.n_m >= SP_KM) { sp_flag(SPF_COMPLEX, 4); return; }
        sst(&sl.m_x[sl.n_m * sl.stride], x); sst(&sl.m_off[sl.n_m * sl.stride], off);
        sl.mmask |= sp_bit(x);
        sl.n_m += 1;
    }
    POA_HD bool reached_any(uint32_t row) {  // !reached_offsets.is_empty()
        const uint32_t x = gld(&G.exit_idx[row]);
        for (uint32_t i = 0; i < W.swpn; ++i) if (rsw(x, i)) return true;
        EX_TRACE_REACH(row, 0, 0xFFFFFFFFu, 0);
        if (in_spec()) note_marks(x, 0, 0xFFFFFFFFu);   // (any mark on this row would change the answer)
        return false;
    }
    // largest reached offset < t, or EX_NIL  (row must be an exit row)
    POA_HD uint32_t reached_before(uint32_t row, uint32_t t) {
        const uint32_t r = reached_before_(row, t);
        EX_TRACE_REACH(row, r == EX_NIL ? 0u : r, t ? t - 1 : 0u, 0);
        if (in_spec() && t) note_marks(gld(&G.exit_idx[row]), r == EX_NIL ? 0u : r, t - 1);
        return r;
    }
    POA_HD uint32_t reached_before_(uint32_t row, uint32_t t) const {
        if (t == 0) return EX_NIL;
        const uint32_t x = gld(&G.exit_idx[row]);
        uint32_t last = t - 1;
        if ((last >> 6) >= W.wpn) last = W.wpn * 64 - 1;
        const uint32_t wi = last >> 6;
        const uint64_t w = rword(x, wi) & (~0ull >> (63 - (last & 63)));
        if (w) return wi * 64 + 63 - (uint32_t)clz64(w);
        if (wi == 0) return EX_NIL;
        // non-empty words below wi, through the summary
        const uint32_t lw = wi - 1;
        int32_t si = (int32_t)(lw >> 6);
        uint64_t sw = rsw(x, (uint32_t)si) & (~0ull >> (63 - (lw & 63)));
        for (;;) {
            if (sw) {
                const uint32_t fw = (uint32_t)si * 64 + 63 - (uint32_t)clz64(sw);
                return fw * 64 + 63 - (uint32_t)clz64(rword(x, fw));
            }
            if (--si < 0) return EX_NIL;
            sw = rsw(x, (uint32_t)si);
        }
    }
    // smallest reached offset >= t, or EX_NIL  (row must be an exit row)
    POA_HD uint32_t reached_from(uint32_t row, uint32_t t) {
        const uint32_t r = reached_from_(row, t);
        EX_TRACE_REACH(row, t, r, 0);
        if (in_spec()) note_marks(gld(&G.exit_idx[row]), t, r);   // (EX_NIL: everything from t on)
        return r;
    }
    POA_HD uint32_t reached_from_(uint32_t row, uint32_t t) const {
        const uint32_t wi = t >> 6;
        if (wi >= W.wpn) return EX_NIL;
        const uint32_t x = gld(&G.exit_idx[row]);
        const uint64_t w = rword(x, wi) & (~0ull << (t & 63));
        if (w) return wi * 64 + (uint32_t)ctz64(w);
        const uint32_t nw = wi + 1;
        if (nw >= W.wpn) return EX_NIL;
        uint32_t si = nw >> 6;
        uint64_t sw = rsw(x, si) & (~0ull << (nw & 63));
        for (;;) {
            if (sw) {
                const uint32_t fw = si * 64 + (uint32_t)ctz64(sw);
                return fw * 64 + (uint32_t)ctz64(rword(x, fw));
            }
            if (++si >= W.swpn) return EX_NIL;
            sw = rsw(x, si);
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
        uint32_t mn, mx;
        if (use_rec()) {
            // (dmin | dmax << 16 of the row's record: one 4-byte load, already minus one)
#if defined(__HIP_DEVICE_COMPILE__)
            typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
            const uint32_t dd = ((lds_cu32*)G.rec)[8 * row + 4];
#else
            const uint32_t dd = (uint32_t)G.rec[row].dmin | ((uint32_t)G.rec[row].dmax << 16);
#endif
            mn = dd & 0xFFFFu; mx = dd >> 16;
        } else {
        mn = gld(&G.dist_min[row]); mn = mn ? mn - 1 : 0;
        mx = gld(&G.dist_max[row]); mx = mx ? mx - 1 : 0;
        }
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
        uint32_t mde = gld(&G.dist_min[ex]); mde = mde ? mde - 1 : 0;
        if (tmax > L) return true;
        uint32_t prev = reached_before(ex, tmin);
        bool have_last = false;
        uint32_t last_offset = 0;
        if (tmin > tmax) { err = EX_PANIC; return true; }
        for (uint32_t next = reached_from(ex, tmin); next != EX_NIL && next <= tmax; next = reached_from(ex, next + 1)) {
            uint32_t offset1 = tmin;
            if (prev != EX_NIL) { const uint32_t pv = prev + 1u; offset1 = tmin > pv ? tmin : pv; }
            if (st == EX_ST_D || (TP && st == EX_ST_D2)) {   // reached.rs:84-101
                const uint32_t cst = get_score(ex, next, EX_ST_M);
                if (score_add(cst, st == EX_ST_D ? C.o : C.o2) > current) return true;
            }
            if (prev != EX_NIL && (st == EX_ST_I || (TP && st == EX_ST_I2))) {   // reached.rs:104-124
                const uint32_t cst = get_score(ex, prev, EX_ST_M);
                if (score_add(cst, st == EX_ST_I ? C.o : C.o2) > current) return true;
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
        if (prev != EX_NIL && (st == EX_ST_I || (TP && st == EX_ST_I2))) {   // reached.rs:165-186
            const uint32_t cst = get_score(ex, prev, EX_ST_M);
            if (score_add(cst, st == EX_ST_I ? C.o : C.o2) > current) return true;
        }
        return false;
    }

    POA_HD bool prune(uint32_t score, uint32_t row, uint32_t off, uint32_t st) {
        const uint32_t b0 = gld(&G.nbm_off[row]), b1 = gld(&G.nbm_off[row + 1]);
        if (b0 == b1) return false;
        for (uint32_t k = b0; k < b1; ++k)
            if (!can_improve_bubble(gld(&G.nbm[k]), row, off, st, score)) return true;
        return false;
    }

    // ---- bucket queue (queue.rs:31-70; gap_affine.rs:945-966) ----------------------------------
    // bucket-queue state (wave search).  A descriptor word is top << 6 | n_top; BQ_EMPTY = no chunk.  A stack that runs empty
    // keeps its last chunk (n_top == 0) until the search moves past its priority: the stacks of the current priority are
    // refilled all the time (a state pushed into the bucket being drained), and would otherwise take a new chunk each time.
    // Freed chunks go on a free list threaded through slot 0.
    uint32_t bq_live = 0, bq_chunk_top = 0, bq_hi = 0, bq_free = EX_NIL;
    bool bq_wr = true;  // wave kernel: in the sections every lane executes alike, only one lane stores
    bool mark_on = true;   // same for the reached marks (atomics: sixty-four of them on one word otherwise)
    POA_HD uint32_t bq_alloc() {
        if (bq_free != EX_NIL) { const uint32_t c = bq_free; bq_free = W.bq_chunks[(uint64_t)BQ_CHUNK * c].x; return c; }
        if (bq_chunk_top >= W.bq_chunk_cap || bq_chunk_top >= (1u << 26)) { err = EX_POOL_FULL; return EX_NIL; }
        return bq_chunk_top++;
    }
    POA_HD void bq_release(uint32_t c) {
        if (bq_wr) W.bq_chunks[(uint64_t)BQ_CHUNK * c].x = bq_free;
        bq_free = c;
    }
    POA_HD void bq_push(uint32_t prio, uint32_t st, uint32_t score, uint32_t row, uint32_t off) {
        if (bq_live == 0) { layer_min = prio; bq_hi = prio; }  // queue.rs:32-35 (chunks kept by drained stacks are reused in place)
        else {
            const uint32_t lo = prio < layer_min ? prio : layer_min, hi = prio > bq_hi ? prio : bq_hi;
            if (hi - lo >= W.bq_win) { err = EX_POOL_FULL; return; }
            layer_min = lo; bq_hi = hi;
        }
        uint32_t* d = &W.bq_desc[3 * (prio & (W.bq_win - 1)) + st];
        const uint32_t dw = rld(d);
        uint32_t top = dw == BQ_EMPTY ? EX_NIL : dw >> 6, n = dw == BQ_EMPTY ? 0u : dw & 63u;
        if (top == EX_NIL || n == BQ_CHUNK - 1) {
            const uint32_t c = bq_alloc();
            if (c == EX_NIL) return;
            if (bq_wr) W.bq_chunks[(uint64_t)BQ_CHUNK * c] = ExU4{top, 0, 0, 0};
            top = c; n = 0;
        }
        n += 1;
        if (bq_wr) { W.bq_chunks[(uint64_t)BQ_CHUNK * top + n] = ExU4{score, row, off, 0}; rst(d, top << 6 | n); }
        bq_live += 1;
    }
    // the stack the next pop comes from: lowest live priority, Match before Deletion before Insertion
    POA_HD bool bq_current(uint32_t& st, BqDesc& d) {
        if (bq_live == 0) return false;
        for (;;) {
            uint32_t* b = &W.bq_desc[3 * (layer_min & (W.bq_win - 1))];
            const uint32_t d0 = rld(b), d1 = rld(b + 1), d2 = rld(b + 2);
            if (d0 != BQ_EMPTY && (d0 & 63u)) { st = 0; d = BqDesc{d0 >> 6, d0 & 63u}; return true; }
            if (d1 != BQ_EMPTY && (d1 & 63u)) { st = 1; d = BqDesc{d1 >> 6, d1 & 63u}; return true; }
            if (d2 != BQ_EMPTY && (d2 & 63u)) { st = 2; d = BqDesc{d2 >> 6, d2 & 63u}; return true; }
            // nothing left at this priority: its (empty) chunks go back, the ring slot is free for priority + bq_win
            if (d0 != BQ_EMPTY) { bq_release(d0 >> 6); if (bq_wr) rst(b, BQ_EMPTY); }
            if (d1 != BQ_EMPTY) { bq_release(d1 >> 6); if (bq_wr) rst(b + 1, BQ_EMPTY); }
            if (d2 != BQ_EMPTY) { bq_release(d2 >> 6); if (bq_wr) rst(b + 2, BQ_EMPTY); }
            if (layer_min >= bq_hi) return false;   // (cannot happen while bq_live counts right: never walk off the ring)
            layer_min += 1;
        }
    }
    // remove the `pops` top entries of the current stack (pops <= d.n_top); `prev` = slot 0 of its top chunk
    POA_HD void bq_drop_at(uint32_t prio, uint32_t st, BqDesc d, uint32_t pops, uint32_t prev) {
        d.n_top -= pops;
        if (d.n_top == 0 && prev != EX_NIL) { bq_release(d.top); d.top = prev; d.n_top = BQ_CHUNK - 1; }
        if (bq_wr) rst(&W.bq_desc[3 * (prio & (W.bq_win - 1)) + st], d.top << 6 | d.n_top);
        bq_live -= pops;
    }
    POA_HD void bq_drop(uint32_t st, BqDesc d, uint32_t pops, uint32_t prev) {
        d.n_top -= pops;
        if (d.n_top == 0 && prev != EX_NIL) { bq_release(d.top); d.top = prev; d.n_top = BQ_CHUNK - 1; }
        if (bq_wr) rst(&W.bq_desc[3 * (layer_min & (W.bq_win - 1)) + st], d.top << 6 | d.n_top);
        bq_live -= pops;
    }

    POA_HD void queue_state(uint32_t row, uint32_t off, uint32_t st, uint32_t new_score) {
        const uint32_t pr64 = new_score + h(row, off, st);
        num_queued += 1;
        if (in_spec()) { spec_queue(pr64, st, new_score, row, off); return; }
        if (W.bq_desc) { bq_push(pr64, st, new_score, row, off); return; }
        if (pr64 >= W.n_prio) { err = EX_POOL_FULL; return; }
        // a popped entry's slot is reused (free list threaded through `next`): the pool holds the entries that are LIVE at
        // once, not every entry ever queued (a cell is queued again each time its score drops)
        uint32_t e;
        if (pool_free != EX_NIL) { e = pool_free; pool_free = W.pool[e].next; }
        else { if (pool_top >= W.pool_cap) { err = EX_POOL_FULL; return; } e = pool_top++; }
        const uint32_t prio = pr64;
        if (n_layers == 0) { n_layers = 1; layer_min = prio; }
        else if (prio < layer_min) { n_layers += layer_min - prio; layer_min = prio; }
        else if (prio >= layer_min + n_layers) { n_layers = prio - layer_min + 1; }
        uint32_t* hd = &W.head[NST * (uint64_t)prio + stack_slot(st)];
        W.pool[e] = ExQEntry{new_score, row, off, *hd};
        *hd = e;
    }
    // log mode: an entry that would be popped before the next entry of the stack the step pops from stays with the lane
    // (lower bucket; same bucket and a state of equal or higher pop priority: M before D before I, gap_affine.rs:954-966);
    // everything else is logged and pushed when the step commits
    POA_HD void spec_queue(uint32_t prio, uint32_t st, uint32_t score, uint32_t row, uint32_t off) {
        if (prio >= (1u << 26)) { sp_flag(SPF_COMPLEX, 9); return; }
        const uint32_t key = prio << 2 | st;
        if (sl.flat) {
            if (sl.n_p >= SP_KP) { sp_flag(SPF_COMPLEX, 5); return; }
            sl.p[sl.n_p * sl.stride] = ExU4{score, row, off, key};
            sl.n_p += 1;
            if (key < sl.min_child) sl.min_child = key;
            return;
        }
        if (prio < sl.cur_f || (prio == sl.cur_f && st <= sl.root_st)) {
            if (sl.n_pd >= SP_KPD) { sp_flag(SPF_COMPLEX, 6); return; }
#pragma unroll
            for (uint32_t k = 0; k < SP_KPD; ++k) if (k == sl.n_pd) { sl.pd_key[k] = key; sl.pd_score[k] = score; sl.pd_row[k] = row; sl.pd_off[k] = off; }
            sl.n_pd += 1;
            return;
        }
        if (sl.n_p >= SP_KP) { sp_flag(SPF_COMPLEX, 5); return; }
        sl.p[sl.n_p * sl.stride] = ExU4{score, row, off, key};
        sl.n_p += 1;
    }
    // the pending entry the queue would pop next: lowest (priority, state), the most recently pushed among equals
    POA_HD bool spec_pending_pop(uint32_t& score, uint32_t& row, uint32_t& off, uint32_t& st) {
        if (sl.n_pd == 0) return false;
        uint32_t best = 0, bkey = sl.pd_key[0];
#pragma unroll
        for (uint32_t k = 1; k < SP_KPD; ++k) if (k < sl.n_pd && sl.pd_key[k] <= bkey) { best = k; bkey = sl.pd_key[k]; }
        uint32_t key = 0;
#pragma unroll
        for (uint32_t k = 0; k < SP_KPD; ++k) if (k == best) { key = sl.pd_key[k]; score = sl.pd_score[k]; row = sl.pd_row[k]; off = sl.pd_off[k]; }
        st = key & 3u;
        // remove it, keeping the push order of the others
#pragma unroll
        for (uint32_t k = 0; k + 1 < SP_KPD; ++k) if (k >= best) { sl.pd_key[k] = sl.pd_key[k + 1]; sl.pd_score[k] = sl.pd_score[k + 1]; sl.pd_row[k] = sl.pd_row[k + 1]; sl.pd_off[k] = sl.pd_off[k + 1]; }
        sl.n_pd -= 1;
        return true;
    }
    // Pop order of the stacks of a priority: Match, Deletion, Insertion (gap_affine.rs:954-966); two-piece: Match, Deletion,
    // Deletion2, Insertion, Insertion2 (gap_affine_2piece.rs:1069-1097).  The table below maps a state to its stack and back.
    static POA_HD uint32_t stack_slot(uint32_t st) {
        if constexpr (TP) return st == EX_ST_I ? 3u : st == EX_ST_D2 ? 2u : st;
        else return st;
    }
    POA_HD bool layer_empty(uint32_t prio) const {
        const uint32_t* hd = &W.head[NST * (uint64_t)prio];
        for (uint32_t s = 0; s < NST; ++s) if (hd[s] != EX_NIL) return false;
        return true;
    }
    POA_HD bool pop_state(uint32_t& score, uint32_t& row, uint32_t& off, uint32_t& st) {
        if (n_layers == 0) return false;
        uint32_t* hd = &W.head[NST * (uint64_t)layer_min];
        bool got = false;
        for (uint32_t s = 0; s < NST; ++s) {  // the stacks in pop order (stack_slot is its own inverse)
            if (hd[s] != EX_NIL) {
                const uint32_t ix = hd[s];
                const ExQEntry e = W.pool[ix];
                hd[s] = e.next;
                W.pool[ix].next = pool_free; pool_free = ix;
                score = e.score; row = e.row; off = e.offset; st = stack_slot(s);
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
        if (C.qfe_kind == EX_BOUND_UNBOUNDED) q_ok = TP ? (off >= L || L == 0)       // gap_affine_2piece.rs:246-250
                                                          : (off > 0 || L == 0);     // sic: any consumed prefix may end
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
        const uint32_t ng = score_add(score_add(score, C.o), C.e);
        if (err) return;
        if (coff >= W.pitch || poff + 1 >= W.pitch) {  // (a write past the row is a workspace overflow, see update_if_lower)
            if (update_if_lower(crow, coff, EX_ST_M, nm)) queue_state(crow, coff, EX_ST_M, nm);
            if (update_if_lower(prow, poff + 1, EX_ST_I, ng)) queue_state(prow, poff + 1, EX_ST_I, ng);
            if (update_if_lower(crow, poff, EX_ST_D, ng)) queue_state(crow, poff, EX_ST_D, ng);
            return;
        }
        // three different cells: read them together, then relax in the reference's order (gap_affine.rs:393-430)
        EX_TRACE_CELL(crow, coff, EX_ST_M, 0); EX_TRACE_CELL(prow, poff + 1, EX_ST_I, 0); EX_TRACE_CELL(crow, poff, EX_ST_D, 0);
        const uint32_t im = cix(crow, coff, EX_ST_M), ii = cix(prow, poff + 1, EX_ST_I), id = cix(crow, poff, EX_ST_D);
        uint32_t vm = W.T[im], vi = W.T[ii], vd = W.T[id];
        if (in_spec()) {
            note_cell(im); note_cell(ii); note_cell(id);
            if (sl.n_w) { vm = ovl(im, vm); vi = ovl(ii, vi); vd = ovl(id, vd); }
        }
        if (nm < vm) { wr(crow, coff, EX_ST_M, nm); queue_state(crow, coff, EX_ST_M, nm); }
        if (ng < vi) { wr(prow, poff + 1, EX_ST_I, ng); queue_state(prow, poff + 1, EX_ST_I, ng); }
        if (ng < vd) { wr(crow, poff, EX_ST_D, ng); queue_state(crow, poff, EX_ST_D, ng); }
    }
    POA_HD void expand_all(uint32_t score, uint32_t row, uint32_t off, uint32_t st) {
        if (update_if_lower(row, off, EX_ST_M, score)) queue_state(row, off, EX_ST_M, score);
        if constexpr (TP) {
            // gap_affine_2piece.rs:346-431: a state of the first piece extends in its piece (extend1) and moves to the second
            // (extend2); a state of the second piece extends there; open2 is never charged
            const uint32_t n1 = score_add(score, C.e), n2 = score_add(score, C.e2);
            if (err) return;
            if (st == EX_ST_I || st == EX_ST_I2) {
                if (st == EX_ST_I && off < L && update_if_lower(row, off + 1, EX_ST_I, n1)) queue_state(row, off + 1, EX_ST_I, n1);
                if (off < L && update_if_lower(row, off + 1, EX_ST_I2, n2)) queue_state(row, off + 1, EX_ST_I2, n2);
            } else {
                for (uint32_t e = gld(&G.succ_off[row]); e < gld(&G.succ_off[row + 1]); ++e) {
                    const uint32_t c = gld(&G.succ[e]);
                    if (st == EX_ST_D && update_if_lower(c, off, EX_ST_D, n1)) queue_state(c, off, EX_ST_D, n1);
                    if (update_if_lower(c, off, EX_ST_D2, n2)) queue_state(c, off, EX_ST_D2, n2);
                    if (err) return;
                }
            }
            return;
        }
        if (st == EX_ST_I) {
            const uint32_t ns = score_add(score, C.e);
            if (err) return;
            if (off < L && update_if_lower(row, off + 1, EX_ST_I, ns)) queue_state(row, off + 1, EX_ST_I, ns);
        } else {
            for (uint32_t e = gld(&G.succ_off[row]); e < gld(&G.succ_off[row + 1]); ++e) {
                const uint32_t ns = score_add(score, C.e);
                if (err) return;
                if (update_if_lower(gld(&G.succ[e]), off, EX_ST_D, ns)) queue_state(gld(&G.succ[e]), off, EX_ST_D, ns);
            }
        }
    }

    // ---- depth-first greedy extension (dfa.rs:138-250) -----------------------------------------
    enum : uint32_t { EV_NONE = 0, EV_REF_GRAPH_END = 1, EV_QUERY_END = 2, EV_MISMATCH = 3 };
    struct Event { uint32_t kind, prow, poff, crow, coff; };
    // The stack of the extension (dfa.rs:86-134).  Its top entry lives in registers (dfa_top), W.stack holds what is below
    // it: the loop below works on the top only, and a read-after-write of a stack slot in memory costs a full round trip.
    // A parent that has no successor left once a child is taken is not kept under that child (popping it later has no
    // effect in the reference: SuccessorsExhausted, dfa.rs:246) — along chains the stack stays one entry deep.  The stack
    // length is therefore not the reference's; the one place that reads it (the offset-0 special case, dfa.rs:146) also
    // requires the bottom entry itself, which alone has offset 0.
    uint32_t sp = 0, dfa_visited = 0;
    uint32_t dfa_score = 0;
    ExStackEntry dfa_top{0, 0, 0};
    POA_HD void dfa_start(uint32_t row, uint32_t off) {
        sp = 1;
        dfa_top = ExStackEntry{row, off, gld(&G.succ_off[row])};
    }
    POA_HD void dfa_push(uint32_t row, uint32_t off, bool parent_exhausted) {
        if (!parent_exhausted) {
            if (in_spec()) {   // (the stack in memory is the query's; a lane keeps a few entries of its own)
                if (sp - 1 >= SP_KDS) { sp_flag(SPF_COMPLEX, 7); return; }
                sl.dstack[(sp - 1) * sl.stride] = dfa_top;
                sp += 1;
                dfa_top = ExStackEntry{row, off, gld(&G.succ_off[row])};
                return;
            }
            // (a select, not a store under a branch: the optimiser otherwise merges that store with the one to dfa_top
            // below into one store through a pointer phi, and both members then live in scratch memory)
            const bool full = sp >= W.stack_cap;
            err = full ? (uint32_t)EX_POOL_FULL : err;
            if (full) return;
            W.stack[sp - 1] = dfa_top;
            sp += 1;
        }
        dfa_top = ExStackEntry{row, off, gld(&G.succ_off[row])};
    }
    POA_HD void dfa_pop() {
        sp -= 1;
        if (sp != 0) dfa_top = in_spec() ? sl.dstack[(sp - 1) * sl.stride] : W.stack[sp - 1];
    }

    POA_HD Event dfa_extend() {
        if (sp == 1 && L != 0) {
            const ExStackEntry init = dfa_top;
            if (init.offset == 0 && is_symbol_equal(init.row, seq[0])) {
                if (update_if_lower(init.row, 1, EX_ST_M, dfa_score)) {
                    dfa_top = ExStackEntry{init.row, 1, gld(&G.succ_off[init.row])};
                    mark_reached(init.row, 1, EX_ST_M);
                    dfa_visited += 1;
                    if (1 == L) return Event{EV_REF_GRAPH_END, init.row, 0, init.row, 1};
                }
            }
        }
        while (sp != 0) {
            if (in_spec() && (sl.flags & SPF_COMPLEX)) break;
            ExStackEntry& parent = dfa_top;
            const uint32_t cend = gld(&G.succ_off[parent.row + 1]);
            bool again = false;
            while (parent.it < cend) {
                const uint32_t child = gld(&G.succ[parent.it++]);
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
                        dfa_push(child, coff, parent.it >= cend);
                        if (err) return Event{EV_NONE, 0, 0, 0, 0};
                        again = true;
                        break;
                    }
                } else {
                    return Event{EV_MISMATCH, parent.row, parent.offset, child, coff};
                }
            }
            if (err) return Event{EV_NONE, 0, 0, 0, 0};
            if (!again) dfa_pop();
        }
        return Event{EV_NONE, 0, 0, 0, 0};
    }
    uint32_t num_pruned_dfa = 0;  // DFA-internal count (dfa.rs:103); NOT added to AstarResult::num_pruned


    // ---- one-round-trip path for the common shape of a popped state ----------------------------------
    // A node with a single successor whose bubbles (other than the one it exits itself) all lie one fixed distance ahead —
    // every node of a chain, every allele of a SNP bubble — needs, to be tested and expanded, a fixed set of cells: its own,
    // the three (two) cells it may relax, the query symbol, and of the exit row the reached bits around the target offset
    // t with the Match scores next to t.  inspect_fast issues ALL those loads before the first use, so that the whole test
    // costs one memory round trip (the generic code chases pointer after pointer: word of the set, then the cell it names,
    // then the next word ...).  Whatever does not fit the shape — several successors, a greedy match to extend, the
    // end row, a bubble with paths of different lengths, far-away reached offsets, ends-free spans — takes the generic code:
    // same results, more round trips.  The logic below is reached.rs:38-255 specialised to tmin == tmax.
    uint32_t n_fast = 0;      // states tested on this path (statistics)
#if defined(POA_EXACT_DIAG)
    uint32_t diag[8][3] = {};   // [why inspect_fast declined][state]
#define EXD(r, st) (diag[r][st]++)
#else
#define EXD(r, st) ((void)0)
#endif
    // the test of one bubble whose exit lies a fixed distance ahead (tmin == tmax == t): reached bits around t and the
    // Match scores next to t, loaded together
    struct Probe {
        uint32_t on;          // 0: nothing to test (no such bubble, pruning off, t beyond the query)
        uint32_t ex, t, wi;
        uint32_t x, mde;      // index of the exit row among the exit rows; max(dist_min[exit] - 1, 0)
        uint64_t w1, sum;     // the word of the reached set that holds t; the summary word (which words are non-empty)
        uint32_t ta, tb, tc;  // Match score of the exit row at t - 1, t, t + 1
    };
    // false: row v has not this shape (several bubbles ahead, or paths of different lengths through one)
    // (pop_level: the test at a pop is subject to enable_pruning, astar.rs:155; the one inside the greedy extension is not, dfa.rs:185)
    // Up to two such bubbles (the first node of an inserted branch lies in its own one-step bubble and in the branch's): P,
    // then Q in the reference's order — a state is pruned as soon as one of its bubbles cannot be improved (gap_affine.rs:780-792).
    POA_HD bool probe_setup(uint32_t v, uint32_t j, Probe& P, Probe& Q, bool pop_level) const {
        uint32_t ex[2] = {EX_NIL, EX_NIL}, dist[2] = {0, 0}, n = 0;
        for (uint32_t k = gld(&G.nbm_off[v]), k1 = gld(&G.nbm_off[v + 1]); k < k1; ++k) {
            const FlatGraph::NodeBubble b = gld(&G.nbm[k]);
            if (b.exit_row == v) continue;                  // reached.rs:56-58
            if (n == 2 || b.min_dist != b.max_dist) return false;
            ex[n] = b.exit_row; dist[n] = b.min_dist; n += 1;
        }
        P.ex = ex[0]; P.t = j + dist[0]; P.wi = P.t >> 6;
        P.on = ((C.prune || !pop_level) && ex[0] != EX_NIL && P.t <= L) ? 1u : 0u;   // reached.rs:63-65: tmax > len -> can improve
        Q.ex = ex[1]; Q.t = j + dist[1]; Q.wi = Q.t >> 6;
        Q.on = ((C.prune || !pop_level) && ex[1] != EX_NIL && Q.t <= L) ? 1u : 0u;
        P.x = P.mde = Q.x = Q.mde = 0;
        if (P.on) { P.x = gld(&G.exit_idx[P.ex]); const uint32_t m = gld(&G.dist_min[P.ex]); P.mde = m ? m - 1 : 0; }
        if (Q.on) { Q.x = gld(&G.exit_idx[Q.ex]); const uint32_t m = gld(&G.dist_min[Q.ex]); Q.mde = m ? m - 1 : 0; }
        return !((P.on && P.t + 1 >= W.pitch) || (Q.on && Q.t + 1 >= W.pitch));   // (the loads below read t + 1)
    }
    // the same from the row's record (FlatGraph::RowRec: one 32-byte load instead of the walk through the bubble map, the exit
    // index and the exit's distance); false also where the record does not hold the whole truth (the generic code decides)
    POA_HD bool probe_setup_rec(const FlatGraph::RowRec& rr, uint32_t j, Probe& P, Probe& Q, bool pop_level) const {
        if (!(rr.flags & FlatGraph::RR_PROBE_OK) || rr.d0min != rr.d0max || rr.d1min != rr.d1max) return false;
        const bool en = C.prune || !pop_level;
        P.ex = rr.e0 == 0xFFFFu ? EX_NIL : (uint32_t)rr.e0; P.t = j + rr.d0min; P.wi = P.t >> 6; P.x = rr.x0; P.mde = rr.mde0;
        P.on = (en && rr.e0 != 0xFFFFu && P.t <= L) ? 1u : 0u;
        Q.ex = rr.e1 == 0xFFFFu ? EX_NIL : (uint32_t)rr.e1; Q.t = j + rr.d1min; Q.wi = Q.t >> 6; Q.x = rr.x1; Q.mde = rr.mde1;
        Q.on = (en && rr.e1 != 0xFFFFu && Q.t <= L) ? 1u : 0u;
        return !((P.on && P.t + 1 >= W.pitch) || (Q.on && Q.t + 1 >= W.pitch));
    }
    // both tests: 0 can improve, 2 pruned, 3 undecided (the generic code decides)
    // (a branch-free restatement of probe_decide — selects instead of its nested conditions — was measured 2.5 % SLOWER in the
    // wave kernel: most probes leave at one of the early exits)
    POA_HD uint32_t probe_decide2(const Probe& P, const Probe& Q, uint32_t g, uint32_t st) {
        const uint32_t r = probe_decide(P, g, st);
        if (r != 0 || !Q.on) return r;
        return probe_decide(Q, g, st);
    }
    POA_HD void probe_load(Probe& P) const {
        P.w1 = P.sum = 0; P.ta = P.tb = P.tc = EX_INF;
        if (!P.on) return;
        const uint32_t x = P.x;
        P.sum = rsw(x, 0);
        P.w1 = rword(x, P.wi);
        const uint32_t ia = cix(P.ex, P.t ? P.t - 1 : 0, EX_ST_M), ib = cix(P.ex, P.t, EX_ST_M), ic = cix(P.ex, P.t + 1, EX_ST_M);
        const uint32_t va = W.T[ia], vb = W.T[ib], vc = W.T[ic];
        if (P.t) P.ta = va;
        P.tb = vb; P.tc = vc;
        if (in_spec() && sl.n_w) { if (P.t) P.ta = ovl(ia, va); P.tb = ovl(ib, vb); P.tc = ovl(ic, vc); }
    }
    // 0: can improve (not pruned); 2: pruned; 3: something the generic code has to look at (a reached cell without a score).
    // reached.rs:38-255 specialised to tmin == tmax.
    POA_HD uint32_t probe_decide(const Probe& P, uint32_t g, uint32_t st) {
        if (!P.on) return 0;                        // no bubble to test
        if (P.sum == 0) { if (in_spec()) note_marks(P.x, 0, 0xFFFFFFFFu); return 0; }   // nothing reached at the exit yet (reached.rs:52-54)
        const uint32_t t = P.t, wi = P.wi, ex = P.ex;
        uint32_t prev = EX_NIL, nxt = EX_NIL;
        const bool at_t = (P.w1 >> (t & 63)) & 1;
        const uint64_t lo = (t & 63) ? (P.w1 & (~0ull >> (64 - (t & 63)))) : 0ull;
        // (a neighbour outside t's word — t next to a word boundary, or a sparse set — is found through the summary: one more load)
        if (lo) prev = wi * 64 + 63 - (uint32_t)clz64(lo);
        else if (wi && (P.sum & (~0ull >> (64 - wi)))) prev = reached_before(ex, wi * 64);
        const uint64_t hi = (t & 63) != 63 ? (P.w1 & (~0ull << ((t & 63) + 1))) : 0ull;
        if (hi) nxt = wi * 64 + (uint32_t)ctz64(hi);
        else if (wi + 1 < W.wpn && (P.sum >> (wi + 1))) nxt = reached_from(ex, (wi + 1) * 64);
#if defined(POA_EXACT_DIAG)
        { const uint32_t dl = prev == EX_NIL ? 99 : t - prev, dr = (at_t || nxt == EX_NIL) ? 99 : nxt - t;
          const uint32_t bl = dl == 99 ? 7 : dl > 8 ? 6 : dl > 4 ? 5 : dl > 2 ? 4 : dl, br = dr == 99 ? 7 : dr > 8 ? 6 : dr > 4 ? 5 : dr > 2 ? 4 : dr;
          pd_hist[0][bl] += 1; pd_hist[1][br] += 1; }
#endif
        // what the decision read of the exit row: the marks between the two neighbours and the scores at them (reached at t:
        // the neighbour above is not looked at, reached.rs:139)
        if (in_spec()) {
            note_marks(P.x, prev == EX_NIL ? 0u : prev, at_t ? t : nxt);
            if (prev != EX_NIL && prev < W.pitch) note_cell(cix(ex, prev, EX_ST_M));
            if (at_t) note_cell(cix(ex, t, EX_ST_M));
            else if (nxt != EX_NIL && nxt < W.pitch) note_cell(cix(ex, nxt, EX_ST_M));
        }
        // scores of the two neighbours: next to t they are loaded already; else both loads go out together
        const uint32_t iq = cix(ex, prev != EX_NIL && prev < W.pitch ? prev : t, EX_ST_M), jq = cix(ex, nxt != EX_NIL && nxt < W.pitch ? nxt : t, EX_ST_M);
        uint32_t lq = W.T[iq], rq = W.T[jq];
        if (in_spec() && sl.n_w) { lq = ovl(iq, lq); rq = ovl(jq, rq); }
        uint32_t ls = 0, rs = 0;
        if (prev != EX_NIL) ls = prev + 1 == t ? P.ta : lq;
        if (nxt != EX_NIL) rs = nxt == t + 1 ? P.tc : (nxt < W.pitch ? rq : EX_INF);
        const uint32_t tb = P.tb;
        // a reached cell holds a score; if one does not, the generic code reports what the reference would (a panic)
        if ((prev != EX_NIL && ls == EX_INF) || (nxt != EX_NIL && rs == EX_INF) || (at_t && tb == EX_INF)) return 3;
        const uint32_t mde = P.mde;
        bool improve;
        if (at_t) {
            // the loop body of reached.rs:67-141 runs once with next == t; afterwards prev == t (reached.rs:139, :177-186)
            improve = (st == EX_ST_D && tb + C.o > g) || (prev != EX_NIL && st == EX_ST_I && ls + C.o > g);
            if (!improve) {
                const uint32_t fl = prev != EX_NIL ? ls + gap_cost(EX_ST_M, t - prev) : EX_INF;
                improve = g < (fl < tb ? fl : tb);
            }
            if (!improve && st == EX_ST_I) improve = tb + C.o > g;
        } else {
            // reached.rs:143-186 with an empty range: one test at t between prev and nxt
            if (prev != EX_NIL && nxt != EX_NIL) {
                const uint32_t fl = ls + gap_cost(EX_ST_M, t - prev), fr = rs + gap_cost(EX_ST_M, nxt - t);
                improve = g < ((nxt - t > mde) ? fl : (fl < fr ? fl : fr));
            } else if (nxt != EX_NIL) {
                improve = (nxt - t > mde) ? true : g < rs + gap_cost(EX_ST_M, nxt - t);
            } else if (prev != EX_NIL) {
                improve = g < ls + gap_cost(EX_ST_M, t - prev);
            } else improve = true;
            if (!improve && prev != EX_NIL && st == EX_ST_I) improve = ls + C.o > g;
        }
        return improve ? 0 : 2;
    }

    struct FastItem {
        uint32_t kind;        // 0: generic path; 1: expands without a greedy match; 2: Match state whose first successor matches
        uint32_t c, c1;       // the successor rows (c1 == EX_NIL: one successor)
        uint32_t t0, t1, t2;  // M: M[c][j+1], I[v][j+1], D[c][j];  I: M[v][j], I[v][j+1];  D: M[v][j], D[c][j]
        uint32_t t3, t4;      // second successor — M: M[c1][j+1], D[c1][j];  D: D[c1][j]
    };
    // Test of a popped state: 0 goes on to process_fast; 1 stale; 2 pruned; 3 not this shape (F.kind == 0: inspect_skip decides).
    // The shape: one or two successors, none of them the end row (an Insertion state: any), for a Match state query symbols left
    // and a second successor that does not match; the bubbles decide only HOW the pruning test is made (probes, else the
    // generic code).
    POA_HD uint32_t inspect_fast(uint32_t g, uint32_t v, uint32_t j, uint32_t st, FastItem& F) { return inspect_fast_t<false>(g, v, j, st, F); }
    // RANGE: also the bubbles whose paths differ in length (the range form of the test: three dependent loads).  The wave kernel
    // keeps that out of the test its lanes run together — every lane would wait for the one that has such a row — and
    // asks again with RANGE for the entry its run stopped at.
    template <bool RANGE>
    POA_HD uint32_t inspect_fast_t(uint32_t g, uint32_t v, uint32_t j, uint32_t st, FastItem& F) {
        F.kind = 0;
        if constexpr (TP) return 3;   // (two-piece model: the generic code)
        uint32_t c, c1, kind = 1;
        Probe P, Q;
        bool probes, use_range = false;
        FlatGraph::RowRec rr_keep{};
        if (use_rec()) {
            // everything the test needs to know of the row in one record: successors, their symbols, the bubbles ahead
            const FlatGraph::RowRec rr = lrec(v);
            const uint32_t fl = rr.flags;
            if (C.ends_free || (fl & FlatGraph::RR_END) || rr.c0 == 0xFFFFu || (st != EX_ST_I && !(fl & FlatGraph::RR_SUCC_OK)) || W.swpn != 1 || j + 2 >= W.pitch || g >= 0xFFFF0000u) { EXD((st != EX_ST_I && !(fl & FlatGraph::RR_SUCC_OK) && !(fl & FlatGraph::RR_END)) ? 1 : 2, st); return 3; }
            c = rr.c0;
            c1 = ((fl & FlatGraph::RR_HAS_C1) && st != EX_ST_I) ? (uint32_t)rr.c1 : EX_NIL;
            if (st == EX_ST_M) {
                if (j >= L) { EXD(4, st); return 3; }
                if (j == 0 && L != 0 && rr.sym == seq[0]) { EXD(4, st); return 3; }   // the offset-0 special case, dfa.rs:146-167
                const uint8_t qc = seq[j];
                if (rr.sym0 == qc) kind = 2;
                if (c1 != EX_NIL && rr.sym1 == qc) { EXD(1, st); return 3; }
            }
            probes = probe_setup_rec(rr, j, P, Q, true);
            if (RANGE && !probes && (fl & FlatGraph::RR_PROBE_OK)) {
                // bubbles whose paths differ in length by one or two (an inserted branch ahead): the range form of the test
                // (three dependent loads instead of one round trip; the generic code costs several times that)
                use_range = true; rr_keep = rr;
            }
        } else {
        const uint32_t s0 = gld(&G.succ_off[v]), s1 = gld(&G.succ_off[v + 1]);
        const uint32_t ns = s1 - s0;
        // (an Insertion state never looks at the successors — expand_all, gap_affine.rs:307-341)
        if (C.ends_free || ((ns == 0 || ns > 2) && (st != EX_ST_I || ns == 0)) || v == G.end_row || W.swpn != 1 || j + 2 >= W.pitch || g >= 0xFFFF0000u) { EXD(ns != 1 ? 1 : 2, st); return 3; }
        c = gld(&G.succ[s0]);
        c1 = (ns == 2 && st != EX_ST_I) ? gld(&G.succ[s0 + 1]) : EX_NIL;
        if (st != EX_ST_I && (c == G.end_row || c1 == G.end_row)) { EXD(4, st); return 3; }
        if (st == EX_ST_M) {
            // a Match state goes through the greedy extension
            if (j >= L) { EXD(4, st); return 3; }
            if (j == 0 && L != 0 && is_symbol_equal(v, seq[0])) { EXD(4, st); return 3; }  // the offset-0 special case, dfa.rs:146-167
            const uint8_t qc = seq[j];
            if (gld(&G.sym[c]) == qc) kind = 2;
            if (c1 != EX_NIL && gld(&G.sym[c1]) == qc) { EXD(1, st); return 3; }   // (a second branch to extend: the generic code keeps the stack)
        }
        probes = probe_setup(v, j, P, Q, true);
        }
        LProbe R0, R1;
        if constexpr ((AS & EX_AS_NO_SPEC) != 0) {
            if (use_range) {
                lp_setup(R0, rr_keep, 0, j, C.prune != 0); lp_setup(R1, rr_keep, 1, j, C.prune != 0);
                if ((R0.on && R0.tmax + 1 >= W.pitch) || (R1.on && R1.tmax + 1 >= W.pitch)) { EXD(3, st); return 3; }
            } else if (!probes) { EXD(3, st); return 3; }   // (wave kernels: the generic code takes it, nothing loaded twice)
        }
        // ---- every load of the step, before any use ----
        const uint32_t i_own = cix(v, j, st);
        uint32_t i0, i1, i2 = i_own, i3 = i_own, i4 = i_own;
        if (st == EX_ST_M) { i0 = cix(c, j + 1, EX_ST_M); i1 = cix(v, j + 1, EX_ST_I); i2 = cix(c, j, EX_ST_D); if (c1 != EX_NIL) { i3 = cix(c1, j + 1, EX_ST_M); i4 = cix(c1, j, EX_ST_D); } }
        else if (st == EX_ST_I) { i0 = cix(v, j, EX_ST_M); i1 = cix(v, j < L ? j + 1 : j, EX_ST_I); }
        else { i0 = cix(v, j, EX_ST_M); i1 = cix(c, j, EX_ST_D); if (c1 != EX_NIL) i3 = cix(c1, j, EX_ST_D); }
        uint32_t own = W.T[i_own];
        uint32_t t0 = W.T[i0], t1 = W.T[i1], t2 = st == EX_ST_M ? W.T[i2] : EX_INF;
        uint32_t t3 = c1 != EX_NIL ? W.T[i3] : EX_INF, t4 = (c1 != EX_NIL && st == EX_ST_M) ? W.T[i4] : EX_INF;
        if (probes) { probe_load(P); probe_load(Q); }
        if (use_range) { lp_load1(R0); lp_load1(R1); }
        if (in_spec()) {
            // (the probes note what they looked at when they decide)
            note_cell(i_own); note_cell(i0); if (st != EX_ST_I || j < L) note_cell(i1); if (st == EX_ST_M) note_cell(i2);
            if (c1 != EX_NIL) { note_cell(i3); if (st == EX_ST_M) note_cell(i4); }
            if (sl.n_w) {
                own = ovl(i_own, own); t0 = ovl(i0, t0); t1 = ovl(i1, t1); if (st == EX_ST_M) t2 = ovl(i2, t2);
                if (c1 != EX_NIL) { t3 = ovl(i3, t3); if (st == EX_ST_M) t4 = ovl(i4, t4); }
            }
        }
        if (st == EX_ST_I && j >= L) t1 = EX_INF;
        n_fast += 1;
        if (g > own) return 1;                      // stale (astar.rs:146)
        uint32_t r = probes ? probe_decide2(P, Q, g, st) : 3u;
        if (use_range) {
            lp_locate1(R0); lp_locate1(R1); lp_load2(R0); lp_load2(R1); lp_locate2(R0); lp_locate2(R1); lp_load3(R0); lp_load3(R1);
            r = lp_decide(R0, g, st);
            if (r == 0) r = lp_decide(R1, g, st);
        }
        if constexpr ((AS & EX_AS_NO_SPEC) != 0) { if (r == 3) { EXD(probes ? 5 : 3, st); return 3; } }   // (wave kernel: the entry's own lane takes the generic code after the run)
        if (r == 3) { PF_TICK(1); r = (C.prune && prune(g, v, j, st)) ? 2u : 0u; PF_TICK(5); }   // bubbles of another shape: the generic test (astar.rs:155)
        if (err) return 3;
        EXD(r == 0 ? (kind == 2 ? 7 : 6) : 0, st);
        F.kind = kind; F.c = c; F.c1 = c1; F.t0 = t0; F.t1 = t1; F.t2 = t2; F.t3 = t3; F.t4 = t4;
        return r;
    }
    // The expansion of a state inspect_fast let through.  true: the search ends here (only through the generic tail).
    POA_HD bool process_fast(uint32_t g, uint32_t v, uint32_t j, uint32_t st, const FastItem& F, ExactResult& R, uint32_t& end_score) {
        mark_reached(v, j, st);
        num_visited += 1;
        if (err) return false;
        const uint32_t c = F.c, c1 = F.c1;
        if (st == EX_ST_I) {
            if (g < F.t0) { wr(v, j, EX_ST_M, g); queue_state(v, j, EX_ST_M, g); }
            const uint32_t ns = g + C.e;
            if (j < L && ns < F.t1) { wr(v, j + 1, EX_ST_I, ns); queue_state(v, j + 1, EX_ST_I, ns); }
            return false;
        }
        if (st == EX_ST_D) {
            if (g < F.t0) { wr(v, j, EX_ST_M, g); queue_state(v, j, EX_ST_M, g); }
            const uint32_t ns = g + C.e;
            if (ns < F.t1) { wr(c, j, EX_ST_D, ns); queue_state(c, j, EX_ST_D, ns); }
            if (c1 != EX_NIL && ns < F.t3) { wr(c1, j, EX_ST_D, ns); queue_state(c1, j, EX_ST_D, ns); }
            return false;
        }
        const uint32_t nm = g + C.x, ng = g + C.o + C.e;
        bool i_open = false;   // I[v][j+1] relaxed by a mismatch event of this row (a second one cannot lower it again)
        if (F.kind == 1) {
            // the first successor mismatches -> expand_mismatch (gap_affine.rs:393-430)
            if (nm < F.t0) { wr(c, j + 1, EX_ST_M, nm); queue_state(c, j + 1, EX_ST_M, nm); }
            if (ng < F.t1) { wr(v, j + 1, EX_ST_I, ng); queue_state(v, j + 1, EX_ST_I, ng); }
            if (ng < F.t2) { wr(c, j, EX_ST_D, ng); queue_state(c, j, EX_ST_D, ng); }
            i_open = true;
        } else {
        // Greedy extension along single-successor rows (dfa.rs:138-250).  A parent with one successor has nothing left once
        // that successor is taken, so it need not stay on the stack: the walk keeps only its tip.  One round trip per
        // matched base: the tip's cell, the bubble test of the tip and the cells a mismatch would relax, all loaded together.
        // (The row the walk starts from may have a second successor: it is looked at when the walk is over, below.)
        uint32_t cj = j, cc = c, tm = F.t0;
        dfa_visited = 0; dfa_score = g;
        for (;;) {
            // here: cc is a successor of the row before, not the end row, cj < L, sym(cc) == seq[cj]
            const uint32_t nj = cj + 1;
            if (in_spec() && (sl.flags & SPF_COMPLEX)) break;   // (the lane is cut off: whatever it does from here on is discarded)
            if (!(g < tm)) break;                          // already there with this score or better: not extended (dfa.rs:242)
            wr(cc, nj, EX_ST_M, g);
            // what the tip needs next: its own successor, its bubble test, and what a mismatch there relaxes
            uint32_t nc, nsym = 0;
            Probe P, Q;
            bool shaped;
            if (use_rec()) {
                const FlatGraph::RowRec rt = lrec(cc);
                nc = ((rt.flags & (FlatGraph::RR_SUCC_OK | FlatGraph::RR_HAS_C1)) == FlatGraph::RR_SUCC_OK) ? (uint32_t)rt.c0 : EX_NIL;   // the single successor, not the end row
                nsym = rt.sym0;
                shaped = probe_setup_rec(rt, nj, P, Q, false);
            } else {
                const uint32_t s0 = gld(&G.succ_off[cc]), s1 = gld(&G.succ_off[cc + 1]);
                nc = s1 - s0 == 1 ? gld(&G.succ[s0]) : EX_NIL;
                shaped = probe_setup(cc, nj, P, Q, false);
            }
            const bool walk_on = shaped && nc != EX_NIL && nc != G.end_row && nj < L && nj + 2 < W.pitch;
            uint32_t n0 = EX_INF, n1 = EX_INF, n2 = EX_INF;
            if (shaped) { probe_load(P); probe_load(Q); }
            if (walk_on) {
                const uint32_t j0 = cix(nc, nj + 1, EX_ST_M), j1 = cix(cc, nj + 1, EX_ST_I), j2 = cix(nc, nj, EX_ST_D);
                n0 = W.T[j0]; n1 = W.T[j1]; n2 = W.T[j2];
                if (in_spec()) {
                    note_cell(j0); note_cell(j1); note_cell(j2);
                    if (sl.n_w) { n0 = ovl(j0, n0); n1 = ovl(j1, n1); n2 = ovl(j2, n2); }
                }
            }
            uint32_t pr = shaped ? probe_decide2(P, Q, g, EX_ST_M) : 3u;
            if (pr == 3) pr = prune(g, cc, nj, EX_ST_M) ? 2u : 0u;
            if (err) break;
            if (pr == 2) { num_pruned_dfa += 1; break; }   // scored, not extended (dfa.rs:185-188)
            mark_reached(cc, nj, EX_ST_M);
            dfa_visited += 1;
            if (!walk_on) {
                // the tip is not a plain chain row: the generic extension goes on from it (its ancestors have nothing left)
                dfa_start(cc, nj);
                if (dfa_events(g, R, end_score)) return true;
                break;
            }
            if ((use_rec() ? nsym : (uint32_t)gld(&G.sym[nc])) != seq[nj]) {
                const uint32_t xm = g + C.x, xg = g + C.o + C.e;
                if (xm < n0) { wr(nc, nj + 1, EX_ST_M, xm); queue_state(nc, nj + 1, EX_ST_M, xm); }
                if (xg < n1) { wr(cc, nj + 1, EX_ST_I, xg); queue_state(cc, nj + 1, EX_ST_I, xg); }
                if (xg < n2) { wr(nc, nj, EX_ST_D, xg); queue_state(nc, nj, EX_ST_D, xg); }
                break;
            }
            cj = nj; cc = nc; tm = n0;
        }
        num_visited += dfa_visited;
        if (err) return false;
        }
        if (c1 != EX_NIL) {
            // the second successor (it mismatches: inspect_fast) -> its mismatch event, after whatever the first one led to
            if (nm < F.t3) { wr(c1, j + 1, EX_ST_M, nm); queue_state(c1, j + 1, EX_ST_M, nm); }
            if (!i_open && ng < F.t1) { wr(v, j + 1, EX_ST_I, ng); queue_state(v, j + 1, EX_ST_I, ng); }
            if (ng < F.t4) { wr(c1, j, EX_ST_D, ng); queue_state(c1, j, EX_ST_D, ng); }
        }
        return false;
    }

    // ---- one popped state (astar.rs:141-216) ------------------------------------------------------
    // 0: goes on to process_popped; 1: stale entry (astar.rs:146); 2: pruned (astar.rs:155-158).  Reads only.
    POA_HD uint32_t inspect_skip(uint32_t score, uint32_t row, uint32_t off, uint32_t st) {
        if (score > get_score(row, off, st)) return 1;
        if (is_end(row, off, st)) return 0;
        if (C.prune && prune(score, row, off, st)) return 2;
        return 0;
    }
    // the events of the greedy extension that stands on the stack (dfa_top + W.stack) (astar.rs:167-204); true: the search ends (R.end_* set)
    POA_HD bool dfa_events(uint32_t score, ExactResult& R, uint32_t& end_score) {
        for (;;) {
            const Event ev = dfa_extend();
            if (err || ev.kind == EV_NONE) break;
            if (ev.kind == EV_REF_GRAPH_END) {
                if (is_end(ev.crow, ev.coff, EX_ST_M)) { end_score = score; R.end_row = ev.crow; R.end_off = ev.coff; return true; }
                expand_ref_graph_end(ev.prow, ev.poff, score);
            } else if (ev.kind == EV_QUERY_END) {
                expand_query_end(ev.poff, ev.crow, score);
            } else {
                expand_mismatch(ev.prow, ev.poff, ev.crow, ev.coff, score);
            }
            if (err) break;
        }
        return false;
    }
    // everything after the stale / prune tests; true when the search ends here (R.end_* set)
    POA_HD bool process_popped(uint32_t score, uint32_t row, uint32_t off, uint32_t st, ExactResult& R, uint32_t& end_score) {
        if (is_end(row, off, st)) { num_visited += 1; end_score = score; R.end_row = row; R.end_off = off; return true; }
        if (err) return false;
        mark_reached(row, off, st);
        num_visited += 1;
        if (st == EX_ST_M) {
            dfa_visited = 0; dfa_score = score;
            dfa_start(row, off);
            if (dfa_events(score, R, end_score)) return true;
            num_visited += dfa_visited;  // skipped by `break 'main` (astar.rs:172,:205)
        } else {
            expand_all(score, row, off, st);
        }
        return false;
    }
    // the initial states (astar.rs:133-139)
    POA_HD void push_initial_states() {
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
    }

    // ---- main loop (astar.rs:124-226) -----------------------------------------------------------
    POA_HD ExactResult run() {
        ExactResult R{EX_OK, EX_INF, 0, 0, 0, G.end_row, L};
        push_initial_states();
        uint32_t end_score = EX_INF;
        bool found = false;
        while (!found && !err) {
            uint32_t score, row, off, st;
            if (!pop_state(score, row, off, st)) { err = EX_PANIC; break; }  // "Could not align sequence!"
            const uint32_t sk = inspect_skip(score, row, off, st);
            if (sk == 2) num_pruned += 1;
            if (sk || err) continue;
            found = process_popped(score, row, off, st, R, end_score);
        }
        R.status = err ? err : (found ? EX_OK : EX_PANIC);
        R.score = end_score;
        R.num_queued = num_queued; R.num_visited = num_visited; R.num_pruned = num_pruned;
        return R;
    }

    // The same search over the bucket queue, in the step structure of the wave kernel (poa_wsearch.hpp), one lane at a time:
    // a step looks at the top `batch` entries of the current stack; the leading entries that are stale or pruned change
    // nothing (inspect_skip only reads), so they are all popped together with the first entry that is neither, which is
    // then processed.  Order of every write and of every push is the reference's.  (Host build: the executable
    // specification of the kernel's schedule; tests/test_exact_replay.py diffs it against the oracle.)
    POA_HD ExactResult run_buckets(uint32_t batch, bool use_fast = true) {
        ExactResult R{EX_OK, EX_INF, 0, 0, 0, G.end_row, L};
        push_initial_states();
        uint32_t end_score = EX_INF;
        bool found = false;
        while (!found && !err) {
            uint32_t st; BqDesc d;
            if (!bq_current(st, d)) { err = EX_PANIC; break; }
            const uint32_t nb = d.n_top < batch ? d.n_top : batch;
            const ExU4* ch = W.bq_chunks + (uint64_t)BQ_CHUNK * d.top;
            uint32_t n = nb;
            ExU4 e{0, 0, 0, 0};
            FastItem F{0, 0, 0, 0, 0, 0, 0, 0};
            for (uint32_t i = 0; i < nb && !err; ++i) {
                e = ch[d.n_top - i];
                uint32_t sk = use_fast ? inspect_fast(e.x, e.y, e.z, st, F) : 3u;
                if (sk == 3 && use_fast && use_rec() && !err) sk = inspect_fast_t<true>(e.x, e.y, e.z, st, F);   // (the wave kernel's second ask)
                if (sk == 3) sk = inspect_skip(e.x, e.y, e.z, st);
                if (sk == 0) { n = i; break; }
                if (sk == 2) num_pruned += 1;
            }
            if (err) break;
            bq_drop(st, d, n < nb ? n + 1 : nb, ch[0].x);
            if (n < nb) {
                if (F.kind) found = process_fast(e.x, e.y, e.z, st, F, R, end_score);
                else found = process_popped(e.x, e.y, e.z, st, R, end_score);
            }
        }
        R.status = err ? err : (found ? EX_OK : EX_PANIC);
        R.score = end_score;
        R.num_queued = num_queued; R.num_visited = num_visited; R.num_pruned = num_pruned;
        return R;
    }

    // ---- log mode: one lane's share of a step (poa_psearch.hpp) ------------------------------------------------------------
    // Processes the entry (g, v, j) popped from stack (f, st) and the entries its expansion puts in front of the next entry
    // of that stack, at most `rmax` of them, reading the table as it stands (plus the lane's own log) and writing only to the
    // log.  sl.flags says how it ended; the counters of the search are left untouched (sl.dq / dv / dp carry the deltas).
    POA_HD void spec_group(uint32_t g, uint32_t v, uint32_t j, uint32_t st, uint32_t f, uint32_t rmax, ExactResult& R, uint32_t& end_score,
                           bool use_fast = true) {
        sl.n_w = sl.n_m = sl.n_p = sl.n_pd = 0; sl.flags = 0; sl.n_rc = sl.n_rm = 0; sl.n_ent = 0; sl.wmask = sl.mmask = 0;
        rc_x = rs_x = EX_NIL;
        sl.cur_f = f; sl.root_st = st;
        const uint32_t q0 = num_queued, v0 = num_visited, p0 = num_pruned;
        spec = true;
        for (;;) {
            // (an entry after the first that does not fit the logs is put back: it and what else is pending go to the queue)
            PF_START();
            const SpecLane snap = sl;
            const uint32_t q1 = num_queued, v1 = num_visited, p1 = num_pruned;
            if (sl.n_ent) spec_pending_pop(g, v, j, st);
            FastItem F{0, 0, 0, 0, 0, 0, 0, 0};
            PF_TICK(0);
            uint32_t sk = use_fast ? inspect_fast(g, v, j, st, F) : 3u;
            PF_TICK(1);
            if (sk == 3 && !err && !(sl.flags & SPF_COMPLEX)) sk = inspect_skip(g, v, j, st);
            PF_TICK(2);
            bool found = false;
            if (!err && !(sl.flags & SPF_COMPLEX)) {
                if (sk == 2) num_pruned += 1;
                if (sk == 0) {
                    if (F.kind) { found = process_fast(g, v, j, st, F, R, end_score); PF_TICK(3); }
                    else { found = process_popped(g, v, j, st, R, end_score); PF_TICK(4); }
                }
            }
            if (err || (sl.flags & SPF_COMPLEX)) {
                if (sl.n_ent == 0) break;
                sl = snap; err = EX_OK; num_queued = q1; num_visited = v1; num_pruned = p1;
                sl.flags |= SPF_LEFTOVER;
                break;
            }
            sl.n_ent += 1;
            if (found) { sl.flags |= SPF_FOUND; break; }
            if (sl.n_pd == 0) break;
            if (sl.n_ent >= rmax) { sl.flags |= SPF_LEFTOVER; break; }
        }
        spec = false;
        if (err) { sl.flags |= SPF_COMPLEX; err = EX_OK; }   // the sequential code meets the same condition and reports it
        sl.dq = num_queued - q0; sl.dv = num_visited - v0; sl.dp = num_pruned - p0;
        num_queued = q0; num_visited = v0; num_pruned = p0;
    }
    // ---- the lean step (poa_fsearch.hpp): one entry, or one more row of a greedy extension, per lane and step --------------------
    // Straight-line code over the per-row records (FlatGraph::RowRec) for the shapes that make up nearly all of a search: rows
    // with one or two successors, at most two bubbles ahead with exits a (nearly) fixed distance away.  Every load of a phase is
    // issued before the first use (own cell, relax targets, the words of the reached sets; then the words the nearest marks lie
    // in; then the scores at them), all decisions follow, then the writes — so nothing a lane reads can be its own write, and a
    // step costs the same few memory round trips whatever the lane met.  Whatever does not fit is flagged SPF_COMPLEX: the
    // entry (or the rest of its extension) takes the generic code of this file, sequentially.
    //
    // A greedy extension (dfa.rs:138-250) is not walked to its end by the lane that started it: after every matched base the
    // lane hands back the state of the walk (LeanWalk) and the next step of its group goes on from there — no lane of a wave
    // waits for another one's long extension.
    struct LeanWalk {
        uint32_t on = 0;          // an extension is under way: (r, j) is the row / offset whose successors are looked at next
        uint32_t r = 0, j = 0, k = 0, g = 0;   // k: index of the next successor; g: the score of the extension
        uint32_t iopen = 0;       // the insertion I[r][j+1] was already relaxed by a mismatch event of this row
        uint32_t sib_n = 0;       // rows left behind with their second successor still to look at (dfa.rs:86-134, the stack)
        uint32_t sib_r[2] = {0, 0}, sib_j[2] = {0, 0}, sib_io[2] = {0, 0};
        uint32_t dv = 0;          // rows this extension has visited so far (AstarResult::num_visited counts them when it ends)
    };
    struct LProbe {
        uint32_t on;              // 0: nothing to test
        uint32_t e, x, tmin, tmax, mde;
        uint64_t S, W0, W1;       // summary (which words are non-empty), the words of tmin and tmax
        uint32_t cw[5];           // Match scores of the exit row at tmin-1 .. tmax+1
        uint32_t prev, nxt;       // nearest marks below tmin / above tmax (EX_NIL: none)
        uint32_t wl, wu;          // words they lie in when not W0 / W1 (EX_NIL: no need)
        uint64_t Wl, Wu;
        uint32_t sp, sn;          // scores at prev / nxt
    };
    POA_HD FlatGraph::RowRec lrec(uint32_t row) const {
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr ((AS & EX_AS_REC_LDS) != 0) {
            typedef __attribute__((address_space(3))) const uint4 lds_cu4;
            const uint4 a = ((lds_cu4*)G.rec)[2 * row], b = ((lds_cu4*)G.rec)[2 * row + 1];
            FlatGraph::RowRec rr;
            __builtin_memcpy(&rr, &a, 16); __builtin_memcpy(reinterpret_cast<char*>(&rr) + 16, &b, 16);
            return rr;
        }
#endif
        return G.rec[row];
    }
    POA_HD uint32_t lh(uint32_t dmn, uint32_t dmx, uint32_t off, uint32_t st) const {   // h() over a record's distances
        if (C.heuristic == EX_H_DIJKSTRA) return 0;
        const uint32_t tmin = off + dmn, tmax = off + dmx;
        uint32_t gap;
        if (tmin > L) { gap = tmin - L; if (st != EX_ST_D) st = EX_ST_M; }
        else if (tmax < L) { gap = L - tmax; if (st != EX_ST_I) st = EX_ST_M; }
        else gap = 0;
        return gap_cost(st, gap);
    }
    // probe of bubble b (0 / 1) of the row of `rr` for a state at offset q
    POA_HD void lp_setup(LProbe& P, const FlatGraph::RowRec& rr, uint32_t b, uint32_t q, bool enabled) const {
        const uint32_t e = b ? rr.e1 : rr.e0;
        P.e = e; P.x = b ? rr.x1 : rr.x0; P.tmin = q + (b ? rr.d1min : rr.d0min); P.tmax = q + (b ? rr.d1max : rr.d0max); P.mde = b ? rr.mde1 : rr.mde0;
        P.on = (enabled && e != 0xFFFFu && P.tmax <= L) ? 1u : 0u;    // reached.rs:63-65: tmax > len -> can improve
        P.S = P.W0 = P.W1 = P.Wl = P.Wu = 0; P.prev = P.nxt = P.wl = P.wu = EX_NIL; P.sp = P.sn = EX_INF;
#pragma unroll
        for (int k = 0; k < 5; ++k) P.cw[k] = EX_INF;
    }
    POA_HD void lp_load1(LProbe& P) const {
        if (!P.on) return;
        P.S = W.rsum[(uint64_t)P.x * W.swpn];
        P.W0 = W.reached[(uint64_t)P.x * W.wpn + (P.tmin >> 6)];
        P.W1 = W.reached[(uint64_t)P.x * W.wpn + (P.tmax >> 6)];
        const uint32_t n = P.tmax - P.tmin + 3;   // cw[k]: offset tmin - 1 + k, up to tmax + 1 (< pitch: checked by the caller)
#pragma unroll
        for (uint32_t k = 0; k < 5; ++k) if (k < n && (k != 0 || P.tmin != 0)) P.cw[k] = W.T[cix(P.e, P.tmin + k - 1, EX_ST_M)];
    }
    // nearest marks from the words at hand; else the word to look in next
    POA_HD void lp_locate1(LProbe& P) const {
        if (!P.on || P.S == 0) return;
        const uint32_t w0 = P.tmin >> 6, w1 = P.tmax >> 6;
        const uint64_t below = (P.tmin & 63) ? (P.W0 & (~0ull >> (64 - (P.tmin & 63)))) : 0ull;
        if (below) P.prev = w0 * 64 + 63 - (uint32_t)clz64(below);
        else if (w0) { const uint64_t sb = P.S & (~0ull >> (64 - w0)); if (sb) P.wl = 63 - (uint32_t)clz64(sb); }
        const uint64_t above = (P.tmax & 63) != 63 ? (P.W1 & (~0ull << ((P.tmax & 63) + 1))) : 0ull;
        if (above) P.nxt = w1 * 64 + (uint32_t)ctz64(above);
        else if (w1 + 1 < W.wpn && w1 + 1 < 64) { const uint64_t sa = P.S >> (w1 + 1); if (sa) P.wu = w1 + 1 + (uint32_t)ctz64(sa); }
    }
    POA_HD void lp_load2(LProbe& P) const {
        if (P.wl != EX_NIL) P.Wl = W.reached[(uint64_t)P.x * W.wpn + P.wl];
        if (P.wu != EX_NIL) P.Wu = W.reached[(uint64_t)P.x * W.wpn + P.wu];
    }
    POA_HD void lp_locate2(LProbe& P) const {
        if (P.wl != EX_NIL && P.Wl) P.prev = P.wl * 64 + 63 - (uint32_t)clz64(P.Wl);
        if (P.wu != EX_NIL && P.Wu) P.nxt = P.wu * 64 + (uint32_t)ctz64(P.Wu);
    }
    POA_HD void lp_load3(LProbe& P) const {
        // scores at the two neighbours: next to the range they are loaded already
        // (the members are read before any branch: loads in the arms of a conditional get merged into one load through a
        // selected ADDRESS, and a probe whose address is taken lives in scratch memory, every member of it)
        const uint32_t c0 = P.cw[0], c2 = P.cw[2], c3 = P.cw[3], c4 = P.cw[4];
        if (P.prev != EX_NIL) { if (P.prev + 1 == P.tmin) P.sp = c0; else if (P.prev < W.pitch) P.sp = W.T[cix(P.e, P.prev, EX_ST_M)]; }
        if (P.nxt != EX_NIL) {
            const uint32_t wd = P.tmax - P.tmin;
            if (P.nxt == P.tmax + 1) P.sn = wd == 0 ? c2 : wd == 1 ? c3 : c4;
            else if (P.nxt < W.pitch) P.sn = W.T[cix(P.e, P.nxt, EX_ST_M)];
        }
    }
    // reached.rs:191-255 on the values at hand: left / right = (offset, score) of the nearest marks, EX_NIL: none
    POA_HD bool lp_improve_at(uint32_t t, uint32_t score, uint32_t left, uint32_t ls, uint32_t right, uint32_t rs, uint32_t mde) const {
        if (left != EX_NIL && right != EX_NIL) {
            const uint32_t fl = ls + gap_cost(EX_ST_M, t - left), fr = rs + gap_cost(EX_ST_M, right - t);
            return score < ((right - t > mde) ? fl : (fl < fr ? fl : fr));
        }
        if (right != EX_NIL) return (right - t > mde) ? true : score < rs + gap_cost(EX_ST_M, right - t);
        if (left != EX_NIL) return score < ls + gap_cost(EX_ST_M, t - left);
        return true;
    }
    // reached.rs:38-189 over the probe's data.  0: can improve; 2: pruned; 3: a mark without a score — the generic code reports
    // what the reference would.  Notes what the decision depended on.
    POA_HD uint32_t lp_decide(const LProbe& P, uint32_t g, uint32_t st) {
        if (!P.on) return 0;
        if (P.S == 0) { if (in_spec()) note_marks(P.x, 0, 0xFFFFFFFFu); return 0; }   // nothing reached at the exit yet (reached.rs:52-54)
        if (in_spec()) {
            note_marks(P.x, P.prev == EX_NIL ? 0u : P.prev, P.nxt);
            if (P.prev != EX_NIL && P.prev < W.pitch) note_cell(cix(P.e, P.prev, EX_ST_M));
            if (P.nxt != EX_NIL && P.nxt < W.pitch) note_cell(cix(P.e, P.nxt, EX_ST_M));
        }
        if ((P.prev != EX_NIL && P.sp == EX_INF) || (P.nxt != EX_NIL && P.sn == EX_INF) || g >= 0xFFFF0000u) return 3;
        uint32_t prev = P.prev, ps = P.sp;
        bool have_last = false; uint32_t last_offset = 0;
        const uint32_t n_in = P.tmax - P.tmin + 1;
#pragma unroll
        for (uint32_t k = 0; k < 3; ++k) {
            if (k >= n_in) break;
            const uint32_t next = P.tmin + k;
            const uint64_t wd = (next >> 6) == (P.tmin >> 6) ? P.W0 : P.W1;
            if (!((wd >> (next & 63)) & 1)) continue;
            const uint32_t ns = P.cw[k + 1];
            if (in_spec()) note_cell(cix(P.e, next, EX_ST_M));
            if (ns == EX_INF) return 3;
            const uint32_t offset1 = prev != EX_NIL ? (P.tmin > prev + 1 ? P.tmin : prev + 1) : P.tmin;
            if (st == EX_ST_D && ns + C.o > g) return 0;
            if (prev != EX_NIL && st == EX_ST_I && ps + C.o > g) return 0;
            if (lp_improve_at(offset1, g, prev, ps, next, ns, P.mde)) return 0;
            const uint32_t nm1 = next - 1u;
            const uint32_t mx = P.tmin > nm1 ? P.tmin : nm1;
            const uint32_t offset2 = P.tmax < mx ? P.tmax : mx;
            if (offset2 != offset1 && lp_improve_at(offset2, g, prev, ps, next, ns, P.mde)) return 0;
            prev = next; ps = ns;
            last_offset = offset2; have_last = true;
        }
        if (!have_last && lp_improve_at(P.tmin, g, prev, ps, P.nxt, P.sn, P.mde)) return 0;
        if ((!have_last || last_offset < P.tmax) && lp_improve_at(P.tmax, g, prev, ps, P.nxt, P.sn, P.mde)) return 0;
        if (prev != EX_NIL && st == EX_ST_I && ps + C.o > g) return 0;
        return 2;
    }
    // queue a state whose row's record is at hand (the heuristic reads its distances)
    POA_HD void lqueue(uint32_t row, uint32_t dmn, uint32_t dmx, uint32_t off, uint32_t st, uint32_t score) {
        num_queued += 1;
        spec_queue(score + lh(dmn, dmx, off, st), st, score, row, off);
    }

    // One step of a lane in log mode: the entry (g, v, j, st) if wk.on == 0, else one more row of the extension wk.  On return
    // sl.flags has SPF_COMPLEX (nothing was logged: the generic code takes over — the entry, or the extension from wk on) or not;
    // wk.on says whether the extension goes on.  Counters: num_queued / num_visited / num_pruned of the search are advanced.
    POA_HD void lean_unit(uint32_t g, uint32_t v, uint32_t j, uint32_t st, LeanWalk& wk) {
        PF_START();
        const bool cont = wk.on != 0;
        const uint32_t r = cont ? wk.r : v, q = cont ? wk.j : j, k0 = cont ? wk.k : 0u, gg = cont ? wk.g : g;
        const uint32_t est = cont ? (uint32_t)EX_ST_M : st;
        if (G.rec == nullptr || C.ends_free || W.swpn != 1 || q + 4 >= W.pitch || gg >= 0xFFFF0000u) { sp_flag(SPF_COMPLEX, 10); return; }
        const FlatGraph::RowRec rr = lrec(r);
        const uint32_t fl = rr.flags;
        if ((fl & FlatGraph::RR_END) || (est != EX_ST_I && !(fl & FlatGraph::RR_SUCC_OK)) || (!cont && C.prune && !(fl & FlatGraph::RR_PROBE_OK))) { sp_flag(SPF_COMPLEX, 10); return; }
        if (est == EX_ST_M && !cont && q == 0 && L != 0 && rr.sym == seq[0]) { sp_flag(SPF_COMPLEX, 10); return; }   // dfa.rs:146-167
        const uint32_t c0 = rr.c0, c1 = (fl & FlatGraph::RR_HAS_C1) ? (uint32_t)rr.c1 : EX_NIL;
        // ---- Match state / extension: which successors match the query symbol; the first one from k0 on is followed ----
        bool m0 = false, m1 = false;
        uint32_t cm = EX_NIL;   // the successor followed (probes of its row are loaded)
        FlatGraph::RowRec rm{};
        if (est == EX_ST_M && q < L) {
            const uint8_t qc = seq[q];
            m0 = k0 == 0 && rr.sym0 == qc; m1 = c1 != EX_NIL && rr.sym1 == qc;
            cm = m0 ? c0 : (m1 ? c1 : EX_NIL);
            if (cm != EX_NIL) {
                rm = lrec(cm);
                if (!(rm.flags & FlatGraph::RR_PROBE_OK)) { sp_flag(SPF_COMPLEX, 10); return; }
            }
        }
        // distances to the end of the rows that may be queued (heuristic)
        const FlatGraph::RowRec r0 = (est != EX_ST_I) ? (cm == c0 && cm != EX_NIL ? rm : lrec(c0)) : rr;
        const FlatGraph::RowRec r1 = (est != EX_ST_I && c1 != EX_NIL) ? (cm == c1 ? rm : lrec(c1)) : rr;
        // ---- probes: of the state itself (a fresh entry), of the successor the extension would step on ----
        LProbe P0, P1, Q0, Q1;
        lp_setup(P0, rr, 0, q, !cont && C.prune != 0); lp_setup(P1, rr, 1, q, !cont && C.prune != 0);
        lp_setup(Q0, rm, 0, q + 1, cm != EX_NIL); lp_setup(Q1, rm, 1, q + 1, cm != EX_NIL);   // (the test inside the extension is not subject to enable_pruning, dfa.rs:185)
        PF_TICK(0);
        // ---- phase 1: every cell the step may look at, the words of the probes ----
        const uint32_t i_own = cix(r, q, est);
        uint32_t own = cont ? gg : W.T[i_own];
        uint32_t iA = i_own, iB = i_own, iC = i_own, iD = i_own, iE = i_own;
        if (est == EX_ST_M) {
            if (q < L) { iA = cix(c0, q + 1, EX_ST_M); iB = cix(r, q + 1, EX_ST_I); iC = cix(c0, q, EX_ST_D); if (c1 != EX_NIL) { iD = cix(c1, q + 1, EX_ST_M); iE = cix(c1, q, EX_ST_D); } }
            else { iC = cix(c0, q, EX_ST_D); if (c1 != EX_NIL) iE = cix(c1, q, EX_ST_D); }
        } else if (est == EX_ST_I) { iA = cix(r, q, EX_ST_M); iB = cix(r, q < L ? q + 1 : q, EX_ST_I); }
        else { iA = cix(r, q, EX_ST_M); iC = cix(c0, q, EX_ST_D); if (c1 != EX_NIL) iE = cix(c1, q, EX_ST_D); }
        uint32_t vA = W.T[iA], vB = W.T[iB], vC = W.T[iC], vD = W.T[iD], vE = W.T[iE];
        lp_load1(P0); lp_load1(P1); lp_load1(Q0); lp_load1(Q1);
        PF_TICK(1);
        // ---- phase 2: the words the nearest marks lie in; phase 3: the scores there ----
        lp_locate1(P0); lp_locate1(P1); lp_locate1(Q0); lp_locate1(Q1);
        lp_load2(P0); lp_load2(P1); lp_load2(Q0); lp_load2(Q1);
        PF_TICK(2);
        lp_locate2(P0); lp_locate2(P1); lp_locate2(Q0); lp_locate2(Q1);
        lp_load3(P0); lp_load3(P1); lp_load3(Q0); lp_load3(Q1);
        PF_TICK(3);
        // ---- decisions ----
        if (in_spec()) {
            if (!cont) note_cell(i_own);
            if (est == EX_ST_M) {
                if (q < L) { if (k0 == 0) { note_cell(iA); note_cell(iC); } note_cell(iB); if (c1 != EX_NIL) { note_cell(iD); note_cell(iE); } }
                else { if (k0 == 0) note_cell(iC); if (c1 != EX_NIL) note_cell(iE); }
            } else if (est == EX_ST_I) { note_cell(iA); if (q < L) note_cell(iB); }
            else { note_cell(iA); note_cell(iC); if (c1 != EX_NIL) note_cell(iE); }
        }
        PF_TICK(4);
        if (!cont) {
            if (gg > own) return;                                   // stale (astar.rs:146)
            uint32_t pr = lp_decide(P0, gg, est);
            if (pr == 0) pr = lp_decide(P1, gg, est);
            if (pr == 3) { sp_flag(SPF_COMPLEX, 10); return; }
            if (sl.flags & SPF_COMPLEX) return;
            if (pr == 2) { num_pruned += 1; return; }               // astar.rs:155-158
            mark_reached(r, q, est);                                // astar.rs:160 (Match states at bubble exits)
            num_visited += 1;
        }
        if (est == EX_ST_I) {   // expand_all, gap_affine.rs:307-322
            if (gg < vA) { wr(r, q, EX_ST_M, gg); lqueue(r, rr.dmin, rr.dmax, q, EX_ST_M, gg); }
            const uint32_t ns = gg + C.e;
            if (q < L && ns < vB) { wr(r, q + 1, EX_ST_I, ns); lqueue(r, rr.dmin, rr.dmax, q + 1, EX_ST_I, ns); }
            return;
        }
        if (est == EX_ST_D) {   // expand_all, gap_affine.rs:323-340
            if (gg < vA) { wr(r, q, EX_ST_M, gg); lqueue(r, rr.dmin, rr.dmax, q, EX_ST_M, gg); }
            const uint32_t ns = gg + C.e;
            if (ns < vC) { wr(c0, q, EX_ST_D, ns); lqueue(c0, r0.dmin, r0.dmax, q, EX_ST_D, ns); }
            if (c1 != EX_NIL && ns < vE) { wr(c1, q, EX_ST_D, ns); lqueue(c1, r1.dmin, r1.dmax, q, EX_ST_D, ns); }
            return;
        }
        PF_TICK(5);
        // ---- Match state: the successors of (r, q) from k0 on, in order (dfa.rs:210-250) ----
        const uint32_t nm = gg + C.x, ng = gg + C.o + C.e;
        uint32_t io = cont ? wk.iopen : 0u;
        bool descended = false;
        uint32_t kk = k0;
        const uint32_t nchild = c1 != EX_NIL ? 2u : 1u;
#pragma unroll
        for (uint32_t i = 0; i < 2; ++i) {
            if (i < k0 || i >= nchild || descended || kk != i) continue;
            const uint32_t c = i ? c1 : c0;
            const FlatGraph::RowRec& rc = i ? r1 : r0;
            const uint32_t vM = i ? vD : vA, vDd = i ? vE : vC;
            if (q >= L) {   // QueryEnd -> expand_query_end (gap_affine.rs:369-391)
                if (ng < vDd) { wr(c, q, EX_ST_D, ng); lqueue(c, rc.dmin, rc.dmax, q, EX_ST_D, ng); }
                kk = i + 1;
                continue;
            }
            const bool mt = i ? m1 : m0;
            if (!mt) {      // Mismatch -> expand_mismatch (gap_affine.rs:393-430)
                if (nm < vM) { wr(c, q + 1, EX_ST_M, nm); lqueue(c, rc.dmin, rc.dmax, q + 1, EX_ST_M, nm); }
                if (!io) { if (ng < vB) { wr(r, q + 1, EX_ST_I, ng); lqueue(r, rr.dmin, rr.dmax, q + 1, EX_ST_I, ng); } io = 1; }
                if (ng < vDd) { wr(c, q, EX_ST_D, ng); lqueue(c, rc.dmin, rc.dmax, q, EX_ST_D, ng); }
                kk = i + 1;
                continue;
            }
            if (c != cm) break;   // a second successor to follow: its probes are not at hand — the next step starts here (kk == i)
            kk = i + 1;
            if (!(gg < vM)) continue;                               // already there with this score or better (dfa.rs:242)
            wr(c, q + 1, EX_ST_M, gg);
            uint32_t pr = lp_decide(Q0, gg, EX_ST_M);
            if (pr == 0) pr = lp_decide(Q1, gg, EX_ST_M);
            if (pr == 3) { sp_flag(SPF_COMPLEX, 10); return; }
            if (sl.flags & SPF_COMPLEX) return;
            if (pr == 2) { num_pruned_dfa += 1; continue; }        // scored, not extended (dfa.rs:185-188)
            mark_reached(c, q + 1, EX_ST_M);
            num_visited += 1; wk.dv += 1;                           // (dfa.num_visited, added when the extension ends: astar.rs:205)
            descended = true;
        }
        if (sl.flags & SPF_COMPLEX) return;
        if (descended) {
            // the row just stepped on is looked at next; the row left behind stays on the stack if it has a successor left
            if (kk < nchild) {
                if (wk.sib_n >= 2) { sp_flag(SPF_COMPLEX, 7); return; }
#pragma unroll
                for (uint32_t s2 = 0; s2 < 2; ++s2) if (s2 == wk.sib_n) { wk.sib_r[s2] = r; wk.sib_j[s2] = q; wk.sib_io[s2] = io; }
                wk.sib_n += 1;
            }
            wk.on = 1; wk.r = cm; wk.j = q + 1; wk.k = 0; wk.g = gg; wk.iopen = 0;
            return;
        }
        if (kk < nchild) { wk.on = 1; wk.r = r; wk.j = q; wk.k = kk; wk.g = gg; wk.iopen = io; return; }   // (second successor to follow next)
        // successors exhausted: back to the last row left behind, if any (its second successor is looked at next)
        if (wk.sib_n) {
            wk.sib_n -= 1;
            uint32_t br = 0, bj = 0, bio = 0;
#pragma unroll
            for (uint32_t s2 = 0; s2 < 2; ++s2) if (s2 == wk.sib_n) { br = wk.sib_r[s2]; bj = wk.sib_j[s2]; bio = wk.sib_io[s2]; }
            wk.on = 1; wk.r = br; wk.j = bj; wk.k = 1; wk.g = gg; wk.iopen = bio;
            return;
        }
        wk.on = 0;
    }
    // The generic code finishes an extension the lean step gave up on (direct mode): the walk's state becomes the stack of
    // dfa_extend.  true: the search ends here.
    POA_HD bool lean_walk_generic(LeanWalk& wk, ExactResult& R, uint32_t& end_score) {
        dfa_score = wk.g; dfa_visited = 0;
        sp = 0;
        // the rows left behind, oldest first, each with its second successor next
        for (uint32_t s2 = 0; s2 < wk.sib_n; ++s2) {
            const ExStackEntry en{wk.sib_r[s2], wk.sib_j[s2], gld(&G.succ_off[wk.sib_r[s2]]) + 1};
            if (sp != 0) { if (sp >= W.stack_cap) { err = EX_POOL_FULL; return false; } W.stack[sp - 1] = dfa_top; }
            dfa_top = en; sp += 1;
        }
        {
            const ExStackEntry en{wk.r, wk.j, gld(&G.succ_off[wk.r]) + wk.k};
            if (sp != 0) { if (sp >= W.stack_cap) { err = EX_POOL_FULL; return false; } W.stack[sp - 1] = dfa_top; }
            dfa_top = en; sp += 1;
        }
        // (the insertion of the row on top may have been relaxed already: relaxing it again cannot succeed — same value)
        const uint32_t dv0 = wk.dv;
        wk = LeanWalk();
        if (dfa_events(dfa_score, R, end_score)) { num_visited -= dv0; return true; }   // skipped by `break 'main` (astar.rs:172,:205)
        num_visited += dfa_visited;
        return false;
    }

    // ---- flat schedule: ONE entry per lane and step, everything it pushes is logged (poa_fsearch.hpp) ----------------------------
    POA_HD void spec_entry(uint32_t g, uint32_t v, uint32_t j, uint32_t st, ExactResult& R, uint32_t& end_score, bool use_fast = true) {
        sl.n_w = sl.n_m = sl.n_p = sl.n_pd = 0; sl.flags = 0; sl.n_rc = sl.n_rm = 0; sl.n_ent = 0; sl.wmask = sl.mmask = 0;
        sl.flat = 1; sl.min_child = 0xFFFFFFFFu;
        rc_x = rs_x = EX_NIL;
        const uint32_t q0 = num_queued, v0 = num_visited, p0 = num_pruned;
        spec = true;
        FastItem F{0, 0, 0, 0, 0, 0, 0, 0};
        uint32_t sk = use_fast ? inspect_fast(g, v, j, st, F) : 3u;
        if (sk == 3 && !err && !(sl.flags & SPF_COMPLEX)) sk = inspect_skip(g, v, j, st);
        bool found = false;
        if (!err && !(sl.flags & SPF_COMPLEX)) {
            if (sk == 2) num_pruned += 1;
            if (sk == 0) found = F.kind ? process_fast(g, v, j, st, F, R, end_score) : process_popped(g, v, j, st, R, end_score);
        }
        spec = false; sl.flat = 0;
        sl.n_ent = 1;
        if (err) { sl.flags |= SPF_COMPLEX; err = EX_OK; }   // the sequential code meets the same condition and reports it
        else if (found) sl.flags |= SPF_FOUND;
        sl.dq = num_queued - q0; sl.dv = num_visited - v0; sl.dp = num_pruned - p0;
        num_queued = q0; num_visited = v0; num_pruned = p0;
    }

    // the lean step of one lane in log mode: wk is advanced only if the step stands (no SPF_COMPLEX)
    POA_HD void spec_lean(uint32_t g, uint32_t v, uint32_t j, uint32_t st, LeanWalk& wk) {
        sl.n_w = sl.n_m = sl.n_p = sl.n_pd = 0; sl.flags = 0; sl.n_rc = sl.n_rm = 0; sl.n_ent = 1; sl.wmask = sl.mmask = 0;
        sl.flat = 1; sl.min_child = 0xFFFFFFFFu;
        const uint32_t q0 = num_queued, v0 = num_visited, p0 = num_pruned;
        spec = true;
        LeanWalk w2 = wk;
        lean_unit(g, v, j, st, w2);
        spec = false; sl.flat = 0;
        if (err) { sl.flags |= SPF_COMPLEX; err = EX_OK; }
        if (sl.flags & SPF_COMPLEX) { sl.n_w = sl.n_m = sl.n_p = 0; sl.min_child = 0xFFFFFFFFu; sl.dq = sl.dv = sl.dp = 0; }
        else { wk = w2; sl.dq = num_queued - q0; sl.dv = num_visited - v0; sl.dp = num_pruned - p0; }
        num_queued = q0; num_visited = v0; num_pruned = p0;
    }

    // did this lane read a cell or a mark that `o` (an earlier lane of the step) logged a write to?
    POA_HD bool spec_reads_what(const SpecLane& o) const {
        for (uint32_t a = 0; a < sl.n_rc; ++a)
            for (uint32_t b = 0; b < o.n_w; ++b)
                if (sl.rc[a * sl.stride] == sp_hash16(o.w_idx[b * o.stride])) return true;
        for (uint32_t a = 0; a < sl.n_rm; ++a)
            for (uint32_t b = 0; b < o.n_m; ++b)
                if (sl.rm_x[a * sl.stride] == o.m_x[b * o.stride] && sl.rm_lo[a * sl.stride] <= o.m_off[b * o.stride] && o.m_off[b * o.stride] <= sl.rm_hi[a * sl.stride]) return true;
        return false;
    }
    // commit of one lane's log: cell writes, marks, counters (queue pushes are the caller's: they need the lanes' order)
    POA_HD void spec_commit(const SpecLane& l) {
        for (uint32_t k = 0; k < l.n_w; ++k) W.T[sld(&l.w_idx[k * l.stride])] = sld(&l.w_val[k * l.stride]);
        for (uint32_t k = 0; k < l.n_m; ++k) mark_word(sld(&l.m_x[k * l.stride]), sld(&l.m_off[k * l.stride]));
        num_queued += l.dq; num_visited += l.dv; num_pruned += l.dp;
    }

#if !defined(__HIP_DEVICE_COMPILE__)
    uint32_t par_steps = 0, par_seq = 0, par_entries = 0, par_cut_complex = 0, par_cut_conflict = 0, par_cut_leftover = 0;   // statistics
    uint32_t par_hist[65] = {}, par_hist_conf[65] = {}, par_hist_nb[65] = {};
    // The schedule of the wave kernel poa_psearch.hpp, one lane after the other (host build: its executable specification;
    // tests/test_exact_replay.py diffs it against the oracle).  A step takes the top `lanes` entries of the current stack;
    // every one is processed in log mode against the table as the step found it; the step commits the lanes before the first
    // one that (a) read what an earlier lane wrote, or (b) needs the sequential code; a lane that ends the search or leaves
    // entries pending is the last one committed.  A first lane that needs the sequential code is run by it, alone.
    ExactResult run_parallel(uint32_t lanes, uint32_t rmax, bool use_fast = true) {
        ExactResult R{EX_OK, EX_INF, 0, 0, 0, G.end_row, L};
        push_initial_states();
        uint32_t end_score = EX_INF;
        bool found = false;
        if (lanes > 63) lanes = 63;
        struct LaneBuf { uint32_t w_idx[SP_KW], w_val[SP_KW], m_x[SP_KM], m_off[SP_KM], rm_x[SP_KRM], rm_lo[SP_KRM], rm_hi[SP_KRM]; uint16_t rc[SP_KRC]; ExU4 p[SP_KP]; ExStackEntry ds[SP_KDS]; };
        LaneBuf* buf = new LaneBuf[64];
        SpecLane* ls = new SpecLane[64];
        ExactResult* rs = new ExactResult[64];
        uint32_t* es = new uint32_t[64];
        while (!found && !err) {
            uint32_t st; BqDesc d;
            if (!bq_current(st, d)) { err = EX_PANIC; break; }
            par_steps += 1;
            const uint32_t f = layer_min;
            const uint32_t nb = d.n_top < lanes ? d.n_top : lanes;
            const ExU4* ch = W.bq_chunks + (uint64_t)BQ_CHUNK * d.top;
            const uint32_t prev_chunk = ch[0].x;
            uint32_t n_commit = 0;       // lanes committed
            bool seq0 = false;
            for (uint32_t i = 0; i < nb; ++i) {
                const ExU4 e = ch[d.n_top - i];
                sl = SpecLane();
                sl.w_idx = buf[i].w_idx; sl.w_val = buf[i].w_val; sl.m_x = buf[i].m_x; sl.m_off = buf[i].m_off; sl.p = buf[i].p; sl.stride = 1;
                sl.rc = buf[i].rc; sl.rm_x = buf[i].rm_x; sl.rm_lo = buf[i].rm_lo; sl.rm_hi = buf[i].rm_hi; sl.dstack = buf[i].ds;
                rs[i] = R; es[i] = end_score;
                spec_group(e.x, e.y, e.z, st, f, rmax, rs[i], es[i], use_fast);
                ls[i] = sl;
                bool cut_before = (sl.flags & SPF_COMPLEX) != 0;
                for (uint32_t a = 0; a < i && !cut_before; ++a) {
                    cut_before = spec_reads_what(ls[a]);
#if defined(POA_EXACT_DIAG)
                    if (cut_before && getenv("EXH_CONFLICTS")) {
                        const ExU4 ea = ch[d.n_top - a];
                        bool cellhit = false;
                        for (uint32_t x_ = 0; x_ < sl.n_rc; ++x_) for (uint32_t y_ = 0; y_ < ls[a].n_w; ++y_) if (sl.rc[x_] == sp_hash16(ls[a].w_idx[y_])) cellhit = true;
                        fprintf(stderr, "conflict st %u lanes %u<-%u  reader (row %u off %u g %u) writer (row %u off %u g %u) %s  writer: %u cells %u marks %u entries\n", st, i, a, e.y, e.z, e.x, ea.y, ea.z, ea.x,
                                cellhit ? "CELL" : "MARK", ls[a].n_w, ls[a].n_m, ls[a].n_ent);
                    }
#endif
                }
                if (cut_before) { seq0 = i == 0; if (i) { if (sl.flags & SPF_COMPLEX) par_cut_complex += 1; else { par_cut_conflict += 1; par_hist_conf[i] += 1; } } break; }
                n_commit = i + 1; par_entries += sl.n_ent;
                if (sl.flags & SPF_LEFTOVER) par_cut_leftover += 1;
                if (sl.flags & (SPF_FOUND | SPF_LEFTOVER)) break;
            }
            if (seq0) {
                // the sequential code for the top entry
                par_seq += 1;
                const ExU4 e = ch[d.n_top];
                bq_drop(st, d, 1, prev_chunk);
                const uint32_t sk = inspect_skip(e.x, e.y, e.z, st);
                if (sk == 2) num_pruned += 1;
                if (sk == 0 && !err) found = process_popped(e.x, e.y, e.z, st, R, end_score);
                continue;
            }
            par_hist[n_commit] += 1; par_hist_nb[nb] += 1;
            bq_drop(st, d, n_commit, prev_chunk);
            for (uint32_t i = 0; i < n_commit; ++i) {
                spec_commit(ls[i]);
                // what the lane left pending goes on the queue in its push order (only the last lane can have any)
                for (uint32_t k = 0; k < ls[i].n_pd; ++k) bq_push(ls[i].pd_key[k] >> 2, ls[i].pd_key[k] & 3u, ls[i].pd_score[k], ls[i].pd_row[k], ls[i].pd_off[k]);
                for (uint32_t k = 0; k < ls[i].n_p; ++k) { const ExU4 q = ls[i].p[k]; bq_push(q.w >> 2, q.w & 3u, q.x, q.y, q.z); }
                if (ls[i].flags & SPF_FOUND) { found = true; R.end_row = rs[i].end_row; R.end_off = rs[i].end_off; end_score = es[i]; }
            }
        }
        delete[] buf; delete[] ls; delete[] rs; delete[] es;
        R.status = err ? err : (found ? EX_OK : EX_PANIC);
        R.score = end_score;
        R.num_queued = num_queued; R.num_visited = num_visited; R.num_pruned = num_pruned;
        return R;
    }

    // The schedule of the wave kernel poa_fsearch.hpp, one lane after the other (host build: its executable specification).
    // A step offers the next `lanes` entries in the queue's pop order — the stacks of the current bucket one after the other,
    // Match, Deletion, Insertion (gap_affine.rs:954-966), each from its top — one entry per lane, processed in log mode against
    // the table as the step found it.  Committed: the lanes before the first one that read what an earlier lane wrote or needs
    // the sequential code, and before the first one whose (priority, state) is not below that of something an earlier
    // committed lane pushed (that push would be popped first).  Every push goes to the queue.
    ExactResult run_flat(uint32_t lanes, bool use_fast = true, bool lean = false) {
        ExactResult R{EX_OK, EX_INF, 0, 0, 0, G.end_row, L};
        push_initial_states();
        uint32_t end_score = EX_INF;
        bool found = false;
        if (lanes > 63) lanes = 63;
        struct LaneBuf { uint32_t w_idx[SP_KW], w_val[SP_KW], m_x[SP_KM], m_off[SP_KM], rm_x[SP_KRM], rm_lo[SP_KRM], rm_hi[SP_KRM]; uint16_t rc[SP_KRC]; ExU4 p[SP_KP]; ExStackEntry ds[SP_KDS]; };
        LaneBuf* buf = new LaneBuf[64];
        SpecLane* ls = new SpecLane[64];
        ExactResult* rs = new ExactResult[64];
        uint32_t* es = new uint32_t[64];
        ExU4* ent = new ExU4[64];
        uint32_t* est = new uint32_t[64];
        LeanWalk* wks = new LeanWalk[64];
        LeanWalk walk;   // the extension under way (lean steps): the first lane of the next step goes on with it
        while (!found && !err) {
            uint32_t st0 = 0; BqDesc d0{0, 0};
            const bool have_q = bq_current(st0, d0);
            if (!have_q && !walk.on) { err = EX_PANIC; break; }
            par_steps += 1;
            const uint32_t f = layer_min;
            // the entries offered: the extension under way first; then the stacks st0.. of bucket f — a stack is left for the
            // next one only when it is taken whole
            uint32_t n_off = 0, taken[3] = {0, 0, 0}, prevc[3] = {EX_NIL, EX_NIL, EX_NIL}, beg[3] = {0, 0, 0};
            BqDesc ds3[3] = {BqDesc{0, 0}, BqDesc{0, 0}, BqDesc{0, 0}};
            const uint32_t w_on = walk.on ? 1u : 0u;
            if (w_on) { ent[0] = ExU4{0, 0, 0, 0}; est[0] = 0; n_off = 1; }
            for (uint32_t s2 = st0; have_q && s2 < 3 && n_off < lanes; ++s2) {
                const uint32_t dw = W.bq_desc[3 * (f & (W.bq_win - 1)) + s2];
                beg[s2] = n_off;
                if (dw == BQ_EMPTY || (dw & 63u) == 0) continue;
                ds3[s2] = BqDesc{dw >> 6, dw & 63u};
                const ExU4* ch = W.bq_chunks + (uint64_t)BQ_CHUNK * ds3[s2].top;
                prevc[s2] = ch[0].x;
                const uint32_t take = ds3[s2].n_top < lanes - n_off ? ds3[s2].n_top : lanes - n_off;
                for (uint32_t i = 0; i < take; ++i) { ent[n_off] = ch[ds3[s2].n_top - i]; est[n_off] = s2; n_off += 1; }
                taken[s2] = take;
                if (take < ds3[s2].n_top || prevc[s2] != EX_NIL) { for (uint32_t s3 = s2 + 1; s3 < 3; ++s3) beg[s3] = n_off; break; }   // more of this stack is left
                for (uint32_t s3 = s2 + 1; s3 < 3; ++s3) beg[s3] = n_off;
            }
            uint32_t n_commit = 0, limit = n_off;
            bool seq0 = false;
            for (uint32_t i = 0; i < n_off && i < limit; ++i) {
                sl = SpecLane();
                sl.w_idx = buf[i].w_idx; sl.w_val = buf[i].w_val; sl.m_x = buf[i].m_x; sl.m_off = buf[i].m_off; sl.p = buf[i].p; sl.stride = 1;
                sl.rc = buf[i].rc; sl.rm_x = buf[i].rm_x; sl.rm_lo = buf[i].rm_lo; sl.rm_hi = buf[i].rm_hi; sl.dstack = buf[i].ds;
                rs[i] = R; es[i] = end_score;
                wks[i] = (i == 0 && w_on) ? walk : LeanWalk();
                if (lean) spec_lean(ent[i].x, ent[i].y, ent[i].z, est[i], wks[i]);
                else spec_entry(ent[i].x, ent[i].y, ent[i].z, est[i], rs[i], es[i], use_fast);
                ls[i] = sl;
                bool cut_before = (sl.flags & SPF_COMPLEX) != 0;
                for (uint32_t a = 0; a < i && !cut_before; ++a) cut_before = spec_reads_what(ls[a]);
                if (cut_before) { seq0 = i == 0; if (i) { if (sl.flags & SPF_COMPLEX) par_cut_complex += 1; else par_cut_conflict += 1; } break; }
                n_commit = i + 1; par_entries += 1;
                if (sl.flags & SPF_FOUND) break;
                if (wks[i].on) break;   // an extension under way: nothing is popped before it is over
                // what this lane pushed is popped before every later entry whose (priority, state) is not below it
                if (sl.min_child != 0xFFFFFFFFu) {
                    uint32_t lim = i + 1;
                    while (lim < n_off && (f << 2 | est[lim]) < sl.min_child) lim += 1;
                    if (lim < limit) { limit = lim; }
                }
            }
            if (n_commit < n_off && n_commit == limit) par_cut_leftover += 1;
            if (seq0) {
                // the generic code, sequentially: the rest of the extension, or the first entry
                par_seq += 1;
                if (w_on) found = lean_walk_generic(walk, R, end_score);
                else {
                    bq_drop(st0, d0, 1, prevc[st0]);
                    const uint32_t sk = inspect_skip(ent[0].x, ent[0].y, ent[0].z, st0);
                    if (sk == 2) num_pruned += 1;
                    if (sk == 0 && !err) found = process_popped(ent[0].x, ent[0].y, ent[0].z, st0, R, end_score);
                }
                continue;
            }
            par_hist[n_commit] += 1; par_hist_nb[n_off] += 1;
            // the committed entries leave their stacks (the extension under way is not one of them)
            {
                uint32_t left = n_commit - w_on;
                for (uint32_t s2 = st0; have_q && s2 < 3 && left; ++s2) {
                    const uint32_t k = taken[s2] < left ? taken[s2] : left;
                    if (k) bq_drop_at(f, s2, ds3[s2], k, prevc[s2]);
                    left -= k;
                }
            }
            walk = wks[n_commit - 1];   // (on only if the last committed lane left its extension under way)
            for (uint32_t i = 0; i < n_commit; ++i) {
                spec_commit(ls[i]);
                for (uint32_t k = 0; k < ls[i].n_p; ++k) { const ExU4 q = ls[i].p[k]; bq_push(q.w >> 2, q.w & 3u, q.x, q.y, q.z); }
                if (ls[i].flags & SPF_FOUND) { found = true; R.end_row = rs[i].end_row; R.end_off = rs[i].end_off; end_score = es[i]; }
            }
        }
        delete[] buf; delete[] ls; delete[] rs; delete[] es; delete[] ent; delete[] est; delete[] wks;
        R.status = err ? err : (found ? EX_OK : EX_PANIC);
        R.score = end_score;
        R.num_queued = num_queued; R.num_visited = num_visited; R.num_pruned = num_pruned;
        return R;
    }
#endif
};
using ExactSearch = ExactSearchT<0>;

}  // namespace poa_amd
