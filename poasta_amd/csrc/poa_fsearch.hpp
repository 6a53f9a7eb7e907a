// Replay of the reference's search (exact / hybrid mode): several queries per wavefront, the next entries of each query's queue
// expanded AT ONCE, one per lane.
//
// The reference pops one state at a time (astar.rs:141-216) and its tie-breaks are decided by that order, so the replay has to
// produce exactly the writes and pushes of that loop.  But the entries that follow one another in the queue's pop order (the
// stacks of a bucket: Match, Deletion, Insertion, each from its top, gap_affine.rs:929-1013) rarely have anything to do with each
// other: on the 1 kbp reads of the benchmark a bucket holds some 400 states spread over a band of the table.  A group of GS lanes
// owns a query, and a step of the group
//
//   1. reads the next entries in pop order, one per lane (the stacks of the current bucket one after the other);
//   2. lets every lane process ITS entry in the log mode of the search object (poa_exact.hpp, SpecLane / spec_entry): the
//      ordinary code — stale test, pruning, greedy extension, relaxations — reading the table as the step found it and writing
//      cells, reached marks and pushes to a per-lane log;
//   3. finds the first lane that read a cell or a reached mark an EARLIER lane of the step logged a write to (two tables in LDS
//      keyed by independent hashes of the 64-cell block / the word of the reached set hold the lowest writing lane; a chance
//      match only ends the step early), and the first lane that something an earlier lane pushed would be popped before;
//   4. commits the logs of the lanes before those, in lane order — the writes and pushes of the sequential loop — pops their
//      entries, and leaves the rest for the next step.  A lane that needs what the log mode does not do takes the sequential
//      code once it is the first lane of a step.
//
// No lane ever works through a chain of entries alone (an entry's children are ordinary entries of a later step), and the
// groups of a wave step through one instruction stream: that is what keeps the lanes of an instruction busy.
// ExactSearch::run_flat (poa_exact.hpp) is this schedule one lane after the other; compiled for the host it is diffed against
// the oracle (tests/test_exact_replay.py), this kernel against both on the GPU.
#pragma once
#include <hip/hip_runtime.h>

#include "poa_exact_kernel.hpp"

namespace poa_amd {

constexpr uint32_t PS_TAB = 512;       // slots of each of the two conflict tables (lowest writing lane per key), per wave
constexpr uint32_t PS_TAB_BITS = 9;

struct FSearchParams {
    ExactParams E;            // graph, queries, planes, reached sets, costs, status / end cell
    ExU4* chunks;             // per slot: chunk_cap chunks of BQ_CHUNK slots
    uint32_t chunk_cap;
    uint32_t win;             // descriptor ring: priorities per wave (power of two)
    uint32_t* ring_global;    // null: the rings live in LDS; else [slots * 3 * win] in global memory
    uint32_t graph_lds;       // bytes of the staged graph arrays (exact_lds_bytes), 0: read them from global memory
    uint32_t rec_lds;         // bytes of the staged row records (lean step), 0: none staged
    uint32_t lean;            // 1: the lean step (needs E.G.rec), 0: the generic code in log mode
    uint32_t waves_per_block;
    uint32_t group;           // lanes per query: 64 (one query per wave), 16, 8 or 4
    uint32_t max_lanes;       // entries per step (<= group)
    uint32_t* scratch;        // per slot: ps_scratch_words() words — the lanes' push logs and extension stacks
    uint32_t* work_counter;   // persistent scheduling (see poa_wsearch.hpp); null: query = block / wave index
    const uint32_t* order;
    uint32_t* counters;       // optional [4 * total]: num_queued, num_visited, num_pruned, steps
    unsigned long long* prof; // optional [8 * total]: cycles per phase
};

// per wave, in words of global memory: pushes (4 words each), extension stack (3 words)
__host__ __device__ inline uint32_t ps_scratch_words() { return 64u * (4 * SP_KP + 3 * SP_KDS); }
// per wave, in bytes of LDS: two conflict tables | cell-write log (index, value) | marks (exit, offset) | mark ranges read (exit, lo, hi) |
// read cells (16-bit hashes) | counts | sequence -> (lane, slot) map of the step's pushes | the lanes found in conflict
constexpr uint32_t PS_O_W = 2u * 4u * PS_TAB;
constexpr uint32_t PS_O_M = PS_O_W + 4u * 64u * 2 * SP_KW;
constexpr uint32_t PS_O_RM = PS_O_M + 4u * 64u * 2 * SP_KM;
constexpr uint32_t PS_O_RC = PS_O_RM + 4u * 64u * 3 * SP_KRM;
constexpr uint32_t PS_O_CNT = PS_O_RC + 2u * 64u * SP_KRC;
constexpr uint32_t PS_O_MAP = PS_O_CNT + 4u * 64u;
constexpr uint32_t PS_O_CONF = PS_O_MAP + 2u * 64u * SP_KP;
__host__ __device__ inline uint32_t ps_lds_bytes() { return PS_O_CONF + 16u; }

__device__ __forceinline__ uint32_t ps_bcast(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane); }
__device__ __forceinline__ uint32_t ps_wave_sum(uint32_t v) {
    for (int o = 32; o; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, 64);
    return v;
}
// two independent table slots per key (second table behind the first): a cell by the 16-bit hash of its index, a word of a reached set
__device__ __forceinline__ uint32_t ps_key1(uint32_t k) { return (k * 0x9E3779B1u) >> (32 - PS_TAB_BITS); }
__device__ __forceinline__ uint32_t ps_key2(uint32_t k) { return PS_TAB + ((k * 0x85EBCA6Bu + 0x165667B1u) >> (32 - PS_TAB_BITS)); }
__device__ __forceinline__ uint32_t ps_mark_id(uint32_t x, uint32_t word) { return 0x80000000u | (x * 0x3D4D51CBu + word * 0xC2B2AE35u); }
constexpr uint32_t PS_ALL_WORDS = 0xFFFFFFu;   // "some word of this exit's set": key of a range too wide to name its words

// The generic code for one entry (or the rest of a greedy extension) — direct mode, one lane, sequential.  It is a function of
// its own, working on a copy of the search's state in memory, so that the step kernel keeps its registers for the step: a lane
// comes here for a row in a thousand.
struct FsFallbackIO {
    ExactGraph G; ExactWork W; const uint8_t* seq; uint32_t L; ExactCosts C;
    uint32_t layer_min, bq_live, bq_hi, bq_chunk_top, bq_free, err;     // queue state, in / out
    uint32_t walk_on; ExactSearchT<EX_AS_NO_SPEC>::LeanWalk wk; ExU4 e; uint32_t st;  // what: the extension wk, or the entry e of state st
    uint32_t found, end_score, end_row, end_off, dq, dv, dp;              // out: end of the search; counters to add
};
__device__ __attribute__((noinline)) void fs_fallback(FsFallbackIO* io) {
    ExactSearchT<EX_AS_NO_SPEC> S(io->G, io->W, io->seq, io->L, io->C);
    S.layer_min = io->layer_min; S.bq_live = io->bq_live; S.bq_hi = io->bq_hi; S.bq_chunk_top = io->bq_chunk_top; S.bq_free = io->bq_free;
    S.err = io->err; S.bq_wr = true;
    ExactResult R{EX_OK, EX_INF, 0, 0, 0, io->G.end_row, io->L};
    uint32_t end_score = EX_INF;
    bool found = false;
    if (io->walk_on) found = S.lean_walk_generic(io->wk, R, end_score);
    else {
        const uint32_t sk = S.inspect_skip(io->e.x, io->e.y, io->e.z, io->st);
        if (sk == 2) S.num_pruned += 1;
        if (sk == 0 && !S.err) found = S.process_popped(io->e.x, io->e.y, io->e.z, io->st, R, end_score);
    }
    io->layer_min = S.layer_min; io->bq_live = S.bq_live; io->bq_hi = S.bq_hi; io->bq_chunk_top = S.bq_chunk_top; io->bq_free = S.bq_free; io->err = S.err;
    io->found = found ? 1u : 0u; io->end_score = end_score; io->end_row = R.end_row; io->end_off = R.end_off;
    io->dq = S.num_queued; io->dv = S.num_visited; io->dp = S.num_pruned;
}

template <int AS, int GS, bool LEAN>
__device__ __forceinline__ void fs_search(const FSearchParams& P, const ExactGraph& G, uint8_t* wbase, uint32_t lane, uint32_t wave);

template <bool LEAN>
__device__ __forceinline__ void fs_kernel_body(const FSearchParams& P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const ExactParams& E = P.E;
    ExactGraph G = E.G;
    const uint32_t nthreads = blockDim.x;
    uint32_t at = 0;
    auto stage = [&](const void* src, uint64_t bytes) {
        uint8_t* dst = lds + at;
        const uint32_t words = (uint32_t)((bytes + 3) / 4);
        const uint32_t* s32 = static_cast<const uint32_t*>(src);
        for (uint32_t i = threadIdx.x; i < words; i += nthreads) reinterpret_cast<uint32_t*>(dst)[i] = s32[i];
        at += (uint32_t)((bytes + 15) & ~15ull);
        return dst;
    };
    if (P.graph_lds) {
        const uint32_t n = E.G.n_rows;
        G.sym = stage(E.G.sym, n);
        G.succ_off = reinterpret_cast<const uint32_t*>(stage(E.G.succ_off, 4ull * (n + 1)));
        G.nbm_off = reinterpret_cast<const uint32_t*>(stage(E.G.nbm_off, 4ull * (n + 1)));
        G.succ = reinterpret_cast<const uint32_t*>(stage(E.G.succ, 4ull * E.n_succ));
        G.dist_min = reinterpret_cast<const uint32_t*>(stage(E.G.dist_min, 4ull * n));
        G.dist_max = reinterpret_cast<const uint32_t*>(stage(E.G.dist_max, 4ull * n));
        G.exit_idx = reinterpret_cast<const uint32_t*>(stage(E.G.exit_idx, 4ull * n));
        G.nbm = reinterpret_cast<const FlatGraph::NodeBubble*>(stage(E.G.nbm, sizeof(FlatGraph::NodeBubble) * (uint64_t)E.n_nbm));
    }
    if (P.rec_lds) G.rec = reinterpret_cast<const FlatGraph::RowRec*>(stage(E.G.rec, sizeof(FlatGraph::RowRec) * (uint64_t)E.G.n_rows));
    const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // per wave: [descriptor rings of its queries (unless global)] [conflict tables | logs | read sets | push map]
    const uint32_t qpw = 64u / P.group;
    const uint32_t ring_bytes = P.ring_global ? 0u : ((qpw * 3u * P.win * 4u + 15u) & ~15u);
    uint8_t* wbase = lds + P.graph_lds + P.rec_lds + (uint64_t)wave * (ring_bytes + ps_lds_bytes());
    __syncthreads();
    // what is staged in LDS is read through typed LDS accesses (a pointer that may be global or LDS compiles to FLAT loads)
    const bool lds_ring = !P.ring_global;
#define FS_CALL(GS_) do { \
        if constexpr (LEAN) { \
            if (P.rec_lds && lds_ring) fs_search<EX_AS_REC_LDS | EX_AS_RING_LDS | EX_AS_READSET_LDS, GS_, true>(P, G, wbase, lane, wave); \
            else fs_search<EX_AS_READSET_LDS, GS_, true>(P, G, wbase, lane, wave); \
        } else { \
            if (P.graph_lds && lds_ring) fs_search<EX_AS_GRAPH_LDS | EX_AS_RING_LDS | EX_AS_READSET_LDS, GS_, false>(P, G, wbase, lane, wave); \
            else fs_search<EX_AS_READSET_LDS, GS_, false>(P, G, wbase, lane, wave); \
        } } while (0)
    // (group sizes built: 8 lanes per query for the lean step, 16 for the generic code in log mode — the engine asks for these)
    if constexpr (LEAN) FS_CALL(8); else FS_CALL(16);
#undef FS_CALL
}
__global__ __launch_bounds__(512) void poa_fsearch_kernel(FSearchParams P) { fs_kernel_body<false>(P); }        // generic code in log mode
__global__ __launch_bounds__(512) void poa_fsearch_lean_kernel(FSearchParams P) { fs_kernel_body<true>(P); }    // the lean step

// GS lanes per query, 64 / GS queries per wave, all stepping through one instruction stream: what is uniform over a wave with one
// query (queue state, the cut of a step) is uniform over a group here and travels by shuffles inside the group.
template <int AS, int GS, bool LEAN>
__device__ __forceinline__ void fs_search(const FSearchParams& P, const ExactGraph& G, uint8_t* wbase, uint32_t lane, uint32_t wave) {
    const ExactParams& E = P.E;
    constexpr uint32_t QPW = 64u / GS;
    const uint32_t grp = lane / GS, gl = lane % GS, gbase = grp * GS;
    constexpr unsigned long long GM = GS == 64 ? ~0ull : ((1ull << (GS & 63)) - 1);
    const uint32_t slot = (blockIdx.x * P.waves_per_block + wave) * QPW + grp;   // this group's workspace (reached sets, chunks, ring)
    const uint32_t wslot = blockIdx.x * P.waves_per_block + wave;               // the wave's (push logs)
    typedef __attribute__((address_space(3))) unsigned long long lds_u64;
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    typedef __attribute__((address_space(3))) uint16_t lds_u16;
    const uint32_t ring_bytes = P.ring_global ? 0u : ((QPW * 3u * P.win * 4u + 15u) & ~15u);
    uint32_t* ring = P.ring_global ? P.ring_global + (uint64_t)slot * 3 * P.win : reinterpret_cast<uint32_t*>(wbase) + (uint64_t)grp * 3 * P.win;
    uint8_t* wlds = wbase + ring_bytes;
    lds_u32* tab = (lds_u32*)wlds;   // [PS_TAB | PS_TAB]
    uint32_t* l_w = reinterpret_cast<uint32_t*>(wlds + PS_O_W);      // [w_idx: SP_KW x 64][w_val: SP_KW x 64]
    uint32_t* l_m = reinterpret_cast<uint32_t*>(wlds + PS_O_M);      // [m_x][m_off]
    uint32_t* l_rm = reinterpret_cast<uint32_t*>(wlds + PS_O_RM);    // [rm_x][rm_lo][rm_hi]
    uint16_t* l_rc = reinterpret_cast<uint16_t*>(wlds + PS_O_RC);
    lds_u16* l_map = (lds_u16*)(wlds + PS_O_MAP) + gbase * SP_KP;    // this group's part: GS * SP_KP entries
    uint32_t* sc = P.scratch + (uint64_t)wslot * ps_scratch_words();
    const unsigned long long lanebit = 1ull << lane;
    const uint32_t gsalt = grp * 0x632BE5ABu;   // keeps the groups apart in the shared conflict tables
    auto gsh = [&](uint32_t v, uint32_t src_gl) { return (uint32_t)__shfl((int)v, (int)(gbase + src_gl), 64); };
    auto gballot = [&](bool p) { return (__ballot(p) >> gbase) & GM; };

    // workspace of the group: constant over its queries but for the planes
    ExactWork W;
    W.T = E.planes;
    W.n_rows = E.G.n_rows;
    W.pitch = 64;
    W.reached = E.reached + (uint64_t)slot * E.G.n_exit * E.wpn;
    W.rsum = E.rsum + (uint64_t)slot * E.G.n_exit * E.swpn;
    W.wpn = E.wpn; W.swpn = E.swpn;
    W.head = nullptr; W.n_prio = 0xFFFFFFFFu;
    W.pool = nullptr; W.pool_cap = 0;
    W.stack = E.stack + (uint64_t)slot * E.stack_cap;
    W.stack_cap = E.stack_cap;
    W.bq_desc = ring; W.bq_win = P.win;
    W.bq_chunks = P.chunks + (uint64_t)slot * P.chunk_cap * BQ_CHUNK;
    W.bq_chunk_cap = P.chunk_cap;
    ExactSearchT<AS> S(G, W, E.qseq, 0, E.C);
    // the lane's logs (element k at base[k * 64])
    S.sl.stride = 64;
    S.sl.w_idx = l_w + lane; S.sl.w_val = l_w + 64u * SP_KW + lane;
    S.sl.m_x = l_m + lane; S.sl.m_off = l_m + 64u * SP_KM + lane;
    S.sl.rm_x = l_rm + lane; S.sl.rm_lo = l_rm + 64u * SP_KRM + lane; S.sl.rm_hi = l_rm + 64u * 2 * SP_KRM + lane;
    S.sl.rc = l_rc + lane;
    S.sl.p = reinterpret_cast<ExU4*>(sc) + lane;
    S.sl.dstack = reinterpret_cast<ExStackEntry*>(sc + 64u * 4 * SP_KP) + lane;
    ExactResult R{EX_OK, EX_INF, 0, 0, 0, G.end_row, 0};
    uint32_t end_score = EX_INF, found = 0, steps = 0, qi = 0;
    typename ExactSearchT<AS>::LeanWalk wk;   // the greedy extension under way (lean step), group-uniform
    bool have = false, more = slot < E.n_queries, first = true;   // (the workspace has a slot per query of the chunk at least)

    // group-uniform state lives identically in every lane of the group; what one lane changes alone is handed round afterwards
    auto adopt = [&](uint32_t from) {
        S.err = gsh(S.err, from);
        S.layer_min = gsh(S.layer_min, from);
        S.bq_live = gsh(S.bq_live, from);
        S.bq_hi = gsh(S.bq_hi, from);
        S.bq_chunk_top = gsh(S.bq_chunk_top, from);
        S.bq_free = gsh(S.bq_free, from);
        found = gsh(found, from);
        end_score = gsh(end_score, from);
        R.end_row = gsh(R.end_row, from);
        R.end_off = gsh(R.end_off, from);
    };
    auto adopt_walk = [&](const typename ExactSearchT<AS>::LeanWalk& w, uint32_t from) {
        wk.on = gsh(w.on, from); wk.r = gsh(w.r, from); wk.j = gsh(w.j, from); wk.k = gsh(w.k, from); wk.g = gsh(w.g, from);
        wk.iopen = gsh(w.iopen, from); wk.sib_n = gsh(w.sib_n, from); wk.dv = gsh(w.dv, from);
        wk.sib_r[0] = gsh(w.sib_r[0], from); wk.sib_r[1] = gsh(w.sib_r[1], from);
        wk.sib_j[0] = gsh(w.sib_j[0], from); wk.sib_j[1] = gsh(w.sib_j[1], from);
        wk.sib_io[0] = gsh(w.sib_io[0], from); wk.sib_io[1] = gsh(w.sib_io[1], from);
    };

    unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool prof = P.prof != nullptr;
#define PS_TICK(k) do { if (prof) { const unsigned long long now_ = clock64(); pc[k] += now_ - t_last; t_last = now_; } } while (0)
    unsigned long long t_last = prof ? clock64() : 0;
    for (;;) {
        // ---- groups without a search take the next query (persistent scheduling: longest expected search first) ----
        if (__any(!have && more)) {
            uint32_t t = EX_NIL;
            if (!have && more && gl == 0) {
                if (P.work_counter) t = atomicAdd(P.work_counter, 1u);
                else if (first && slot < E.n_queries) t = slot;
            }
            const bool want = !have && more;
            t = gsh(t, 0);
            first = false;
            if (want) {
                if (t >= E.n_queries) more = false;
                else {
                    const uint32_t pos = P.order ? P.order[t] : t;
                    qi = E.first_query + pos;
                    if (!(E.hybrid && E.dense_flags[qi] == 0)) {
                        // the group's workspace: reached sets back to empty, ring back to empty (the host clears nothing)
                        uint64_t* z = W.reached;
                        for (uint64_t i = gl; i < (uint64_t)E.G.n_exit * E.wpn; i += GS) z[i] = 0;
                        uint64_t* zs = W.rsum;
                        for (uint64_t i = gl; i < (uint64_t)E.G.n_exit * E.swpn; i += GS) zs[i] = 0;
                        for (uint32_t i = gl; i < 3 * P.win; i += GS) ring[i] = BQ_EMPTY;
                        const uint64_t qbeg = E.qoff[qi];
                        W.T = E.planes + E.plane_off[qi];
                        W.pitch = E.pitch[qi];
                        S.begin_query(E.qseq + qbeg, (uint32_t)(E.qoff[qi + 1] - qbeg));
                        R = ExactResult{EX_OK, EX_INF, 0, 0, 0, G.end_row, S.L};
                        end_score = EX_INF; found = 0; steps = 0;
                        wk = typename ExactSearchT<AS>::LeanWalk();
                        have = true;
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        S.bq_wr = true;
                        if (gl == 0) S.push_initial_states();
                    }
                }
            }
            adopt(0);   // (of the lanes that did not take a query: their own lane 0's values — nothing changes)
            S.bq_wr = gl == 0;
            if (__any(!have && more)) continue;   // (a group skipped a query the dense pass certified: it asks again)
        }
        if (!__any(have)) break;

        // ---- one step of every group that has a search ----
        const bool go = have && !found && !S.err;
        const uint32_t w_on = (go && wk.on) ? 1u : 0u;        // an extension is under way: the group's first lane goes on with it
        uint32_t st0 = 0; BqDesc d0{0, 0};
        bool have_q = false;
        if (go) { have_q = S.bq_current(st0, d0); if (!have_q && !w_on) S.err = EX_PANIC; }   // "Could not align sequence!" (astar.rs:142-144)
        const bool run = go && !S.err;
        const uint32_t f = S.layer_min;
        const uint32_t cap = P.max_lanes < (uint32_t)GS ? P.max_lanes : (uint32_t)GS;
        // the stacks of bucket f from st0 on: how many entries each offers (its top chunk), in pop order
        uint32_t dsw[3] = {BQ_EMPTY, BQ_EMPTY, BQ_EMPTY}, cnt3[3] = {0, 0, 0}, beg3[3] = {w_on, w_on, w_on};   // beg3: first lane of a state's entries
        if (run && have_q) {
            uint32_t* b3 = &W.bq_desc[3 * (f & (W.bq_win - 1))];
#pragma unroll
            for (uint32_t s2 = 0; s2 < 3; ++s2) { const uint32_t dw = S.rld(b3 + s2); dsw[s2] = dw; cnt3[s2] = (s2 >= st0 && dw != BQ_EMPTY) ? (dw & 63u) : 0u; }
            beg3[1] = w_on + cnt3[0]; beg3[2] = w_on + cnt3[0] + cnt3[1];
        }
        // lane gl: which stack, which entry from its top
        uint32_t lst = 0, lidx = gl - w_on;
        if (gl >= beg3[1]) { lst = 1; lidx = gl - beg3[1]; }
        if (gl >= beg3[2]) { lst = 2; lidx = gl - beg3[2]; }
        const uint32_t ltop = lst == 0 ? (dsw[0] >> 6) : lst == 1 ? (dsw[1] >> 6) : (dsw[2] >> 6);
        const uint32_t lcnt = lst == 0 ? cnt3[0] : lst == 1 ? cnt3[1] : cnt3[2];
        bool act = run && gl >= w_on && gl < cap && lidx < lcnt;
        ExU4 e{0, 0, 0, 0};
        uint32_t lprev = EX_NIL;
        if (act) {
            const ExU4* ch = W.bq_chunks + (uint64_t)BQ_CHUNK * ltop;
            e = ch[lcnt - lidx];
            lprev = ch[0].x;   // {previous chunk} of that stack
        }
        // a stack is left for the next one only when it was taken whole: the lanes of the stacks behind one that has more to
        // offer (entries beyond the cap, or an earlier chunk) do not take part
        const uint32_t pv0 = gsh(lprev, beg3[0] < GS ? beg3[0] : 0), pv1 = gsh(lprev, beg3[1] < GS ? beg3[1] : 0), pv2 = gsh(lprev, beg3[2] < GS ? beg3[2] : 0);
        {
            const bool more0 = cnt3[0] != 0 && (beg3[1] > cap || pv0 != EX_NIL);
            const bool more1 = cnt3[1] != 0 && (beg3[2] > cap || pv1 != EX_NIL);
            if (lst >= 1 && more0) act = false;
            if (lst >= 2 && (more0 || more1)) act = false;
        }
        if (run && w_on && gl == 0) act = true;   // (the extension under way)
        const uint32_t n_off = (uint32_t)__builtin_popcountll(gballot(act));   // (the active lanes are the first n_off of the group)
        // the conflict tables of this step (4 KB: 8 bytes per lane, eight times)
        {
            lds_u64* t8 = (lds_u64*)wlds;
#pragma unroll
            for (uint32_t k = 0; k < 2 * PS_TAB / 128; ++k) t8[lane + 64 * k] = ~0ull;
        }
        if (run) steps += 1;
        PS_TICK(0);
        // ---- every lane: its entry in log mode ----
        ExactResult Rl = R; uint32_t esl = end_score;
        S.sl.flags = 0; S.sl.n_w = S.sl.n_m = S.sl.n_p = S.sl.n_pd = S.sl.n_rc = S.sl.n_rm = 0; S.sl.n_ent = 0; S.sl.min_child = 0xFFFFFFFFu;
        typename ExactSearchT<AS>::LeanWalk wl = wk;
        if (!(w_on && gl == 0)) wl = typename ExactSearchT<AS>::LeanWalk();
        if (act) { if constexpr (LEAN) S.spec_lean(e.x, e.y, e.z, lst, wl); else S.spec_entry(e.x, e.y, e.z, lst, Rl, esl); }
        PS_TICK(1);
        // ---- who read what an earlier lane wrote ----
        // Writers enter their lane (lowest wins) under two independent hashes of every cell they logged / every word of a
        // reached set they marked (and under "some word of that set", for readers of wide ranges); a reader that finds a lower
        // lane under BOTH hashes of something it depended on is taken to conflict.  A chance match ends the step early, nothing
        // else: the lanes before the first such reader commit.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        for (uint32_t k = 0; k < S.sl.n_w; ++k) {
            const uint32_t h16 = ExactSearchT<AS>::sp_hash16(((lds_u32*)l_w)[k * 64 + lane]) ^ gsalt;
            __hip_atomic_fetch_min(&tab[ps_key1(h16)], lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_min(&tab[ps_key2(h16)], lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        for (uint32_t k = 0; k < S.sl.n_m; ++k) {
            const uint32_t x = ((lds_u32*)l_m)[k * 64 + lane], off = ((lds_u32*)l_m)[64 * SP_KM + k * 64 + lane];
            const uint32_t ka = ps_mark_id(x, off >> 6) ^ gsalt, kb = ps_mark_id(x, PS_ALL_WORDS) ^ gsalt;
            __hip_atomic_fetch_min(&tab[ps_key1(ka)], lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_min(&tab[ps_key2(ka)], lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_min(&tab[ps_key1(kb)], lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_min(&tab[ps_key2(kb)], lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        bool clash = false;
        for (uint32_t k0 = 0; k0 < S.sl.n_rc; k0 += 4) {   // (four at a time: eight table reads in flight)
            uint32_t hh[4], ta[4], tb[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) hh[u] = ((lds_u16*)l_rc)[(k0 + u < SP_KRC ? k0 + u : SP_KRC - 1) * 64 + lane] ^ gsalt;
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) { ta[u] = tab[ps_key1(hh[u])]; tb[u] = tab[ps_key2(hh[u])]; }
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) clash = clash || (k0 + u < S.sl.n_rc && ta[u] < lane && tb[u] < lane);
        }
        for (uint32_t k = 0; k < S.sl.n_rm; ++k) {
            const uint32_t x = ((lds_u32*)l_rm)[k * 64 + lane], lo = ((lds_u32*)l_rm)[64 * SP_KRM + k * 64 + lane], hi = ((lds_u32*)l_rm)[128 * SP_KRM + k * 64 + lane];
            const uint32_t w0 = lo >> 6, w1 = hi >> 6;
            if (w1 - w0 < 3) {
                for (uint32_t w = w0; w <= w1; ++w) { const uint32_t ka = ps_mark_id(x, w) ^ gsalt; clash = clash || (tab[ps_key1(ka)] < lane && tab[ps_key2(ka)] < lane); }
            } else {
                const uint32_t kb = ps_mark_id(x, PS_ALL_WORDS) ^ gsalt;
                clash = clash || (tab[ps_key1(kb)] < lane && tab[ps_key2(kb)] < lane);
            }
        }
        PS_TICK(2);
        // ---- how many lanes of the group commit ----
        // (a) before the first lane that clashes or needs the sequential code; through the first one that ends the search;
        const unsigned long long m_cx = gballot(act && (clash || (S.sl.flags & SPF_COMPLEX)));
        const unsigned long long m_fd = gballot(act && (S.sl.flags & SPF_FOUND));
        uint32_t n_commit = m_cx ? (uint32_t)__builtin_ctzll(m_cx) : n_off;
        if (m_fd) { const uint32_t a = (uint32_t)__builtin_ctzll(m_fd) + 1; n_commit = a < n_commit ? a : n_commit; }
        {   // through the first one that leaves an extension under way: nothing is popped before that is over
            const unsigned long long m_wk = gballot(act && wl.on != 0);
            if (m_wk) { const uint32_t a = (uint32_t)__builtin_ctzll(m_wk) + 1; n_commit = a < n_commit ? a : n_commit; }
        }
        // (b) what a committed lane pushed is popped before every later entry whose (priority, state) is not below it: the first
        //     such lane ends the step.  The lanes' keys are (f, state) with the states in order, so that lane is where the
        //     child's state begins among the lanes (or the very next lane).
        {
            const uint32_t mc = S.sl.min_child;
            uint32_t lim = n_off;
            if (act && mc != 0xFFFFFFFFu) {
                if (mc <= (f << 2 | 0u)) lim = gl + 1;
                else if (mc <= (f << 2 | 1u)) lim = beg3[1] > gl + 1 ? beg3[1] : gl + 1;
                else if (mc <= (f << 2 | 2u)) lim = beg3[2] > gl + 1 ? beg3[2] : gl + 1;
            }
#pragma unroll
            for (uint32_t i = 0; i + 1 < (uint32_t)GS; ++i) {   // lane i's limit counts once lane i itself commits
                const uint32_t li = gsh(lim, i);
                if (i < n_commit && li < n_commit) n_commit = li;
            }
        }
        const bool seq = run && n_commit == 0;   // the group's first lane needs the sequential code (direct mode): its entry alone
        if (__any(seq)) {
            if (seq) {
                if (!w_on) S.bq_drop(st0, d0, 1, pv0);
                if (gl == 0) {
                    FsFallbackIO io;
                    io.G = P.E.G; io.W = W; io.seq = S.seq; io.L = S.L; io.C = S.C;
                    io.layer_min = S.layer_min; io.bq_live = S.bq_live; io.bq_hi = S.bq_hi; io.bq_chunk_top = S.bq_chunk_top; io.bq_free = S.bq_free; io.err = S.err;
                    io.walk_on = w_on;
                    io.wk.on = wk.on; io.wk.r = wk.r; io.wk.j = wk.j; io.wk.k = wk.k; io.wk.g = wk.g; io.wk.iopen = wk.iopen; io.wk.sib_n = wk.sib_n; io.wk.dv = wk.dv;
                    io.wk.sib_r[0] = wk.sib_r[0]; io.wk.sib_r[1] = wk.sib_r[1]; io.wk.sib_j[0] = wk.sib_j[0]; io.wk.sib_j[1] = wk.sib_j[1]; io.wk.sib_io[0] = wk.sib_io[0]; io.wk.sib_io[1] = wk.sib_io[1];
                    io.e = e; io.st = st0;
                    fs_fallback(&io);
                    S.layer_min = io.layer_min; S.bq_live = io.bq_live; S.bq_hi = io.bq_hi; S.bq_chunk_top = io.bq_chunk_top; S.bq_free = io.bq_free; S.err = io.err;
                    S.num_queued += io.dq; S.num_visited += io.dv; S.num_pruned += io.dp;
                    if (io.found) { found = 1; end_score = io.end_score; R.end_row = io.end_row; R.end_off = io.end_off; }
                    wk.on = 0; wk.sib_n = 0; wk.dv = 0;
                }
            }
            S.bq_wr = gl == 0;
            adopt(0);
            adopt_walk(wk, 0);
            PS_TICK(4);
            if (prof && seq) pc[6] += 1;
        }
        const bool com = run && n_commit != 0;
        // ---- commit: table, marks, counters of the lanes before the cut; their entries leave their stacks ----
        if (com && gl < n_commit) S.spec_commit(S.sl);
        if (com) {
#pragma unroll
            for (uint32_t s2 = 0; s2 < 3; ++s2) {
                const uint32_t hi = beg3[s2] + cnt3[s2] < n_commit ? beg3[s2] + cnt3[s2] : n_commit;   // (lanes [beg3, beg3 + cnt3) hold this stack's entries)
                const uint32_t k = hi > beg3[s2] ? hi - beg3[s2] : 0u;
                if (k) S.bq_drop_at(f, s2, BqDesc{dsw[s2] >> 6, dsw[s2] & 63u}, k, s2 == 0 ? pv0 : s2 == 1 ? pv1 : pv2);
            }
        }
        const uint32_t last = com ? n_commit - 1 : 0u;
        {   // the extension the last committed lane left under way (or none) is the group's from here on
            typename ExactSearchT<AS>::LeanWalk keep = wk;
            adopt_walk(wl, last);
            if (!com) wk = keep;
        }
        PS_TICK(3);
        // ---- the logged pushes, in the order the sequential loop makes them: lane after lane, each lane's in its own order ----
        {
            const uint32_t np_l = (com && gl < n_commit) ? S.sl.n_p : 0u;
            uint32_t incl = np_l;
#pragma unroll
            for (int dd = 1; dd < GS; dd <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, dd, 64); if ((int)gl >= dd) incl += t; }
            const uint32_t seq0 = incl - np_l, N = gsh(incl, GS - 1);
            for (uint32_t k = 0; k < np_l; ++k) l_map[seq0 + k] = (uint16_t)(lane | k << 8);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const ExU4* plog = reinterpret_cast<const ExU4*>(sc);
            for (uint32_t base = 0; __any(base < N); base += GS) {
                const uint32_t i = base + gl;
                bool pend = i < N;
                const uint32_t mk = pend ? (uint32_t)l_map[i] : 0u;
                const ExU4 q = plog[(mk >> 8) * 64 + (mk & 63u)];   // (the idle lanes read the first slot of lane 0)
                unsigned long long pm = gballot(pend);
                while (__any(pm != 0 && !S.err)) {
                    const bool on = pm != 0 && !S.err;
                    // all entries of the group for one stack, in sequence order
                    const uint32_t key = gsh(q.w, on ? (uint32_t)__builtin_ctzll(pm) : 0u);
                    const bool mine = on && pend && q.w == key;
                    const unsigned long long mm = gballot(mine);
                    if (on) {
                    const uint32_t cnt = (uint32_t)__builtin_popcountll(mm);
                    const uint32_t rank = (uint32_t)__builtin_popcountll(mm & ((1ull << gl) - 1));
                    const uint32_t prio = key >> 2, pst = key & 3u;
                    // queue.rs:31-54: the layers the queue spans (here: the ring's window)
                    bool ok = true;
                    if (S.bq_live == 0) { S.layer_min = prio; S.bq_hi = prio; }
                    else {
                        const uint32_t lo = prio < S.layer_min ? prio : S.layer_min, hi = prio > S.bq_hi ? prio : S.bq_hi;
                        if (hi - lo >= W.bq_win) { S.err = EX_POOL_FULL; ok = false; }
                        else { S.layer_min = lo; S.bq_hi = hi; }
                    }
                    if (ok) {
                        uint32_t* dp = &W.bq_desc[3 * (prio & (W.bq_win - 1)) + pst];
                        const uint32_t dw = S.rld(dp);
                        uint32_t top = dw == BQ_EMPTY ? EX_NIL : dw >> 6, n = dw == BQ_EMPTY ? 0u : dw & 63u;
                        uint32_t done = 0;
                        while (done < cnt) {
                            if (top == EX_NIL || n == BQ_CHUNK - 1) {
                                const uint32_t c = S.bq_alloc();
                                if (c == EX_NIL) break;
                                if (gl == 0) W.bq_chunks[(uint64_t)BQ_CHUNK * c] = ExU4{top, 0, 0, 0};
                                top = c; n = 0;
                            }
                            const uint32_t room = BQ_CHUNK - 1 - n, take = cnt - done < room ? cnt - done : room;
                            if (mine && rank >= done && rank < done + take) W.bq_chunks[(uint64_t)BQ_CHUNK * top + n + 1 + (rank - done)] = ExU4{q.x, q.y, q.z, 0};
                            n += take; done += take;
                        }
                        if (!S.err) {
                            if (gl == 0) S.rst(dp, top << 6 | n);
                            S.bq_live += cnt;
                        }
                    }
                    pend = pend && !mine;
                    pm &= ~mm;
                    }
                }
            }
        }
        {
            const uint32_t fl = gsh(S.sl.flags, last);
            const uint32_t es2 = gsh(esl, last), er = gsh(Rl.end_row, last), eo = gsh(Rl.end_off, last);
            if (com && (fl & SPF_FOUND)) { found = 1; end_score = es2; R.end_row = er; R.end_off = eo; }
        }
        PS_TICK(5);
        if (prof && com) pc[7] += n_commit;

        // ---- searches that ended: results out, the group is free for the next query ----
        if (__any(have && (found || S.err))) {
            const bool fin = have && (found || S.err);
            uint32_t nq = S.num_queued, nv = S.num_visited, np = S.num_pruned;
#pragma unroll
            for (int o = GS / 2; o; o >>= 1) {
                nq += (uint32_t)__shfl_xor((int)nq, o, 64); nv += (uint32_t)__shfl_xor((int)nv, o, 64); np += (uint32_t)__shfl_xor((int)np, o, 64);
            }
            if (fin && gl == 0) {
                E.status[qi] = S.err ? S.err : EX_OK;
                E.end_cell[2 * qi] = R.end_row;
                E.end_cell[2 * qi + 1] = R.end_off;
                if (P.counters) {
                    if (prof) for (int k = 0; k < 8; ++k) P.prof[8 * (uint64_t)qi + k] = pc[k];
#if defined(POA_PS_PROF_FINE)
                    if (prof) for (int k = 0; k < 6; ++k) P.prof[8 * (uint64_t)qi + k] = S.pf[k];
#endif
                    P.counters[4 * qi] = nq; P.counters[4 * qi + 1] = nv; P.counters[4 * qi + 2] = np; P.counters[4 * qi + 3] = steps;
                }
            }
            if (fin) {
                have = false; S.num_queued = S.num_visited = S.num_pruned = 0;
                for (int k = 0; k < 8; ++k) pc[k] = 0;
#if defined(POA_PS_PROF_FINE)
                for (int k = 0; k < 8; ++k) S.pf[k] = 0;
#endif
            }
        }
    }
#undef PS_TICK
}

}  // namespace poa_amd
