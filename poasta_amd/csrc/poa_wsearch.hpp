// Wave-per-query replay of the reference's search (exact / hybrid mode).
//
// The reference's tie-breaks are decided by the order in which its best-first search pops states (bucket queue, three LIFO
// stacks per bucket, pruning against the bubble exits reached so far), so an exact replay has to keep that order.  What it
// does not have to keep is one memory round trip after another per pop.  Here a wavefront owns a query and a step is:
//
//   1. the wave finds the stack the next pop comes from (descriptor ring in LDS) and reads its top entries, one per lane
//      (the entries of a stack are contiguous in 64-slot chunks: one coalesced load);
//   2. every lane tests ITS entry as if it were popped now: stale (a lower score is on the table, astar.rs:146) or pruned
//      (reached.rs:38-189) — reads only, all lanes' loads in flight together;
//   3. the leading run of stale / pruned entries changes nothing, so the whole run is popped at once together with the
//      first entry that survives; that one is expanded by its own lane (greedy extension, relaxations, pushes) while the
//      table and the queue are exactly what the reference's would be at that pop.
//
// The search of one query keeps one or two lanes busy, and the replay is bound by instruction issue, so a
// wave carries SEVERAL queries: groups of GS lanes (64 / 32 / 16 / 8), one query per group, all groups stepping through the
// same instruction stream — the test of the top entries is per lane anyway, and the expansion of each group's surviving
// entry runs in that group's lane beside the other groups'.  What is wave-uniform with one query per wave (queue state,
// the popped run) is group-uniform here and travels by shuffles inside the group.
//
// ExactSearch::run_buckets (poa_exact.hpp) is the same schedule one lane at a time; compiled for the host it is diffed
// against the oracle (tests/test_exact_replay.py), and this kernel is diffed against both on the GPU.
#pragma once
// 1: everything outside the test of the entries runs on every lane with the same values (scalar branches instead of one lane's
// divergent code and a broadcast of the state afterwards).  Built, bit-identical on the GPU tests, measured 11 % SLOWER on
// configs[1] (0.775 s against 0.698 s): sixty-four lanes' worth of addresses per load and store, and 35 more spilled registers.
#if !defined(POA_WS_UNIFORM)
#define POA_WS_UNIFORM 0
#endif
#include <hip/hip_runtime.h>

#include "poa_exact_kernel.hpp"

namespace poa_amd {

struct WSearchParams {
    ExactParams E;            // graph, queries, planes, reached sets, DFA stacks, costs, status / end cell
    ExU4* chunks;             // per slot: chunk_cap chunks of BQ_CHUNK slots
    uint32_t chunk_cap;
    uint32_t win;             // descriptor ring: priorities per wave (power of two)
    uint32_t* ring_global;    // null: the rings live in LDS; else [slots * 3 * win] in global memory
    uint32_t graph_lds;       // bytes of the staged graph arrays (exact_lds_bytes), 0: read them from global memory
    uint32_t waves_per_block;
    uint32_t rec_lds;         // bytes of dynamic LDS behind graph_lds that hold the per-row records (0: not staged)
    uint32_t max_lanes;       // entries tested per step (<= 63)
    uint32_t adapt_lanes;     // 0: always max_lanes; else the step after an expansion tests this many, and a run that used up its lanes four times as many
    uint32_t group;           // lanes per query: 64 (one query per wave), 32, 16 or 8
    // persistent scheduling (group == 64): the launch holds as many waves as are resident at once; each takes the next
    // query of `order` (longest expected search first) from `work_counter` until none is left, so that no wave waits for
    // the slowest member of its block and the longest searches do not start last.  Null: query = block / wave index.
    uint32_t* work_counter;
    const uint32_t* order;    // [n_queries] positions within the chunk, or null = identity
    uint32_t* counters;       // optional [4 * total]: num_queued, num_visited, num_pruned, steps (null: not kept)
    unsigned long long* prof; // optional [8 * total] cycles: queue+entries, parallel test, drop, fast expand, generic test+expand; counts
};

__device__ __forceinline__ uint32_t ws_bcast(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane); }
__device__ __forceinline__ uint32_t ws_wave_sum(uint32_t v) {
    for (int o = 32; o; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, 64);
    return v;
}

template <int AS>
__device__ __forceinline__ void ws_search_query(const WSearchParams& P, const ExactGraph& G, uint32_t* ring, uint32_t lane, uint32_t wave);
template <int AS, int GS>
__device__ __forceinline__ void ws_search_groups(const WSearchParams& P, const ExactGraph& G, uint32_t* ring0, uint32_t lane, uint32_t wave);


// graph staging + ring set-up shared by the two kernels; GROUPS: several queries per wave (its own kernel, so that the
// default one-query-per-wave kernel keeps its registers: the union of both needs scratch)
// MODE 0: one query per wave, graph arrays + records + descriptor rings in LDS (the default); 1: several queries per wave;
// 2: one query per wave with whatever does not fit read through generic pointers.  Three kernels, so that each keeps its
// registers (the union of two instantiations of the search in one kernel spills for both).
template <int MODE>
__device__ __forceinline__ void ws_kernel_body(const WSearchParams& P) {
    constexpr bool GROUPS = MODE == 1;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const ExactParams& E = P.E;
    ExactGraph G = E.G;
    const uint32_t nthreads = blockDim.x;
    if (P.graph_lds) {
        uint32_t at = 0;
        auto stage = [&](const void* src, uint64_t bytes) {
            uint8_t* dst = lds + at;
            const uint32_t words = (uint32_t)((bytes + 3) / 4);
            const uint32_t* s32 = static_cast<const uint32_t*>(src);
            for (uint32_t i = threadIdx.x; i < words; i += nthreads) reinterpret_cast<uint32_t*>(dst)[i] = s32[i];
            at += (uint32_t)((bytes + 15) & ~15ull);
            return dst;
        };
        const uint32_t n = E.G.n_rows;
        G.sym = stage(E.G.sym, n);
        G.succ_off = reinterpret_cast<const uint32_t*>(stage(E.G.succ_off, 4ull * (n + 1)));
        G.nbm_off = reinterpret_cast<const uint32_t*>(stage(E.G.nbm_off, 4ull * (n + 1)));
        G.succ = reinterpret_cast<const uint32_t*>(stage(E.G.succ, 4ull * E.n_succ));
        G.dist_min = reinterpret_cast<const uint32_t*>(stage(E.G.dist_min, 4ull * n));
        G.dist_max = reinterpret_cast<const uint32_t*>(stage(E.G.dist_max, 4ull * n));
        G.exit_idx = reinterpret_cast<const uint32_t*>(stage(E.G.exit_idx, 4ull * n));
        G.nbm = reinterpret_cast<const FlatGraph::NodeBubble*>(stage(E.G.nbm, sizeof(FlatGraph::NodeBubble) * (uint64_t)E.n_nbm));
        // the per-row records of the one-round-trip path, behind the graph arrays (P.rec_lds bytes; 0: that path reads the arrays)
        if (P.rec_lds) G.rec = reinterpret_cast<const FlatGraph::RowRec*>(stage(E.G.rec, sizeof(FlatGraph::RowRec) * (uint64_t)n));
    }
    if (!P.rec_lds) G.rec = nullptr;
    // (the wave index is uniform: say so, and every per-query pointer below lives in scalar registers)
    const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // descriptor rings of this wave (one per query it carries): LDS, or (a priority range too wide for it) slices of P.ring_global
    const uint32_t qpw = 64u / P.group;   // queries per wave
    uint32_t* ring = P.ring_global ? nullptr : reinterpret_cast<uint32_t*>(lds + P.graph_lds + P.rec_lds) + (uint64_t)wave * qpw * 3 * P.win;
    const uint32_t slot0 = (blockIdx.x * P.waves_per_block + wave) * qpw;   // first query slot of this wave
    if (P.ring_global && slot0 < P.E.n_queries) ring = P.ring_global + (uint64_t)slot0 * 3 * P.win;
    if (ring) {
        const uint32_t nring = min(qpw, P.E.n_queries > slot0 ? P.E.n_queries - slot0 : 0u) * 3 * P.win;
        for (uint32_t i = lane; i < nring; i += 64) ring[i] = BQ_EMPTY;
    }
    __syncthreads();
    if (slot0 >= P.E.n_queries) return;
    // graph arrays and descriptor ring both in LDS: typed LDS accesses (no FLAT instructions); else generic pointers
    const bool lds_all = P.graph_lds && !P.ring_global;
    if constexpr (MODE == 0) {
        if (lds_all) ws_search_query<EX_AS_GRAPH_LDS | EX_AS_RING_LDS | EX_AS_REC_LDS | EX_AS_NO_SPEC>(P, G, ring, lane, wave);   // (the host launches this kernel only then)
    } else if constexpr (MODE == 2) {
        ws_search_query<EX_AS_NO_SPEC>(P, G, ring, lane, wave);
    } else {
        // (the host asks for groups only with graph and rings in LDS)
        if (P.group == 32) ws_search_groups<EX_AS_GRAPH_LDS | EX_AS_RING_LDS | EX_AS_NO_SPEC, 32>(P, G, ring, lane, wave);
        else if (P.group == 16) ws_search_groups<EX_AS_GRAPH_LDS | EX_AS_RING_LDS | EX_AS_NO_SPEC, 16>(P, G, ring, lane, wave);
        else ws_search_groups<EX_AS_GRAPH_LDS | EX_AS_RING_LDS | EX_AS_NO_SPEC, 8>(P, G, ring, lane, wave);
    }
}

__global__ __launch_bounds__(1024) void poa_wsearch_kernel(WSearchParams P) { ws_kernel_body<0>(P); }
__global__ __launch_bounds__(1024) void poa_wsearch_groups_kernel(WSearchParams P) { ws_kernel_body<1>(P); }
__global__ __launch_bounds__(1024) void poa_wsearch_global_kernel(WSearchParams P) { ws_kernel_body<2>(P); }



// ---- several queries per wave: one per group of GS lanes ------------------------------------------------------------------
template <int AS, int GS>
__device__ __forceinline__ void ws_search_groups(const WSearchParams& P, const ExactGraph& G, uint32_t* ring0, uint32_t lane, uint32_t wave) {
    const ExactParams& E = P.E;
    constexpr uint32_t QPW = 64u / GS;
    const uint32_t grp = lane / GS, gl = lane % GS, gbase = grp * GS;
    const uint32_t slot = (blockIdx.x * P.waves_per_block + wave) * QPW + grp;
    const bool have = slot < E.n_queries;
    const uint32_t qi = E.first_query + (have ? slot : 0u);
    bool live = have && !(E.hybrid && E.dense_flags[qi] == 0);
    const uint64_t qbeg = E.qoff[qi];
    const uint32_t L = (uint32_t)(E.qoff[qi + 1] - qbeg);
    const uint32_t wslot = have ? slot : 0u;   // a group without a query computes addresses of slot 0 and never uses them
    ExactWork W;
    W.T = E.planes + E.plane_off[qi];
    W.n_rows = E.G.n_rows;
    W.pitch = E.pitch[qi];
    W.reached = E.reached + (uint64_t)wslot * E.G.n_exit * E.wpn;
    W.rsum = E.rsum + (uint64_t)wslot * E.G.n_exit * E.swpn;
    W.wpn = E.wpn; W.swpn = E.swpn;
    W.head = nullptr; W.n_prio = 0xFFFFFFFFu;
    W.pool = nullptr; W.pool_cap = 0;
    W.stack = E.stack + (uint64_t)wslot * E.stack_cap;
    W.stack_cap = E.stack_cap;
    W.bq_desc = ring0 + (uint64_t)grp * 3 * P.win; W.bq_win = P.win;
    W.bq_chunks = P.chunks + (uint64_t)wslot * P.chunk_cap * BQ_CHUNK;
    W.bq_chunk_cap = P.chunk_cap;

    ExactSearchT<AS> S(G, W, E.qseq + qbeg, L, E.C);
    ExactResult R{EX_OK, EX_INF, 0, 0, 0, G.end_row, L};
    uint32_t end_score = EX_INF, found = 0, steps = 0;
    // group-uniform state lives identically in every lane of the group; what one lane changes alone is handed round afterwards
    auto from_lane = [&](uint32_t v, uint32_t src) { return (uint32_t)__shfl((int)v, (int)(gbase + src), 64); };
    auto adopt = [&](uint32_t src) {
        S.err = from_lane(S.err, src);
        S.layer_min = from_lane(S.layer_min, src);
        S.bq_live = from_lane(S.bq_live, src);
        S.bq_hi = from_lane(S.bq_hi, src);
        S.bq_chunk_top = from_lane(S.bq_chunk_top, src);
        S.bq_free = from_lane(S.bq_free, src);
        found = from_lane(found, src);
        end_score = from_lane(end_score, src);
        R.end_row = from_lane(R.end_row, src);
        R.end_off = from_lane(R.end_off, src);
    };
    if (live && gl == 0) S.push_initial_states();
    adopt(0);
    S.bq_wr = gl == 0;
    const uint32_t cap = P.max_lanes < (uint32_t)GS ? P.max_lanes : (uint32_t)GS;

    // every lane runs the loop until no group of the wave has work left (shuffles need the whole wave inside)
    while (__any(live && !found && !S.err)) {
        const bool go = live && !found && !S.err;
        uint32_t st = 0; BqDesc d{0, 0};
        if (go && !S.bq_current(st, d)) S.err = EX_PANIC;   // "Could not align sequence!" (astar.rs:142-144)
        const bool run = go && !S.err;
        const uint32_t nb = run ? (d.n_top < cap ? d.n_top : cap) : 0u;
        const ExU4* ch = W.bq_chunks + (uint64_t)BQ_CHUNK * d.top;
        const bool act = gl < nb;
        ExU4 e{0, 0, 0, 0};
        uint32_t prev = EX_NIL;
        if (run) { prev = ch[0].x; if (act) e = ch[d.n_top - gl]; }
        uint32_t sk = 1;
        typename ExactSearchT<AS>::FastItem F{0, 0, 0, 0, 0};
        if (act) sk = S.inspect_fast(e.x, e.y, e.z, st, F);
        const uint64_t stop_w = __ballot(act && (sk == 0 || sk == 3 || S.err != 0));
        const uint32_t stop = (uint32_t)((stop_w >> gbase) & ((GS == 64) ? ~0ull : ((1ull << GS) - 1)));
        const uint32_t n = stop ? (uint32_t)__builtin_ctz(stop) : nb;
        if (run) {
            if (gl < n && sk == 2) S.num_pruned += 1;   // per-lane tallies, summed at the end
            if (gl > n) S.err = 0;                      // tests beyond the run are discarded with whatever they hit
            S.bq_drop(st, d, n < nb ? n + 1 : nb, prev);
            steps += 1;
        }
        const bool expand = run && n < nb;
        if (expand && gl == n && !S.err) {
            S.bq_wr = true;
            if (F.kind) found = S.process_fast(e.x, e.y, e.z, st, F, R, end_score) ? 1u : 0u;
            else {
                if (sk == 3) sk = S.inspect_skip(e.x, e.y, e.z, st);
                if (sk == 2) S.num_pruned += 1;
                if (sk == 0 && !S.err) found = S.process_popped(e.x, e.y, e.z, st, R, end_score) ? 1u : 0u;
            }
            S.bq_wr = gl == 0;
        }
        // (every lane shuffles; a group that did not expand takes its own lane 0's values, i.e. nothing changes)
        adopt(expand ? n : 0u);
    }

    uint32_t nq = S.num_queued, nv = S.num_visited, np = S.num_pruned;
    for (int o = GS / 2; o; o >>= 1) {
        nq += (uint32_t)__shfl_xor((int)nq, o, 64); nv += (uint32_t)__shfl_xor((int)nv, o, 64); np += (uint32_t)__shfl_xor((int)np, o, 64);
    }
    if (live && gl == 0) {
        E.status[qi] = S.err ? S.err : (found ? EX_OK : EX_PANIC);
        E.end_cell[2 * qi] = R.end_row;
        E.end_cell[2 * qi + 1] = R.end_off;
        if (P.counters) { P.counters[4 * qi] = nq; P.counters[4 * qi + 1] = nv; P.counters[4 * qi + 2] = np; P.counters[4 * qi + 3] = steps; }
    }
}

template <int AS>
__device__ __forceinline__ void ws_search_query(const WSearchParams& P, const ExactGraph& G, uint32_t* ring, uint32_t lane, uint32_t wave) {
    const ExactParams& E = P.E;
    const uint32_t slot = blockIdx.x * P.waves_per_block + wave;   // this wave's workspace (reached sets, stack, chunks, ring)
    bool first = true;
    for (;;) {
    uint32_t pos = slot;
    if (P.work_counter) {
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(P.work_counter, 1u);
        pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
        if (pos >= E.n_queries) return;
        if (P.order) pos = P.order[pos];
    } else if (!first) return;
    const uint32_t qi = E.first_query + pos;
    if (E.hybrid && E.dense_flags[qi] == 0) { first = false; continue; }
    if (!first || P.work_counter) {
        // the workspace of the previous search of this wave: reached sets back to empty, ring back to empty
        // (the host clears nothing in persistent mode)
        uint64_t* z = E.reached + (uint64_t)slot * E.G.n_exit * E.wpn;
        for (uint64_t i = lane; i < (uint64_t)E.G.n_exit * E.wpn; i += 64) z[i] = 0;
        uint64_t* zs = E.rsum + (uint64_t)slot * E.G.n_exit * E.swpn;
        for (uint64_t i = lane; i < (uint64_t)E.G.n_exit * E.swpn; i += 64) zs[i] = 0;
        for (uint32_t i = lane; i < 3 * P.win; i += 64) ring[i] = BQ_EMPTY;
    }
    first = false;
    const uint64_t qbeg = E.qoff[qi];
    const uint32_t L = (uint32_t)(E.qoff[qi + 1] - qbeg);
    ExactWork W;
    W.T = E.planes + E.plane_off[qi];
    W.n_rows = E.G.n_rows;
    W.pitch = E.pitch[qi];
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

    ExactSearchT<AS> S(G, W, E.qseq + qbeg, L, E.C);
    ExactResult R{EX_OK, EX_INF, 0, 0, 0, G.end_row, L};
    uint32_t end_score = EX_INF, found = 0, steps = 0;

    // uniform state lives identically in every lane; whatever one lane changes alone is broadcast afterwards
    auto adopt = [&](uint32_t from) {
        S.err = ws_bcast(S.err, from);
        S.layer_min = ws_bcast(S.layer_min, from);
        S.bq_live = ws_bcast(S.bq_live, from);
        S.bq_hi = ws_bcast(S.bq_hi, from);
        S.bq_chunk_top = ws_bcast(S.bq_chunk_top, from);
        S.bq_free = ws_bcast(S.bq_free, from);
        found = ws_bcast(found, from);
        if (found) { end_score = ws_bcast(end_score, from); R.end_row = ws_bcast(R.end_row, from); R.end_off = ws_bcast(R.end_off, from); }
    };
#if POA_WS_UNIFORM
    // Everything outside the test of the entries runs on EVERY lane with the same values (the popped entry and what its test
    // found are broadcast from its lane): uniform control flow — scalar branches, no exec-mask bookkeeping, no copies where
    // divergent paths merge, which is what the single-lane form of this code mostly consisted of.  Only lane 0 stores to the
    // queue and sets marks (bq_wr); table cells are stored by all lanes alike (same address, same value).
    (void)adopt;
    S.bq_wr = lane == 0; S.mark_on = lane == 0;
    S.push_initial_states();
    uint32_t pruned_lane = 0;   // per-lane tally of the entries the test found pruned
#else
    if (lane == 0) S.push_initial_states();
    adopt(0);
    S.bq_wr = lane == 0;
#endif

    unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool prof = P.prof != nullptr;
#define WS_TICK(k) do { if (prof) { const unsigned long long now_ = clock64(); pc[k] += now_ - t_last; t_last = now_; } } while (0)
    unsigned long long t_last = prof ? clock64() : 0;
    uint32_t lanes_now = P.adapt_lanes ? (P.adapt_lanes < P.max_lanes ? P.adapt_lanes : P.max_lanes) : P.max_lanes;
    while (!found && !S.err) {
        uint32_t st; BqDesc d;
        if (!S.bq_current(st, d)) { S.err = EX_PANIC; break; }  // "Could not align sequence!" (astar.rs:142-144)
        const uint32_t nb = d.n_top < lanes_now ? d.n_top : lanes_now;  // <= 63 entries in the top chunk: lane i takes the i-th from the top
        const ExU4* ch = W.bq_chunks + (uint64_t)BQ_CHUNK * d.top;
        const bool act = lane < nb;
        const ExU4 e = ch[act ? d.n_top - lane : 0];  // the idle lanes read slot 0: {previous chunk}
        if (prof) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        WS_TICK(0);
        uint32_t sk = 1;
        typename ExactSearchT<AS>::FastItem F{0, 0, 0, 0, 0};
        if (act) sk = S.inspect_fast(e.x, e.y, e.z, st, F);
        // first entry that is neither stale nor pruned; an entry of another shape (sk == 3) ends the run too and is tested
        // by its own lane below: the lanes stay on the one-round-trip path together
        const uint64_t stop = __ballot(act && (sk == 0 || sk == 3 || S.err != 0));
        WS_TICK(1);
        const uint32_t n = stop ? (uint32_t)__builtin_ctzll(stop) : nb;
        // the next step: after an expansion what lies on top was just pushed (rarely stale): few lanes — each tested entry costs
        // loads and its own branch of the test; a run of stale / pruned entries that used up its lanes goes on wider
        if (P.adapt_lanes) { lanes_now = n < nb ? P.adapt_lanes : 4 * lanes_now; if (lanes_now > P.max_lanes) lanes_now = P.max_lanes; }
#if POA_WS_UNIFORM
        if (lane < n && sk == 2) pruned_lane += 1;
        S.err = n < nb ? ws_bcast(S.err, n) : 0u;     // what the run's last test hit; tests beyond the run are discarded
        const uint32_t prev = ws_bcast(e.x, 63);      // lane 63 is never active (nb <= 63)
        S.bq_drop(st, d, n < nb ? n + 1 : nb, prev);
        steps += 1;
        WS_TICK(2);
        if (n < nb && !S.err) {
            const uint32_t ux = ws_bcast(e.x, n), uy = ws_bcast(e.y, n), uz = ws_bcast(e.z, n);
            typename ExactSearchT<AS>::FastItem Fu{ws_bcast(F.kind, n), ws_bcast(F.c, n), ws_bcast(F.c1, n), ws_bcast(F.t0, n), ws_bcast(F.t1, n),
                                                   ws_bcast(F.t2, n), ws_bcast(F.t3, n), ws_bcast(F.t4, n)};
            uint32_t sku = ws_bcast(sk, n);
            const uint32_t kind0 = Fu.kind;
            if (Fu.kind) found = S.process_fast(ux, uy, uz, st, Fu, R, end_score) ? 1u : 0u;
            else {
                // a row whose bubbles need the range form of the test: asked here
                if (sku == 3 && S.use_rec()) sku = S.template inspect_fast_t<true>(ux, uy, uz, st, Fu);
                if (sku == 0 && Fu.kind && !S.err) found = S.process_fast(ux, uy, uz, st, Fu, R, end_score) ? 1u : 0u;
                else {
                    if (sku == 3) sku = S.inspect_skip(ux, uy, uz, st);
                    if (sku == 2) S.num_pruned += 1;
                    if (sku == 0 && !S.err) found = S.process_popped(ux, uy, uz, st, R, end_score) ? 1u : 0u;
                }
            }
            if (prof) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); WS_TICK(kind0 ? 3 : 4); pc[kind0 ? 5 : 6] += 1; }
        }
    }

    // (the counters of the uniform part hold the same value in every lane)
    const uint32_t nq = S.num_queued, nv = S.num_visited, np = S.num_pruned + ws_wave_sum(pruned_lane);
#else
        if (lane < n && sk == 2) S.num_pruned += 1;   // per-lane tallies, summed at the end
        if (lane > n) S.err = 0;                      // tests beyond the run are discarded with whatever they hit
        const uint32_t prev = ws_bcast(e.x, 63);      // lane 63 is never active (nb <= 63)
        S.bq_drop(st, d, n < nb ? n + 1 : nb, prev);
        steps += 1;
        WS_TICK(2);
        if (n < nb) {
            if (lane == n && !S.err) {
                S.bq_wr = true;
                if (F.kind) found = S.process_fast(e.x, e.y, e.z, st, F, R, end_score) ? 1u : 0u;
                else {
                    // a row whose bubbles need the range form of the test: asked here, by the entry's own lane alone
                    if (sk == 3 && S.use_rec()) sk = S.template inspect_fast_t<true>(e.x, e.y, e.z, st, F);
                    if (sk == 0 && F.kind && !S.err) found = S.process_fast(e.x, e.y, e.z, st, F, R, end_score) ? 1u : 0u;
                    else {
                    if (sk == 3) sk = S.inspect_skip(e.x, e.y, e.z, st);
                    if (sk == 2) S.num_pruned += 1;
                    if (sk == 0 && !S.err) found = S.process_popped(e.x, e.y, e.z, st, R, end_score) ? 1u : 0u;
                    }
                }
                S.bq_wr = lane == 0;
            }
            adopt(n);
            if (prof) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); const uint32_t k_ = ws_bcast(F.kind, n); WS_TICK(k_ ? 3 : 4); pc[k_ ? 5 : 6] += 1; }
        }
    }

    const uint32_t nq = ws_wave_sum(S.num_queued), nv = ws_wave_sum(S.num_visited), np = ws_wave_sum(S.num_pruned);
#endif
    if (lane == 0) {
        E.status[qi] = S.err ? S.err : (found ? EX_OK : EX_PANIC);
        E.end_cell[2 * qi] = R.end_row;
        E.end_cell[2 * qi + 1] = R.end_off;
        if (P.counters) {
            if (prof) for (int k = 0; k < 8; ++k) P.prof[8 * (uint64_t)qi + k] = pc[k];
            P.counters[4 * qi] = nq; P.counters[4 * qi + 1] = nv; P.counters[4 * qi + 2] = np; P.counters[4 * qi + 3] = steps;
        }
    }
    }  // next query of this wave
}

}  // namespace poa_amd
