// Replay of the reference's search, one wavefront per query, the top entries of a stack expanded AT ONCE (exact / hybrid mode).
//
// The reference pops one state at a time (astar.rs:141-216) and its tie-breaks are decided by that order, so the replay has to
// produce exactly the writes and pushes of that loop.  But the entries that sit on top of one another in a stack of the bucket
// queue (gap_affine.rs:929-1013) rarely have anything to do with each other: on the 1 kbp reads of the benchmark a bucket holds
// some 400 states spread over a band of the table.  A step of this kernel therefore
//
//   1. reads the top entries of the current stack, one per lane;
//   2. lets every lane process ITS entry in the log mode of the search object (poa_exact.hpp, SpecLane): the ordinary code —
//      stale test, pruning, greedy extension, relaxations — reading the table as the step found it (plus the lane's own log)
//      and writing cells, reached marks and pushes to a per-lane log; entries the expansion puts in front of the next entry of
//      the stack (same bucket, state of equal or higher pop priority) stay with the lane and are processed by it, in the
//      queue's order;
//   3. finds the first lane that read a cell or a reached mark an EARLIER lane of the step logged a write to (a table of lane
//      masks in LDS, keyed by 64-cell blocks / words of the reached sets, names the candidates; the test itself is exact);
//   4. commits the logs of the lanes before it in lane order — the writes and pushes of the sequential loop — pops their
//      entries, and leaves the rest for the next step.  A lane that ends the search or has entries of its group left is the
//      last one committed (what is left goes on the queue); a lane that needs what the log mode does not do takes the
//      sequential code once it is the first lane of a step.
//
// ExactSearch::run_parallel (poa_exact.hpp) is this schedule one lane after the other; compiled for the host it is diffed
// against the oracle (tests/test_exact_replay.py), this kernel against both on the GPU.
#pragma once
#include <hip/hip_runtime.h>

#include "poa_exact_kernel.hpp"

namespace poa_amd {

constexpr uint32_t PS_TAB = 512;   // slots of the conflict table (lane masks), per wave

struct PSearchParams {
    ExactParams E;            // graph, queries, planes, reached sets, costs, status / end cell
    ExU4* chunks;             // per slot: chunk_cap chunks of BQ_CHUNK slots
    uint32_t chunk_cap;
    uint32_t win;             // descriptor ring: priorities per wave (power of two)
    uint32_t* ring_global;    // null: the rings live in LDS; else [slots * 3 * win] in global memory
    uint32_t graph_lds;       // bytes of the staged graph arrays (exact_lds_bytes), 0: read them from global memory
    uint32_t waves_per_block;
    uint32_t max_lanes;       // entries per step (<= 63)
    uint32_t rmax;            // entries a lane may process per step (its own and what they push in front of the next)
    uint32_t* scratch;        // per slot: ps_scratch_words() words — the lanes' logs
    uint32_t* work_counter;   // persistent scheduling (see poa_wsearch.hpp); null: query = block / wave index
    const uint32_t* order;
    uint32_t* counters;       // optional [4 * total]: num_queued, num_visited, num_pruned, steps
    unsigned long long* prof; // optional [8 * total]: cycles per phase
};

// per wave, in words: cell-write log (index, value), marks (exit, offset), pushes (4 words), extension stack (3 words)
__host__ __device__ inline uint32_t ps_scratch_words() { return 64u * (2 * SP_KW + 2 * SP_KM + 4 * SP_KP + 3 * SP_KDS); }
// per wave, in bytes of LDS: conflict table, read cells, mark ranges (exit, lo, hi), their counts, one word for the lanes found in conflict
__host__ __device__ inline uint32_t ps_lds_bytes() { return 8u * PS_TAB + 4u * 64u * (SP_KRC + 3 * SP_KRM + 1) + 16u; }

__device__ __forceinline__ uint32_t ps_bcast(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane); }
__device__ __forceinline__ uint32_t ps_wave_sum(uint32_t v) {
    for (int o = 32; o; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, 64);
    return v;
}
__device__ __forceinline__ uint32_t ps_key_cell(uint32_t ix) { return ((ix >> 6) * 0x9E3779B1u) >> (32 - 9); }          // PS_TAB == 512
__device__ __forceinline__ uint32_t ps_key_mark(uint32_t x, uint32_t word) { return ((x * 0x85EBCA6Bu + word * 0xC2B2AE35u + 0x27D4EB2Fu) * 0x9E3779B1u) >> (32 - 9); }
constexpr uint32_t PS_ALL_WORDS = 0xFFFFFFu;   // "some word of this exit's set": key of a range too wide to name its words

template <int AS>
__device__ __forceinline__ void ps_search_query(const PSearchParams& P, const ExactGraph& G, uint32_t* ring, uint8_t* wlds, uint32_t lane, uint32_t wave);

__global__ __launch_bounds__(512) void poa_psearch_kernel(PSearchParams P) {
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
    }
    const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // per wave: [descriptor ring (unless global)] [conflict table | read cells | mark ranges | conflict word]
    const uint32_t ring_bytes = P.ring_global ? 0u : 3u * P.win * 4u;
    const uint32_t per_wave = ((ring_bytes + 15u) & ~15u) + ps_lds_bytes();
    uint8_t* wbase = lds + P.graph_lds + (uint64_t)wave * per_wave;
    uint32_t* ring = P.ring_global ? nullptr : reinterpret_cast<uint32_t*>(wbase);
    uint8_t* wlds = wbase + ((ring_bytes + 15u) & ~15u);
    const uint32_t slot = blockIdx.x * P.waves_per_block + wave;
    if (P.ring_global) ring = P.ring_global + (uint64_t)slot * 3 * P.win;
    for (uint32_t i = lane; i < 3 * P.win; i += 64) ring[i] = BQ_EMPTY;
    __syncthreads();
    const bool lds_all = P.graph_lds && !P.ring_global;
    if (lds_all) ps_search_query<EX_AS_GRAPH_LDS | EX_AS_RING_LDS>(P, G, ring, wlds, lane, wave);
    else ps_search_query<0>(P, G, ring, wlds, lane, wave);
}

template <int AS>
__device__ __forceinline__ void ps_search_query(const PSearchParams& P, const ExactGraph& G, uint32_t* ring, uint8_t* wlds, uint32_t lane, uint32_t wave) {
    const ExactParams& E = P.E;
    const uint32_t slot = blockIdx.x * P.waves_per_block + wave;   // this wave's workspace (reached sets, chunks, ring, logs)
    typedef __attribute__((address_space(3))) unsigned long long lds_u64;
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    lds_u64* tab = (lds_u64*)wlds;
    uint32_t* l_rc = reinterpret_cast<uint32_t*>(wlds + 8u * PS_TAB);
    uint32_t* l_rm = l_rc + 64u * SP_KRC;
    lds_u32* l_cnt = (lds_u32*)(l_rm + 64u * 3 * SP_KRM);   // n_rc | n_rm << 16 of every lane
    lds_u64* l_conf = (lds_u64*)(wlds + 8u * PS_TAB + 4u * 64u * (SP_KRC + 3 * SP_KRM + 1));
    uint32_t* sc = P.scratch + (uint64_t)slot * ps_scratch_words();
    const unsigned long long lanebit = 1ull << lane;
    const unsigned long long above = lane == 63 ? 0ull : (~0ull << (lane + 1));
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
    else if (slot >= E.n_queries) return;
    const uint32_t qi = E.first_query + pos;
    if (E.hybrid && E.dense_flags[qi] == 0) { first = false; continue; }
    if (!first || P.work_counter) {
        // the workspace of the previous search of this wave: reached sets back to empty, ring back to empty
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
    // the lane's logs (element k at base[k * 64])
    S.sl.stride = 64;
    S.sl.w_idx = sc + lane; S.sl.w_val = sc + 64u * SP_KW + lane;
    S.sl.m_x = sc + 64u * 2 * SP_KW + lane; S.sl.m_off = sc + 64u * (2 * SP_KW + SP_KM) + lane;
    S.sl.p = reinterpret_cast<ExU4*>(sc + 64u * (2 * SP_KW + 2 * SP_KM)) + lane;
    S.sl.dstack = reinterpret_cast<ExStackEntry*>(sc + 64u * (2 * SP_KW + 2 * SP_KM + 4 * SP_KP)) + lane;
    S.sl.rc = l_rc + lane; S.sl.rm_x = l_rm + lane; S.sl.rm_lo = l_rm + 64u * SP_KRM + lane; S.sl.rm_hi = l_rm + 64u * 2 * SP_KRM + lane;
    ExactResult R{EX_OK, EX_INF, 0, 0, 0, G.end_row, L};
    uint32_t end_score = EX_INF, found = 0, steps = 0;

    // uniform state lives identically in every lane; whatever one lane changes alone is broadcast afterwards
    auto adopt = [&](uint32_t from) {
        S.err = ps_bcast(S.err, from);
        S.layer_min = ps_bcast(S.layer_min, from);
        S.bq_live = ps_bcast(S.bq_live, from);
        S.bq_hi = ps_bcast(S.bq_hi, from);
        S.bq_chunk_top = ps_bcast(S.bq_chunk_top, from);
        S.bq_free = ps_bcast(S.bq_free, from);
        found = ps_bcast(found, from);
        end_score = ps_bcast(end_score, from);
        R.end_row = ps_bcast(R.end_row, from);
        R.end_off = ps_bcast(R.end_off, from);
    };
    if (lane == 0) S.push_initial_states();
    adopt(0);
    S.bq_wr = lane == 0;

    unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool prof = P.prof != nullptr;
#define PS_TICK(k) do { if (prof) { const unsigned long long now_ = clock64(); pc[k] += now_ - t_last; t_last = now_; } } while (0)
    unsigned long long t_last = prof ? clock64() : 0;
    while (!found && !S.err) {
        uint32_t st; BqDesc d;
        if (!S.bq_current(st, d)) { S.err = EX_PANIC; break; }   // "Could not align sequence!" (astar.rs:142-144)
        const uint32_t f = S.layer_min;
        const uint32_t nb = d.n_top < P.max_lanes ? d.n_top : P.max_lanes;   // <= 63 entries in the top chunk: lane i takes the i-th from the top
        const ExU4* ch = W.bq_chunks + (uint64_t)BQ_CHUNK * d.top;
        const bool act = lane < nb;
        const ExU4 e = ch[act ? d.n_top - lane : 0];   // the idle lanes read slot 0: {previous chunk}
        const uint32_t prev = ps_bcast(e.x, 63);         // lane 63 is never active (nb <= 63)
        // the conflict table of this step
        tab[lane] = 0; tab[lane + 64] = 0; tab[lane + 128] = 0; tab[lane + 192] = 0;
        tab[lane + 256] = 0; tab[lane + 320] = 0; tab[lane + 384] = 0; tab[lane + 448] = 0;
        if (lane == 0) *l_conf = 0;
        steps += 1;
        PS_TICK(0);
        // ---- every lane: its entry (and what that puts in front of the next one) in log mode ----
        ExactResult Rl = R; uint32_t esl = end_score;
        S.sl.flags = 0; S.sl.n_w = S.sl.n_m = S.sl.n_p = S.sl.n_pd = S.sl.n_rc = S.sl.n_rm = 0; S.sl.n_ent = 0;
        if (act) S.spec_group(e.x, e.y, e.z, st, f, P.rmax, Rl, esl);
        PS_TICK(1);
        // ---- who read what an earlier lane wrote ----
        // readers enter their lane under the 64-cell block of every cell / the word of every mark range they depended on ...
        l_cnt[lane] = S.sl.n_rc | S.sl.n_rm << 16;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        for (uint32_t k = 0; k < S.sl.n_rc; ++k) __hip_atomic_fetch_or(&tab[ps_key_cell(((lds_u32*)S.sl.rc)[k * 64])], lanebit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        for (uint32_t k = 0; k < S.sl.n_rm; ++k) {
            const uint32_t x = ((lds_u32*)S.sl.rm_x)[k * 64], lo = ((lds_u32*)S.sl.rm_lo)[k * 64], hi = ((lds_u32*)S.sl.rm_hi)[k * 64];
            const uint32_t w0 = lo >> 6, w1 = hi >> 6;
            if (w1 - w0 < 3) for (uint32_t w = w0; w <= w1; ++w) __hip_atomic_fetch_or(&tab[ps_key_mark(x, w)], lanebit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            else __hip_atomic_fetch_or(&tab[ps_key_mark(x, PS_ALL_WORDS)], lanebit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // ... writers look their writes up and check the lanes above them they find there, exactly
        unsigned long long hit = 0;
        for (uint32_t k = 0; k < S.sl.n_w; ++k) {
            const uint32_t ix = S.sl.w_idx[k * 64];
            unsigned long long m = tab[ps_key_cell(ix)] & above;
            while (m) {
                const uint32_t b = (uint32_t)__builtin_ctzll(m);
                m &= m - 1;
                const uint32_t nrc = l_cnt[b] & 0xFFFFu;   // (a shuffle would read an inactive lane here)
                bool h = false;
                for (uint32_t r = 0; r < nrc; ++r) h = h || ((lds_u32*)l_rc)[r * 64 + b] == ix;
                if (h) hit |= 1ull << b;
            }
        }
        for (uint32_t k = 0; k < S.sl.n_m; ++k) {
            const uint32_t x = S.sl.m_x[k * 64], off = S.sl.m_off[k * 64];
            unsigned long long m = (tab[ps_key_mark(x, off >> 6)] | tab[ps_key_mark(x, PS_ALL_WORDS)]) & above;
            while (m) {
                const uint32_t b = (uint32_t)__builtin_ctzll(m);
                m &= m - 1;
                const uint32_t nrm = l_cnt[b] >> 16;
                bool h = false;
                for (uint32_t r = 0; r < nrm; ++r)
                    h = h || (((lds_u32*)l_rm)[r * 64 + b] == x && ((lds_u32*)l_rm)[64 * SP_KRM + r * 64 + b] <= off && off <= ((lds_u32*)l_rm)[128 * SP_KRM + r * 64 + b]);
                if (h) hit |= 1ull << b;
            }
        }
        if (hit) __hip_atomic_fetch_or(l_conf, hit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const unsigned long long conflict = *l_conf;
        PS_TICK(2);
        // ---- how many lanes commit ----
        const unsigned long long active = nb >= 64 ? ~0ull : ((1ull << nb) - 1);
        const unsigned long long m_cx = __ballot(act && (S.sl.flags & SPF_COMPLEX)) | (conflict & active);
        const unsigned long long m_last = __ballot(act && (S.sl.flags & (SPF_FOUND | SPF_LEFTOVER)));
        uint32_t n_commit = m_cx ? (uint32_t)__builtin_ctzll(m_cx) : nb;
        if (m_last) { const uint32_t a = (uint32_t)__builtin_ctzll(m_last) + 1; n_commit = a < n_commit ? a : n_commit; }
        if (n_commit == 0) {
            // the first lane needs the sequential code (direct mode): its entry alone
            S.bq_drop(st, d, 1, prev);
            if (lane == 0) {
                S.bq_wr = true;
                const uint32_t sk = S.inspect_skip(e.x, e.y, e.z, st);
                if (sk == 2) S.num_pruned += 1;
                if (sk == 0 && !S.err) found = S.process_popped(e.x, e.y, e.z, st, R, end_score) ? 1u : 0u;
                S.bq_wr = false;
            }
            S.bq_wr = lane == 0;
            adopt(0);
            PS_TICK(4);
            if (prof) pc[6] += 1;
            continue;
        }
        // ---- commit: table, marks, counters of the lanes before the cut; their entries leave the stack ----
        if (lane < n_commit) S.spec_commit(S.sl);
        S.bq_drop(st, d, n_commit, prev);
        // what the last lane left pending goes on the queue in its push order
        const uint32_t last = n_commit - 1;
        const uint32_t npd = ps_bcast(S.sl.n_pd, last);
        if (npd) {
            if (lane == last) {
                S.bq_wr = true;
                if (S.sl.n_pd > 0) S.bq_push(S.sl.pd_key[0] >> 2, S.sl.pd_key[0] & 3u, S.sl.pd_score[0], S.sl.pd_row[0], S.sl.pd_off[0]);
                if (S.sl.n_pd > 1) S.bq_push(S.sl.pd_key[1] >> 2, S.sl.pd_key[1] & 3u, S.sl.pd_score[1], S.sl.pd_row[1], S.sl.pd_off[1]);
                if (S.sl.n_pd > 2) S.bq_push(S.sl.pd_key[2] >> 2, S.sl.pd_key[2] & 3u, S.sl.pd_score[2], S.sl.pd_row[2], S.sl.pd_off[2]);
                if (S.sl.n_pd > 3) S.bq_push(S.sl.pd_key[3] >> 2, S.sl.pd_key[3] & 3u, S.sl.pd_score[3], S.sl.pd_row[3], S.sl.pd_off[3]);
            }
            S.bq_wr = lane == 0;
            adopt(last);
        }
        PS_TICK(3);
        // the logged pushes, lane after lane (every lane runs the queue code alike; lane 0 stores)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (uint32_t a = 0; a < n_commit; ++a) {
            const uint32_t np = ps_bcast(S.sl.n_p, a);
            const ExU4* pa = reinterpret_cast<const ExU4*>(sc + 64u * (2 * SP_KW + 2 * SP_KM)) + a;
            for (uint32_t k = 0; k < np; ++k) {
                const ExU4 q = pa[k * 64];
                const uint32_t qx = ps_bcast(q.x, 0), qy = ps_bcast(q.y, 0), qz = ps_bcast(q.z, 0), qw = ps_bcast(q.w, 0);
                S.bq_push(qw >> 2, qw & 3u, qx, qy, qz);
            }
        }
        if (m_last && (uint32_t)__builtin_ctzll(m_last) == last) {
            const uint32_t fl = ps_bcast(S.sl.flags, last);
            if (fl & SPF_FOUND) {
                found = 1;
                end_score = ps_bcast(esl, last);
                R.end_row = ps_bcast(Rl.end_row, last);
                R.end_off = ps_bcast(Rl.end_off, last);
            }
        }
        PS_TICK(5);
        if (prof) pc[7] += n_commit;
    }

    const uint32_t nq = ps_wave_sum(S.num_queued), nv = ps_wave_sum(S.num_visited), np = ps_wave_sum(S.num_pruned);
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
#undef PS_TICK
}

}  // namespace poa_amd
