// Device wrapper of the exact-replay search (poa_exact.hpp): one query per GPU thread.
// The search is inherently sequential per query (a priority-queue driven best-first search whose
// tie-breaks are the point), so parallelism comes only from the batch: every lane runs its own
// search on its own workspace slice; memory latency, not arithmetic, bounds it.
#pragma once
#include <hip/hip_runtime.h>

#include "poa_exact.hpp"

namespace poa_amd {

struct ExactParams {
    ExactGraph G;
    uint32_t first_query, n_queries;  // chunk
    uint32_t hybrid;                  // 1: only queries whose dense flags are non-zero
    const uint32_t* dense_flags;      // [total]
    const uint8_t* qseq;
    const uint64_t* qoff;
    const uint32_t* pitch;
    const uint64_t* plane_off;
    uint32_t* planes;                 // u32 layout, INF-filled for the selected queries
    uint64_t* reached;                // per slot: n_exit * wpn words
    uint64_t* rsum;                   // per slot: n_exit * swpn words
    uint32_t wpn, swpn;               // sized for the longest query of the batch
    uint32_t* head; uint32_t n_prio;  // per slot: 3 * n_prio
    ExQEntry* pool; uint32_t pool_cap;
    ExStackEntry* stack; uint32_t stack_cap;
    ExactCosts C;
    uint32_t* status;                 // [total] EX_* (0xFFFFFFFF = not replayed)
    uint32_t* end_cell;               // [2 * total] (row, offset) the search stopped at
    uint32_t lanes_per_wave;          // active lanes per wave (divergence vs occupancy knob)
    uint32_t lds_graph;               // 1: the launch carries enough dynamic LDS to hold the graph arrays
    uint32_t n_succ, n_nbm;           // lengths of G.succ / G.nbm
};

// bytes of dynamic LDS the graph arrays need (each array padded to 16 bytes)
__host__ __device__ inline uint32_t exact_lds_bytes(uint32_t n_rows, uint32_t n_succ, uint32_t n_nbm) {
    auto pad = [](uint64_t b) { return (uint32_t)((b + 15) & ~15ull); };
    return pad(n_rows) + 2 * pad(4ull * (n_rows + 1)) + pad(4ull * n_succ) + 3 * pad(4ull * n_rows) +
           pad(sizeof(FlatGraph::NodeBubble) * (uint64_t)n_nbm);
}

// Every search of a block reads the same graph: stage its arrays in LDS once (latency, not bandwidth, bounds the search, and
// an LDS read is an order of magnitude closer than an L2 hit).  All threads of the block call this; G receives the LDS
// addresses (generic pointers), src holds the global ones.
__device__ inline void exact_stage_graph(ExactGraph& G, const ExactGraph& src, uint8_t* lds, uint32_t n_succ, uint32_t n_nbm) {
    uint32_t at = 0;
    auto stage = [&](const void* from, uint64_t bytes) {
        uint8_t* dst = lds + at;
        const uint32_t words = (uint32_t)((bytes + 3) / 4);
        const uint32_t* s32 = static_cast<const uint32_t*>(from);
        for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) reinterpret_cast<uint32_t*>(dst)[i] = s32[i];
        at += (uint32_t)((bytes + 15) & ~15ull);
        return dst;
    };
    const uint32_t n = src.n_rows;
    G.sym = stage(src.sym, n);
    G.succ_off = reinterpret_cast<const uint32_t*>(stage(src.succ_off, 4ull * (n + 1)));
    G.nbm_off = reinterpret_cast<const uint32_t*>(stage(src.nbm_off, 4ull * (n + 1)));
    G.succ = reinterpret_cast<const uint32_t*>(stage(src.succ, 4ull * n_succ));
    G.dist_min = reinterpret_cast<const uint32_t*>(stage(src.dist_min, 4ull * n));
    G.dist_max = reinterpret_cast<const uint32_t*>(stage(src.dist_max, 4ull * n));
    G.exit_idx = reinterpret_cast<const uint32_t*>(stage(src.exit_idx, 4ull * n));
    G.nbm = reinterpret_cast<const FlatGraph::NodeBubble*>(stage(src.nbm, sizeof(FlatGraph::NodeBubble) * (uint64_t)n_nbm));
    __syncthreads();
}

constexpr int EXACT_BLOCK = 256;  // 4 waves share one LDS copy of the graph

__global__ __launch_bounds__(EXACT_BLOCK) void poa_exact_kernel(ExactParams P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    ExactGraph G = P.G;
    if (P.lds_graph) exact_stage_graph(G, P.G, lds, P.n_succ, P.n_nbm);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane >= P.lanes_per_wave) return;
    const uint32_t slot = (blockIdx.x * (EXACT_BLOCK / 64) + wave) * P.lanes_per_wave + lane;
    if (slot >= P.n_queries) return;
    const uint32_t qi = P.first_query + slot;
    if (P.hybrid && P.dense_flags[qi] == 0) return;
    const uint64_t qbeg = P.qoff[qi];
    const uint32_t L = (uint32_t)(P.qoff[qi + 1] - qbeg);
    const uint32_t pitch = P.pitch[qi];
    ExactWork W;
    W.T = P.planes + P.plane_off[qi];
    W.n_rows = P.G.n_rows;
    W.pitch = pitch;
    W.reached = P.reached + (uint64_t)slot * P.G.n_exit * P.wpn;
    W.rsum = P.rsum + (uint64_t)slot * P.G.n_exit * P.swpn;
    W.wpn = P.wpn; W.swpn = P.swpn;
    W.head = P.head + (uint64_t)slot * 3 * P.n_prio;
    W.n_prio = P.n_prio;
    W.pool = P.pool + (uint64_t)slot * P.pool_cap;
    W.pool_cap = P.pool_cap;
    W.stack = P.stack + (uint64_t)slot * P.stack_cap;
    W.stack_cap = P.stack_cap;
    ExactSearchT<EX_AS_NO_SPEC> S(G, W, P.qseq + qbeg, L, P.C);
    const ExactResult R = S.run();
    P.status[qi] = R.status;
    P.end_cell[2 * qi] = R.end_row;
    P.end_cell[2 * qi + 1] = R.end_off;
}

// INF-fill the u32 planes of the queries the replay will run on (coalesced, one block column per query)
__global__ __launch_bounds__(256) void poa_fill_planes_kernel(uint32_t* planes, const uint64_t* plane_off, const uint32_t* pitch,
                                                              uint32_t n_rows, uint32_t first_query, uint32_t hybrid,
                                                              const uint32_t* dense_flags) {
    const uint32_t qi = first_query + blockIdx.y;
    if (hybrid && dense_flags[qi] == 0) return;
    const uint64_t n4 = 3ull * n_rows * pitch[qi] / 4;  // pitch is a multiple of 64
    uint4* p = reinterpret_cast<uint4*>(planes + plane_off[qi]);
    const uint4 v = make_uint4(EX_INF, EX_INF, EX_INF, EX_INF);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * blockDim.x) p[i] = v;
}

}  // namespace poa_amd
