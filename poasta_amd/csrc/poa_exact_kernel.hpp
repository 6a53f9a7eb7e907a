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
    uint64_t* reached; uint64_t reached_stride;   // per slot (u64 words)
    uint32_t* rcnt;                   // per slot: n_rows
    uint32_t* head; uint32_t n_prio;  // per slot: 3 * n_prio
    ExQEntry* pool; uint32_t pool_cap;
    ExStackEntry* stack; uint32_t stack_cap;
    ExactCosts C;
    uint32_t* status;                 // [total] EX_* (0xFFFFFFFF = not replayed)
    uint32_t lanes_per_wave;          // active lanes per wave (divergence vs occupancy knob)
};

__global__ __launch_bounds__(64) void poa_exact_kernel(ExactParams P) {
    const uint32_t lane = threadIdx.x & 63u;
    if (lane >= P.lanes_per_wave) return;
    const uint32_t slot = blockIdx.x * P.lanes_per_wave + lane;
    if (slot >= P.n_queries) return;
    const uint32_t qi = P.first_query + slot;
    if (P.hybrid && P.dense_flags[qi] == 0) return;
    const uint64_t qbeg = P.qoff[qi];
    const uint32_t L = (uint32_t)(P.qoff[qi + 1] - qbeg);
    const uint32_t pitch = P.pitch[qi];
    const uint64_t RP = (uint64_t)P.G.n_rows * pitch;
    ExactWork W;
    W.M = P.planes + P.plane_off[qi];
    W.I = W.M + RP;
    W.D = W.I + RP;
    W.pitch = pitch;
    W.reached = P.reached + (uint64_t)slot * P.reached_stride;
    W.reached_cnt = P.rcnt + (uint64_t)slot * P.G.n_rows;
    W.wpn = (L + 1 + 63) / 64;
    W.head = P.head + (uint64_t)slot * 3 * P.n_prio;
    W.n_prio = P.n_prio;
    W.pool = P.pool + (uint64_t)slot * P.pool_cap;
    W.pool_cap = P.pool_cap;
    W.stack = P.stack + (uint64_t)slot * P.stack_cap;
    W.stack_cap = P.stack_cap;
    ExactSearch S(P.G, W, P.qseq + qbeg, L, P.C);
    const ExactResult R = S.run();
    P.status[qi] = R.status;
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
