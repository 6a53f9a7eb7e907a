// gfx950 (CDNA4) kernels of the gap-affine POA alignment engine.  Product code.
//
// What is computed (derivation: DESIGN.md §2; reference edge set: /root/reference/src/aligner/
// dfa.rs:210-250 and scoring/gap_affine.rs:307-430): for every row v (graph node in topological
// order) and query column j in [0, L], the three u32 min-plus planes
//
//   D[v][j] = min( PD[j] + e,  openD(v,j) ? PM[j] + o + e : INF )
//   H[v][j] = min( PM[j-1] + (sym(v) != q[j-1] ? x : 0),  D[v][j] )          (H[start][0] = 0)
//   I[v][j+1] = min( I[v][j] + e,  openI(v,j) ? H[v][j] + o + e : INF )       (I[v][0] = INF)
//   M[v][j] = min( H[v][j], I[v][j] )
//   end row:  D[j] = PD[j] + e ;  M[j] = min( PM[j], D[j] ) ;  I = INF
//
// with PM/PD the minima over the predecessors' M/D rows, openD(v,j) = (j >= L || sym(v) != q[j]),
// openI(v,j) = (j < L && (v -> end exists || some non-end child's symbol != q[j])).
// INF = 0xFFFFFFFF == Score::Unvisited (scoring/mod.rs:64-70); every add saturates
// (`v_add_u32 ... clamp`), so INF is absorbing and no overflow handling is needed.
//
// Mapping: ONE WAVEFRONT (64 lanes) PER QUERY, walking the rows in topological order over strips of
// up to 1024 columns ("quad-striped": lane l owns 4 consecutive columns in each 256-column quad, so
// every plane store is a contiguous 1 KiB per wave-instruction).  The previous row's M and D stay
// in registers (chains: predecessor == previous row); other predecessors are re-read from the score
// planes (L2/MALL hits, they were just written).  The diagonal term needs one cross-lane value per
// quad and row (DPP wave_shr:1); the insertion row is a min-plus prefix scan: in-lane chain +
// 6-step DPP scan per quad + uniform carry over the quads + in-lane fix-up.  No MFMA, no LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/poasta_amd.h"
#include "poa_graph.hpp"

namespace poa_amd {

constexpr uint32_t INF = 0xFFFFFFFFu;

struct FwdParams {
    const RowMeta* rows;        // [n_rows]
    const uint32_t* pred_rows;  // [n_edges]
    uint32_t n_rows;
    uint32_t first_query;       // first query of this chunk
    uint32_t n_queries;         // queries in this chunk
    const uint8_t* qseq;
    const uint64_t* qoff;       // [total+1]
    const uint32_t* pitch;      // [total] columns per plane row (multiple of 32)
    const uint64_t* plane_off;  // [total] element offset of the query's M plane in `planes`
    uint32_t* planes;           // workspace: per query [M | I | D], each n_rows * pitch
    uint32_t* strip_carry;      // [n_queries_in_chunk * n_rows] I carried between strips (long queries)
    uint32_t cost_x, cost_oe, cost_e;
};

struct TbParams {
    const RowMeta* rows;
    const uint32_t* pred_rows;
    uint32_t n_rows, start_row, end_row;
    uint32_t first_query, n_queries;
    const uint8_t* qseq;
    const uint64_t* qoff;
    const uint32_t* pitch;
    const uint64_t* plane_off;
    const uint32_t* planes;
    const uint64_t* scratch_off;  // [total+1] per-query region in `scratch` (capacity len + n_rows)
    uint2* scratch;               // pairs written from the BACK of each region
    uint32_t* score;              // [total]
    uint32_t* flags;              // [total]
    uint32_t* n_pairs;            // [total]
    uint32_t cost_x, cost_o, cost_e;
};

__device__ __forceinline__ uint32_t sat_add(uint32_t a, uint32_t b) {
    return __builtin_elementwise_add_sat(a, b);  // v_add_u32 ... clamp
}
__device__ __forceinline__ uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }

// lane l receives x from lane l-1; lane 0 receives `fill`  (v_mov_b32_dpp wave_shr:1)
__device__ __forceinline__ uint32_t wave_shr1(uint32_t x, uint32_t fill) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)x, 0x138, 0xF, 0xF, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_or_inf(uint32_t x) {
    // lanes without a source (or masked rows) receive INF
    return (uint32_t)__builtin_amdgcn_update_dpp((int)INF, (int)x, CTRL, ROW_MASK, 0xF, false);
}

// Inclusive min-plus scan over the 64 lanes: P(l) = min_{i<=l} ( t(i) + (l-i)*step ).
// DPP only (row_shr 1/2/4/8 inside 16-lane rows, then row_bcast:15 / row_bcast:31): no LDS pipe.
// w15 = ((l&15)+1)*step and w31 = (l-31)*step are per-lane constants.
__device__ __forceinline__ uint32_t wave_scan_min_plus(uint32_t t, uint32_t step, uint32_t w15, uint32_t w31) {
    uint32_t P = t;
    P = umin(P, sat_add(dpp_or_inf<0x111, 0xF>(P), step));
    P = umin(P, sat_add(dpp_or_inf<0x112, 0xF>(P), 2 * step));
    P = umin(P, sat_add(dpp_or_inf<0x114, 0xF>(P), 4 * step));
    P = umin(P, sat_add(dpp_or_inf<0x118, 0xF>(P), 8 * step));
    P = umin(P, sat_add(dpp_or_inf<0x142, 0xA>(P), w15));  // row_bcast:15 -> rows 1,3
    P = umin(P, sat_add(dpp_or_inf<0x143, 0xC>(P), w31));  // row_bcast:31 -> rows 2,3
    return P;
}

__device__ __forceinline__ uint32_t qbyte(uint32_t packed, int k) { return (packed >> (8 * k)) & 0xFFu; }

// ---------------------------------------------------------------------------------------------
// Forward pass.  One wave per query; 4 waves (queries) per 256-thread block.
//
// Layout ("quad-striped"): a strip is W = Q*256 columns; lane l owns, in each of the Q quads,
// the 4 consecutive columns  s*W + m*256 + 4*l + {0,1,2,3}.  Every global_load/store_dwordx4 of a
// quad therefore covers 1 KiB contiguous bytes per wave-instruction (8 full 128-B lines), the
// four quads' insertion scans are independent chains, and the column-(j-1) neighbour is an
// in-register value except for k = 0 (one DPP wave_shr:1 per quad).
template <int Q>
__global__ __launch_bounds__(256) void poa_forward_kernel(FwdParams P) {
    constexpr int C = 4 * Q;
    constexpr uint32_t W = 256 * Q;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wq = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;  // wave-uniform
    if (wq >= P.n_queries) return;
    const uint32_t qi = P.first_query + wq;
    const uint64_t qbeg = P.qoff[qi];
    const uint32_t L = (uint32_t)(P.qoff[qi + 1] - qbeg);
    const uint8_t* __restrict__ q = P.qseq + qbeg;
    const uint32_t pitch = P.pitch[qi];
    const uint64_t RP = (uint64_t)P.n_rows * pitch;
    uint32_t* __restrict__ Mp = P.planes + P.plane_off[qi];
    uint32_t* __restrict__ Ip = Mp + RP;
    uint32_t* __restrict__ Dp = Ip + RP;
    uint32_t* __restrict__ carry = P.strip_carry + (uint64_t)wq * P.n_rows;
    const uint32_t x = P.cost_x, oe = P.cost_oe, e = P.cost_e;
    const uint32_t n_strips = (pitch + W - 1) / W;
    const uint32_t step = 4 * e;                      // one lane == 4 columns
    const uint32_t w15 = ((lane & 15u) + 1u) * step;
    const uint32_t w31 = (lane - 31u) * step;         // only used by lanes >= 32
    const uint32_t lane_off = 4 * lane * e;           // cost of extending an insertion to my first column of a quad

    for (uint32_t s = 0; s < n_strips; ++s) {
        const uint32_t sbase = s * W;
        bool act[Q];          // my 4 columns of quad m lie inside the plane row
        uint32_t qcp[Q];      // my 4 query symbols of quad m, one per byte; 0 (never a symbol) past the end
        uint32_t ql[Q];       // query symbol left of my first column of quad m
#pragma unroll
        for (int m = 0; m < Q; ++m) {
            const uint32_t c0 = sbase + m * 256 + 4 * lane;
            act[m] = c0 < pitch;
            uint32_t pk = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) pk |= ((c0 + k < L) ? (uint32_t)q[c0 + k] : 0u) << (8 * k);
            qcp[m] = pk;
            ql[m] = (c0 > 0 && c0 - 1 < L) ? (uint32_t)q[c0 - 1] : 0u;
        }

        uint32_t Mprev[C], Dprev[C];
#pragma unroll
        for (int k = 0; k < C; ++k) { Mprev[k] = INF; Dprev[k] = INF; }

        for (uint32_t r = 0; r < P.n_rows; ++r) {
            const RowMeta meta = P.rows[r];
            const uint32_t sym = meta.sym;
            const uint64_t rbase = (uint64_t)r * pitch + sbase + 4 * lane;
            uint32_t PM[C], PD[C], PMl[Q];

            const bool chain = (meta.pred_count == 1) && (P.pred_rows[meta.pred_begin] + 1 == r);
            if (chain) {
                // fast path: the only predecessor is the previous row, still in registers
                uint32_t edge = INF;
                if (s > 0) edge = Mp[(uint64_t)(r - 1) * pitch + sbase - 1];  // uniform address
#pragma unroll
                for (int m = 0; m < Q; ++m) {
                    PMl[m] = wave_shr1(Mprev[4 * m + 3], edge);
                    edge = (uint32_t)__builtin_amdgcn_readlane((int)Mprev[4 * m + 3], 63);
                }
#pragma unroll
                for (int k = 0; k < C; ++k) { PM[k] = Mprev[k]; PD[k] = Dprev[k]; }
            } else {
#pragma unroll
                for (int k = 0; k < C; ++k) { PM[k] = INF; PD[k] = INF; }
#pragma unroll
                for (int m = 0; m < Q; ++m) PMl[m] = INF;
                // rows written earlier by this wave are re-read below by other lanes of the wave:
                // drain the stores first (write-through L1; the lines are then fetched from L2).
                if (meta.pred_count > 0) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                for (uint32_t pe = 0; pe < meta.pred_count; ++pe) {
                    const uint32_t pr = P.pred_rows[meta.pred_begin + pe];
                    const uint64_t pbase = (uint64_t)pr * pitch + sbase + 4 * lane;
                    uint32_t tm[C], td[C];
                    if (pr + 1 == r) {
#pragma unroll
                        for (int k = 0; k < C; ++k) { tm[k] = Mprev[k]; td[k] = Dprev[k]; }
                    } else {
#pragma unroll
                        for (int m = 0; m < Q; ++m) {
                            uint4 a = make_uint4(INF, INF, INF, INF), b = a;
                            if (act[m]) {
                                a = *reinterpret_cast<const uint4*>(Mp + pbase + m * 256);
                                b = *reinterpret_cast<const uint4*>(Dp + pbase + m * 256);
                            }
                            tm[4 * m] = a.x; tm[4 * m + 1] = a.y; tm[4 * m + 2] = a.z; tm[4 * m + 3] = a.w;
                            td[4 * m] = b.x; td[4 * m + 1] = b.y; td[4 * m + 2] = b.z; td[4 * m + 3] = b.w;
                        }
                    }
                    uint32_t edge = INF;
                    if (s > 0) edge = Mp[(uint64_t)pr * pitch + sbase - 1];
#pragma unroll
                    for (int m = 0; m < Q; ++m) {
                        PMl[m] = umin(PMl[m], wave_shr1(tm[4 * m + 3], edge));
                        edge = (uint32_t)__builtin_amdgcn_readlane((int)tm[4 * m + 3], 63);
                    }
#pragma unroll
                    for (int k = 0; k < C; ++k) { PM[k] = umin(PM[k], tm[k]); PD[k] = umin(PD[k], td[k]); }
                }
            }

            uint32_t Mc[C], Ic[C], Dc[C];
            if (meta.flags & ROW_END) {
#pragma unroll
                for (int k = 0; k < C; ++k) {
                    Dc[k] = sat_add(PD[k], e);
                    Mc[k] = umin(PM[k], Dc[k]);
                    Ic[k] = INF;
                }
            } else {
                const bool open_always = (meta.flags & ROW_OPENI_ALWAYS) != 0;
                const bool open_never = (meta.flags & ROW_OPENI_NEVER) != 0;
                const uint32_t csym = meta.child_sym;
                uint32_t H[C], T[Q];
#pragma unroll
                for (int m = 0; m < Q; ++m) {
                    uint32_t t = INF;  // in-lane insertion chain, carry-in INF
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int i = 4 * m + k;
                        const uint32_t qk = qbyte(qcp[m], k);
                        const uint32_t open = (qk != sym) ? sat_add(PM[i], oe) : INF;
                        Dc[i] = umin(sat_add(PD[i], e), open);
                        const uint32_t pm_left = (k == 0) ? PMl[m] : PM[i - 1];
                        const uint32_t q_left = (k == 0) ? ql[m] : qbyte(qcp[m], k - 1);
                        H[i] = umin(sat_add(pm_left, (q_left != sym) ? x : 0u), Dc[i]);
                        if (m == 0 && k == 0 && (meta.flags & ROW_START) && sbase == 0 && lane == 0) H[i] = 0;
                        Ic[i] = t;  // value entering column i from my own earlier columns (INF for k == 0)
                        const bool op = !open_never && (open_always || qk != csym);
                        t = umin(sat_add(t, e), op ? sat_add(H[i], oe) : INF);
                    }
                    T[m] = t;  // leaves my last column of quad m (carry-in INF)
                }
                // cross-lane: independent scans per quad, then a uniform carry chain over the quads
                uint32_t cq = (s > 0) ? carry[r] : INF;  // insertion value entering column sbase
#pragma unroll
                for (int m = 0; m < Q; ++m) {
                    const uint32_t Pm = wave_scan_min_plus(T[m], step, w15, w31);
                    const uint32_t excl = wave_shr1(Pm, INF);             // from earlier lanes of this quad
                    const uint32_t cin = umin(excl, sat_add(cq, lane_off));  // ... or from before the quad
                    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)Pm, 63);
                    cq = umin(sat_add(cq, 256 * e), total);
                    Ic[4 * m] = cin;
#pragma unroll
                    for (int k = 1; k < 4; ++k) Ic[4 * m + k] = umin(Ic[4 * m + k], sat_add(cin, (uint32_t)k * e));
                }
                if (n_strips > 1 && lane == 0) carry[r] = cq;  // I[r][(s+1)*W]
#pragma unroll
                for (int k = 0; k < C; ++k) Mc[k] = umin(H[k], Ic[k]);
            }

#pragma unroll
            for (int m = 0; m < Q; ++m) {
                if (act[m]) {
                    *reinterpret_cast<uint4*>(Mp + rbase + m * 256) = make_uint4(Mc[4 * m], Mc[4 * m + 1], Mc[4 * m + 2], Mc[4 * m + 3]);
                    *reinterpret_cast<uint4*>(Ip + rbase + m * 256) = make_uint4(Ic[4 * m], Ic[4 * m + 1], Ic[4 * m + 2], Ic[4 * m + 3]);
                    *reinterpret_cast<uint4*>(Dp + rbase + m * 256) = make_uint4(Dc[4 * m], Dc[4 * m + 1], Dc[4 * m + 2], Dc[4 * m + 3]);
                }
            }
#pragma unroll
            for (int k = 0; k < C; ++k) { Mprev[k] = Mc[k]; Dprev[k] = Dc[k]; }
        }
        if (n_strips > 1) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // carry[] and edge columns for the next strip
    }
}

// ---------------------------------------------------------------------------------------------
// Traceback: the reference's score-based rule (scoring/gap_affine.rs:550-657, :804-915) applied to
// the dense planes.  Every test of a step is evaluated so that the certificate "exactly one
// candidate, no phantom below target" can be decided.  One thread per query.
struct TbCtx {
    const RowMeta* rows;
    const uint32_t* pred_rows;
    const uint32_t* M;
    const uint32_t* I;
    const uint32_t* D;
    const uint8_t* q;
    uint32_t L, pitch, start_row, end_row;
    uint32_t x, o, e;
};

struct TbStep {
    uint32_t row, j, st;  // st: 0 M, 1 D, 2 I
    bool found;
};

__device__ __forceinline__ uint32_t pl(const uint32_t* p, uint32_t pitch, uint32_t row, uint32_t j) {
    return p[(uint64_t)row * pitch + j];
}

__device__ inline bool tb_open_i(const TbCtx& c, const RowMeta& m, uint32_t j) {
    if (j >= c.L) return false;
    if (m.flags & ROW_OPENI_ALWAYS) return true;
    if (m.flags & ROW_OPENI_NEVER) return false;
    return (uint32_t)m.child_sym != (uint32_t)c.q[j];
}

__device__ inline TbStep tb_step(const TbCtx& c, uint32_t row, uint32_t j, uint32_t st, uint32_t& n_cand,
                                 bool& bad, bool& panic) {
    TbStep first{0, 0, 0, false};
    n_cand = 0;
    const RowMeta m = c.rows[row];
    const bool is_end = (m.flags & ROW_END) != 0;
    auto sub = [&](uint32_t a, uint32_t b) { uint32_t r = a - b; if (r == INF) panic = true; return r; };
    auto cand = [&](uint32_t r2, uint32_t j2, uint32_t s2) {
        if (!first.found) first = TbStep{r2, j2, s2, true};
        n_cand++;
    };
    if (st == 0) {
        const uint32_t cs = pl(c.M, c.pitch, row, j);
        if (cs == INF) return first;
        if (j > 0) {
            const bool moe = is_end || ((uint32_t)m.sym == (uint32_t)c.q[j - 1]);
            const uint32_t pj = is_end ? j : j - 1;
            const uint32_t target = moe ? cs : sub(cs, c.x);
            for (uint32_t pe = 0; pe < m.pred_count; ++pe) {
                const uint32_t pr = c.pred_rows[m.pred_begin + pe];
                if (pl(c.M, c.pitch, pr, pj) == target) cand(pr, pj, 0);
            }
        }
        if (pl(c.D, c.pitch, row, j) == cs) cand(row, j, 1);
        if (pl(c.I, c.pitch, row, j) == cs) cand(row, j, 2);
    } else if (st == 1) {
        const uint32_t cs = pl(c.D, c.pitch, row, j);
        if (cs == INF) return first;
        if (m.pred_count == 0) return first;
        const uint32_t t_open = sub(sub(cs, c.o), c.e), t_ext = sub(cs, c.e);
        const bool real_open = !is_end && (j >= c.L || (uint32_t)m.sym != (uint32_t)c.q[j]);
        for (uint32_t pe = 0; pe < m.pred_count; ++pe) {
            const uint32_t pr = c.pred_rows[m.pred_begin + pe];
            const uint32_t ps = pl(c.M, c.pitch, pr, j);
            if (ps == t_open) cand(pr, j, 0);
            else if (!real_open && ps < t_open) bad = true;  // phantom edge the reference does not re-check
        }
        for (uint32_t pe = 0; pe < m.pred_count; ++pe) {
            const uint32_t pr = c.pred_rows[m.pred_begin + pe];
            if (pl(c.D, c.pitch, pr, j) == t_ext) cand(pr, j, 1);
        }
    } else {
        const uint32_t cs = pl(c.I, c.pitch, row, j);
        if (cs == INF) return first;
        if (j > 0) {
            const uint32_t t_open = sub(sub(cs, c.o), c.e), t_ext = sub(cs, c.e);
            const uint32_t pm = pl(c.M, c.pitch, row, j - 1);
            const uint32_t pi = pl(c.I, c.pitch, row, j - 1);
            if (pm == t_open) cand(row, j - 1, 0);
            else if (!tb_open_i(c, m, j - 1) && pm < t_open) bad = true;
            if (pi == t_ext) {
                const bool only = (n_cand == 0);
                cand(row, j - 1, 0);  // sic: the reference returns Match here (gap_affine.rs:649)
                if (only && pm != pi) bad = true;  // the hop lands on M[row][j-1] which is not this I value
            }
        }
    }
    return first;
}

__global__ __launch_bounds__(64) void poa_traceback_kernel(TbParams P) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= P.n_queries) return;
    const uint32_t qi = P.first_query + t;
    const uint64_t qbeg = P.qoff[qi];
    TbCtx c;
    c.rows = P.rows; c.pred_rows = P.pred_rows;
    c.L = (uint32_t)(P.qoff[qi + 1] - qbeg);
    c.q = P.qseq + qbeg;
    c.pitch = P.pitch[qi];
    const uint64_t RP = (uint64_t)P.n_rows * c.pitch;
    c.M = P.planes + P.plane_off[qi];
    c.I = c.M + RP;
    c.D = c.I + RP;
    c.start_row = P.start_row; c.end_row = P.end_row;
    c.x = P.cost_x; c.o = P.cost_o; c.e = P.cost_e;
    const uint32_t L = c.L;

    uint2* out = P.scratch + P.scratch_off[qi];
    const uint32_t cap = (uint32_t)(P.scratch_off[qi + 1] - P.scratch_off[qi]);
    uint32_t cnt = 0;
    uint32_t flags = 0;
    auto emit = [&](uint32_t rpos, uint32_t qpos) {
        if (cnt < cap) out[cap - 1 - cnt] = make_uint2(rpos, qpos);
        cnt++;
    };

    P.score[qi] = pl(c.M, c.pitch, c.end_row, L);
    const uint32_t end_node = c.rows[c.end_row].node;

    bool done = false;
    if (L == 0) done = true;
    if (!done && L == 1) {
        // gap_affine.rs:812-824: the end node equals every symbol -> always [(end, 0)]
        flags |= POA_FLAG_SHORT_QUERY;
        emit(end_node, 0);
        done = true;
    }
    if (!done) {
        uint32_t nc; bool bad = false, pn = false;
        TbStep cur = tb_step(c, c.end_row, L, 0, nc, bad, pn);
        if (pn) flags |= POA_FLAG_REF_PANIC;
        if (cur.found && (nc != 1 || bad)) flags |= POA_FLAG_AMBIGUOUS;
        if (!cur.found) {
            // .or_else(Insertion).or_else(Deletion), gap_affine.rs:832-835
            cur = tb_step(c, c.end_row, L, 2, nc, bad, pn);
            if (!cur.found) cur = tb_step(c, c.end_row, L, 1, nc, bad, pn);
            if (!cur.found) {
                flags |= POA_FLAG_REF_PANIC;
                if (L <= 3) for (uint32_t i = 0; i < L; ++i) emit(end_node, L - 1 - i);
                done = true;
            } else {
                flags |= POA_FLAG_AMBIGUOUS;
            }
        }
        if (!done) {
            uint32_t crow = cur.row, cj = cur.j, cst = cur.st;
            bool reached_start = false;
            for (;;) {
                bad = false; pn = false;
                const TbStep bt = tb_step(c, crow, cj, cst, nc, bad, pn);
                if (pn) flags |= POA_FLAG_REF_PANIC;
                if (!bt.found) break;
                if (nc != 1 || bad) flags |= POA_FLAG_AMBIGUOUS;
                if (cst == 0 && bt.st != 0) { crow = bt.row; cj = bt.j; cst = bt.st; continue; }
                const uint32_t node = c.rows[crow].node;
                if (cst == 0) emit(node, cj - 1);
                else if (cst == 2) emit(POA_NONE, cj - 1);
                else emit(node, POA_NONE);
                // start-quirk certificate (dfa.rs:146-167): this step used an out-edge of
                // (bt.row, 0, M) whose node symbol equals q[0]
                if (bt.st == 0 && bt.j == 0 && bt.row != c.start_row && cst != 1 &&
                    (uint32_t)c.rows[bt.row].sym == (uint32_t)c.q[0])
                    flags |= POA_FLAG_START_QUIRK;
                if (bt.row == c.start_row) { reached_start = true; break; }
                crow = bt.row; cj = bt.j; cst = bt.st;
            }
            if (!reached_start) flags |= POA_FLAG_TRUNCATED;
        }
    }
    P.flags[qi] = flags;
    P.n_pairs[qi] = cnt < cap ? cnt : cap;
}

// exclusive prefix sum of n_pairs -> pair_off[n+1]; single block.
__global__ __launch_bounds__(1024) void poa_scan_kernel(const uint32_t* __restrict__ n_pairs, uint64_t* __restrict__ pair_off,
                                                        uint32_t n) {
    __shared__ uint64_t part[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (n + 1023) / 1024;
    const uint32_t b = t * per, e = (b + per < n) ? b + per : n;
    uint64_t s = 0;
    for (uint32_t i = b; i < e; ++i) s += n_pairs[i];
    part[t] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint64_t v = (t >= d) ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint64_t run = part[t] - s;
    for (uint32_t i = b; i < e; ++i) { pair_off[i] = run; run += n_pairs[i]; }
    if (t == 1023) pair_off[n] = part[1023];
}

// copy each query's pairs (written from the back of its scratch region) to its compact slot
__global__ __launch_bounds__(256) void poa_compact_kernel(const uint2* __restrict__ scratch,
                                                          const uint64_t* __restrict__ scratch_off,
                                                          const uint32_t* __restrict__ n_pairs,
                                                          const uint64_t* __restrict__ pair_off, uint2* __restrict__ pairs,
                                                          uint32_t n) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (w >= n) return;
    const uint32_t k = n_pairs[w];
    const uint64_t cap = scratch_off[w + 1] - scratch_off[w];
    const uint2* src = scratch + scratch_off[w] + (cap - k);
    uint2* dst = pairs + pair_off[w];
    for (uint32_t i = lane; i < k; i += 64) dst[i] = src[i];
}

}  // namespace poa_amd
