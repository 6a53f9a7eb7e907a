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
// Mapping: ONE WAVEFRONT (64 lanes) PER QUERY.  Lane l owns C consecutive columns of a strip of
// W = 64*C columns; the wave walks the rows in topological order.  The previous row's M and D
// stay in registers (chains: predecessor == previous row); other predecessors are re-read from the
// score planes (L2/MALL hits, they were just written).  The diagonal term needs one cross-lane
// value per row (DPP wave_shr:1); the insertion row is a min-plus prefix scan: in-lane serial pass
// + 6-step cross-lane scan + in-lane fix-up.  No MFMA: integer min/add only.
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

template <int C>
__device__ __forceinline__ void load_row(const uint32_t* __restrict__ p, uint32_t (&v)[C]) {
    static_assert(C % 4 == 0, "C must be a multiple of 4");
#pragma unroll
    for (int k = 0; k < C; k += 4) {
        uint4 t = *reinterpret_cast<const uint4*>(p + k);
        v[k] = t.x; v[k + 1] = t.y; v[k + 2] = t.z; v[k + 3] = t.w;
    }
}
template <int C>
__device__ __forceinline__ void store_row(uint32_t* __restrict__ p, const uint32_t (&v)[C]) {
#pragma unroll
    for (int k = 0; k < C; k += 4) {
        *reinterpret_cast<uint4*>(p + k) = make_uint4(v[k], v[k + 1], v[k + 2], v[k + 3]);
    }
}

// ---------------------------------------------------------------------------------------------
// Forward pass.  grid: one wave per query of the chunk, 4 waves (queries) per 256-thread block.
template <int C>
__global__ __launch_bounds__(256) void poa_forward_kernel(FwdParams P) {
    constexpr uint32_t W = 64 * C;
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

    for (uint32_t s = 0; s < n_strips; ++s) {
        const uint32_t col0 = s * W + lane * C;
        const bool active = col0 < pitch;  // pitch is a multiple of 32 >= C: a lane is all in or all out
        // query symbols of my columns; 0xFFFF (never a symbol) beyond the query end, so that
        // "mismatch" holds there: openD(v, j >= L) is true, as the recurrence wants.
        uint32_t qc[C];
#pragma unroll
        for (int k = 0; k < C; ++k) qc[k] = (col0 + k < L) ? (uint32_t)q[col0 + k] : 0xFFFFu;
        const uint32_t qleft = (col0 > 0 && col0 - 1 < L) ? (uint32_t)q[col0 - 1] : 0xFFFFu;

        uint32_t Mprev[C], Dprev[C];
#pragma unroll
        for (int k = 0; k < C; ++k) { Mprev[k] = INF; Dprev[k] = INF; }

        for (uint32_t r = 0; r < P.n_rows; ++r) {
            const RowMeta meta = P.rows[r];
            const uint32_t sym = meta.sym;
            uint32_t PM[C], PD[C];
            uint32_t PMl = INF;  // min over predecessors of M[p][col0 - 1]
#pragma unroll
            for (int k = 0; k < C; ++k) { PM[k] = INF; PD[k] = INF; }

            bool need_fence = false;
            for (uint32_t pe = 0; pe < meta.pred_count; ++pe)
                need_fence |= (P.pred_rows[meta.pred_begin + pe] + 1 != r);
            // rows written earlier by this wave are re-read below by OTHER lanes of the wave:
            // drain the stores first (the L1 is write-through; lines are fetched from L2 afterwards).
            if (need_fence || s > 0) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");

            for (uint32_t pe = 0; pe < meta.pred_count; ++pe) {
                const uint32_t pr = P.pred_rows[meta.pred_begin + pe];
                const uint64_t ro = (uint64_t)pr * pitch;
                if (pr + 1 == r) {
                    // chain: predecessor is the previous row, still in registers
                    uint32_t edge = INF;
                    if (s > 0 && lane == 0) edge = Mp[ro + col0 - 1];
                    const uint32_t left = wave_shr1(Mprev[C - 1], edge);
                    PMl = umin(PMl, left);
#pragma unroll
                    for (int k = 0; k < C; ++k) { PM[k] = umin(PM[k], Mprev[k]); PD[k] = umin(PD[k], Dprev[k]); }
                } else if (active) {
                    uint32_t tm[C], td[C];
                    load_row<C>(Mp + ro + col0, tm);
                    load_row<C>(Dp + ro + col0, td);
                    const uint32_t left = col0 > 0 ? Mp[ro + col0 - 1] : INF;
                    PMl = umin(PMl, left);
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
                uint32_t H[C];
                // D and H
#pragma unroll
                for (int k = 0; k < C; ++k) {
                    const uint32_t open = (qc[k] != sym) ? sat_add(PM[k], oe) : INF;
                    Dc[k] = umin(sat_add(PD[k], e), open);
                    const uint32_t pm_left = (k == 0) ? PMl : PM[k - 1];
                    const uint32_t q_left = (k == 0) ? qleft : qc[k - 1];
                    const uint32_t diag = sat_add(pm_left, (q_left != sym) ? x : 0u);
                    H[k] = umin(diag, Dc[k]);
                }
                if ((meta.flags & ROW_START) && col0 == 0) H[0] = 0;
                // insertion row: I[j+1] = min(I[j] + e, A[j]),  A[j] = openI ? H[j] + oe : INF
                // in-lane pass with carry-in INF
                uint32_t t = INF;
                Ic[0] = INF;
#pragma unroll
                for (int k = 0; k < C; ++k) {
                    const bool op = !open_never && (open_always || qc[k] != csym);
                    const uint32_t a = op ? sat_add(H[k], oe) : INF;
                    t = umin(sat_add(t, e), a);
                    if (k + 1 < C) Ic[k + 1] = t;
                }
                // cross-lane: carry(l+1) = min(carry(l) + C*e, t(l))
                const uint32_t c0 = (s > 0) ? carry[r] : INF;  // I[r][s*W], uniform load
                uint32_t Pv = t;
                if (lane == 0) Pv = umin(Pv, sat_add(c0, C * e));
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t up = __shfl_up(Pv, d);
                    if (lane >= (uint32_t)d) Pv = umin(Pv, sat_add(up, (uint32_t)d * C * e));
                }
                uint32_t cin = __shfl_up(Pv, 1);
                if (lane == 0) cin = c0;
                if (n_strips > 1 && lane == 63) carry[r] = Pv;  // I[r][(s+1)*W] for the next strip
                Ic[0] = cin;
#pragma unroll
                for (int k = 1; k < C; ++k) Ic[k] = umin(Ic[k], sat_add(cin, (uint32_t)k * e));
#pragma unroll
                for (int k = 0; k < C; ++k) Mc[k] = umin(H[k], Ic[k]);
            }

            if (active) {
                const uint64_t ro = (uint64_t)r * pitch + col0;
                store_row<C>(Mp + ro, Mc);
                store_row<C>(Ip + ro, Ic);
                store_row<C>(Dp + ro, Dc);
            }
#pragma unroll
            for (int k = 0; k < C; ++k) { Mprev[k] = Mc[k]; Dprev[k] = Dc[k]; }
        }
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
