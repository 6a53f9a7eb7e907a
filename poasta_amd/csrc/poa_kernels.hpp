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

#ifndef POA_FWD_MIN_WAVES
#define POA_FWD_MIN_WAVES 1
#endif

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
    const uint32_t* pitch;      // [total] columns per plane row (multiple of 64)
    const uint64_t* plane_off;  // [total] element offset of the query's M plane in `planes`
    uint32_t* planes;           // workspace: per query [M | I | D], each n_rows * pitch
    uint32_t* strip_carry;      // [n_queries_in_chunk * n_rows] I carried between strips (long queries)
    uint32_t cost_x, cost_oe, cost_e;
    // pairs-across-quads kernels: deletion / insertion costs apart, and e * pred_k[edge] added to every predecessor value
    // (relative encoding, FlatGraph::row_depth); absolute encoding: cost_de = cost_ie = cost_e, cost_doe = cost_ioe = cost_oe,
    // pred_k = nullptr.  The end row's deletion always costs cost_e.
    uint32_t cost_de, cost_doe, cost_ie, cost_ioe;
    const uint32_t* pred_k;     // [n_edges] or nullptr
    // compact layout (FlatGraph::d_slot / pred_dslot): where the kept D rows live; nullptr: stored by row
    const uint32_t* d_slot;     // [n_rows]
    const uint32_t* pred_dslot; // [n_edges]
    uint32_t* pipeline_error;   // one word: set when a wave of the multi-wave pipeline gave up waiting (a protocol bug, not an input)
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
    uint32_t spec_depth;          // lanes that speculate per traceback round (1..64)
    // exact-replay pass: the planes hold the table the replayed search filled; only queries whose
    // replay succeeded are (re)traced, the others keep their dense result and get a flag.
    uint32_t exact_pass;
    const uint32_t* ex_status;    // [total] 0 ok, 1 reference panic, 2 workspace overflow, 0xFFFFFFFF not replayed
    const uint32_t* ex_end;       // [2 * total] (row, offset) of the end cell the replayed search stopped at
    uint32_t code_fmt;            // compact layout: 0 = one nibble per cell, 1 = bit-planes (poa_forward_px_kernel<0>, poa_forward_pxmw_kernel),
                                  // 2 = flags A, C in bits 14, 15 of the stored M value, B, D as bit-planes (poa_forward_px_kernel<1>)
                                  // 3 = flags B, D, A, C in bits 12..15 of the stored M value, no flag words (poa_forward_px_kernel<2>)
    const uint32_t* row_depth;    // relative encoding: stored value = score - e * (row_depth[row] - column); nullptr: absolute
    const uint32_t* d_slot;       // compact layout: slot of a row's kept D row (FlatGraph::d_slot / pred_dslot)
    const uint32_t* pred_dslot;
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
// Plane element types.  u32 planes hold Score values verbatim (INF = 0xFFFFFFFF).  u16 planes are
// used when every finite score of the batch provably fits ( (rows + L + 2) * max(x, o+e) <= 65534 ):
// INF is stored as 0xFFFF and restored on load; arithmetic is always done on u32 registers.
template <typename T> struct PlaneIO;
template <> struct PlaneIO<uint32_t> {
    static constexpr int K = 4;  // columns per lane per quad == one 16-byte access
    static __device__ __forceinline__ uint32_t get(const uint32_t* p) { return *p; }
    static __device__ __forceinline__ void load(const uint32_t* p, uint32_t (&v)[4]) {
        const uint4 t = *reinterpret_cast<const uint4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void store(uint32_t* p, const uint32_t* v) {
        *reinterpret_cast<uint4*>(p) = make_uint4(v[0], v[1], v[2], v[3]);
    }
};
template <> struct PlaneIO<uint16_t> {
    static constexpr int K = 8;
    static __device__ __forceinline__ uint32_t widen(uint32_t h) { return h == 0xFFFFu ? INF : h; }
    static __device__ __forceinline__ uint32_t get(const uint16_t* p) { return widen(*p); }
    static __device__ __forceinline__ void load(const uint16_t* p, uint32_t (&v)[8]) {
        const uint4 t = *reinterpret_cast<const uint4*>(p);
        const uint32_t d[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = widen(d[i] & 0xFFFFu); v[2 * i + 1] = widen(d[i] >> 16); }
    }
    static __device__ __forceinline__ void store(uint16_t* p, const uint32_t* v) {
        uint32_t d[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) d[i] = umin(v[2 * i], 0xFFFFu) | (umin(v[2 * i + 1], 0xFFFFu) << 16);
        *reinterpret_cast<uint4*>(p) = make_uint4(d[0], d[1], d[2], d[3]);
    }
};

// ---------------------------------------------------------------------------------------------
// Traceback: the reference's score-based rule (scoring/gap_affine.rs:550-657, :804-915) applied to
// the dense planes.  Every test of a step is evaluated so that the certificate "exactly one
// candidate, no phantom below target" can be decided.  One thread per query.
template <typename T>
struct TbCtx {
    const RowMeta* rows;
    const uint32_t* pred_rows;
    const T* M;
    const T* I;
    const T* D;
    const uint32_t* codes;  // compact layout: 4 flag bits per cell at the I plane's place
    uint32_t code_fmt;      // 0: nibble per cell, 1: bit-planes, 2: A, C in the M value + B, D bit-planes
    uint32_t tiled;         // 1: the table of the replayed search (u32, ex_cell_index layout) at M
    uint32_t n_rows;
    const uint8_t* q;
    uint32_t L, pitch, start_row, end_row;
    uint32_t x, o, e;
    const uint32_t* row_depth;  // relative encoding (TbParams::row_depth) or nullptr
    const uint32_t* d_slot;     // compact layout: c.D holds only the kept D rows, row r at slot d_slot[r]
    const uint32_t* pred_dslot;
};

struct TbStep {
    uint32_t row, j, st;  // st: 0 M, 1 D, 2 I
    bool found;
    uint32_t cs;          // score of the cell the step started from
    uint32_t node;        // node of the row the step started from (for the emitted pair: no second, dependent load)
};

template <typename T>
__device__ __forceinline__ uint32_t pl(const T* p, uint32_t pitch, uint32_t row, uint32_t j) {
    return PlaneIO<T>::get(p + (uint64_t)row * pitch + j);
}

template <typename T>
__device__ inline bool tb_open_i(const TbCtx<T>& c, const RowMeta& m, uint32_t j) {
    if (j >= c.L) return false;
    if (m.flags & ROW_OPENI_ALWAYS) return true;
    if (m.flags & ROW_OPENI_NEVER) return false;
    return (uint32_t)m.child_sym != (uint32_t)c.q[j];
}

// code_fmt 2: a stored M value carries two flags — bit 14: I == M, bit 15: D == M — over a 14-bit score (0x3FFF = INF)
constexpr uint32_t MF_MASK = 0x3FFFu;
// code_fmt 3: four flags — bit 12: I[j] == I[j-1] + e, bit 13: D == PD + e, bit 14: I == M, bit 15: D == M — over a 12-bit score
constexpr uint32_t MF4_MASK = 0x0FFFu;
__device__ __forceinline__ uint32_t mf4_value(uint32_t raw) { const uint32_t v = raw & MF4_MASK; return v == MF4_MASK ? INF : v; }
// dwords of [B, D] flag bit-planes per row: one per lane that owns columns of the row (a short row has fewer than 64)
__host__ __device__ __forceinline__ uint32_t mf_code_stride(uint32_t pitch) { return pitch / 8 < 64u ? pitch / 8 : 64u; }
__device__ __forceinline__ uint32_t mf_value(uint32_t raw) { const uint32_t v = raw & MF_MASK; return v == MF_MASK ? INF : v; }
// the score of cell (row, j) of the M plane, whatever the format
template <typename T>
__device__ __forceinline__ uint32_t pl_tiled(const TbCtx<T>& c, uint32_t row, uint32_t j, uint32_t st) {
    return reinterpret_cast<const uint32_t*>(c.M)[ex_cell_index(row, j, st, c.n_rows, c.pitch)];  // st: 0 M, 1 D, 2 I
}
// relative encoding -> the score itself: + e * (row_depth[row] - column), in wrapping u32 arithmetic (the sum is >= 0)
template <typename T>
__device__ __forceinline__ uint32_t tb_abs(const TbCtx<T>& c, uint32_t v, uint32_t depth, uint32_t j) {
    return v == INF ? INF : v + c.e * depth - c.e * j;
}
template <typename T>
__device__ __forceinline__ uint32_t plM(const TbCtx<T>& c, uint32_t row, uint32_t j) {
    if (c.tiled) return pl_tiled(c, row, j, 0);
    if (c.code_fmt == 2) return mf_value((uint32_t)c.M[(uint64_t)row * c.pitch + j]);
    if (c.code_fmt == 3) return mf4_value((uint32_t)c.M[(uint64_t)row * c.pitch + j]);
    if (c.row_depth) {
        const uint32_t depth = c.row_depth[row];  // independent of the plane load: one round trip for both
        return tb_abs(c, PlaneIO<T>::get(c.M + (uint64_t)row * c.pitch + j), depth, j);
    }
    return PlaneIO<T>::get(c.M + (uint64_t)row * c.pitch + j);
}
template <typename T>
__device__ __forceinline__ uint32_t plD(const TbCtx<T>& c, uint32_t row, uint32_t j) {
    if (c.tiled) return pl_tiled(c, row, j, 1);
    if (c.row_depth) {
        const uint32_t depth = c.row_depth[row];
        return tb_abs(c, PlaneIO<T>::get(c.D + (uint64_t)row * c.pitch + j), depth, j);
    }
    return PlaneIO<T>::get(c.D + (uint64_t)row * c.pitch + j);
}
// compact layout: D of the kept row stored at `slot` (the row itself only enters the relative encoding)
template <typename T>
__device__ __forceinline__ uint32_t plDslot(const TbCtx<T>& c, uint32_t slot, uint32_t row, uint32_t j) {
    const uint32_t v = PlaneIO<T>::get(c.D + (uint64_t)slot * c.pitch + j);
    return c.row_depth ? tb_abs(c, v, c.row_depth[row], j) : v;
}
template <typename T>
__device__ __forceinline__ uint32_t plI(const TbCtx<T>& c, uint32_t row, uint32_t j) {
    return c.tiled ? pl_tiled(c, row, j, 2) : PlaneIO<T>::get(c.I + (uint64_t)row * c.pitch + j);
}

template <typename T>
__device__ __forceinline__ uint32_t tb_code(const TbCtx<T>& c, uint32_t row, uint32_t j) {
    if (c.code_fmt == 3) {  // all four flags in the cell's M value
        const uint32_t raw = (uint32_t)c.M[(uint64_t)row * c.pitch + j];
        return ((raw >> 14) & 1u) | (((raw >> 12) & 1u) << 1) | (((raw >> 15) & 1u) << 2) | (((raw >> 13) & 1u) << 3);
    }
    if (c.code_fmt == 2) {  // one dword per lane and row: bytes [B quad 0, D quad 0, B quad 1, D quad 1], bit k = column 8l + k
        const uint32_t w = c.codes[(uint64_t)row * mf_code_stride(c.pitch) + ((j & 511u) >> 3)];
        const uint32_t sh = (j >> 9) * 16u + (j & 7u);
        return (((w >> sh) & 1u) << 1) | (((w >> (sh + 8u)) & 1u) << 3);
    }
    // 8 cells per dword; the nibble of column k = j & 7 sits at position (k >> 1) + 4 * (k & 1)
    const uint32_t k = j & 7u;
    const uint32_t w = c.codes[(uint64_t)row * (c.pitch / 8) + (j >> 3)];
    if (c.code_fmt == 1) {  // bit-planes: byte b holds flag bit b of the 8 cells, bit k = column k
        const uint32_t v = w >> k;
        return (v & 1u) | ((v >> 7) & 2u) | ((v >> 14) & 4u) | ((v >> 21) & 8u);
    }
    return (w >> (4 * ((k >> 1) + 4 * (k & 1)))) & 0xFu;
}
// gap_cs: the score of the current D / I cell, carried along the walk (compact layout has no I plane and
// only some D rows; with full planes it equals the stored value and the stored value is used).
template <typename T, bool COMPACT>
__device__ inline TbStep tb_step(const TbCtx<T>& c, uint32_t row, uint32_t j, uint32_t st, uint32_t gap_cs, uint32_t& n_cand,
                                 bool& bad, bool& panic) {
    TbStep first{0, 0, 0, false, 0, 0};
    n_cand = 0;
    const RowMeta m = c.rows[row];
    first.node = m.node;
    const bool is_end = (m.flags & ROW_END) != 0;
    auto sub = [&](uint32_t a, uint32_t b) { uint32_t r = a - b; if (r == INF) panic = true; return r; };
    auto cand = [&](uint32_t r2, uint32_t j2, uint32_t s2) {
        if (!first.found) { first.row = r2; first.j = j2; first.st = s2; first.found = true; }
        n_cand++;
    };
    if (st == 0) {
        // all loads of the step are issued before the first use (one memory round-trip for chain rows)
        uint32_t cs, dv = 0, iv = 0, code = 0;
        if (COMPACT && c.code_fmt >= 2) {
            // the two flags a Match-state step needs travel with the score: no code load on the common path
            const uint32_t raw = (uint32_t)c.M[(uint64_t)row * c.pitch + j];
            cs = c.code_fmt == 3 ? mf4_value(raw) : mf_value(raw);
            code = ((raw >> 14) & 1u) | (((raw >> 15) & 1u) << 2);
        } else {
            cs = plM(c, row, j);
            if (COMPACT) code = tb_code(c, row, j);
            else { dv = plD(c, row, j); iv = plI(c, row, j); }
        }
        const uint32_t up = (row > 0 && j > 0) ? plM(c, row - 1, j - 1) : INF;  // the usual diagonal predecessor
        first.cs = cs;
        if (cs == INF) return first;
        if (j > 0) {
            const bool moe = is_end || ((uint32_t)m.sym == (uint32_t)c.q[j - 1]);
            const uint32_t pj = is_end ? j : j - 1;
            // the reference evaluates `curr_score - mismatch` per predecessor: never for a row without predecessors
            const uint32_t target = (moe || m.pred_count == 0) ? cs : sub(cs, c.x);
            if ((m.flags & ROW_CHAIN) && !is_end) {  // the end row reads its predecessor at the SAME column
                if (up == target) cand(row - 1, pj, 0);
            } else {
                // predecessors four at a time: the row indices, then their scores, then the tests in trait order
                for (uint32_t pe0 = 0; pe0 < m.pred_count; pe0 += 4) {
                    uint32_t prs[4], vals[4];
#pragma unroll
                    for (int b = 0; b < 4; ++b) prs[b] = (pe0 + b < m.pred_count) ? c.pred_rows[m.pred_begin + pe0 + b] : 0u;
#pragma unroll
                    for (int b = 0; b < 4; ++b) vals[b] = (pe0 + b < m.pred_count) ? plM(c, prs[b], pj) : INF;
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        if (pe0 + b < m.pred_count && vals[b] == target) cand(prs[b], pj, 0);
                }
            }
        }
        if (COMPACT ? (code & 4u) != 0 : dv == cs) cand(row, j, 1);
        if (COMPACT ? (code & 1u) != 0 : iv == cs) cand(row, j, 2);
    } else if (st == 1) {
        const uint32_t cs = COMPACT ? gap_cs : plD(c, row, j);
        first.cs = cs;
        if (cs == INF) return first;
        if (m.pred_count == 0) return first;
        const uint32_t t_open = sub(sub(cs, c.o), c.e), t_ext = sub(cs, c.e);
        const bool real_open = !is_end && (j >= c.L || (uint32_t)m.sym != (uint32_t)c.q[j]);
        for (uint32_t pe0 = 0; pe0 < m.pred_count; pe0 += 4) {
            uint32_t prs[4], vals[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) prs[b] = (pe0 + b < m.pred_count) ? c.pred_rows[m.pred_begin + pe0 + b] : 0u;
#pragma unroll
            for (int b = 0; b < 4; ++b) vals[b] = (pe0 + b < m.pred_count) ? plM(c, prs[b], j) : INF;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (pe0 + b >= m.pred_count) continue;
                if (vals[b] == t_open) cand(prs[b], j, 0);
                else if (!real_open && vals[b] < t_open) bad = true;  // phantom edge the reference does not re-check
            }
        }
        if (COMPACT && (m.flags & ROW_CHAIN)) {
            // single predecessor right above: D[row][j] == D[row-1][j] + e is code bit 3
            if (tb_code(c, row, j) & 8u) cand(row - 1, j, 1);
        } else {
            for (uint32_t pe0 = 0; pe0 < m.pred_count; pe0 += 4) {
                uint32_t prs[4], vals[4], slots[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) prs[b] = (pe0 + b < m.pred_count) ? c.pred_rows[m.pred_begin + pe0 + b] : 0u;
#pragma unroll
                for (int b = 0; b < 4; ++b) slots[b] = (COMPACT && c.pred_dslot && pe0 + b < m.pred_count) ? c.pred_dslot[m.pred_begin + pe0 + b] : prs[b];
#pragma unroll
                for (int b = 0; b < 4; ++b)  // predecessors of a non-chain row keep their D row
                    vals[b] = (pe0 + b < m.pred_count) ? (COMPACT ? plDslot(c, slots[b], prs[b], j) : plD(c, prs[b], j)) : INF;
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if (pe0 + b < m.pred_count && vals[b] == t_ext) cand(prs[b], j, 1);
            }
        }
    } else {
        const uint32_t cs = COMPACT ? gap_cs : plI(c, row, j);
        first.cs = cs;
        if (cs == INF) return first;
        if (j > 0) {
            const uint32_t t_open = sub(sub(cs, c.o), c.e), t_ext = sub(cs, c.e);
            const uint32_t pm = plM(c, row, j - 1);
            if (pm == t_open) cand(row, j - 1, 0);
            else if (!tb_open_i(c, m, j - 1) && pm < t_open) bad = true;
            const bool ext = COMPACT ? (tb_code(c, row, j) & 2u) != 0 : plI(c, row, j - 1) == t_ext;
            if (ext) {
                const bool only = (n_cand == 0);
                cand(row, j - 1, 0);  // sic: the reference returns Match here (gap_affine.rs:649)
                if (only && pm != t_ext) bad = true;  // the hop lands on M[row][j-1] which is not that I value
            }
        }
    }
    return first;
}

// One WAVE per query.  A traceback is a chain of dependent reads (~2 us each from HBM): to cut the
// chain, lane i speculatively evaluates the step at cell (row - i, j - i) — where the path is if
// the previous i steps were all (mis)match steps to the previous row — and the longest prefix of
// lanes whose step really is that diagonal move is accepted at once.  The first lane that deviates
// (gap open/close, bubble predecessor, start reached) is then handled exactly like the sequential
// rule, so the emitted alignment and flags are identical to a step-by-step walk.
// GW lanes per query: 64 (one wave per query) or 16 (four queries per wave — the walk is a chain of dependent memory
// round-trips, so what sets the throughput is how many walks are in flight; `lane` is the lane within the group).
template <typename T, bool COMPACT, int GW = 64>
__device__ __forceinline__ void traceback_wave(const TbParams& P, const uint32_t qi, const uint32_t lane) {
    const uint32_t gbase = (threadIdx.x & 63u) - lane;  // first lane of my group within the wave
    auto gballot = [&](bool pred) -> uint64_t {
        const uint64_t b = __ballot(pred);
        return GW == 64 ? b : ((b >> gbase) & ((1ull << (GW & 63)) - 1ull));
    };
    if (P.exact_pass) {
        const uint32_t stt = P.ex_status[qi];
        if (stt != 0) {
            if (lane == 0 && stt == 1) P.flags[qi] |= POA_FLAG_REF_PANIC;
            if (lane == 0 && stt == 2) P.flags[qi] |= POA_FLAG_EXACT_OVERFLOW;
            return;
        }
    }
    const uint64_t qbeg = P.qoff[qi];
    TbCtx<T> c;
    c.rows = P.rows; c.pred_rows = P.pred_rows;
    c.L = (uint32_t)(P.qoff[qi + 1] - qbeg);
    c.q = P.qseq + qbeg;
    c.pitch = P.pitch[qi];
    const uint64_t RP = (uint64_t)P.n_rows * c.pitch;
    c.M = reinterpret_cast<const T*>(P.planes) + P.plane_off[qi];
    c.I = c.M + RP;
    c.D = c.I + (COMPACT ? RP / 4 : RP);   // compact: the flag codes take a quarter plane, then the kept D rows (compact_plane_elems)
    c.codes = reinterpret_cast<const uint32_t*>(c.I);
    c.d_slot = P.d_slot; c.pred_dslot = P.pred_dslot;
    c.code_fmt = P.code_fmt;
    c.tiled = P.exact_pass;  // the replayed search leaves its table tiled
    c.n_rows = P.n_rows;
    c.start_row = P.start_row; c.end_row = P.end_row;
    c.x = P.cost_x; c.o = P.cost_o; c.e = P.cost_e;
    c.row_depth = P.row_depth;
    const uint32_t L = c.L;

    uint2* out = P.scratch + P.scratch_off[qi];
    const uint32_t cap = (uint32_t)(P.scratch_off[qi + 1] - P.scratch_off[qi]);
    uint32_t cnt = 0;    // wave-uniform
    uint32_t flags = 0;  // wave-uniform
    auto emit_at = [&](uint32_t pos, uint32_t rpos, uint32_t qpos) {
        if (pos < cap) out[cap - 1 - pos] = make_uint2(rpos, qpos);
    };
    auto bc = [&](uint32_t v, uint32_t src) { return (uint32_t)__shfl((int)v, (int)(gbase + src)); };

    // the cell the backtrace starts from: (end row, L) for Global; where the replayed search stopped for ends-free
    const uint32_t tb_row = (P.exact_pass && P.ex_end) ? P.ex_end[2 * qi] : c.end_row;
    const uint32_t tb_off = (P.exact_pass && P.ex_end) ? P.ex_end[2 * qi + 1] : L;
    if (lane == 0) P.score[qi] = plM(c, tb_row, tb_off);
    const uint32_t end_node = c.rows[tb_row].node;

    bool done = false;
    uint32_t crow = 0, cj = 0, cst = 0;
    uint32_t gcs = INF;  // score of the current D / I cell (wave-uniform)
    if (L == 0) done = true;
    if (!done && L == 1 && tb_off == 1 && (tb_row == c.end_row || (uint32_t)c.rows[tb_row].sym == (uint32_t)c.q[0])) {
        // gap_affine.rs:812-824: "single nucleotide perfect match"; the end node equals every symbol, so in Global
        // mode this always yields [(end, 0)]
        flags |= POA_FLAG_SHORT_QUERY;
        if (lane == 0) emit_at(0, end_node, 0);
        cnt = 1;
        done = true;
    }
    if (!done) {
        // first hop from the end cell: Match, .or_else(Insertion), .or_else(Deletion) (gap_affine.rs:832-835)
        uint32_t f0 = 0, fr = 0, fj = 0, fs = 0, fallback = 0, fg = INF;
        if (lane == 0) {
            uint32_t nc; bool bad = false, pn = false;
            TbStep cur = tb_step<T, COMPACT>(c, tb_row, tb_off, 0, INF, nc, bad, pn);
            fg = cur.cs;
            if (cur.found && !pn && (nc != 1 || bad)) f0 |= POA_FLAG_AMBIGUOUS;
            if (!cur.found && !pn) {
                // the end row has no insertion state (I[end] = INF) and keeps its D row in every layout
                cur = tb_step<T, COMPACT>(c, tb_row, tb_off, 2, INF, nc, bad, pn);  // (full planes: the stored I value is used)
                if (!cur.found && !pn) {
                    const uint32_t d_end = COMPACT ? plDslot(c, c.d_slot ? c.d_slot[tb_row] : tb_row, tb_row, tb_off) : plD(c, tb_row, tb_off);
                    cur = tb_step<T, COMPACT>(c, tb_row, tb_off, 1, d_end, nc, bad, pn);
                    fg = cur.cs;
                }
                // no backtrace from the end cell: the reference builds a 'simple alignment' for len <= 3 and panics otherwise
                // (gap_affine.rs:838-853); on a replayed table that is exactly what happened, so only len > 3 is a panic
                if (!pn) {
                    if (!cur.found) { if (!(P.exact_pass && L <= 3)) f0 |= POA_FLAG_REF_PANIC; fallback = 1; }
                    else f0 |= POA_FLAG_AMBIGUOUS;
                    if (cur.found && cur.st == 1) fg = cur.cs - c.e;  // stepped D -> D
                }
            }
            // a Score subtraction wrapped onto u32::MAX: the reference dies here, nothing is emitted
            if (pn) { f0 |= POA_FLAG_REF_PANIC; fallback = 2; }
            fr = cur.row; fj = cur.j; fs = cur.st;
        }
        flags |= bc(f0, 0);
        if (bc(fallback, 0)) {
            if (bc(fallback, 0) == 2) flags |= POA_FLAG_TRUNCATED;
            if (bc(fallback, 0) == 1 && L <= 3) {
                if (lane == 0) for (uint32_t i = 0; i < L; ++i) emit_at(i, end_node, L - 1 - i);
                cnt = L;
            }
            done = true;
        } else {
            crow = bc(fr, 0); cj = bc(fj, 0); cst = bc(fs, 0); gcs = bc(fg, 0);
        }
    }
    bool reached_start = done;  // nothing to truncate in the special cases
    uint32_t d_run = 0;         // deletion steps taken in the current run (wave-uniform)
    uint32_t i_run = 0;         // insertion cycles taken in the current run (wave-uniform)
    while (!done) {
        // speculation: in Match state lane i assumes i (mis)matches into the previous row came before it, in Deletion
        // state i deletion-extensions into the previous row (long deletion runs: a read against a much longer graph)
        uint32_t depth = 1;
        const uint32_t max_depth = P.spec_depth < (uint32_t)GW ? P.spec_depth : (uint32_t)GW;
        // Insertion runs (a read's unaligned tail: tens of bases).  The reference's insertion step lands on the MATCH cell to
        // its left (gap_affine.rs:649) and re-enters the Insertion state from there through the zero-cost gap close, so one
        // inserted base is two steps of the walk: I(row, j) -> M(row, j-1) -> I(row, j-1).  Lane i evaluates that pair of
        // steps at column j - i, assuming i such cycles came before it (the Insertion cell it starts from then holds the Match
        // cell's score); the longest prefix of lanes whose cycle is exactly that — both steps found, no reference panic — is
        // taken at once, with the ambiguity / start-quirk flags of both steps.  Anything else (run ends, gap opens, a panic)
        // is left to the ordinary one-step code below, which makes the same decisions one at a time.
        if (cst == 2 && crow != c.start_row && max_depth > 1 && cj > 1) {
            uint32_t idepth = i_run < max_depth ? (i_run ? i_run : 1u) : max_depth;   // ramp up: most insertions are one base long
            if (idepth < 2) idepth = 2;
            if (cj < idepth) idepth = cj;
            const bool iact = lane < idepth;
            const uint32_t jj = cj - lane;
            TbStep sI{0, 0, 0, false, 0, 0}, sM{0, 0, 0, false, 0, 0};
            uint32_t ncI = 0, ncM = 0;
            bool badI = false, pnI = false, badM = false, pnM = false;
            if (iact) {
                const uint32_t gcs_i = lane == 0 ? gcs : plM(c, crow, jj);
                sI = tb_step<T, COMPACT>(c, crow, jj, 2, gcs_i, ncI, badI, pnI);
                if (sI.found && !pnI && sI.st == 0) sM = tb_step<T, COMPACT>(c, sI.row, sI.j, 0, INF, ncM, badM, pnM);
            }
            const bool cyc = iact && sI.found && !pnI && sI.st == 0 && sI.row == crow && sI.j + 1 == jj &&
                             sM.found && !pnM && sM.st == 2 && sM.row == crow && sM.j == sI.j;
            const uint64_t cb = gballot(cyc);
            const uint32_t ip = (cb == ~0ull) ? 64u : (uint32_t)__builtin_ctzll(~cb);   // accepted cycles, <= idepth
            if (ip > 0) {
                const uint64_t low = (ip >= 64) ? ~0ull : ((1ull << ip) - 1ull);
                const bool ambI = iact && ((ncI != 1 || badI) || (ncM != 1 || badM));
                const bool quirkI = iact && sI.j == 0 && (uint32_t)c.rows[crow].sym == (uint32_t)c.q[0];   // (crow is not the start row here)
                if (gballot(ambI) & low) flags |= POA_FLAG_AMBIGUOUS;
                if (gballot(quirkI) & low) flags |= POA_FLAG_START_QUIRK;
                if (lane < ip) emit_at(cnt + lane, POA_NONE, jj - 1);
                cnt += ip;
                cj -= ip;
                gcs = bc(sM.cs, ip - 1);   // the Insertion cell entered last has the Match cell's score
                i_run += ip;
                continue;
            }
            // no full cycle at lane 0 (the run ends here): its Insertion step is evaluated already — take it as the ordinary
            // code below would (one round instead of two for the last base of every run)
            if (bc((sI.found && !pnI) ? 1u : 0u, 0)) {
                if (bc((ncI != 1 || badI) ? 1u : 0u, 0)) flags |= POA_FLAG_AMBIGUOUS;
                if (lane == 0) emit_at(cnt, POA_NONE, cj - 1);
                cnt += 1;
                const uint32_t n_row = bc(sI.row, 0), n_j = bc(sI.j, 0), n_st = bc(sI.st, 0);
                if (n_st == 0 && n_j == 0 && n_row != c.start_row && (uint32_t)c.rows[n_row].sym == (uint32_t)c.q[0]) flags |= POA_FLAG_START_QUIRK;
                if (n_row == c.start_row) { reached_start = true; break; }
                crow = n_row; cj = n_j; cst = n_st;
                i_run = 0;
                continue;
            }
        }
        if (cst != 2) i_run = 0;
        if (cst == 0) {
            depth = max_depth;
            if (crow + 1 < depth) depth = crow + 1;
            if (cj + 1 < depth) depth = cj + 1;
        } else if (cst == 1) {
            // ramp up with the length of the run so far: most deletions are one or two rows long
            depth = d_run < max_depth ? (d_run ? d_run : 1u) : max_depth;
            if (crow + 1 < depth) depth = crow + 1;
        }
        if (cst != 1) d_run = 0;
        const bool active = lane < depth;
        const uint32_t my_row = crow - lane, my_j = (cst == 0) ? cj - lane : cj;
        const uint32_t my_gcs = (cst == 1) ? gcs - lane * c.e : gcs;
        TbStep bt{0, 0, 0, false, 0, 0};
        uint32_t nc = 0;
        bool bad = false, pn = false;
        if (active) bt = tb_step<T, COMPACT>(c, my_row, my_j, cst, my_gcs, nc, bad, pn);
        const bool amb = active && bt.found && (nc != 1 || bad);
        const bool quirk = active && bt.found && bt.st == 0 && bt.j == 0 && bt.row != c.start_row && cst != 1 &&
                           (uint32_t)c.rows[bt.row].sym == (uint32_t)c.q[0];
        // a "regular" step: (mis)match into exactly the cell the next lane speculated on, not yet at start
        const bool regular = active && bt.found && bt.row + 1 == my_row && bt.row != c.start_row &&
                             ((cst == 0 && bt.st == 0 && bt.j + 1 == my_j) || (cst == 1 && bt.st == 1 && bt.j == my_j));
        const uint64_t rb = gballot(regular);
        uint32_t p = (rb == ~0ull) ? 64u : (uint32_t)__builtin_ctzll(~rb);  // accepted prefix, <= depth
        // a Score subtraction that wraps onto u32::MAX kills the reference at that step: the steps before it stand,
        // nothing after it is defined
        const uint64_t pnb = gballot(active && pn);
        const uint32_t first_pn = pnb ? (uint32_t)__builtin_ctzll(pnb) : 64u;
        const bool dies = pnb != 0 && first_pn <= p;   // (p can be 64: the sentinel must not count)
        if (dies) p = first_pn;
        const uint64_t low = (p >= 64) ? ~0ull : ((1ull << p) - 1ull);
        if (gballot(amb) & low) flags |= POA_FLAG_AMBIGUOUS;
        if (gballot(quirk) & low) flags |= POA_FLAG_START_QUIRK;
        if (dies) {
            if (lane < p) emit_at(cnt + lane, bt.node, cst == 0 ? my_j - 1 : POA_NONE);
            cnt += p;
            flags |= POA_FLAG_REF_PANIC;
            break;
        }
        if (lane < p) emit_at(cnt + lane, bt.node, cst == 0 ? my_j - 1 : POA_NONE);
        cnt += p;
        if (p == depth) {
            // every speculated step was regular: continue below the last one, in the same state
            crow -= depth;
            if (cst == 0) cj -= depth;
            else { gcs -= depth * c.e; d_run += depth; }
            continue;
        }
        // lane p deviates: replay the sequential rule with its results
        const uint32_t d_found = bc(bt.found ? 1u : 0u, p), d_row = bc(bt.row, p), d_j = bc(bt.j, p), d_st = bc(bt.st, p);
        const uint32_t d_amb = bc(amb ? 1u : 0u, p), d_quirk = bc(quirk ? 1u : 0u, p);
        const uint32_t d_cs = bc(bt.cs, p), d_node = bc(bt.node, p);
        const uint32_t cur_j = (cst == 0) ? cj - p : cj, cur_st = cst;  // the cell is (crow - p, cur_j); speculation keeps the state
        if (!d_found) break;
        if (d_amb) flags |= POA_FLAG_AMBIGUOUS;
        if (cur_st == 0 && d_st != 0) {  // zero-cost gap close: no pair (gap_affine.rs:871-875)
            crow = d_row; cj = d_j; cst = d_st;
            gcs = d_cs;  // the gap cell has the Match cell's score
            continue;
        }
        if (lane == 0) {
            const uint32_t node = d_node;
            if (cur_st == 0) emit_at(cnt, node, cur_j - 1);
            else if (cur_st == 2) emit_at(cnt, POA_NONE, cur_j - 1);
            else emit_at(cnt, node, POA_NONE);
        }
        cnt += 1;
        if (d_quirk) flags |= POA_FLAG_START_QUIRK;
        if (d_row == c.start_row) { reached_start = true; break; }
        if (cur_st == 1 && d_st == 1) gcs = d_cs - c.e;  // D -> D: one more extension
        if (cur_st == 1) d_run += p + 1;
        crow = d_row; cj = d_j; cst = d_st;
    }
    if (!reached_start) flags |= POA_FLAG_TRUNCATED;
    if (lane == 0) {
        // after an exact replay the table IS the reference's: ties and quirks are resolved exactly as it does
        P.flags[qi] = P.exact_pass ? (flags & (POA_FLAG_REF_PANIC | POA_FLAG_TRUNCATED)) : flags;
        P.n_pairs[qi] = cnt < cap ? cnt : cap;
    }
}

template <typename T, bool COMPACT, int GW = 64>
__global__ __launch_bounds__(256) void poa_traceback_kernel(TbParams P) {
    const uint32_t lane = threadIdx.x & (uint32_t)(GW - 1);
    const uint32_t wq = (blockIdx.x * blockDim.x + threadIdx.x) / (uint32_t)GW;  // uniform over the group
    if (wq >= P.n_queries) return;
    traceback_wave<T, COMPACT, GW>(P, P.first_query + wq, lane);
}

// ---------------------------------------------------------------------------------------------
// Forward pass.  One wave per query; 4 waves (queries) per 256-thread block.
//
// Layout ("quad-striped"): a strip is W = Q*64*K columns; lane l owns, in each of the Q quads,
// the K consecutive columns  s*W + m*64*K + K*l + {0..K-1}  (K*sizeof(T) == 16 bytes).  Every
// global_load/store_dwordx4 of a quad therefore covers 1 KiB contiguous bytes per wave-instruction
// (8 full 128-B lines), the quads' insertion scans are independent chains, and the column-(j-1)
// neighbour is an in-register value except for k = 0 (one DPP wave_shr:1 per quad).
// FUSE_TB: the wave traces its own query right after its last row (the latency-bound traceback then overlaps
// other waves' HBM-bound forward work instead of running as a separate launch).
// COMPACT (u16 planes only): the I plane is replaced by a 4-bit code per cell and D rows are written only where a
// later row reads them back (ROW_STORE_D).  Codes, one nibble per cell, 8 cells per dword, row-major at the I plane's
// place:  bit0 I==M   bit1 I[j]==I[j-1]+e   bit2 D==M   bit3 D==PD+e; within a dword the
// nibble of column k (0..7) sits at position (k >> 1) + 4 * (k & 1).  They are exactly the predicates the
// traceback evaluates on I and (for chain rows) D; see traceback_wave.
// MW: one workgroup per query, its strips pipelined over the waves — described at poa_forward_packed_kernel.
constexpr int MW_MAX_WAVES = 16;
constexpr uint32_t MW_RING = 64;
static_assert(MW_RING >= 2 * ROW_NEAR, "ring must hold the look-back window plus slack");

__device__ __forceinline__ void mw_wait_gt(uint32_t* p, uint32_t v, uint32_t* pipeline_error) {
    uint32_t spins = 0;
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= v) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 27)) {
            // never reached in a correct pipeline; bounds a protocol bug to seconds AND reports it: the wave goes on with
            // stale hand-over data, so the run's results must not be used (poa_batch_fetch / poa_batch_stats return POA_ERR_HIP)
            if (pipeline_error) __hip_atomic_store(pipeline_error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <int Q, typename T, bool FUSE_TB, bool COMPACT, bool MW = false>
__global__ __launch_bounds__(MW ? (Q >= 4 ? 640 : 1024) : 256, POA_FWD_MIN_WAVES) void poa_forward_kernel(FwdParams P, TbParams TP) {
    static_assert(!COMPACT || PlaneIO<T>::K == 8, "compact codes assume 8 columns per lane and quad");
    using IO = PlaneIO<T>;
    constexpr int K = IO::K;
    constexpr int C = K * Q;
    constexpr uint32_t QW = 64 * K;  // columns per quad
    constexpr uint32_t W = QW * Q;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t S = MW ? (blockDim.x >> 6) : 1u;  // strips in flight per query
    __shared__ uint32_t mw_progress[MW ? MW_MAX_WAVES : 1];
    __shared__ uint32_t mw_ring[MW ? MW_MAX_WAVES : 1][MW ? MW_RING : 1][4];  // {scan carry, I last col, M last col, -}
    if (MW) {
        if (threadIdx.x < MW_MAX_WAVES) mw_progress[threadIdx.x] = 0;
        __syncthreads();
    }
    const uint32_t wq = MW ? blockIdx.x : (blockIdx.x * blockDim.x + threadIdx.x) >> 6;  // wave-uniform
    if (wq >= P.n_queries) return;
    const uint32_t qi = P.first_query + wq;
    const uint64_t qbeg = P.qoff[qi];
    const uint32_t L = (uint32_t)(P.qoff[qi + 1] - qbeg);
    const uint8_t* __restrict__ q = P.qseq + qbeg;
    const uint32_t pitch = P.pitch[qi];
    const uint64_t RP = (uint64_t)P.n_rows * pitch;
    T* __restrict__ Mp = reinterpret_cast<T*>(P.planes) + P.plane_off[qi];
    T* __restrict__ Ip = Mp + RP;
    T* __restrict__ Dp = Ip + (COMPACT ? RP / 4 : RP);   // compact: codes, then the kept D rows at their slots
    uint32_t* __restrict__ carry = P.strip_carry + 2ull * wq * P.n_rows;  // [2r]: I entering the next strip, [2r+1]: I of my last column
    const uint32_t x = P.cost_x, oe = P.cost_oe, e = P.cost_e;
    const uint32_t n_strips = (pitch + W - 1) / W;
    const uint32_t step = K * e;                      // one lane == K columns
    const uint32_t w15 = ((lane & 15u) + 1u) * step;
    const uint32_t w31 = (lane - 31u) * step;         // only used by lanes >= 32
    const uint32_t lane_off = K * lane * e;           // cost of extending an insertion to my first column of a quad

    const uint32_t n_groups = (n_strips + S - 1) / S;
    for (uint32_t g = 0; g < n_groups; ++g) {
        const uint32_t s = g * S + (MW ? wave : 0u);
        if (MW && s >= n_strips) break;
        const bool from_ring = MW && wave > 0;                          // strip s - 1 is being computed by wave - 1 right now
        const bool from_global = s > 0 && !from_ring;                   // ... or was finished earlier (carry array + planes)
        const bool to_ring = MW && wave + 1 < S && s + 1 < n_strips;
        const bool to_global = s + 1 < n_strips && !to_ring;
        const uint32_t prog_base = g * P.n_rows;
        uint32_t m_edge_prev = INF;                                     // from_ring: M[r-1][sbase-1]
        if (MW && from_global) {
            mw_wait_gt(&mw_progress[S - 1], prog_base - 1, P.pipeline_error);             // the whole previous group is done and released
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        const uint32_t sbase = s * W;
        bool act[Q];            // my K columns of quad m lie inside the plane row
        uint32_t qcp[C / 4];    // my query symbols, one per byte; 0 (never a symbol) past the end
        uint32_t ql[Q];         // query symbol left of my first column of quad m
#pragma unroll
        for (int m = 0; m < Q; ++m) {
            const uint32_t c0 = sbase + m * QW + K * lane;
            act[m] = c0 < pitch;
#pragma unroll
            for (int w = 0; w < K / 4; ++w) {
                uint32_t pk = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t c = c0 + 4 * w + k;
                    pk |= ((c < L) ? (uint32_t)q[c] : 0u) << (8 * k);
                }
                qcp[m * (K / 4) + w] = pk;
            }
            ql[m] = (c0 > 0 && c0 - 1 < L) ? (uint32_t)q[c0 - 1] : 0u;
        }
        auto qsym = [&](int i) -> uint32_t { return qbyte(qcp[i >> 2], i & 3); };  // i = m*K + k

        uint32_t Mprev[C], Dprev[C];
#pragma unroll
        for (int k = 0; k < C; ++k) { Mprev[k] = INF; Dprev[k] = INF; }

        for (uint32_t r = 0; r < P.n_rows; ++r) {
            const RowMeta meta = P.rows[r];
            const uint32_t sym = meta.sym;
            const uint64_t rbase = (uint64_t)r * pitch + sbase + K * lane;
            uint32_t PM[C], PD[C], PMl[Q];
            uint32_t in_cq = INF, in_ilast = INF, in_mlast = INF, cq_out = INF;
            if (from_ring) {
                mw_wait_gt(&mw_progress[wave - 1], prog_base + r, P.pipeline_error);
                const uint32_t* slot = mw_ring[wave - 1][(prog_base + r) % MW_RING];
                in_cq = slot[0]; in_ilast = slot[1]; in_mlast = slot[2];
            }

            const bool chain = (meta.flags & ROW_CHAIN) != 0;
            if (chain) {
                // fast path: the only predecessor is the previous row, still in registers
                uint32_t edge = INF;
                if (from_ring) edge = m_edge_prev;
                else if (from_global) edge = IO::get(Mp + (uint64_t)(r - 1) * pitch + sbase - 1);  // uniform address
#pragma unroll
                for (int m = 0; m < Q; ++m) {
                    PMl[m] = wave_shr1(Mprev[K * m + K - 1], edge);
                    edge = (uint32_t)__builtin_amdgcn_readlane((int)Mprev[K * m + K - 1], 63);
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
                // I read back rows this wave stored itself (wavefront scope); MW: a far predecessor's edge column comes
                // from the planes of wave - 1, released by its fence on this same graph row
                if (MW && (meta.flags & ROW_FAR_PRED)) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                else if (meta.pred_count > 0) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                for (uint32_t pe = 0; pe < meta.pred_count; ++pe) {
                    const uint32_t pr = P.pred_rows[meta.pred_begin + pe];
                    const uint64_t pbase = (uint64_t)pr * pitch + sbase + K * lane;
                    const uint64_t pbase_d = (COMPACT && P.pred_dslot) ? (uint64_t)P.pred_dslot[meta.pred_begin + pe] * pitch + sbase + K * lane : pbase;
                    uint32_t tm[C], td[C];
                    if (pr + 1 == r) {
#pragma unroll
                        for (int k = 0; k < C; ++k) { tm[k] = Mprev[k]; td[k] = Dprev[k]; }
                    } else {
#pragma unroll
                        for (int m = 0; m < Q; ++m) {
                            uint32_t a[K], b[K];
#pragma unroll
                            for (int k = 0; k < K; ++k) { a[k] = INF; b[k] = INF; }
                            if (act[m]) {
                                IO::load(Mp + pbase + m * QW, a);
                                IO::load(Dp + pbase_d + m * QW, b);
                            }
#pragma unroll
                            for (int k = 0; k < K; ++k) { tm[K * m + k] = a[k]; td[K * m + k] = b[k]; }
                        }
                    }
                    uint32_t edge = INF;
                    if (from_ring && r - pr <= ROW_NEAR) edge = mw_ring[wave - 1][(prog_base + pr) % MW_RING][2];
                    else if (s > 0) edge = IO::get(Mp + (uint64_t)pr * pitch + sbase - 1);
#pragma unroll
                    for (int m = 0; m < Q; ++m) {
                        PMl[m] = umin(PMl[m], wave_shr1(tm[K * m + K - 1], edge));
                        edge = (uint32_t)__builtin_amdgcn_readlane((int)tm[K * m + K - 1], 63);
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
                uint32_t H[C], Tq[Q];
#pragma unroll
                for (int m = 0; m < Q; ++m) {
                    uint32_t t = INF;  // in-lane insertion chain, carry-in INF
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const int i = K * m + k;
                        const uint32_t qk = qsym(i);
                        const uint32_t open = (qk != sym) ? sat_add(PM[i], oe) : INF;
                        Dc[i] = umin(sat_add(PD[i], e), open);
                        const uint32_t pm_left = (k == 0) ? PMl[m] : PM[i - 1];
                        const uint32_t q_left = (k == 0) ? ql[m] : qsym(i - 1);
                        H[i] = umin(sat_add(pm_left, (q_left != sym) ? x : 0u), Dc[i]);
                        if (m == 0 && k == 0 && (meta.flags & ROW_START) && sbase == 0 && lane == 0) H[i] = 0;
                        Ic[i] = t;  // value entering column i from my own earlier columns (INF for k == 0)
                        const bool op = !open_never && (open_always || qk != csym);
                        t = umin(sat_add(t, e), op ? sat_add(H[i], oe) : INF);
                    }
                    Tq[m] = t;  // leaves my last column of quad m (carry-in INF)
                }
                // cross-lane: independent scans per quad, then a uniform carry chain over the quads
                uint32_t cq = from_ring ? in_cq : (from_global ? carry[2 * r] : INF);  // insertion value entering column sbase
#pragma unroll
                for (int m = 0; m < Q; ++m) {
                    const uint32_t Pm = wave_scan_min_plus(Tq[m], step, w15, w31);
                    const uint32_t excl = wave_shr1(Pm, INF);                // from earlier lanes of this quad
                    const uint32_t cin = umin(excl, sat_add(cq, lane_off));  // ... or from before the quad
                    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)Pm, 63);
                    cq = umin(sat_add(cq, QW * e), total);
                    Ic[K * m] = cin;
#pragma unroll
                    for (int k = 1; k < K; ++k) Ic[K * m + k] = umin(Ic[K * m + k], sat_add(cin, (uint32_t)k * e));
                }
                if (to_global && lane == 0) carry[2 * r] = cq;  // I[r][(s+1)*W]
                cq_out = cq;
#pragma unroll
                for (int k = 0; k < C; ++k) Mc[k] = umin(H[k], Ic[k]);
            }

            if (!COMPACT) {
#pragma unroll
                for (int m = 0; m < Q; ++m) {
                    if (act[m]) {
                        IO::store(Mp + rbase + m * QW, &Mc[K * m]);
                        IO::store(Ip + rbase + m * QW, &Ic[K * m]);
                        IO::store(Dp + rbase + m * QW, &Dc[K * m]);
                    }
                }
            } else {
                uint32_t* __restrict__ codes = reinterpret_cast<uint32_t*>(Ip) + (uint64_t)r * (pitch / 8) + sbase / 8 + lane;
                const bool keep_d = (meta.flags & ROW_STORE_D) != 0;
                uint32_t edge_i = INF;  // I of the column left of my first column of the quad
                if (from_ring) edge_i = in_ilast;
                else if (from_global) edge_i = carry[2 * r + 1];
                // (for s > 0 the value was written by lane 63 at the end of the previous strip of this row)
#pragma unroll
                for (int m = 0; m < Q; ++m) {
                    const uint32_t i_left = wave_shr1(Ic[K * m + K - 1], edge_i);
                    edge_i = (uint32_t)__builtin_amdgcn_readlane((int)Ic[K * m + K - 1], 63);
                    uint32_t code = 0;
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const int i = K * m + k;
                        const uint32_t ip = (k == 0) ? i_left : Ic[i - 1];
                        uint32_t nb = (Ic[i] == Mc[i]) ? 1u : 0u;
                        nb |= (Ic[i] == sat_add(ip, e)) ? 2u : 0u;
                        nb |= (Dc[i] == Mc[i]) ? 4u : 0u;
                        nb |= (Dc[i] == sat_add(PD[i], e)) ? 8u : 0u;
                        code |= nb << (4 * ((k >> 1) + 4 * (k & 1)));
                    }
                    if (act[m]) {
                        IO::store(Mp + rbase + m * QW, &Mc[K * m]);
                        if (keep_d) IO::store(Dp + (uint64_t)(P.d_slot ? P.d_slot[r] : r) * pitch + sbase + K * lane + m * QW, &Dc[K * m]);
                        codes[m * (QW / 8)] = code;
                    }
                }
                if (to_global && lane == 63) carry[2 * r + 1] = Ic[C - 1];  // I[r][(s+1)*W - 1]
            }
            if (MW) {
                if (to_ring) {
                    // back-pressure: the consumer may still look ROW_NEAR rows back from the row it is working on
                    if (prog_base + r + ROW_NEAR >= MW_RING) mw_wait_gt(&mw_progress[wave + 1], prog_base + r + ROW_NEAR - MW_RING, P.pipeline_error);
                    if (lane == 63) {
                        uint32_t* slot = mw_ring[wave][(prog_base + r) % MW_RING];
                        slot[0] = cq_out; slot[1] = Ic[C - 1]; slot[2] = Mc[C - 1];
                    }
                }
                if (to_global && r + 1 == P.n_rows) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
                if (lane == 0) __hip_atomic_store(&mw_progress[wave], prog_base + r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            m_edge_prev = in_mlast;
#pragma unroll
            for (int k = 0; k < C; ++k) { Mprev[k] = Mc[k]; Dprev[k] = Dc[k]; }
        }
        if (n_strips > 1) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // carry[] and edge columns for the next strip
    }
    if (FUSE_TB) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // my plane stores are complete before I read them back
        traceback_wave<T, COMPACT>(TP, qi, lane);
    }
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
