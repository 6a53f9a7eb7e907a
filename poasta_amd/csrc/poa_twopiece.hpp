// Two-piece affine model (/root/reference/src/aligner/scoring/gap_affine_2piece.rs): the dense pass.
// Included by poa_engine.hip (uses its poa_graph, DevBuf, HIP_TRY, fail).
//
// Edge set (gap_affine_2piece.rs:292-516; the tests hold a CPU restatement as the executable specification): a gap opens in the
// first piece exactly as in the one-piece model (open1 + extend1, greedy-match rule), every further step stays in its
// piece or moves from the first to the second at extend2; open2 is never charged.  Five planes per query:
//
//   D1[v][j] = min( PD1[j] + e1 , openD(v,j) ? PM[j] + o1 + e1 : INF )        PM / PD1 / PD2 = min over predecessors
//   D2[v][j] = min( PD1[j], PD2[j] ) + e2
//   H [v][j] = min( PM[j-1] + (mismatch ? x : 0) , D1, D2 )                    H[start][0] = 0
//   I1[v][j+1] = min( I1[v][j] + e1 , openI(v,j) ? H[v][j] + o1 + e1 : INF )   min-plus prefix scan, decay e1
//   I2[v][j+1] = min( I1[v][j], I2[v][j] ) + e2                                min-plus prefix scan over I1, decay e2
//   M [v][j] = min( H, I1, I2 )                                                end row: no I, D1 only by extension
//
// One wavefront per query, rows in topological order, four columns per lane and 256 per pass with the scan carries kept in
// registers; predecessor rows are re-read from the planes (just written by this wave: L2).  u32 arithmetic, INF absorbing.
// This is the u32 kernel of the model (20 bytes written per cell: HBM-bound territory); the packed-u16 pairs-across-quads
// mapping of the one-piece kernels carries over (two more packed recurrences per register) and is the next step for it.
// Traceback: the reference's rule (gap_affine_2piece.rs:639-794, :944-1043) on the five planes with the same uniqueness
// certificate as the one-piece pass; one lane per query (a chain of dependent reads; the speculative walk of
// poa_traceback_kernel is not ported to five planes yet).
#pragma once

namespace poa_amd {

struct TwoPieceParams {
    const RowMeta* rows;
    const uint32_t* pred_rows;
    uint32_t n_rows, start_row, end_row;
    const uint8_t* qseq;
    const uint64_t* qoff;
    uint32_t first_query, n_queries;   // chunk
    uint32_t pitch;                    // columns per row (multiple of 64), same for the whole chunk
    uint32_t* planes;                  // per slot: [5][n_rows][pitch] of the kernel's plane type (u32, or u16 when every score that matters fits): M, I1, D1, I2, D2
    uint32_t x, oe, e1, e2, o1;
    uint32_t* score;                   // [total]
    uint32_t* flags;                   // [total]
    uint32_t* n_pairs;                 // [total]
    poa_aln_pair_t* scratch;           // per slot: n_rows + pitch pairs, written back to front
    uint32_t scratch_stride;
    // replayed search (poa2_exact_kernel ran on u32 planes): where each walk starts and what became of the search
    uint32_t exact_pass;
    const uint32_t* ex_status;         // [total] EX_*
    const uint32_t* ex_end;            // [2 * total] (row, offset) the search stopped at
};

// Replay of the reference's two-piece search (Affine2PieceMinGapCost / Affine2PieceDijkstra with or without pruning,
// config.rs:160-272; astar.rs:124-226 over gap_affine_2piece.rs): the search object of poa_exact.hpp instantiated with
// EX_AS_TWO_PIECE — the generic code, five plain u32 planes (the layout the traceback below reads), linked-list queue with
// five stacks per priority in the reference's pop order.  One search per lane, `lanes_per_wave` lanes of a wave active.
struct TwoPieceExact {
    ExactGraph G;
    ExactCosts C;
    uint64_t* reached; uint64_t* rsum; uint32_t wpn, swpn;   // per slot: n_exit * wpn / n_exit * swpn words
    uint32_t* head; uint32_t n_prio;                          // per slot: 5 * n_prio
    ExQEntry* pool; uint32_t pool_cap;
    ExStackEntry* stack; uint32_t stack_cap;
    uint32_t* status;        // [total]
    uint32_t* end_cell;      // [2 * total]
    uint32_t* counters;      // [4 * total] num_queued, num_visited, num_pruned, queue entries live at once (high water)
    uint32_t lanes_per_wave;
    uint32_t lds_graph, n_succ, n_nbm;   // 1: the launch carries exact_lds_bytes() of dynamic LDS for the graph arrays
};

constexpr int EXACT2_BLOCK = 1024;   // 16 waves share one LDS copy of the graph

__global__ __launch_bounds__(EXACT2_BLOCK) void poa2_exact_kernel(TwoPieceParams P, TwoPieceExact X) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds2[];
    if (X.lds_graph) { const ExactGraph src = X.G; exact_stage_graph(X.G, src, lds2, X.n_succ, X.n_nbm); }
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane >= X.lanes_per_wave) return;
    const uint32_t slot = (blockIdx.x * (EXACT2_BLOCK / 64) + wave) * X.lanes_per_wave + lane;
    if (slot >= P.n_queries) return;
    const uint32_t qi = P.first_query + slot;
    const uint64_t qbeg = P.qoff[qi];
    const uint32_t L = (uint32_t)(P.qoff[qi + 1] - qbeg);
    ExactWork W;
    W.T = P.planes + (uint64_t)slot * 5 * P.n_rows * P.pitch;
    W.n_rows = P.n_rows; W.pitch = P.pitch;
    W.reached = X.reached + (uint64_t)slot * X.G.n_exit * X.wpn;
    W.rsum = X.rsum + (uint64_t)slot * X.G.n_exit * X.swpn;
    W.wpn = X.wpn; W.swpn = X.swpn;
    W.head = X.head + (uint64_t)slot * 5 * X.n_prio; W.n_prio = X.n_prio;
    W.pool = X.pool + (uint64_t)slot * X.pool_cap; W.pool_cap = X.pool_cap;
    W.stack = X.stack + (uint64_t)slot * X.stack_cap; W.stack_cap = X.stack_cap;
    ExactSearchT<EX_AS_NO_SPEC | EX_AS_TWO_PIECE> S(X.G, W, P.qseq + qbeg, L, X.C);
    const ExactResult R = S.run();
    X.status[qi] = R.status;
    X.end_cell[2 * qi] = R.end_row; X.end_cell[2 * qi + 1] = R.end_off;
    X.counters[4 * qi] = R.num_queued; X.counters[4 * qi + 1] = R.num_visited; X.counters[4 * qi + 2] = R.num_pruned;
    X.counters[4 * qi + 3] = S.pool_top;
}

__device__ __forceinline__ uint32_t tp_sat(uint32_t a, uint32_t b) {
    const uint32_t r = a + b;
    return r < a ? 0xFFFFFFFFu : r;
}

// inclusive min-plus scan over the lanes with a constant decay per lane step: out(l) = min over l' <= l of v(l') + (l - l') * step
// (DPP row shifts and broadcasts, no LDS pipe: wave_scan_min_plus of the one-piece kernels)
__device__ __forceinline__ uint32_t tp_scan(uint32_t v, uint32_t step, uint32_t lane) {
    return wave_scan_min_plus(v, step, ((lane & 15u) + 1u) * step, (lane - 31u) * step);
}

// K consecutive columns per lane (4 with u32 planes, 8 with u16), 64 K columns per pass: every plane access is one 16-byte
// load / store per lane, 1 KiB contiguous per wave-instruction; the insertion recurrences run as a K-step chain in the lane
// plus one wave scan per pass and plane (I1 with decay K*e1 per lane, I2 over the finished I1 with decay K*e2), carries
// between passes in registers.  T = uint16_t (same saturation argument as the one-piece u16 planes: poa_batch_run_ex) halves
// the bytes of a kernel that lives on its stores; arithmetic is u32 in registers either way (PlaneIO widens 0xFFFF to INF).
// NP: passes whose previous row stays in registers (rows of up to NP * 64 * K columns): a chain row — one predecessor, the
// previous row — then reads nothing back from the planes (6 of the 16 bytes of traffic per cell with u16 planes).
template <typename T, int NP>
__global__ __launch_bounds__(64) void poa2_forward_kernel(TwoPieceParams P) {
    using IO = PlaneIO<T>;
    constexpr int K = IO::K;
    constexpr uint32_t PW = 64 * K;   // columns per pass
    constexpr uint32_t INF = 0xFFFFFFFFu;
    const uint32_t slot = blockIdx.x, lane = threadIdx.x;
    const uint32_t qi = P.first_query + slot;
    const uint64_t qbeg = P.qoff[qi];
    const uint32_t L = (uint32_t)(P.qoff[qi + 1] - qbeg);
    const uint8_t* q = P.qseq + qbeg;
    const uint64_t plane = (uint64_t)P.n_rows * P.pitch;
    T* M = reinterpret_cast<T*>(P.planes) + (uint64_t)slot * 5 * plane;
    T* I1 = M + plane; T* D1 = I1 + plane; T* I2 = D1 + plane; T* D2 = I2 + plane;
    const uint32_t n_pass = (L + 1 + PW - 1) / PW;   // pitch is a multiple of 64: a pass may end inside the row's padding
    constexpr int NPA = NP > 0 ? NP : 1;
    uint32_t keepM[NPA][K], keepD1[NPA][K], keepD2[NPA][K];   // M, D1, D2 of the previous row, my columns of every pass
    const bool keep = NP > 0 && n_pass <= (uint32_t)NP;
    for (uint32_t r = 0; r < P.n_rows; ++r) {
        const RowMeta rm = P.rows[r];
        const bool is_end = r == P.end_row, is_start = r == P.start_row;
        const uint64_t ro = (uint64_t)r * P.pitch;
        uint32_t c1 = INF, c2 = INF;   // I1 / I2 entering the first column of the pass
        uint32_t cpm = INF;             // min over predecessors of M[p][first column of the pass - 1]
        const bool from_regs = keep && r > 0 && (rm.flags & ROW_CHAIN);
        auto do_pass = [&](const uint32_t ps, uint32_t (&kM)[K], uint32_t (&kD1)[K], uint32_t (&kD2)[K]) {
            const uint32_t j = ps * PW + K * lane;    // my first column
            const bool in = j < P.pitch;              // (whole 16-byte groups lie inside or outside the plane row)
            uint32_t qs[K], qm;                       // q[j + k] (0 past the end: never a symbol), q[j - 1]
#pragma unroll
            for (int k = 0; k < K; ++k) qs[k] = (j + k < L) ? (uint32_t)q[j + k] : 0u;
            qm = (j > 0 && j - 1 < L) ? (uint32_t)q[j - 1] : 0u;
            uint32_t pm[K], pd[K], pd2[K];
#pragma unroll
            for (int k = 0; k < K; ++k) { pm[k] = INF; pd[k] = INF; pd2[k] = INF; }
            if (from_regs) {
#pragma unroll
                for (int k = 0; k < K; ++k) { pm[k] = kM[k]; pd[k] = kD1[k]; pd2[k] = kD2[k]; }
            } else if (in)
                for (uint32_t e = 0; e < rm.pred_count; ++e) {
                    const uint64_t po = (uint64_t)P.pred_rows[rm.pred_begin + e] * P.pitch + j;
                    uint32_t a[K], b[K], c[K];
                    IO::load(M + po, a); IO::load(D1 + po, b); IO::load(D2 + po, c);
#pragma unroll
                    for (int k = 0; k < K; ++k) { pm[k] = min(pm[k], a[k]); pd[k] = min(pd[k], b[k]); pd2[k] = min(pd2[k], c[k]); }
                }
            // M of the predecessors one column to the left of my first column
            const uint32_t pml = wave_shr1(pm[K - 1], cpm);   // lane 0: the last column of the previous pass
            cpm = (uint32_t)__builtin_amdgcn_readlane((int)pm[K - 1], 63);
            uint32_t h[K], d1[K], d2[K], a[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const uint32_t col = j + k;
                const bool col_in = col <= L;
                d1[k] = tp_sat(pd[k], P.e1);
                if (!is_end && (col >= L || rm.sym != qs[k])) d1[k] = min(d1[k], tp_sat(pm[k], P.oe));   // openD
                d2[k] = tp_sat(min(pd[k], pd2[k]), P.e2);
                const uint32_t left = k == 0 ? pml : pm[k - 1];
                const uint32_t ql = k == 0 ? qm : qs[k - 1];
                uint32_t diag = INF;
                if (is_end) diag = pm[k];
                else if (col > 0) diag = tp_sat(left, rm.sym != ql ? P.x : 0u);
                h[k] = min(diag, min(d1[k], d2[k]));
                if (is_start && col == 0) h[k] = 0;
                if (!col_in) { h[k] = INF; d1[k] = INF; d2[k] = INF; }
                bool open_i = false;
                if (col < L && !is_end) {
                    if (rm.flags & ROW_OPENI_ALWAYS) open_i = true;
                    else if (rm.flags & ROW_OPENI_NEVER) open_i = false;
                    else open_i = rm.child_sym != qs[k];
                }
                a[k] = open_i ? tp_sat(h[k], P.oe) : INF;   // what column col opens INTO col + 1
            }
            // I1: chain inside the lane with nothing entering, then what enters from the left
            uint32_t v1[K];
            v1[0] = INF;
#pragma unroll
            for (int k = 1; k < K; ++k) v1[k] = min(tp_sat(v1[k - 1], P.e1), a[k - 1]);
            const uint32_t out1 = min(tp_sat(v1[K - 1], P.e1), a[K - 1]);   // leaves my last column towards the next lane
            const uint32_t s1 = tp_scan(out1, K * P.e1, lane);
            uint32_t in1 = wave_shr1(s1, INF);
            in1 = min(in1, tp_sat(c1, lane * K * P.e1));
            if (ps == 0 && lane == 0) in1 = INF;                            // I1[v][0] = INF
#pragma unroll
            for (int k = 0; k < K; ++k) v1[k] = min(v1[k], tp_sat(in1, (uint32_t)k * P.e1));
            c1 = min((uint32_t)__builtin_amdgcn_readlane((int)s1, 63), tp_sat(c1, 64 * K * P.e1));
            // I2 over the finished I1: I2[c] = min(I1[c-1], I2[c-1]) + e2
            const uint32_t i1_left = in1;                                    // == I1 of my first column
            uint32_t v2[K];
            v2[0] = INF;
#pragma unroll
            for (int k = 1; k < K; ++k) v2[k] = tp_sat(min(v2[k - 1], v1[k - 1]), P.e2);
            const uint32_t out2 = tp_sat(min(v2[K - 1], v1[K - 1]), P.e2);
            const uint32_t s2 = tp_scan(out2, K * P.e2, lane);
            uint32_t in2 = wave_shr1(s2, INF);
            in2 = min(in2, tp_sat(c2, lane * K * P.e2));
            if (ps == 0 && lane == 0) in2 = INF;
            (void)i1_left;
#pragma unroll
            for (int k = 0; k < K; ++k) v2[k] = min(v2[k], tp_sat(in2, (uint32_t)k * P.e2));
            c2 = min((uint32_t)__builtin_amdgcn_readlane((int)s2, 63), tp_sat(c2, 64 * K * P.e2));
            if (is_end) {
#pragma unroll
                for (int k = 0; k < K; ++k) { v1[k] = INF; v2[k] = INF; }
            }
            if (in) {
                uint32_t m[K];
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    m[k] = min(h[k], min(v1[k], v2[k]));
                    if (j + k > L) { m[k] = INF; v1[k] = INF; v2[k] = INF; }   // padding columns read as unvisited
                }
                IO::store(M + ro + j, m); IO::store(I1 + ro + j, v1); IO::store(D1 + ro + j, d1);
                IO::store(I2 + ro + j, v2); IO::store(D2 + ro + j, d2);
                if (keep) {
#pragma unroll
                    for (int k = 0; k < K; ++k) { kM[k] = m[k]; kD1[k] = d1[k]; kD2[k] = d2[k]; }
                }
            } else if (keep) {
#pragma unroll
                for (int k = 0; k < K; ++k) { kM[k] = INF; kD1[k] = INF; kD2[k] = INF; }
            }
        };
        if (keep) {
#pragma unroll
            for (int ps = 0; ps < NPA; ++ps)
                if ((uint32_t)ps < n_pass) do_pass((uint32_t)ps, keepM[ps], keepD1[ps], keepD2[ps]);
        } else {
            for (uint32_t ps = 0; ps < n_pass; ++ps) do_pass(ps, keepM[0], keepD1[0], keepD2[0]);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // the next rows read this one back (same wave)
    }
    if (lane == 0) P.score[qi] = IO::get(M + (uint64_t)P.end_row * P.pitch + L);
}

// One lane per query: the reference's two-piece backtrace on the five planes, every test of a step evaluated so that the
// certificate (exactly one candidate, no phantom below the target of an open test) can be decided.
template <typename T>
__global__ __launch_bounds__(64) void poa2_traceback_kernel(TwoPieceParams P) {
    const uint32_t slot = blockIdx.x * 64 + threadIdx.x;
    if (slot >= P.n_queries) return;
    const uint32_t qi = P.first_query + slot;
    const uint64_t qbeg = P.qoff[qi];
    const uint32_t L = (uint32_t)(P.qoff[qi + 1] - qbeg);
    const uint8_t* q = P.qseq + qbeg;
    const uint64_t plane = (uint64_t)P.n_rows * P.pitch;
    const T* base = reinterpret_cast<const T*>(P.planes) + (uint64_t)slot * 5 * plane;
    enum : uint32_t { SM = 0, SI = 1, SD = 2, SI2 = 3, SD2 = 4 };   // plane order
    // dense pass: five planes; replayed search (u32): [row][offset][state] (ExactSearchT::cix)
    auto S = [&](uint32_t row, uint32_t j, uint32_t st) {
        return P.exact_pass ? PlaneIO<T>::get(base + ((uint64_t)row * P.pitch + j) * 5u + st) : PlaneIO<T>::get(base + st * plane + (uint64_t)row * P.pitch + j);
    };
    const uint32_t INF = 0xFFFFFFFFu;
    poa_aln_pair_t* out = P.scratch + (uint64_t)slot * P.scratch_stride;
    uint32_t n_out = 0, fl = 0;
    auto emit = [&](uint32_t rpos, uint32_t qpos) { if (n_out < P.scratch_stride) out[P.scratch_stride - 1 - n_out] = poa_aln_pair_t{rpos, qpos}; n_out++; };
    auto sym_eq = [&](uint32_t row, uint8_t c) { return row == P.end_row || P.rows[row].sym == c; };
    // the cell the walk starts from: (end row, L) in the dense Global pass; where the replayed search stopped otherwise
    uint32_t tb_row = P.end_row, tb_off = L;
    if (P.exact_pass) {
        const uint32_t stt = P.ex_status[qi];
        if (stt != 0) {   // the reference panics ("Could not align sequence!", a Score overflow) / the workspace ran out
            P.flags[qi] = stt == 1 ? POA_FLAG_REF_PANIC : POA_FLAG_EXACT_OVERFLOW;
            P.n_pairs[qi] = 0; P.score[qi] = 0xFFFFFFFFu;
            return;
        }
        tb_row = P.ex_end[2 * qi]; tb_off = P.ex_end[2 * qi + 1];
        P.score[qi] = S(tb_row, tb_off, SM);
    }
    if (L == 0) { P.flags[qi] = 0; P.n_pairs[qi] = 0; return; }
    if (L == 1 && tb_off == 1 && sym_eq(tb_row, q[0])) {   // gap_affine_2piece.rs:952-965 (Global: the end node equals every symbol)
        emit(P.rows[tb_row].node, 0);
        P.flags[qi] = P.exact_pass ? 0u : POA_FLAG_SHORT_QUERY; P.n_pairs[qi] = 1;
        return;
    }
    struct Step { uint32_t row, j, st; bool found; };
    uint32_t nc; bool plt, pn;
    auto step = [&](uint32_t v, uint32_t j, uint32_t st) -> Step {
        Step first{0, 0, SM, false};
        nc = 0; plt = false; pn = false;
        auto sub = [&](uint32_t a, uint32_t b) { const uint32_t r = a - b; if (r == INF) pn = true; return r; };
        auto cand = [&](uint32_t r2, uint32_t j2, uint32_t s2) { if (!first.found) first = Step{r2, j2, s2, true}; nc++; };
        // every load of a Match-state step on a chain row goes out before the first use: one memory round trip instead of
        // three (row record -> predecessor list -> predecessor cell); the walk is a chain of such steps
        const uint32_t cs = S(v, j, st);
        const RowMeta rm = P.rows[v];
        uint32_t up = INF, gd = INF, gd2 = INF, gi = INF, gi2 = INF;
        if (st == SM) {
            if (v > 0 && j > 0) up = S(v - 1, j - 1, SM);
            gd = S(v, j, SD); gd2 = S(v, j, SD2); gi = S(v, j, SI); gi2 = S(v, j, SI2);
        }
        if (cs == INF) return first;
        if (st == SM) {
            if (j > 0) {
                const bool moe = sym_eq(v, q[j - 1]);
                const uint32_t pj = v == P.end_row ? j : j - 1;
                const uint32_t target = (moe || rm.pred_count == 0) ? cs : sub(cs, P.x);
                if ((rm.flags & ROW_CHAIN) && v != P.end_row) {
                    if (up == target) cand(v - 1, pj, SM);
                } else {
                    for (uint32_t k = 0; k < rm.pred_count; ++k) { const uint32_t p = P.pred_rows[rm.pred_begin + k]; if (S(p, pj, SM) == target) cand(p, pj, SM); }
                }
            }
            if (gd == cs) cand(v, j, SD);
            if (gd2 == cs) cand(v, j, SD2);
            if (gi == cs) cand(v, j, SI);
            if (gi2 == cs) cand(v, j, SI2);
        } else if (st == SD) {
            const uint32_t t_open = sub(sub(cs, P.o1), P.e1), t_ext = sub(cs, P.e1);
            const bool real_open = v != P.end_row && (j >= L || rm.sym != q[j]);
            for (uint32_t k = 0; k < rm.pred_count; ++k) {
                const uint32_t p = P.pred_rows[rm.pred_begin + k];
                const uint32_t ps = S(p, j, SM);
                if (ps == t_open) cand(p, j, SM);
                else if (!real_open && ps < t_open) plt = true;
            }
            for (uint32_t k = 0; k < rm.pred_count; ++k) { const uint32_t p = P.pred_rows[rm.pred_begin + k]; if (S(p, j, SD) == t_ext) cand(p, j, SD); }
        } else if (st == SD2) {
            const uint32_t t = sub(cs, P.e2);
            for (uint32_t k = 0; k < rm.pred_count; ++k) { const uint32_t p = P.pred_rows[rm.pred_begin + k]; if (S(p, j, SD) == t) cand(p, j, SD); }
            for (uint32_t k = 0; k < rm.pred_count; ++k) { const uint32_t p = P.pred_rows[rm.pred_begin + k]; if (S(p, j, SD2) == t) cand(p, j, SD2); }
        } else if (st == SI) {
            if (j > 0) {
                const uint32_t t_open = sub(sub(cs, P.o1), P.e1), t_ext = sub(cs, P.e1);
                const uint32_t ps = S(v, j - 1, SM);
                bool open_i = false;
                if (j - 1 < L && v != P.end_row) {
                    if (rm.flags & ROW_OPENI_ALWAYS) open_i = true;
                    else if (rm.flags & ROW_OPENI_NEVER) open_i = false;
                    else open_i = rm.child_sym != q[j - 1];
                }
                if (ps == t_open) cand(v, j - 1, SM);
                else if (!open_i && ps < t_open) plt = true;
                if (S(v, j - 1, SI) == t_ext) cand(v, j - 1, SI);
            }
        } else {
            if (j > 0) {
                const uint32_t t = sub(cs, P.e2);
                if (S(v, j - 1, SI) == t) cand(v, j - 1, SI);
                if (S(v, j - 1, SI2) == t) cand(v, j - 1, SI2);
            }
        }
        return first;
    };
    Step cur = step(tb_row, tb_off, SM);
    bool dead = false;
    if (pn) { fl |= POA_FLAG_REF_PANIC | POA_FLAG_TRUNCATED; dead = true; }
    else if (cur.found && (nc != 1 || plt)) fl |= POA_FLAG_AMBIGUOUS;
    if (!dead && !cur.found) {
        const uint32_t order[4] = {SI, SI2, SD, SD2};   // gap_affine_2piece.rs:972-978
        for (int k = 0; k < 4 && !cur.found && !dead; ++k) {
            cur = step(tb_row, tb_off, order[k]);
            if (pn) { fl |= POA_FLAG_REF_PANIC | POA_FLAG_TRUNCATED; dead = true; }
        }
        if (!dead && !cur.found) { fl |= POA_FLAG_REF_PANIC; dead = true; }
        if (!dead) fl |= POA_FLAG_AMBIGUOUS;
    }
    if (!dead) {
        uint32_t cn = cur.row, cj = cur.j, cst = cur.st;
        bool reached_start = false;
        for (;;) {
            const Step bt = step(cn, cj, cst);
            if (pn) { fl |= POA_FLAG_REF_PANIC; break; }
            if (!bt.found) break;
            if (nc != 1 || plt) fl |= POA_FLAG_AMBIGUOUS;
            if (cst == SM && bt.st != SM) { cn = bt.row; cj = bt.j; cst = bt.st; continue; }
            if (cst == SM) emit(P.rows[cn].node, cj - 1);
            else if (cst == SI || cst == SI2) emit(POA_NONE, cj - 1);
            else emit(P.rows[cn].node, POA_NONE);
            if (bt.st == SM && bt.j == 0 && bt.row != P.start_row && cst != SD && cst != SD2 && sym_eq(bt.row, q[0])) fl |= POA_FLAG_START_QUIRK;
            if (bt.row == P.start_row) { reached_start = true; break; }
            cn = bt.row; cj = bt.j; cst = bt.st;
        }
        if (!reached_start) fl |= POA_FLAG_TRUNCATED;
    }
    // (a replayed table IS the reference's: its backtrace takes the first candidate, nothing to certify)
    P.flags[qi] = P.exact_pass ? (fl & (POA_FLAG_REF_PANIC | POA_FLAG_TRUNCATED)) : fl;
    P.n_pairs[qi] = n_out;
}

}  // namespace poa_amd
