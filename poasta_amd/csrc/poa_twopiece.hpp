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
// One wavefront per query, rows in topological order, 64 columns per pass with the scan carries kept in registers; predecessor
// rows are re-read from the planes (just written by this wave: L2).  u32 arithmetic, INF absorbing.  This is the plain
// kernel of the model — parity first; the packed-u16 pairs-across-quads mapping of the one-piece kernels carries over
// (two more packed recurrences per register) and is the next step for it.
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
    uint32_t* planes;                  // per slot: [5][n_rows][pitch]: M, I1, D1, I2, D2
    uint32_t x, oe, e1, e2, o1;
    uint32_t* score;                   // [total]
    uint32_t* flags;                   // [total]
    uint32_t* n_pairs;                 // [total]
    poa_aln_pair_t* scratch;           // per slot: n_rows + pitch pairs, written back to front
    uint32_t scratch_stride;
};

__device__ __forceinline__ uint32_t tp_sat(uint32_t a, uint32_t b) {
    const uint32_t r = a + b;
    return r < a ? 0xFFFFFFFFu : r;
}

__global__ __launch_bounds__(64) void poa2_forward_kernel(TwoPieceParams P) {
    const uint32_t slot = blockIdx.x, lane = threadIdx.x;
    const uint32_t qi = P.first_query + slot;
    const uint64_t qbeg = P.qoff[qi];
    const uint32_t L = (uint32_t)(P.qoff[qi + 1] - qbeg);
    const uint8_t* q = P.qseq + qbeg;
    const uint64_t plane = (uint64_t)P.n_rows * P.pitch;
    uint32_t* M = P.planes + (uint64_t)slot * 5 * plane;
    uint32_t* I1 = M + plane; uint32_t* D1 = I1 + plane; uint32_t* I2 = D1 + plane; uint32_t* D2 = I2 + plane;
    const uint32_t INF = 0xFFFFFFFFu;
    for (uint32_t r = 0; r < P.n_rows; ++r) {
        const RowMeta rm = P.rows[r];
        const bool is_end = r == P.end_row, is_start = r == P.start_row;
        const uint64_t ro = (uint64_t)r * P.pitch;
        // scan carries: I1 / I2 of the last column done, and what that column would open
        uint32_t cI1 = INF, cI2 = INF, cA = INF;
        for (uint32_t j0 = 0; j0 <= L; j0 += 64) {
            const uint32_t j = j0 + lane;
            const bool in = j <= L;
            const uint8_t qj = (in && j < L) ? q[j] : 0, qjm = (in && j > 0) ? q[j - 1] : 0;
            uint32_t pm = INF, pd = INF, pd2 = INF, pml = INF;
            if (in)
                for (uint32_t k = 0; k < rm.pred_count; ++k) {
                    const uint64_t po = (uint64_t)P.pred_rows[rm.pred_begin + k] * P.pitch + j;
                    pm = min(pm, M[po]); pd = min(pd, D1[po]); pd2 = min(pd2, D2[po]);
                    if (j > 0) pml = min(pml, M[po - 1]);
                }
            uint32_t d1 = tp_sat(pd, P.e1);
            if (!is_end && (j >= L || rm.sym != qj)) d1 = min(d1, tp_sat(pm, P.oe));   // openD: the row mismatches q[j], or the query is exhausted
            const uint32_t d2 = tp_sat(min(pd, pd2), P.e2);
            uint32_t diag = INF;
            if (is_end) diag = pm;                                                      // M[u][j] -> M[end][j], cost 0
            else if (j > 0) diag = tp_sat(pml, rm.sym != qjm ? P.x : 0u);
            uint32_t h = min(diag, min(d1, d2));
            if (is_start && j == 0) h = 0;
            if (!in) h = INF;
            // openI(v, j): an edge to end, or a non-end child that mismatches q[j]  (RowMeta: ALWAYS / NEVER / the one child symbol)
            bool open_i = false;
            if (in && j < L && !is_end) {
                if (rm.flags & ROW_OPENI_ALWAYS) open_i = true;
                else if (rm.flags & ROW_OPENI_NEVER) open_i = false;
                else open_i = rm.child_sym != qj;
            }
            const uint32_t a = open_i ? tp_sat(h, P.oe) : INF;   // what column j opens INTO column j + 1
            // I1[j] = min over k <= j of (B[k] + (j - k) e1), B[k] = what enters column k from k - 1
            const uint32_t a_left = (uint32_t)__shfl_up((int)a, 1, 64);
            uint32_t v1 = lane == 0 ? min(tp_sat(cI1, P.e1), cA) : a_left;
            if (j0 == 0 && lane == 0) v1 = INF;                   // I1[v][0] = INF
            for (uint32_t s = 1; s < 64; s <<= 1) {
                const uint32_t t = (uint32_t)__shfl_up((int)v1, (int)s, 64);
                if (lane >= s) v1 = min(v1, tp_sat(t, s * P.e1));
            }
            // I2[j] = min over k <= j of (B2[k] + (j - k) e2), B2[k] = I1[k-1] + e2
            const uint32_t i1_left = (uint32_t)__shfl_up((int)v1, 1, 64);
            uint32_t v2 = lane == 0 ? tp_sat(min(cI1, cI2), P.e2) : tp_sat(i1_left, P.e2);
            if (j0 == 0 && lane == 0) v2 = INF;
            for (uint32_t s = 1; s < 64; s <<= 1) {
                const uint32_t t = (uint32_t)__shfl_up((int)v2, (int)s, 64);
                if (lane >= s) v2 = min(v2, tp_sat(t, s * P.e2));
            }
            if (is_end) { v1 = INF; v2 = INF; }
            if (in) {
                M[ro + j] = min(h, min(v1, v2));
                I1[ro + j] = v1; D1[ro + j] = d1; I2[ro + j] = v2; D2[ro + j] = d2;
            }
            cI1 = (uint32_t)__shfl((int)v1, 63, 64); cI2 = (uint32_t)__shfl((int)v2, 63, 64); cA = (uint32_t)__shfl((int)a, 63, 64);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // the next rows read this one back (same wave)
    }
    if (lane == 0) P.score[qi] = M[(uint64_t)P.end_row * P.pitch + L];
}

// One lane per query: the reference's two-piece backtrace on the five planes, every test of a step evaluated so that the
// certificate (exactly one candidate, no phantom below the target of an open test) can be decided.
__global__ __launch_bounds__(64) void poa2_traceback_kernel(TwoPieceParams P) {
    const uint32_t slot = blockIdx.x * 64 + threadIdx.x;
    if (slot >= P.n_queries) return;
    const uint32_t qi = P.first_query + slot;
    const uint64_t qbeg = P.qoff[qi];
    const uint32_t L = (uint32_t)(P.qoff[qi + 1] - qbeg);
    const uint8_t* q = P.qseq + qbeg;
    const uint64_t plane = (uint64_t)P.n_rows * P.pitch;
    const uint32_t* base = P.planes + (uint64_t)slot * 5 * plane;
    enum : uint32_t { SM = 0, SI = 1, SD = 2, SI2 = 3, SD2 = 4 };   // plane order
    auto S = [&](uint32_t row, uint32_t j, uint32_t st) { return base[st * plane + (uint64_t)row * P.pitch + j]; };
    const uint32_t INF = 0xFFFFFFFFu;
    poa_aln_pair_t* out = P.scratch + (uint64_t)slot * P.scratch_stride;
    uint32_t n_out = 0, fl = 0;
    auto emit = [&](uint32_t rpos, uint32_t qpos) { if (n_out < P.scratch_stride) out[P.scratch_stride - 1 - n_out] = poa_aln_pair_t{rpos, qpos}; n_out++; };
    auto sym_eq = [&](uint32_t row, uint8_t c) { return row == P.end_row || P.rows[row].sym == c; };
    if (L == 0) { P.flags[qi] = 0; P.n_pairs[qi] = 0; return; }
    if (L == 1) {   // gap_affine_2piece.rs:952-965: the end node equals every symbol
        emit(P.rows[P.end_row].node, 0);
        P.flags[qi] = POA_FLAG_SHORT_QUERY; P.n_pairs[qi] = 1;
        return;
    }
    struct Step { uint32_t row, j, st; bool found; };
    uint32_t nc; bool plt, pn;
    auto step = [&](uint32_t v, uint32_t j, uint32_t st) -> Step {
        Step first{0, 0, SM, false};
        nc = 0; plt = false; pn = false;
        auto sub = [&](uint32_t a, uint32_t b) { const uint32_t r = a - b; if (r == INF) pn = true; return r; };
        auto cand = [&](uint32_t r2, uint32_t j2, uint32_t s2) { if (!first.found) first = Step{r2, j2, s2, true}; nc++; };
        const uint32_t cs = S(v, j, st);
        if (cs == INF) return first;
        const RowMeta rm = P.rows[v];
        if (st == SM) {
            if (j > 0) {
                const bool moe = sym_eq(v, q[j - 1]);
                const uint32_t pj = v == P.end_row ? j : j - 1;
                const uint32_t target = (moe || rm.pred_count == 0) ? cs : sub(cs, P.x);
                for (uint32_t k = 0; k < rm.pred_count; ++k) { const uint32_t p = P.pred_rows[rm.pred_begin + k]; if (S(p, pj, SM) == target) cand(p, pj, SM); }
            }
            if (S(v, j, SD) == cs) cand(v, j, SD);
            if (S(v, j, SD2) == cs) cand(v, j, SD2);
            if (S(v, j, SI) == cs) cand(v, j, SI);
            if (S(v, j, SI2) == cs) cand(v, j, SI2);
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
    Step cur = step(P.end_row, L, SM);
    bool dead = false;
    if (pn) { fl |= POA_FLAG_REF_PANIC | POA_FLAG_TRUNCATED; dead = true; }
    else if (cur.found && (nc != 1 || plt)) fl |= POA_FLAG_AMBIGUOUS;
    if (!dead && !cur.found) {
        const uint32_t order[4] = {SI, SI2, SD, SD2};   // gap_affine_2piece.rs:972-978
        for (int k = 0; k < 4 && !cur.found && !dead; ++k) {
            cur = step(P.end_row, L, order[k]);
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
    P.flags[qi] = fl;
    P.n_pairs[qi] = n_out;
}

}  // namespace poa_amd
