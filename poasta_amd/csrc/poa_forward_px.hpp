// Packed-u16 forward kernel, "pairs across quads" register mapping (gfx950) — the default for queries of
// 512 < pitch <= 1024 columns (one strip), i.e. the headline workload.
//
// Same recurrences and the same compact plane layout as poa_forward_packed_kernel<2> (M plane u16, D rows
// where ROW_STORE_D, 4 flag bits per cell), but the two 16-bit halves of a VGPR no longer hold ADJACENT
// columns: lane l owns columns 8l..8l+7 of quad 0 and 512+8l..512+8l+7 of quad 1, and register k holds
//      lo half = column 8l + k (quad 0),   hi half = column 512 + 8l + k (quad 1).
// Consequences (456 -> ~330 VALU instructions per 1024-cell row):
//   * "the column to the left" of register k is register k-1 — no v_alignbit shifts; only k = 0 needs the
//     neighbour lane (one DPP wave_shr:1 serves both quads, lane 0's hi half = lane 63's lo half);
//   * the in-lane insertion chain is a plain packed recurrence over k (2 instructions per register instead of 7),
//     both quads advancing together, and ONE packed DPP scan replaces the two per-quad scans;
//   * stores/loads repack with v_perm_b32 (one per dword), the only new cost.
// Flag bits are kept as bit-planes: per 8 columns one dword = [A bits | B bits | C bits | D bits], bit k of a byte =
// column k (TbParams::code_fmt = 1; the nibble format of the other kernels is code_fmt 0).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "poa_forward_packed.hpp"

namespace poa_amd {

typedef uint32_t poa_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(4))) const poa_u32x4 CRowWords;  // a RowMeta as four dwords, constant address space
static_assert(sizeof(RowMeta) == 16, "RowMeta is read as one 16-byte scalar load");
typedef __attribute__((address_space(4))) const uint32_t CU32;

// inclusive min-plus scan over the lanes on both halves at once; the *_2 constants are packed per-lane weights
__device__ __forceinline__ uint32_t wave_scan_min_plus_pk(uint32_t t, uint32_t step2, uint32_t w15_2, uint32_t w31_2) {
    constexpr uint32_t INF2 = 0xFFFFFFFFu;
    uint32_t P = t;
    P = pk_min(P, pk_add_sat((uint32_t)__builtin_amdgcn_update_dpp((int)INF2, (int)P, 0x111, 0xF, 0xF, false), step2));
    P = pk_min(P, pk_add_sat((uint32_t)__builtin_amdgcn_update_dpp((int)INF2, (int)P, 0x112, 0xF, 0xF, false), 2 * step2));
    P = pk_min(P, pk_add_sat((uint32_t)__builtin_amdgcn_update_dpp((int)INF2, (int)P, 0x114, 0xF, 0xF, false), 4 * step2));
    P = pk_min(P, pk_add_sat((uint32_t)__builtin_amdgcn_update_dpp((int)INF2, (int)P, 0x118, 0xF, 0xF, false), 8 * step2));
    P = pk_min(P, pk_add_sat((uint32_t)__builtin_amdgcn_update_dpp((int)INF2, (int)P, 0x142, 0xA, 0xF, false), w15_2));
    P = pk_min(P, pk_add_sat((uint32_t)__builtin_amdgcn_update_dpp((int)INF2, (int)P, 0x143, 0xC, 0xF, false), w31_2));
    return P;
}

// The same scan on complemented values (c = 0xFFFF - v per half): min becomes max, the saturating add a saturating subtract,
// and INF becomes 0 — which is what a DPP move hands a lane without a source under bound_ctrl, so no register has to be
// preset to INF before every step (a v_mov and two wait states each).  All rows take part in the two broadcast steps
// (row_mask 0xF): a row that the classic scan masks out receives a total it already holds or gets again later, harmless
// under an idempotent max; the *_2 weights are the same per-lane constants (0 where a lane must not receive keeps it out:
// callers pass 0xFFFF there, see w31c_2).
__device__ __forceinline__ uint32_t wave_scan_max_minus_pk(uint32_t c, uint32_t step2, uint32_t w15_2, uint32_t w31c_2) {
    uint32_t P = c;
    P = pk_max(P, pk_sub_sat((uint32_t)__builtin_amdgcn_update_dpp(0, (int)P, 0x111, 0xF, 0xF, true), step2));
    P = pk_max(P, pk_sub_sat((uint32_t)__builtin_amdgcn_update_dpp(0, (int)P, 0x112, 0xF, 0xF, true), 2 * step2));
    P = pk_max(P, pk_sub_sat((uint32_t)__builtin_amdgcn_update_dpp(0, (int)P, 0x114, 0xF, 0xF, true), 4 * step2));
    P = pk_max(P, pk_sub_sat((uint32_t)__builtin_amdgcn_update_dpp(0, (int)P, 0x118, 0xF, 0xF, true), 8 * step2));
    P = pk_max(P, pk_sub_sat((uint32_t)__builtin_amdgcn_update_dpp(0, (int)P, 0x142, 0xF, 0xF, true), w15_2));
    P = pk_max(P, pk_sub_sat((uint32_t)__builtin_amdgcn_update_dpp(0, (int)P, 0x143, 0xF, 0xF, true), w31c_2));
    return P;
}

// {a.lo, b.lo} -> (a.lo | b.lo << 16);  {a.hi, b.hi} -> (a.hi | b.hi << 16)
__device__ __forceinline__ uint32_t pk_lo_lo(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }
__device__ __forceinline__ uint32_t pk_hi_hi(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// MF = 2 (every relevant score < 0x0FFF): all four flags ride in the stored M value — bits 12..15 = I[j]==I[j-1]+e, D==PD+e,
// I==M, D==M over a 12-bit score (0x0FFF = INF) — and no flag words are written at all (TbParams::code_fmt 3): a tenth less
// HBM traffic for a kernel whose stores cost a third of its time.
// MF = 1 (needs every relevant score < 0x3FFF, decided by the launcher from the bound on the optimal score): the two flags a
// Match-state traceback step reads — I == M, D == M — are stored in bits 14 and 15 of the M value itself (0x3FFF = INF,
// larger finite values of irrelevant cells are clamped to it), and the flag plane keeps only the two gap-state flags
// (one dword per lane and row).  The traceback then needs ONE load per diagonal step instead of two.
template <int MF>
__global__ __launch_bounds__(256) void poa_forward_px_kernel(FwdParams P) {
    constexpr int K = 8;                 // columns per lane and quad == packed registers per row array
    constexpr uint32_t QW = 64 * K;      // 512 columns per quad
    constexpr uint32_t I16 = 0xFFFFu, INF2 = 0xFFFFFFFFu;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wq = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;  // wave-uniform
    if (wq >= P.n_queries) return;
    const uint32_t qi = P.first_query + wq;
    const uint64_t qbeg = P.qoff[qi];
    const uint32_t L = (uint32_t)(P.qoff[qi + 1] - qbeg);
    const uint8_t* __restrict__ q = P.qseq + qbeg;
    const uint32_t pitch = P.pitch[qi];  // <= 1024 (launcher: one strip)
    const uint64_t RP = (uint64_t)P.n_rows * pitch;
    uint16_t* __restrict__ Mp = reinterpret_cast<uint16_t*>(P.planes) + P.plane_off[qi];
    uint16_t* __restrict__ Ip = Mp + RP;  // holds the flag bit-planes (a quarter plane)
    uint16_t* __restrict__ Dp = Ip + RP / 4;  // the kept D rows, row r at slot d_slot[r] (compact_plane_elems)
    // deletion and insertion costs apart (they differ under the relative encoding, FwdParams::cost_de ..); the end row's
    // deletion always costs the plain e
    const uint32_t e = P.cost_ie, x = P.cost_x;
    const uint32_t e2 = e | (e << 16), ioe2 = P.cost_ioe | (P.cost_ioe << 16), x2 = x | (x << 16);
    const uint32_t de2 = P.cost_de | (P.cost_de << 16), doe2 = P.cost_doe | (P.cost_doe << 16), eend2 = P.cost_e | (P.cost_e << 16);
    auto pack16 = [](uint32_t v) { v = v < I16 ? v : I16; return v | (v << 16); };
    const uint32_t step = K * e;
    const uint32_t step2 = pack16(step);
    const uint32_t w15_2 = pack16(((lane & 15u) + 1u) * step);
    const uint32_t w31_2 = pack16(lane >= 32u ? (lane - 31u) * step : 0xFFFFu);  // only lanes >= 32 receive the row_bcast:31 value (complemented scan: INF keeps the others out)
    const uint32_t lane_off2 = pack16(K * lane * e);
    const uint32_t c_lo = K * lane, c_hi = QW + K * lane;
    const bool act_lo = c_lo < pitch, act_hi = c_hi < pitch;  // my 8 columns of the quad lie inside the plane row
    uint32_t qP[K];                     // my query symbols: lo = q[c_lo + k], hi = q[c_hi + k] (0 past the end: never a symbol)
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const uint32_t a = (c_lo + k < L) ? (uint32_t)q[c_lo + k] : 0u, b = (c_hi + k < L) ? (uint32_t)q[c_hi + k] : 0u;
        qP[k] = a | (b << 16);
    }
    // symbols left of my first columns
    const uint32_t qlE = ((c_lo > 0 && c_lo - 1 < L) ? (uint32_t)q[c_lo - 1] : 0u) | (((c_hi - 1 < L) ? (uint32_t)q[c_hi - 1] : 0u) << 16);

    // Symbol masks per lane, staged once in LDS: mask[s][k] = 0xFFFF per half where my query symbol equals symbol s
    // ("ACGT"[s]).  A row then fetches the masks of its symbol and of its child symbol
    // (4 ds_read_b128) instead of recomputing them (7 VALU instructions per register).  A fifth, all-zero table stands for
    // "no child symbol: an insertion opens everywhere", so that the common row takes one branch for both masks.  Other
    // symbols: computed.
    __shared__ uint4 sym_tab[4 * 5 * 2 * 64];  // 40 KB per block of four waves: four blocks per CU (what the registers allow anyway)
    uint4* my_tab = sym_tab + (threadIdx.x >> 6) * (5 * 2 * 64) + lane;
    {
        const uint32_t letters[4] = {'A', 'C', 'G', 'T'};
#pragma unroll
        for (int si = 0; si < 4; ++si) {
            const uint32_t s2 = letters[si] | (letters[si] << 16);
            uint32_t m[K];
#pragma unroll
            for (int k = 0; k < K; ++k) m[k] = pku(pkv(0u) - pkv(pk_is_zero(qP[k] ^ s2)));
            my_tab[(si * 2 + 0) * 64] = make_uint4(m[0], m[1], m[2], m[3]);
            my_tab[(si * 2 + 1) * 64] = make_uint4(m[4], m[5], m[6], m[7]);
        }
        my_tab[(4 * 2 + 0) * 64] = make_uint4(0u, 0u, 0u, 0u);
        my_tab[(4 * 2 + 1) * 64] = make_uint4(0u, 0u, 0u, 0u);
        // each lane reads back only what it wrote itself: no barrier needed
    }

    // The graph tables are read-only for the whole launch: read them through the constant address space so that
    // they come in over the scalar cache (s_load).  As a plain global load the row record is a VECTOR load, and
    // waiting for it (vmcnt) also waits for every plane store of the previous row.
    const CRowWords* crows = (const CRowWords*)P.rows;
    const CU32* cpred = (const CU32*)P.pred_rows;
    const CU32* cpredk = (const CU32*)P.pred_k;  // relative encoding: e * pred_k is added to what a predecessor hands over
    const CU32* cslot = (const CU32*)P.d_slot;
    const CU32* cpslot = (const CU32*)P.pred_dslot;

    // predecessor minima of the last multi-predecessor row: sibling rows (ROW_SAME_PREDS) reuse them
    uint32_t PMc[K], PDc[K], PMlc = INF2;
#pragma unroll
    for (int k = 0; k < K; ++k) { PMc[k] = INF2; PDc[k] = INF2; }

    // one row: reads the previous row from (Mprev, Dprev), leaves this row in (Mout, Dout) — the caller alternates two
    // register sets so that no row ends with 16 register copies
    // the row record is read one row ahead (four scalar registers): its s_load is not waited for at the head of the row
    poa_u32x4 mw_ahead = crows[0];
    auto do_row = [&](const uint32_t r, const uint32_t (&Mprev)[K], const uint32_t (&Dprev)[K], uint32_t (&Mout)[K], uint32_t (&Dout)[K]) {
        const poa_u32x4 mw = mw_ahead;  // {node, pred_begin, pred_count, sym | child_sym << 8 | flags << 16 | sym_idx << 24}
        mw_ahead = crows[r + 1 < P.n_rows ? r + 1 : r];
        struct { uint32_t pred_begin, pred_count, sym, child_sym, flags, sym_idx; } meta{mw.y, mw.z, mw.w & 0xFFu, (mw.w >> 8) & 0xFFu, (mw.w >> 16) & 0xFFu, mw.w >> 24};
        const uint32_t sym = meta.sym;
        const uint32_t sym2 = sym | (sym << 16);
        const uint64_t rbase = (uint64_t)r * pitch + K * lane;
        uint32_t PMl = INF2;  // min over predecessors of M[p][my first column - 1], both quads

        // lane l <- v of lane l-1; lane 0: lo half <- INF (no column -1), hi half <- lane 63's lo half (column 511)
        auto shr_lane = [&](uint32_t v) {
            const uint32_t last = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
            return pk_wave_shr1(v, I16 | (last << 16));
        };

        auto row_body = [&](const uint32_t (&PM)[K], const uint32_t (&PD)[K]) {
            uint32_t (&Mc)[K] = Mout;
            uint32_t (&Dc)[K] = Dout;
            uint32_t Ic[K], PDe[K], fDv[K];   // fDv: flag D == PD + e, per half
            const uint32_t de_row2 = (meta.flags & ROW_END) ? eend2 : de2;
#pragma unroll
            for (int k = 0; k < K; ++k) PDe[k] = pk_add_sat(PD[k], de_row2);
            if (meta.flags & ROW_END) {
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    Dc[k] = PDe[k];
                    Mc[k] = pk_min(PM[k], Dc[k]);
                    Ic[k] = INF2;
                    fDv[k] = 0x00010001u;
                }
            } else {
                // insertion-open rule, branch-free: "always" == the child symbol 0, which no query symbol equals
                const uint32_t cs1 = (meta.flags & ROW_OPENI_ALWAYS) ? 0u : (uint32_t)meta.child_sym;
                const uint32_t start_keep = ((meta.flags & ROW_START) && lane == 0) ? 0xFFFF0000u : 0xFFFFFFFFu;
                // masks (0xFFFF per half): mD where my query symbol equals the row's symbol, mI where it equals the child symbol
                uint32_t mD[K], mI[K];
                const uint32_t si = meta.sym_idx & 15u, ci = meta.sym_idx >> 4;  // set by build_flat_graph
                if ((meta.sym_idx & 0x88u) == 0) {   // both symbols among ACGT (or no child symbol, table 4): the common row
                    const uint4 a = my_tab[(si * 2 + 0) * 64], b = my_tab[(si * 2 + 1) * 64];
                    const uint4 c = my_tab[(ci * 2 + 0) * 64], d = my_tab[(ci * 2 + 1) * 64];
                    mD[0] = a.x; mD[1] = a.y; mD[2] = a.z; mD[3] = a.w; mD[4] = b.x; mD[5] = b.y; mD[6] = b.z; mD[7] = b.w;
                    mI[0] = c.x; mI[1] = c.y; mI[2] = c.z; mI[3] = c.w; mI[4] = d.x; mI[5] = d.y; mI[6] = d.z; mI[7] = d.w;
                } else {
                    const uint32_t csym2 = cs1 | (cs1 << 16);
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        mD[k] = pku(pkv(0u) - pkv(pk_is_zero(qP[k] ^ sym2)));
                        mI[k] = pku(pkv(0u) - pkv(pk_is_zero(qP[k] ^ csym2)));
                    }
                }
                // The eight columns of a lane are independent until the insertion chain, and gfx950 wants a wait state between
                // a packed-math result and a packed-math use of it: the recurrences are therefore written one OPERATION at a
                // time over all eight columns (the scheduler is told not to move instructions across the steps), so that a
                // dependent pair is seven instructions apart instead of adjacent (~45 s_nop per row otherwise).
                uint32_t Hc[K], u[K], h1[K];
                // (mis)match cost of the column to the left: x where its symbol differs, 0 on a match (x -sat 0xFFFF)
                const uint32_t cost_left0 = pk_sub_sat(x2, pku(pkv(0u) - pkv(pk_is_zero(qlE ^ sym2))));
#pragma unroll
                for (int k = 0; k < K; ++k) u[k] = pk_add_sat(PM[k], doe2);
                h1[0] = cost_left0;
#pragma unroll
                for (int k = 1; k < K; ++k) h1[k] = pk_sub_sat(x2, mD[k - 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < K; ++k) u[k] = pk_max(u[k], mD[k]);   // D: open a deletion only where the symbols differ (or past the query end, where q is 0)
                h1[0] = pk_add_sat(PMl, h1[0]);
#pragma unroll
                for (int k = 1; k < K; ++k) h1[k] = pk_add_sat(PM[k - 1], h1[k]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < K; ++k) Dc[k] = pk_min(PDe[k], u[k]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < K; ++k) Hc[k] = pk_min(h1[k], Dc[k]);
                Hc[0] &= start_keep;  // H[start][0] = 0
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < K; ++k) u[k] = pk_add_sat(Hc[k], ioe2);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < K; ++k) u[k] = pk_max(u[k], mI[k]);   // insertion open: A = (q != child symbol) ? H + oe : INF
                __builtin_amdgcn_sched_barrier(0);
                // in-lane insertion chain, both quads at once; the deletion flag (independent of it) fills its wait states
                uint32_t t = INF2;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const uint32_t te = pk_add_sat(t, e2);
                    const uint32_t dd = pk_sub_sat(PDe[k], Dc[k]);
                    Ic[k] = t;
                    t = pk_min(te, u[k]);
                    fDv[k] = pk_sub_sat(0x00010001u, dd);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const uint32_t Pm = ~wave_scan_max_minus_pk(~t, step2, w15_2, w31_2);
                const uint32_t excl = pk_wave_shr1(Pm, INF2);
                // carry entering quad 1 = everything that leaves quad 0 (nothing enters quad 0: single strip)
                const uint32_t total_lo = (uint32_t)__builtin_amdgcn_readlane((int)Pm, 63) & 0xFFFFu;
                const uint32_t cin = pk_min(excl, pk_add_sat(I16 | (total_lo << 16), lane_off2));
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    Ic[k] = pk_min(Ic[k], pk_add_sat(cin, (uint32_t)k * e2));
                    Mc[k] = pk_min(Hc[k], Ic[k]);
                }
            }

            // flag bit-planes: A: I == M, B: I[j] == I[j-1] + e, C: D == M, D: D == PD + e  (lhs >= rhs by construction)
            uint32_t i_left = shr_lane(Ic[K - 1]);
            uint32_t accA = 0, accB = 0, accC = 0, accD = 0;  // bit k (quad 0) and bit 16 + k (quad 1)
            uint32_t Ms[K];                                   // what is stored for M
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const uint32_t fA = pk_eq_ge(Ic[k], Mc[k]), fC = pk_eq_ge(Dc[k], Mc[k]);
                const uint32_t fB = pk_eq_ge(pk_add_sat(i_left, e2), Ic[k]), fD = fDv[k];
                if (MF == 2) {
                    Ms[k] = (fC << 15) | ((fA << 14) | ((fD << 13) | ((fB << 12) | pk_min(Mc[k], 0x0FFF0FFFu))));
                } else if (MF == 1) {
                    Ms[k] = (fC << 15) | ((fA << 14) | pk_min(Mc[k], 0x3FFF3FFFu));
                    accB |= fB << k;
                    accD |= fD << k;
                } else {
                    Ms[k] = Mc[k];
                    accA |= fA << k;
                    accC |= fC << k;
                    accB |= fB << k;
                    accD |= fD << k;
                }
                i_left = Ic[k];
            }
            const uint32_t ab = __builtin_amdgcn_perm(accB, accA, 0x06020400u);  // [A.q0, B.q0, A.q1, B.q1]
            const uint32_t cd = __builtin_amdgcn_perm(accD, accC, 0x06020400u);  // [C.q0, D.q0, C.q1, D.q1]
            uint32_t* __restrict__ codes = reinterpret_cast<uint32_t*>(Ip) + (MF ? (uint64_t)r * mf_code_stride(pitch) : (uint64_t)r * (pitch / 8)) + lane;
            const bool keep_d = (meta.flags & ROW_STORE_D) != 0;
            const uint64_t dbase = keep_d ? (uint64_t)(cslot ? cslot[r] : r) * pitch + K * lane : 0;
            if (act_lo) {
                *reinterpret_cast<uint4*>(Mp + rbase) =
                    make_uint4(pk_lo_lo(Ms[0], Ms[1]), pk_lo_lo(Ms[2], Ms[3]), pk_lo_lo(Ms[4], Ms[5]), pk_lo_lo(Ms[6], Ms[7]));
                if (MF == 1) codes[0] = __builtin_amdgcn_perm(accD, accB, 0x06020400u);  // [B.q0, D.q0, B.q1, D.q1]
                else if (MF == 0) codes[0] = pk_lo_lo(ab, cd);
                if (keep_d)
                    *reinterpret_cast<uint4*>(Dp + dbase) =
                        make_uint4(pk_lo_lo(Dc[0], Dc[1]), pk_lo_lo(Dc[2], Dc[3]), pk_lo_lo(Dc[4], Dc[5]), pk_lo_lo(Dc[6], Dc[7]));
            }
            if (act_hi) {
                *reinterpret_cast<uint4*>(Mp + rbase + QW) =
                    make_uint4(pk_hi_hi(Ms[0], Ms[1]), pk_hi_hi(Ms[2], Ms[3]), pk_hi_hi(Ms[4], Ms[5]), pk_hi_hi(Ms[6], Ms[7]));
                if (!MF) codes[QW / 8] = pk_hi_hi(ab, cd);
                if (keep_d)
                    *reinterpret_cast<uint4*>(Dp + dbase + QW) =
                        make_uint4(pk_hi_hi(Dc[0], Dc[1]), pk_hi_hi(Dc[2], Dc[3]), pk_hi_hi(Dc[4], Dc[5]), pk_hi_hi(Dc[6], Dc[7]));
            }
        };

        if (meta.flags & ROW_CHAIN) {
            PMl = shr_lane(Mprev[K - 1]);
            row_body(Mprev, Dprev);
        } else if (meta.flags & ROW_SAME_PREDS) {
            PMl = PMlc;
            row_body(PMc, PDc);
        } else {
            uint32_t (&PM)[K] = PMc;
            uint32_t (&PD)[K] = PDc;
#pragma unroll
            for (int k = 0; k < K; ++k) { PM[k] = INF2; PD[k] = INF2; }
            if (meta.pred_count > 0) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // I read back rows this wave stored
            for (uint32_t pe = 0; pe < meta.pred_count; ++pe) {
                const uint32_t pr = cpred[meta.pred_begin + pe];
                uint32_t tm[K], td[K];
                if (pr + 1 == r) {
#pragma unroll
                    for (int k = 0; k < K; ++k) { tm[k] = Mprev[k]; td[k] = Dprev[k]; }
                } else {
                    const uint64_t pbase = (uint64_t)pr * pitch + K * lane;
                    const uint64_t pbase_d = cpslot ? (uint64_t)cpslot[meta.pred_begin + pe] * pitch + K * lane : pbase;
                    uint4 m0 = make_uint4(INF2, INF2, INF2, INF2), d0 = m0, m1 = m0, d1 = m0;
                    if (act_lo) {
                        m0 = *reinterpret_cast<const uint4*>(Mp + pbase);
                        d0 = *reinterpret_cast<const uint4*>(Dp + pbase_d);
                    }
                    if (act_hi) {
                        m1 = *reinterpret_cast<const uint4*>(Mp + pbase + QW);
                        d1 = *reinterpret_cast<const uint4*>(Dp + pbase_d + QW);
                    }
                    const uint32_t a0[4] = {m0.x, m0.y, m0.z, m0.w}, a1[4] = {m1.x, m1.y, m1.z, m1.w};
                    const uint32_t b0[4] = {d0.x, d0.y, d0.z, d0.w}, b1[4] = {d1.x, d1.y, d1.z, d1.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        tm[2 * i] = pk_lo_lo(a0[i], a1[i]); tm[2 * i + 1] = pk_hi_hi(a0[i], a1[i]);
                        td[2 * i] = pk_lo_lo(b0[i], b1[i]); td[2 * i + 1] = pk_hi_hi(b0[i], b1[i]);
                    }
                    if (MF) {
                        // strip the flags; the 14-bit (12-bit) INF becomes the 16-bit one again
                        constexpr uint32_t VM = MF == 2 ? 0x0FFF0FFFu : 0x3FFF3FFFu;
#pragma unroll
                        for (int k = 0; k < K; ++k) {
                            const uint32_t v = tm[k] & VM;
                            const uint32_t is_inf = pku(pkv(0u) - pkv(pk_is_zero(v ^ VM)));  // 0xFFFF per half
                            tm[k] = v | (is_inf & ~VM);
                        }
                    }
                }
                if (!MF && cpredk) {
                    const uint32_t ex2 = pack16(cpredk[meta.pred_begin + pe] * P.cost_e);
                    if (ex2) {
#pragma unroll
                        for (int k = 0; k < K; ++k) { tm[k] = pk_add_sat(tm[k], ex2); td[k] = pk_add_sat(td[k], ex2); }
                    }
                }
                PMl = pk_min(PMl, shr_lane(tm[K - 1]));
#pragma unroll
                for (int k = 0; k < K; ++k) { PM[k] = pk_min(PM[k], tm[k]); PD[k] = pk_min(PD[k], td[k]); }
            }
            PMlc = PMl;
            row_body(PM, PD);
        }
    };

    uint32_t MA[K], DA[K], MB[K], DB[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { MA[k] = INF2; DA[k] = INF2; }
    uint32_t r = 0;
    for (; r + 1 < P.n_rows; r += 2) {
        do_row(r, MA, DA, MB, DB);
        do_row(r + 1, MB, DB, MA, DA);
    }
    if (r < P.n_rows) do_row(r, MA, DA, MB, DB);
}


// ---------------------------------------------------------------------------------------------------------------------
// The same register mapping for queries longer than one strip: one WORKGROUP per query, wave w computes the 1024-column
// strip g*S + w, the strips running as the software pipeline described at poa_forward_packed_kernel (LDS hand-over ring
// with back-pressure, LDS-only fences, ROW_NEAR look-back for multi-predecessor rows, far predecessors and later groups
// through the planes / the carry array).  What differs from the one-strip kernel: values enter quad 0 from the previous
// strip (scan carry, I and M of its last column), and leave quad 1 towards the next one.  Flags: bit-planes (code_fmt 1);
// symbol masks are computed (the LDS holds the ring; 16 waves of tables would not fit).
__global__ __launch_bounds__(1024) void poa_forward_pxmw_kernel(FwdParams P) {
    constexpr int K = 8;
    constexpr uint32_t QW = 64 * K, W = 2 * QW;
    constexpr uint32_t I16 = 0xFFFFu, INF2 = 0xFFFFFFFFu;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t S = blockDim.x >> 6;
    __shared__ uint32_t mw_progress[MW_MAX_WAVES];
    __shared__ uint32_t mw_ring[MW_MAX_WAVES][MW_RING][4];  // {scan carry, I last col, M last col, -}
    if (threadIdx.x < MW_MAX_WAVES) mw_progress[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t wq = blockIdx.x;
    if (wq >= P.n_queries) return;
    const uint32_t qi = P.first_query + wq;
    const uint64_t qbeg = P.qoff[qi];
    const uint32_t L = (uint32_t)(P.qoff[qi + 1] - qbeg);
    const uint8_t* __restrict__ q = P.qseq + qbeg;
    const uint32_t pitch = P.pitch[qi];
    const uint64_t RP = (uint64_t)P.n_rows * pitch;
    uint16_t* __restrict__ Mp = reinterpret_cast<uint16_t*>(P.planes) + P.plane_off[qi];
    uint16_t* __restrict__ Ip = Mp + RP;
    uint16_t* __restrict__ Dp = Ip + RP / 4;  // see poa_forward_px_kernel
    uint32_t* __restrict__ carry = P.strip_carry + 2ull * wq * P.n_rows;
    const uint32_t e = P.cost_ie, x = P.cost_x;   // see poa_forward_px_kernel
    const uint32_t e2 = e | (e << 16), ioe2 = P.cost_ioe | (P.cost_ioe << 16), x2 = x | (x << 16);
    const uint32_t de2 = P.cost_de | (P.cost_de << 16), doe2 = P.cost_doe | (P.cost_doe << 16), eend2 = P.cost_e | (P.cost_e << 16);
    auto pack16 = [](uint32_t v) { v = v < I16 ? v : I16; return v | (v << 16); };
    auto clamp16 = [](uint32_t v) { return v < I16 ? v : I16; };
    const uint32_t step = K * e;
    const uint32_t step2 = pack16(step);
    const uint32_t w15_2 = pack16(((lane & 15u) + 1u) * step);
    const uint32_t w31_2 = pack16(lane >= 32u ? (lane - 31u) * step : 0xFFFFu);  // (complemented scan: INF keeps the lower lanes out)
    const uint32_t lane_off2 = pack16(K * lane * e);
    const uint32_t n_strips = (pitch + W - 1) / W;
    const uint32_t n_groups = (n_strips + S - 1) / S;
    const CRowWords* crows = (const CRowWords*)P.rows;
    const CU32* cpred = (const CU32*)P.pred_rows;
    const CU32* cpredk = (const CU32*)P.pred_k;  // relative encoding: e * pred_k is added to what a predecessor hands over
    const CU32* cslot = (const CU32*)P.d_slot;
    const CU32* cpslot = (const CU32*)P.pred_dslot;

    for (uint32_t g = 0; g < n_groups; ++g) {
        const uint32_t s = g * S + wave;
        if (s >= n_strips) break;
        const bool from_ring = wave > 0;
        const bool from_global = s > 0 && !from_ring;
        const bool to_ring = wave + 1 < S && s + 1 < n_strips;
        const bool to_global = s + 1 < n_strips && !to_ring;
        const uint32_t prog_base = g * P.n_rows;
        uint32_t m_edge_prev = I16;  // from_ring: M[r-1][sbase-1]
        if (from_global) {
            mw_wait_gt(&mw_progress[S - 1], prog_base - 1, P.pipeline_error);  // the whole previous group is done and released
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        const uint32_t sbase = s * W;
        const uint32_t c_lo = sbase + K * lane, c_hi = sbase + QW + K * lane;
        const bool act_lo = c_lo < pitch, act_hi = c_hi < pitch;
        uint32_t qP[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t a = (c_lo + k < L) ? (uint32_t)q[c_lo + k] : 0u, b = (c_hi + k < L) ? (uint32_t)q[c_hi + k] : 0u;
            qP[k] = a | (b << 16);
        }
        const uint32_t qlE = ((c_lo > 0 && c_lo - 1 < L) ? (uint32_t)q[c_lo - 1] : 0u) | (((c_hi - 1 < L) ? (uint32_t)q[c_hi - 1] : 0u) << 16);

        // predecessor minima of the last multi-predecessor row: sibling rows (ROW_SAME_PREDS) reuse them
        uint32_t PMc[K], PDc[K], PMlc = INF2;
#pragma unroll
        for (int k = 0; k < K; ++k) { PMc[k] = INF2; PDc[k] = INF2; }

        // the row record is read one row ahead: with two or three waves per SIMD (a chunk of long queries) nothing else hides
        // the scalar load's latency at the head of every row
        poa_u32x4 mw_ahead = crows[0];
        auto do_row = [&](const uint32_t r, const uint32_t (&Mprev)[K], const uint32_t (&Dprev)[K], uint32_t (&Mout)[K], uint32_t (&Dout)[K]) {
            const poa_u32x4 mw = mw_ahead;
            mw_ahead = crows[r + 1 < P.n_rows ? r + 1 : r];
            struct { uint32_t pred_begin, pred_count, sym, child_sym, flags; } meta{mw.y, mw.z, mw.w & 0xFFu, (mw.w >> 8) & 0xFFu, (mw.w >> 16) & 0xFFu};
            const uint32_t sym2 = meta.sym | (meta.sym << 16);
            const uint64_t rbase = (uint64_t)r * pitch + sbase + K * lane;
            uint32_t in_cq = I16, in_ilast = I16, in_mlast = I16;
            if (from_ring) {
                mw_wait_gt(&mw_progress[wave - 1], prog_base + r, P.pipeline_error);
                const uint32_t* slot = mw_ring[wave - 1][(prog_base + r) % MW_RING];
                in_cq = slot[0]; in_ilast = slot[1]; in_mlast = slot[2];
            }
            // lane l <- v of lane l-1; lane 0: lo half <- `left` (last column of the previous strip), hi half <- lane 63's lo half
            auto shr_lane = [&](uint32_t v, uint32_t left) {
                const uint32_t last = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
                return pk_wave_shr1(v, (left & 0xFFFFu) | (last << 16));
            };
            uint32_t PMl = INF2;
            uint32_t cq_out = I16, i_out = I16, m_out = I16;

            auto row_body = [&](const uint32_t (&PM)[K], const uint32_t (&PD)[K]) {
                uint32_t (&Mc)[K] = Mout;
                uint32_t (&Dc)[K] = Dout;
                uint32_t Ic[K], PDe[K], fDv[K];   // fDv: flag D == PD + e, per half
                const uint32_t de_row2 = (meta.flags & ROW_END) ? eend2 : de2;
#pragma unroll
                for (int k = 0; k < K; ++k) PDe[k] = pk_add_sat(PD[k], de_row2);
                if (meta.flags & ROW_END) {
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        Dc[k] = PDe[k];
                        Mc[k] = pk_min(PM[k], Dc[k]);
                        Ic[k] = INF2;
                        fDv[k] = 0x00010001u;
                    }
                } else {
                    const uint32_t cs1 = (meta.flags & ROW_OPENI_ALWAYS) ? 0u : (uint32_t)meta.child_sym;
                    const uint32_t csym2 = cs1 | (cs1 << 16);
                    const uint32_t start_keep = ((meta.flags & ROW_START) && sbase == 0 && lane == 0) ? 0xFFFF0000u : 0xFFFFFFFFu;
                    // one operation at a time over the eight columns, as in poa_forward_px_kernel (no wait states between a
                    // packed result and its packed use); here the symbol masks are computed, the LDS holds the ring
                    uint32_t Hc[K], u[K], h1[K], mD[K], mI[K];
#pragma unroll
                    for (int k = 0; k < K; ++k) { mD[k] = qP[k] ^ sym2; mI[k] = qP[k] ^ csym2; }
                    h1[0] = qlE ^ sym2;
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < K; ++k) { mD[k] = pk_is_zero(mD[k]); mI[k] = pk_is_zero(mI[k]); }   // 1 where the symbols are equal
                    h1[0] = pk_is_zero(h1[0]);
#pragma unroll
                    for (int k = 0; k < K; ++k) u[k] = pk_add_sat(PM[k], doe2);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < K; ++k) { mD[k] = pku(pkv(0u) - pkv(mD[k])); mI[k] = pku(pkv(0u) - pkv(mI[k])); }   // 0xFFFF where equal
                    h1[0] = pku(pkv(0u) - pkv(h1[0]));
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < K; ++k) u[k] = pk_max(u[k], mD[k]);   // D: open a deletion only where the symbols differ
                    h1[0] = pk_sub_sat(x2, h1[0]);
#pragma unroll
                    for (int k = 1; k < K; ++k) h1[k] = pk_sub_sat(x2, mD[k - 1]);   // (mis)match cost of the column to the left
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < K; ++k) Dc[k] = pk_min(PDe[k], u[k]);
                    h1[0] = pk_add_sat(PMl, h1[0]);
#pragma unroll
                    for (int k = 1; k < K; ++k) h1[k] = pk_add_sat(PM[k - 1], h1[k]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < K; ++k) Hc[k] = pk_min(h1[k], Dc[k]);
                    Hc[0] &= start_keep;
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < K; ++k) u[k] = pk_add_sat(Hc[k], ioe2);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < K; ++k) u[k] = pk_max(u[k], mI[k]);   // insertion open: (q != child symbol) ? H + oe : INF
                    __builtin_amdgcn_sched_barrier(0);
                    uint32_t t = INF2;   // in-lane insertion chain; the deletion flag fills its wait states
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const uint32_t te = pk_add_sat(t, e2);
                        __builtin_amdgcn_sched_barrier(0);
                        uint32_t dd = pk_sub_sat(PDe[k], Dc[k]);
                        asm volatile("" : "+v"(dd));   // (keeps the flag arithmetic here: it would otherwise sink to its use)
                        __builtin_amdgcn_sched_barrier(0);
                        Ic[k] = t;
                        t = pk_min(te, u[k]);
                        __builtin_amdgcn_sched_barrier(0);
                        fDv[k] = pk_sub_sat(0x00010001u, dd);
                        asm volatile("" : "+v"(fDv[k]));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    const uint32_t Pm = ~wave_scan_max_minus_pk(~t, step2, w15_2, w31_2);
                    const uint32_t excl = pk_wave_shr1(Pm, INF2);
                    const uint32_t totals = (uint32_t)__builtin_amdgcn_readlane((int)Pm, 63);
                    // scan carries: into quad 0 from the previous strip, into quad 1 from quad 0, out of quad 1 to the next strip
                    const uint32_t cq_lo = from_ring ? in_cq : (from_global ? carry[2 * r] : I16);
                    const uint32_t cq_hi = umin(clamp16(cq_lo + QW * e), totals & 0xFFFFu);
                    cq_out = umin(clamp16(cq_hi + QW * e), totals >> 16);
                    const uint32_t cin = pk_min(excl, pk_add_sat((cq_lo & 0xFFFFu) | (cq_hi << 16), lane_off2));
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        Ic[k] = pk_min(Ic[k], pk_add_sat(cin, (uint32_t)k * e2));
                        Mc[k] = pk_min(Hc[k], Ic[k]);
                    }
                    if (to_global && lane == 0) carry[2 * r] = cq_out;
                }

                // flag bit-planes
                const uint32_t edge_i = from_ring ? in_ilast : (from_global ? carry[2 * r + 1] : I16);
                uint32_t i_left = shr_lane(Ic[K - 1], edge_i);
                uint32_t accA = 0, accB = 0, accC = 0, accD = 0;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    accA |= pk_eq_ge(Ic[k], Mc[k]) << k;
                    accB |= pk_eq_ge(pk_add_sat(i_left, e2), Ic[k]) << k;
                    accC |= pk_eq_ge(Dc[k], Mc[k]) << k;
                    accD |= fDv[k] << k;
                    i_left = Ic[k];
                }
                const uint32_t ab = __builtin_amdgcn_perm(accB, accA, 0x06020400u);
                const uint32_t cd = __builtin_amdgcn_perm(accD, accC, 0x06020400u);
                uint32_t* __restrict__ codes = reinterpret_cast<uint32_t*>(Ip) + (uint64_t)r * (pitch / 8) + sbase / 8 + lane;
                const bool keep_d = (meta.flags & ROW_STORE_D) != 0;
                const uint64_t dbase = keep_d ? (uint64_t)(cslot ? cslot[r] : r) * pitch + sbase + K * lane : 0;
                if (act_lo) {
                    *reinterpret_cast<uint4*>(Mp + rbase) =
                        make_uint4(pk_lo_lo(Mc[0], Mc[1]), pk_lo_lo(Mc[2], Mc[3]), pk_lo_lo(Mc[4], Mc[5]), pk_lo_lo(Mc[6], Mc[7]));
                    codes[0] = pk_lo_lo(ab, cd);
                    if (keep_d)
                        *reinterpret_cast<uint4*>(Dp + dbase) =
                            make_uint4(pk_lo_lo(Dc[0], Dc[1]), pk_lo_lo(Dc[2], Dc[3]), pk_lo_lo(Dc[4], Dc[5]), pk_lo_lo(Dc[6], Dc[7]));
                }
                if (act_hi) {
                    *reinterpret_cast<uint4*>(Mp + rbase + QW) =
                        make_uint4(pk_hi_hi(Mc[0], Mc[1]), pk_hi_hi(Mc[2], Mc[3]), pk_hi_hi(Mc[4], Mc[5]), pk_hi_hi(Mc[6], Mc[7]));
                    codes[QW / 8] = pk_hi_hi(ab, cd);
                    if (keep_d)
                        *reinterpret_cast<uint4*>(Dp + dbase + QW) =
                            make_uint4(pk_hi_hi(Dc[0], Dc[1]), pk_hi_hi(Dc[2], Dc[3]), pk_hi_hi(Dc[4], Dc[5]), pk_hi_hi(Dc[6], Dc[7]));
                }
                // what the next strip needs of this row: I and M of my last column (quad 1, lane 63, register 7)
                i_out = (uint32_t)__builtin_amdgcn_readlane((int)Ic[K - 1], 63) >> 16;
                m_out = (uint32_t)__builtin_amdgcn_readlane((int)Mc[K - 1], 63) >> 16;
                if (to_global && lane == 63) carry[2 * r + 1] = Ic[K - 1] >> 16;
            };

            if (meta.flags & ROW_CHAIN) {
                uint32_t edge = I16;
                if (from_ring) edge = m_edge_prev;
                else if (from_global) edge = (uint32_t)Mp[(uint64_t)(r - 1) * pitch + sbase - 1];
                PMl = shr_lane(Mprev[K - 1], edge);
                row_body(Mprev, Dprev);
            } else if (meta.flags & ROW_SAME_PREDS) {
                PMl = PMlc;
                row_body(PMc, PDc);
            } else {
                uint32_t (&PM)[K] = PMc;
                uint32_t (&PD)[K] = PDc;
#pragma unroll
                for (int k = 0; k < K; ++k) { PM[k] = INF2; PD[k] = INF2; }
                if (meta.flags & ROW_FAR_PRED) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                else if (meta.pred_count > 0) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                for (uint32_t pe = 0; pe < meta.pred_count; ++pe) {
                    const uint32_t pr = cpred[meta.pred_begin + pe];
                    uint32_t tm[K], td[K];
                    if (pr + 1 == r) {
#pragma unroll
                        for (int k = 0; k < K; ++k) { tm[k] = Mprev[k]; td[k] = Dprev[k]; }
                    } else {
                        const uint64_t pbase = (uint64_t)pr * pitch + sbase + K * lane;
                        const uint64_t pbase_d = cpslot ? (uint64_t)cpslot[meta.pred_begin + pe] * pitch + sbase + K * lane : pbase;
                        uint4 m0 = make_uint4(INF2, INF2, INF2, INF2), d0 = m0, m1 = m0, d1 = m0;
                        if (act_lo) {
                            m0 = *reinterpret_cast<const uint4*>(Mp + pbase);
                            d0 = *reinterpret_cast<const uint4*>(Dp + pbase_d);
                        }
                        if (act_hi) {
                            m1 = *reinterpret_cast<const uint4*>(Mp + pbase + QW);
                            d1 = *reinterpret_cast<const uint4*>(Dp + pbase_d + QW);
                        }
                        const uint32_t a0[4] = {m0.x, m0.y, m0.z, m0.w}, a1[4] = {m1.x, m1.y, m1.z, m1.w};
                        const uint32_t b0[4] = {d0.x, d0.y, d0.z, d0.w}, b1[4] = {d1.x, d1.y, d1.z, d1.w};
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            tm[2 * i] = pk_lo_lo(a0[i], a1[i]); tm[2 * i + 1] = pk_hi_hi(a0[i], a1[i]);
                            td[2 * i] = pk_lo_lo(b0[i], b1[i]); td[2 * i + 1] = pk_hi_hi(b0[i], b1[i]);
                        }
                    }
                    uint32_t edge = I16;
                    if (from_ring && r - pr <= ROW_NEAR) edge = mw_ring[wave - 1][(prog_base + pr) % MW_RING][2];
                    else if (s > 0) edge = (uint32_t)Mp[(uint64_t)pr * pitch + sbase - 1];
                    if (cpredk) {
                        const uint32_t ex = clamp16(cpredk[meta.pred_begin + pe] * P.cost_e);
                        if (ex) {
                            const uint32_t ex2 = ex | (ex << 16);
#pragma unroll
                            for (int k = 0; k < K; ++k) { tm[k] = pk_add_sat(tm[k], ex2); td[k] = pk_add_sat(td[k], ex2); }
                            edge = clamp16((edge & 0xFFFFu) + ex);
                        }
                    }
                    PMl = pk_min(PMl, shr_lane(tm[K - 1], edge));
#pragma unroll
                    for (int k = 0; k < K; ++k) { PM[k] = pk_min(PM[k], tm[k]); PD[k] = pk_min(PD[k], td[k]); }
                }
                PMlc = PMl;
                row_body(PM, PD);
            }

            if (to_ring) {
                // back-pressure: the consumer may still look ROW_NEAR rows back from the row it is working on
                if (prog_base + r + ROW_NEAR >= MW_RING) mw_wait_gt(&mw_progress[wave + 1], prog_base + r + ROW_NEAR - MW_RING, P.pipeline_error);
                if (lane == 63) {
                    uint32_t* slot = mw_ring[wave][(prog_base + r) % MW_RING];
                    slot[0] = cq_out; slot[1] = i_out; slot[2] = m_out;
                }
            }
            if (to_global && r + 1 == P.n_rows) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            if (lane == 0) __hip_atomic_store(&mw_progress[wave], prog_base + r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            m_edge_prev = in_mlast;
        };

        uint32_t MA[K], DA[K], MB[K], DB[K];
#pragma unroll
        for (int k = 0; k < K; ++k) { MA[k] = INF2; DA[k] = INF2; }
        uint32_t r = 0;
        for (; r + 1 < P.n_rows; r += 2) {
            do_row(r, MA, DA, MB, DB);
            do_row(r + 1, MB, DB, MA, DA);
        }
        if (r < P.n_rows) do_row(r, MA, DA, MB, DB);
        if (n_strips > 1) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    }
}

}  // namespace poa_amd
