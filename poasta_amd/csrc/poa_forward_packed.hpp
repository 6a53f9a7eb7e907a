// Packed-u16 forward kernel (gfx950): the same recurrences and the same compact plane layout as
// poa_forward_kernel<Q, uint16_t, *, true> (poa_kernels.hpp), but two columns per VGPR and
// per VALU instruction: v_pk_add_u16 clamp / v_pk_min_u16 / v_pk_max_u16 / v_pk_sub_u16.
// INF is 0xFFFF per half and the clamp saturates there, so INF stays absorbing; the launcher only
// selects this kernel when every finite score provably fits 16 bits (see poa_batch_run_ex).
//
// Everything is written with clang's elementwise vector builtins on a 2 x u16 vector; the 0/1 flags use
// forms LLVM leaves alone (1 -sat d), because the obvious ones (umin(d, 1), compares, 0 - flag) are
// canonicalised into per-half v_cmp + v_cndmask + v_perm.  (An inline-asm version of the same primitives
// measured no faster: the compiler pads every asm statement with s_nop and cannot schedule across them.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "poa_kernels.hpp"

namespace poa_amd {

typedef __attribute__((address_space(4))) const uint32_t CPredRows;  // read-only graph table: scalar loads

typedef unsigned short poa_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ poa_u16x2 pkv(uint32_t a) { return __builtin_bit_cast(poa_u16x2, a); }
__device__ __forceinline__ uint32_t pku(poa_u16x2 a) { return __builtin_bit_cast(uint32_t, a); }
// v_pk_add_u16 clamp / v_pk_sub_u16 clamp / v_pk_min_u16 / v_pk_max_u16 / v_pk_lshlrev_b16
__device__ __forceinline__ uint32_t pk_add_sat(uint32_t a, uint32_t b) { return pku(__builtin_elementwise_add_sat(pkv(a), pkv(b))); }
__device__ __forceinline__ uint32_t pk_sub_sat(uint32_t a, uint32_t b) { return pku(__builtin_elementwise_sub_sat(pkv(a), pkv(b))); }
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) { return pku(__builtin_elementwise_min(pkv(a), pkv(b))); }
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) { return pku(__builtin_elementwise_max(pkv(a), pkv(b))); }
template <int S>
__device__ __forceinline__ uint32_t pk_shl(uint32_t a) { return pku(pkv(a) << (unsigned short)S); }
// 1 per half where the halves of a and b are equal, GIVEN a >= b per half (saturating difference is 0 iff equal).
// Written as 1 -sat (a -sat b): LLVM keeps both as v_pk_sub_u16 ... clamp (an explicit umin(d, 1) or a compare
// is canonicalised into per-half v_cmp + v_cndmask + v_perm, five instructions).
__device__ __forceinline__ uint32_t pk_eq_ge(uint32_t a, uint32_t b) { return pk_sub_sat(0x00010001u, pk_sub_sat(a, b)); }
// 1 per half where the half of z is zero
__device__ __forceinline__ uint32_t pk_is_zero(uint32_t z) { return pk_sub_sat(0x00010001u, z); }
// v where sel1 (0/1 per half) is 0, INF (0xFFFF) where it is 1:  max(v, 0 - sel1)   (v_pk_sub_i16 + v_pk_max_u16)
__device__ __forceinline__ uint32_t pk_inf_where(uint32_t v, uint32_t sel1) { return pk_max(v, pku(pkv(0u) - pkv(sel1))); }

// lane l <- x of lane l-1, lane 0 <- fill
__device__ __forceinline__ uint32_t pk_wave_shr1(uint32_t x, uint32_t fill) { return wave_shr1(x, fill); }
// (hi << 16) | (lo >> 16): the packed register shifted right by one column, `lo` supplying the new low half
__device__ __forceinline__ uint32_t shr_col(uint32_t hi, uint32_t lo) { return __builtin_amdgcn_alignbit(hi, lo, 16); }

__device__ __forceinline__ uint32_t umin3(uint32_t a, uint32_t b, uint32_t c) { return umin(umin(a, b), c); }

// inclusive min-plus scan over the lanes, values clamped to the 16-bit INF
__device__ __forceinline__ uint32_t wave_scan_min_plus16(uint32_t t, uint32_t step, uint32_t w15, uint32_t w31) {
    constexpr uint32_t I16 = 0xFFFFu;
    uint32_t P = t;
    P = umin3(P, (uint32_t)__builtin_amdgcn_update_dpp((int)I16, (int)P, 0x111, 0xF, 0xF, false) + step, I16);
    P = umin3(P, (uint32_t)__builtin_amdgcn_update_dpp((int)I16, (int)P, 0x112, 0xF, 0xF, false) + 2 * step, I16);
    P = umin3(P, (uint32_t)__builtin_amdgcn_update_dpp((int)I16, (int)P, 0x114, 0xF, 0xF, false) + 4 * step, I16);
    P = umin3(P, (uint32_t)__builtin_amdgcn_update_dpp((int)I16, (int)P, 0x118, 0xF, 0xF, false) + 8 * step, I16);
    P = umin3(P, (uint32_t)__builtin_amdgcn_update_dpp((int)I16, (int)P, 0x142, 0xA, 0xF, false) + w15, I16);
    P = umin3(P, (uint32_t)__builtin_amdgcn_update_dpp((int)I16, (int)P, 0x143, 0xC, 0xF, false) + w31, I16);
    return P;
}

// FUSE_TB: the wave traces its own query right after its last row (the latency-bound walk then overlaps the other
// waves' VALU-bound forward work instead of running as a separate launch; the traceback needs fewer registers).
//
// MW ("multi-wave"): one WORKGROUP per query, wave w computes strip g*S + w (S = waves per block) — the strips of
// a long query run as a software pipeline, wave w one or more rows behind wave w - 1.  What strip s needs from
// strip s - 1 for row r is three scalars: the insertion-scan carry, I of its last column (for the code bit of my
// first column) and M of its last column (the diagonal edge of the NEXT chain row).  They travel through an LDS
// ring (MW_RING rows deep, with back-pressure) guarded by a per-wave progress counter; fences on that hand-over are
// LDS-only, so no wave ever waits for its plane stores to be acknowledged.  Rows with several predecessors read the
// edge column from the planes instead: the producer's begin-of-row fence of the same graph row has made its earlier
// rows visible before it publishes that row.  Every wait is on a wave that is resident (same workgroup) and strictly
// earlier in the chain, so the pipeline cannot deadlock; spins are bounded all the same.
template <int Q, bool FUSE_TB, bool MW>
__global__ __launch_bounds__(MW ? 1024 : 256) void poa_forward_packed_kernel(FwdParams P, TbParams TP) {
    constexpr int K = 8;                 // columns per lane and quad
    constexpr int NP = 4 * Q;            // packed registers per row array (2 columns each)
    constexpr uint32_t QW = 64 * K;      // 512 columns per quad
    constexpr uint32_t W = QW * Q;
    constexpr uint32_t I16 = 0xFFFFu, INF2 = 0xFFFFFFFFu;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t S = MW ? (blockDim.x >> 6) : 1u;  // strips in flight per query
    __shared__ uint32_t mw_progress[MW ? MW_MAX_WAVES : 1];          // rows completed by wave w, over all its strips
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
    uint16_t* __restrict__ Mp = reinterpret_cast<uint16_t*>(P.planes) + P.plane_off[qi];
    uint16_t* __restrict__ Ip = Mp + RP;  // holds the 4-bit codes (a quarter plane)
    uint16_t* __restrict__ Dp = Ip + RP / 4;  // the kept D rows, row r at slot d_slot[r] (compact_plane_elems)
    uint32_t* __restrict__ carry = P.strip_carry + 2ull * wq * P.n_rows;
    const uint32_t e = P.cost_e, oe = P.cost_oe, x = P.cost_x;
    const uint32_t e2 = e | (e << 16), oe2 = oe | (oe << 16), x2 = x | (x << 16);
    const uint32_t n_strips = (pitch + W - 1) / W;
    const uint32_t step = K * e;
    const uint32_t w15 = ((lane & 15u) + 1u) * step;
    const uint32_t w31 = (lane - 31u) * step;
    const uint32_t lane_off = K * lane * e;
    uint32_t off2[4];  // cost of extending an insertion from my first column of a quad to columns 2i, 2i+1
#pragma unroll
    for (int i = 0; i < 4; ++i) off2[i] = (2 * i * e) | ((2 * i + 1) * e << 16);
    const uint32_t inf2 = INF2;

    const uint32_t n_groups = (n_strips + S - 1) / S;
    for (uint32_t g = 0; g < n_groups; ++g) {
        const uint32_t s = g * S + (MW ? wave : 0u);
        if (MW && s >= n_strips) break;
        const bool from_ring = MW && wave > 0;                          // strip s - 1 is being computed by wave - 1 right now
        const bool from_global = s > 0 && !from_ring;                   // ... or was finished earlier (carry array + planes)
        const bool to_ring = MW && wave + 1 < S && s + 1 < n_strips;
        const bool to_global = s + 1 < n_strips && !to_ring;
        const uint32_t prog_base = g * P.n_rows;
        uint32_t m_edge_prev = I16;                                     // from_ring: M[r-1][sbase-1]
        if (MW && from_global) {
            // wave 0 of a later group: the last wave must have finished (and released) the whole previous group
            mw_wait_gt(&mw_progress[S - 1], prog_base - 1, P.pipeline_error);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        const uint32_t sbase = s * W;
        bool act[Q];
        uint32_t qP[NP];   // my query symbols, 16 bits each (0 past the end: never a symbol)
        uint32_t qlE[Q];   // symbol left of my first column of quad m, in the HIGH half
#pragma unroll
        for (int m = 0; m < Q; ++m) {
            const uint32_t c0 = sbase + m * QW + K * lane;
            act[m] = c0 < pitch;
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
                const uint32_t c = c0 + 2 * pp;
                const uint32_t lo = (c < L) ? (uint32_t)q[c] : 0u, hi = (c + 1 < L) ? (uint32_t)q[c + 1] : 0u;
                qP[4 * m + pp] = lo | (hi << 16);
            }
            qlE[m] = ((c0 > 0 && c0 - 1 < L) ? (uint32_t)q[c0 - 1] : 0u) << 16;
        }
        uint32_t Mprev[NP], Dprev[NP];
        uint32_t PMc[NP], PDc[NP], PMlc[Q];  // predecessor minima of the last multi-predecessor row (ROW_SAME_PREDS reuses them)
#pragma unroll
        for (int p = 0; p < NP; ++p) { Mprev[p] = inf2; Dprev[p] = inf2; PMc[p] = inf2; PDc[p] = inf2; }
#pragma unroll
        for (int m = 0; m < Q; ++m) PMlc[m] = inf2;

        for (uint32_t r = 0; r < P.n_rows; ++r) {
            const RowMeta meta = P.rows[r];
            const uint32_t sym = meta.sym;
            const uint32_t sym2 = sym | (sym << 16);
            const uint64_t rbase = (uint64_t)r * pitch + sbase + K * lane;
            uint32_t PMl[Q];  // hi half = min over predecessors of M[p][my first column - 1]
            uint32_t in_cq = I16, in_ilast = I16, in_mlast = I16;
            if (from_ring) {
                mw_wait_gt(&mw_progress[wave - 1], prog_base + r, P.pipeline_error);
                const uint32_t* slot = mw_ring[wave - 1][(prog_base + r) % MW_RING];
                in_cq = slot[0]; in_ilast = slot[1]; in_mlast = slot[2];
            }
            // the row itself, given the predecessor minima PM / PD; the chain path passes the previous row's
            // registers themselves (no copies)
            auto row_body = [&](const uint32_t (&PM)[NP], const uint32_t (&PD)[NP]) {
                uint32_t Mc[NP], Ic[NP], Dc[NP], Hc[NP], PDe[NP];
                uint32_t cq_out = I16;  // scan carry leaving this strip
#pragma unroll
                for (int p = 0; p < NP; ++p) PDe[p] = pk_add_sat(PD[p], e2);
                if (meta.flags & ROW_END) {
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        Dc[p] = PDe[p];
                        Mc[p] = pk_min(PM[p], Dc[p]);
                        Ic[p] = inf2;
                        Hc[p] = Mc[p];
                    }
                } else {
                    // insertion-open rule, branch-free: "always" == the child symbol 0, which no query symbol equals
                    // (past the query end q is 0 too: not open there, as the rule wants).  ROW_OPENI_NEVER only
                    // occurs on the end row, handled above.
                    const uint32_t cs1 = (meta.flags & ROW_OPENI_ALWAYS) ? 0u : (uint32_t)meta.child_sym;
                    const uint32_t csym2 = cs1 | (cs1 << 16);
                    const uint32_t start_keep = ((meta.flags & ROW_START) && sbase == 0 && lane == 0) ? 0xFFFF0000u : 0xFFFFFFFFu;
                    uint32_t Tq[Q];
#pragma unroll
                    for (int m = 0; m < Q; ++m) {
                        uint32_t eq_prev = ((qlE[m] >> 16) == sym) ? 0x00010000u : 0u;  // does the column left of the quad match? (hi half)
                        uint32_t pm_prev = PMl[m];
                        uint32_t t = I16;  // in-lane insertion chain (32-bit scalar in the 16-bit domain)
                        uint32_t iloc[4];
#pragma unroll
                        for (int pp = 0; pp < 4; ++pp) {
                            const int p = 4 * m + pp;
                            const uint32_t eq1 = pk_is_zero(qP[p] ^ sym2);  // 1 where the query symbol equals the row's symbol
                            // D: open a deletion only where the symbols differ (or past the query end, where q is 0)
                            Dc[p] = pk_min(PDe[p], pk_inf_where(pk_add_sat(PM[p], oe2), eq1));
                            const uint32_t pm_s = shr_col(PM[p], pm_prev);  // M of the predecessor(s), one column to the left
                            const uint32_t eq_s = shr_col(eq1, eq_prev);
                            // (mis)match from the left column: + x where the symbols differ (x - (eq << 8) saturates to 0 on a match)
                            Hc[p] = pk_min(pk_add_sat(pm_s, pk_sub_sat(x2, pk_shl<8>(eq_s))), Dc[p]);
                            pm_prev = PM[p]; eq_prev = eq1;
                            if (m == 0 && pp == 0) Hc[p] &= start_keep;  // H[start][0] = 0
                            // insertion open: A = (q != child symbol) ? H + oe : INF
                            const uint32_t a = pk_inf_where(pk_add_sat(Hc[p], oe2), pk_is_zero(qP[p] ^ csym2));
                            const uint32_t a_lo = a & 0xFFFFu, a_hi = a >> 16;
                            const uint32_t i_lo = t;
                            t = umin3(t + e, a_lo, I16);
                            iloc[pp] = i_lo | (t << 16);
                            t = umin3(t + e, a_hi, I16);
                        }
                        Tq[m] = t;
#pragma unroll
                        for (int pp = 0; pp < 4; ++pp) Ic[4 * m + pp] = iloc[pp];
                    }
                    uint32_t cq = from_ring ? in_cq : (from_global ? carry[2 * r] : I16);
#pragma unroll
                    for (int m = 0; m < Q; ++m) {
                        const uint32_t Pm = wave_scan_min_plus16(Tq[m], step, w15, w31);
                        const uint32_t excl = wave_shr1(Pm, I16);
                        const uint32_t cin = umin3(excl, cq + lane_off, I16);
                        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)Pm, 63);
                        cq = umin3(cq + QW * e, total, I16);
                        const uint32_t cin2 = cin | (cin << 16);
#pragma unroll
                        for (int pp = 0; pp < 4; ++pp) Ic[4 * m + pp] = pk_min(Ic[4 * m + pp], pk_add_sat(cin2, off2[pp]));
                    }
                    if (to_global && lane == 0) carry[2 * r] = cq;
                    cq_out = cq;
#pragma unroll
                    for (int p = 0; p < NP; ++p) Mc[p] = pk_min(Hc[p], Ic[p]);
                }

                // 4-bit codes (bit set == predicate holds): I==M, I[j]==I[j-1]+e, D==M, D==PD+e; 8 cells per dword, the
                // nibble of my column k at position (k >> 1) + 4 * (k & 1)  (decoder: tb_code)
                uint32_t* __restrict__ codes = reinterpret_cast<uint32_t*>(Ip) + (uint64_t)r * (pitch / 8) + sbase / 8 + lane;
                const bool keep_d = (meta.flags & ROW_STORE_D) != 0;
                uint32_t edge_i = inf2;
                if (from_ring) edge_i = in_ilast << 16;
                else if (from_global) edge_i = carry[2 * r + 1] << 16;
#pragma unroll
                for (int m = 0; m < Q; ++m) {
                    uint32_t i_prev = pk_wave_shr1(Ic[4 * m + 3], edge_i);
                    edge_i = (uint32_t)__builtin_amdgcn_readlane((int)Ic[4 * m + 3], 63);
                    uint32_t fA = 0, fB = 0, fC = 0, fD = 0;  // flag of pair pp at bit 4*pp (even column) and 16 + 4*pp (odd column)
#pragma unroll
                    for (int pp = 0; pp < 4; ++pp) {
                        const int p = 4 * m + pp;
                        const uint32_t i_left = shr_col(Ic[p], i_prev);
                        i_prev = Ic[p];
                        // equality flags (0/1 per half); in every pair below lhs >= rhs holds by construction
                        fA |= pk_eq_ge(Ic[p], Mc[p]) << (4 * pp);                           // I == M   (M = min(H, I) <= I)
                        fB |= pk_eq_ge(pk_add_sat(i_left, e2), Ic[p]) << (4 * pp);          // I[j] == I[j-1] + e
                        fC |= pk_eq_ge(Dc[p], Mc[p]) << (4 * pp);                           // D == M   (M <= H <= D)
                        fD |= pk_eq_ge(PDe[p], Dc[p]) << (4 * pp);                          // D == PD + e
                    }
                    const uint32_t word = fA | (fB << 1) | (fC << 2) | (fD << 3);
                    if (act[m]) {
                        *reinterpret_cast<uint4*>(Mp + rbase + m * QW) = make_uint4(Mc[4 * m], Mc[4 * m + 1], Mc[4 * m + 2], Mc[4 * m + 3]);
                        if (keep_d)
                            *reinterpret_cast<uint4*>(Dp + (uint64_t)(P.d_slot ? P.d_slot[r] : r) * pitch + sbase + K * lane + m * QW) = make_uint4(Dc[4 * m], Dc[4 * m + 1], Dc[4 * m + 2], Dc[4 * m + 3]);
                        codes[m * (QW / 8)] = word;
                    }
                }
                if (to_global && lane == 63) carry[2 * r + 1] = Ic[NP - 1] >> 16;
                if (MW) {
                    if (to_ring) {
                        // back-pressure: the slot I am about to overwrite was read MW_RING rows ago
                        // (the consumer may still look ROW_NEAR rows back from the row it is working on)
                        if (prog_base + r + ROW_NEAR >= MW_RING) mw_wait_gt(&mw_progress[wave + 1], prog_base + r + ROW_NEAR - MW_RING, P.pipeline_error);
                        if (lane == 63) {
                            uint32_t* slot = mw_ring[wave][(prog_base + r) % MW_RING];
                            slot[0] = cq_out; slot[1] = Ic[NP - 1] >> 16; slot[2] = Mc[NP - 1] >> 16;
                        }
                    }
                    // the last row of a strip whose successor reads the carry ARRAY must also release my global stores
                    if (to_global && r + 1 == P.n_rows) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
                    if (lane == 0) __hip_atomic_store(&mw_progress[wave], prog_base + r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
#pragma unroll
                for (int p = 0; p < NP; ++p) { Mprev[p] = Mc[p]; Dprev[p] = Dc[p]; }
            };

            if (meta.flags & ROW_CHAIN) {
                uint32_t edge = inf2;
                if (from_ring) edge = m_edge_prev << 16;
                else if (from_global) edge = (uint32_t)Mp[(uint64_t)(r - 1) * pitch + sbase - 1] << 16;
#pragma unroll
                for (int m = 0; m < Q; ++m) {
                    PMl[m] = pk_wave_shr1(Mprev[4 * m + 3], edge);
                    edge = (uint32_t)__builtin_amdgcn_readlane((int)Mprev[4 * m + 3], 63);
                }
                row_body(Mprev, Dprev);
            } else if (meta.flags & ROW_SAME_PREDS) {
                // a sibling of the previous row (same predecessor set): its predecessor minima are still in registers
#pragma unroll
                for (int m = 0; m < Q; ++m) PMl[m] = PMlc[m];
                row_body(PMc, PDc);
            } else {
                uint32_t (&PM)[NP] = PMc;
                uint32_t (&PD)[NP] = PDc;
#pragma unroll
                for (int p = 0; p < NP; ++p) { PM[p] = inf2; PD[p] = inf2; }
#pragma unroll
                for (int m = 0; m < Q; ++m) PMl[m] = inf2;
                // I read back rows this wave stored itself: wavefront scope orders that.  The column left of my strip
                // was stored by wave - 1 (MW): near rows come through the LDS ring, a far one needs workgroup scope —
                // on the same graph row of BOTH waves, so that the producer has released before it published the row.
                if (MW && (meta.flags & ROW_FAR_PRED)) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                else if (meta.pred_count > 0) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                // Predecessors in batches of PB: all row indices first (scalar loads), then every plane load of the batch,
                // then the reduction — one memory round-trip per batch instead of two per predecessor (a bubble-rich graph
                // reads 2-4 predecessor rows on every row).
                constexpr int PB = Q == 1 ? 4 : 2;
                const CPredRows* cpred = (const CPredRows*)P.pred_rows + meta.pred_begin;
                const CPredRows* cpslot = (const CPredRows*)P.pred_dslot + meta.pred_begin;
                for (uint32_t pe0 = 0; pe0 < meta.pred_count; pe0 += PB) {
                    uint32_t prs[PB];
                    bool valid[PB], in_regs[PB];
#pragma unroll
                    for (int b = 0; b < PB; ++b) {
                        valid[b] = pe0 + b < meta.pred_count;
                        prs[b] = valid[b] ? cpred[pe0 + b] : 0u;
                        in_regs[b] = prs[b] + 1 == r;  // the previous row is still in registers
                    }
                    uint4 la[PB][Q], lb[PB][Q];
                    uint32_t edges[PB];
#pragma unroll
                    for (int b = 0; b < PB; ++b) {
                        const uint64_t pbase = (uint64_t)prs[b] * pitch + sbase + K * lane;
                        const uint64_t pbase_d = P.pred_dslot ? (uint64_t)(valid[b] ? cpslot[pe0 + b] : 0u) * pitch + sbase + K * lane : pbase;
#pragma unroll
                        for (int m = 0; m < Q; ++m) {
                            la[b][m] = make_uint4(inf2, inf2, inf2, inf2); lb[b][m] = la[b][m];
                            if (valid[b] && !in_regs[b] && act[m]) {
                                la[b][m] = *reinterpret_cast<const uint4*>(Mp + pbase + m * QW);
                                lb[b][m] = *reinterpret_cast<const uint4*>(Dp + pbase_d + m * QW);
                            }
                        }
                        edges[b] = inf2;
                        if (valid[b]) {
                            if (from_ring && r - prs[b] <= ROW_NEAR) edges[b] = mw_ring[wave - 1][(prog_base + prs[b]) % MW_RING][2] << 16;
                            else if (s > 0) edges[b] = (uint32_t)Mp[(uint64_t)prs[b] * pitch + sbase - 1] << 16;
                        }
                    }
#pragma unroll
                    for (int b = 0; b < PB; ++b) {
                        if (!valid[b]) continue;
                        uint32_t tm[NP], td[NP];
#pragma unroll
                        for (int m = 0; m < Q; ++m) {
                            tm[4 * m] = la[b][m].x; tm[4 * m + 1] = la[b][m].y; tm[4 * m + 2] = la[b][m].z; tm[4 * m + 3] = la[b][m].w;
                            td[4 * m] = lb[b][m].x; td[4 * m + 1] = lb[b][m].y; td[4 * m + 2] = lb[b][m].z; td[4 * m + 3] = lb[b][m].w;
                        }
                        if (in_regs[b]) {
#pragma unroll
                            for (int p = 0; p < NP; ++p) { tm[p] = Mprev[p]; td[p] = Dprev[p]; }
                        }
                        uint32_t edge = edges[b];
#pragma unroll
                        for (int m = 0; m < Q; ++m) {
                            PMl[m] = pk_min(PMl[m], pk_wave_shr1(tm[4 * m + 3], edge));
                            edge = (uint32_t)__builtin_amdgcn_readlane((int)tm[4 * m + 3], 63);
                        }
#pragma unroll
                        for (int p = 0; p < NP; ++p) { PM[p] = pk_min(PM[p], tm[p]); PD[p] = pk_min(PD[p], td[p]); }
                    }
                }
#pragma unroll
                for (int m = 0; m < Q; ++m) PMlc[m] = PMl[m];
                row_body(PM, PD);
            }
            m_edge_prev = in_mlast;
        }
        if (n_strips > 1) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    }
    if (FUSE_TB) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // my plane stores are complete before I read them back
        traceback_wave<uint16_t, true>(TP, qi, lane);
    }
}

}  // namespace poa_amd
