// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
//
// Dense CPU restatement of the alignment graph that the reference's A* searches
// (SURVEY.md §7.0): the min-plus M/I/D recurrences DERIVED from the reference's
// edge set, evaluated for every (node, offset) cell, plus the reference's
// score-based traceback rule applied to the dense planes.  This is the
// executable specification the HIP kernels are checked against plane-by-plane;
// it is itself checked against the literal A* restatement (astar.hpp) in
// tests/test_oracle_dense_vs_astar.py.
//
// Edge set (all paths relative to /root/reference/src/aligner/):
//   M[u][j] -> M[v][j+1]   0 | x      dfa.rs:234-246, scoring/gap_affine.rs:406-411
//   M[u][j] -> M[end][j]   0          dfa.rs:222-228
//   M[u][j] -> I[u][j+1]   o+e        gap_affine.rs:413-421 (a child mismatches), :360-366 (u -> end)
//   M[u][j] -> D[v][j]     o+e        gap_affine.rs:423-429 (v mismatches q[j]), :384-390 (j == L)
//   I[u][j] -> I[u][j+1]   e          gap_affine.rs:313-321
//   D[u][j] -> D[v][j]     e          gap_affine.rs:329-340 (v may be end)
//   I/D[u][j] -> M[u][j]   0          gap_affine.rs:309,:325
// Traceback rule: gap_affine.rs:550-657 + :804-915.
//
// Two-piece model (Costs::two_piece; gap_affine_2piece.rs:292-516, backtrace :639-794, :944-1043): five planes.  A gap opens
// in the first piece exactly as above (o1 + e1); afterwards
//   I[u][j]  -> I[u][j+1]   e1      I[u][j]  -> I2[u][j+1]   e2      I2[u][j] -> I2[u][j+1]   e2     (gap_affine_2piece.rs:352-388)
//   D[u][j]  -> D[v][j]     e1      D[u][j]  -> D2[v][j]     e2      D2[u][j] -> D2[v][j]     e2     (:390-430)
//   I, I2, D, D2 [u][j] -> M[u][j]  0
// (open2 is never charged: the second piece is entered from the first.)
#pragma once
#include <cstdint>
#include <vector>

#include "astar.hpp"
#include "graph.hpp"

namespace poa_oracle {

// certificate / status flags (same bit meaning as include/poasta_amd.h POA_FLAG_*)
enum : uint32_t {
    DF_AMBIGUOUS = 1u << 0,     // some traceback step had != 1 candidate (tie / phantom / I-extend quirk)
    DF_START_QUIRK = 1u << 1,   // path leaves a (v,0,M) cell with sym(v)==q[0] (dfa.rs:146-167 may hide that edge)
    DF_REF_PANIC = 1u << 2,     // the reference would panic (wrap to u32::MAX, no backtrace)
    DF_SHORT_QUERY = 1u << 3,   // len <= 1: reference special cases (gap_affine.rs:808-824)
    DF_TRUNCATED = 1u << 4,     // get_backtrace returned None before reaching the start node
};

inline uint32_t sat_add(uint32_t a, uint32_t b) {
    uint32_t r = a + b;
    return r < a ? 0xFFFFFFFFu : r;
}

struct DenseResult {
    Score score = UNVISITED;
    std::vector<AlignedPair> alignment;
    uint32_t flags = 0;
    uint32_t n_ambiguous_steps = 0;
    // planes [row][col], row = topological rank, pitch = L+1
    std::vector<Score> M, I, D;
    std::vector<Score> I2, D2;  // two-piece model only
    size_t rows = 0, pitch = 0;
};

class DenseAligner {
public:
    const Graph& g;
    Costs costs;
    std::vector<uint32_t> rank;       // node -> row
    std::vector<uint32_t> row_node;   // row -> node
    std::vector<uint8_t> has_end_child;
    std::vector<uint8_t> child_sym;   // common symbol of the non-end children; 0 = none; 0xFF = >= 2 distinct
    std::vector<uint8_t> has_real_child;

    DenseAligner(const Graph& graph, Costs c) : g(graph), costs(c) {
        rank = g.node_ranks();
        row_node = g.topo;
        size_t n = g.node_count_with_start_and_end();
        has_end_child.assign(n, 0); child_sym.assign(n, 0); has_real_child.assign(n, 0);
        for (uint32_t v = 0; v < n; ++v) {
            for (uint32_t c2 : g.succ[v]) {
                if (c2 == g.end) { has_end_child[v] = 1; continue; }
                has_real_child[v] = 1;
                uint8_t s = g.symbol[c2];
                if (s == 0) { child_sym[v] = 0xFF; continue; }  // symbol-less child never matches
                if (child_sym[v] == 0) child_sym[v] = s;
                else if (child_sym[v] != s) child_sym[v] = 0xFF;
            }
        }
    }

    bool mm(uint32_t v, const uint8_t* q, size_t j) const { return !g.is_symbol_equal(v, q[j]); }
    // openI(u,j), j < L: u -> end exists, or some non-end child mismatches q[j]
    bool open_i(uint32_t u, const uint8_t* q, size_t L, size_t j) const {
        if (j >= L) return false;
        if (has_end_child[u]) return true;
        if (!has_real_child[u]) return false;
        if (child_sym[u] == 0xFF) return true;
        return child_sym[u] != q[j];
    }
    bool open_d(uint32_t v, const uint8_t* q, size_t L, size_t j) const { return j >= L || mm(v, q, j); }

    void forward(const uint8_t* q, size_t L, DenseResult& R) const {
        if (costs.two_piece) { forward2(q, L, R); return; }
        size_t rows = row_node.size(), P = L + 1;
        R.rows = rows; R.pitch = P;
        R.M.assign(rows * P, UNVISITED); R.I.assign(rows * P, UNVISITED); R.D.assign(rows * P, UNVISITED);
        const uint32_t x = costs.mismatch, oe = (uint32_t)costs.gap_open + costs.gap_extend, e = costs.gap_extend;
        std::vector<Score> H(P);
        for (size_t r = 0; r < rows; ++r) {
            uint32_t v = row_node[r];
            Score* Mv = &R.M[r * P]; Score* Iv = &R.I[r * P]; Score* Dv = &R.D[r * P];
            if (v == g.end) {
                for (size_t j = 0; j < P; ++j) {
                    Score pm = UNVISITED, pd = UNVISITED;
                    for (uint32_t p : g.pred[v]) {
                        pm = std::min(pm, R.M[rank[p] * P + j]);
                        pd = std::min(pd, R.D[rank[p] * P + j]);
                    }
                    Dv[j] = sat_add(pd, e);
                    Mv[j] = std::min(pm, Dv[j]);
                }
                continue;
            }
            for (size_t j = 0; j < P; ++j) {
                Score pm = UNVISITED, pd = UNVISITED, pml = UNVISITED;
                for (uint32_t p : g.pred[v]) {
                    pm = std::min(pm, R.M[rank[p] * P + j]);
                    pd = std::min(pd, R.D[rank[p] * P + j]);
                    if (j > 0) pml = std::min(pml, R.M[rank[p] * P + j - 1]);
                }
                Score d = sat_add(pd, e);
                if (open_d(v, q, L, j)) d = std::min(d, sat_add(pm, oe));
                Dv[j] = d;
                Score diag = UNVISITED;
                if (j > 0) diag = sat_add(pml, mm(v, q, j - 1) ? x : 0);
                H[j] = std::min(diag, d);
                if (v == g.start && j == 0) H[j] = 0;
            }
            Iv[0] = UNVISITED;
            for (size_t j = 0; j < L; ++j) {
                Score a = open_i(v, q, L, j) ? sat_add(H[j], oe) : UNVISITED;
                Iv[j + 1] = std::min(sat_add(Iv[j], e), a);
            }
            for (size_t j = 0; j < P; ++j) Mv[j] = std::min(H[j], Iv[j]);
        }
        R.score = R.M[rank[g.end] * P + L];
    }

    // Reference traceback rule on the dense planes, evaluating EVERY test of a step so the
    // certificate (exactly one candidate, no phantom below target) can be decided.
    void traceback(const uint8_t* q, size_t L, DenseResult& R) const {
        if (costs.two_piece) { traceback2(q, L, R); return; }
        const size_t P = R.pitch;
        const uint32_t x = costs.mismatch, o = costs.gap_open, e = costs.gap_extend;
        auto Mx = [&](uint32_t node, size_t j) { return R.M[rank[node] * P + j]; };
        auto Ix = [&](uint32_t node, size_t j) { return R.I[rank[node] * P + j]; };
        auto Dx = [&](uint32_t node, size_t j) { return R.D[rank[node] * P + j]; };
        R.alignment.clear();
        if (L == 0) return;
        if (L == 1) {
            // gap_affine.rs:812-824: always [(end, 0)] in global mode (end equals every symbol)
            R.flags |= DF_SHORT_QUERY;
            if (g.is_symbol_equal(g.end, q[0])) { R.alignment.push_back({g.end, 0}); return; }
        }
        struct Step { uint32_t node; size_t j; AlignState st; bool found; };
        // one get_backtrace evaluation; counts candidates
        auto step = [&](uint32_t v, size_t j, AlignState st, uint32_t& n_cand, bool& phantom_lt,
                        bool& panic) -> Step {
            Step first{0, 0, ST_M, false};
            n_cand = 0; phantom_lt = false; panic = false;
            auto sub = [&](Score a, uint32_t b) { uint32_t r = a - b; if (r == UNVISITED) panic = true; return r; };
            auto cand = [&](uint32_t n2, size_t j2, AlignState s2) {
                if (!first.found) first = {n2, j2, s2, true};
                n_cand++;
            };
            if (st == ST_M) {
                Score cs = Mx(v, j);
                if (cs == UNVISITED) return first;
                if (j > 0) {
                    bool moe = g.is_symbol_equal(v, q[j - 1]) || v == g.end;
                    size_t pj = (v == g.end) ? j : j - 1;
                    Score target = (moe || g.pred[v].empty()) ? cs : sub(cs, x);  // evaluated per predecessor in the reference
                    for (uint32_t p : g.pred[v]) if (Mx(p, pj) == target) cand(p, pj, ST_M);
                }
                if (Dx(v, j) == cs) cand(v, j, ST_D);
                if (Ix(v, j) == cs) cand(v, j, ST_I);
            } else if (st == ST_D) {
                Score cs = Dx(v, j);
                if (cs == UNVISITED) return first;
                Score t_open = sub(sub(cs, o), e), t_ext = sub(cs, e);
                bool real_open = (v != g.end) && open_d(v, q, L, j);
                for (uint32_t p : g.pred[v]) {
                    Score ps = Mx(p, j);
                    if (ps == t_open) cand(p, j, ST_M);
                    else if (!real_open && ps < t_open) phantom_lt = true;
                }
                for (uint32_t p : g.pred[v]) if (Dx(p, j) == t_ext) cand(p, j, ST_D);
            } else {
                Score cs = Ix(v, j);
                if (cs == UNVISITED) return first;
                if (j > 0) {
                    Score t_open = sub(sub(cs, o), e), t_ext = sub(cs, e);
                    Score ps = Mx(v, j - 1);
                    if (ps == t_open) cand(v, j - 1, ST_M);
                    else if (!open_i(v, q, L, j - 1) && ps < t_open) phantom_lt = true;
                    if (Ix(v, j - 1) == t_ext) {
                        cand(v, j - 1, ST_M);  // sic: reference returns Match (gap_affine.rs:649)
                        // the hop lands on M[v][j-1]; certify only if it equals I[v][j-1]
                        if (first.found && first.node == v && first.j == j - 1 && n_cand == 1 && Mx(v, j - 1) != Ix(v, j - 1))
                            phantom_lt = true;
                    }
                }
            }
            return first;
        };

        uint32_t nc; bool plt, pn;
        // first hop from the end cell: M, then I, then D (gap_affine.rs:832-835)
        Step cur = step(g.end, L, ST_M, nc, plt, pn);
        if (pn) { R.flags |= DF_REF_PANIC | DF_TRUNCATED; return; }
        if (cur.found && (nc != 1 || plt)) { R.flags |= DF_AMBIGUOUS; R.n_ambiguous_steps++; }
        if (!cur.found) {
            cur = step(g.end, L, ST_I, nc, plt, pn);
            if (!cur.found && !pn) cur = step(g.end, L, ST_D, nc, plt, pn);
            if (pn) { R.flags |= DF_REF_PANIC | DF_TRUNCATED; return; }
            if (!cur.found) {
                R.flags |= DF_REF_PANIC;
                if (L <= 3) for (size_t i = 0; i < L; ++i) R.alignment.push_back({g.end, (uint32_t)i});
                return;
            }
            R.flags |= DF_AMBIGUOUS;
        }
        uint32_t cn = cur.node; size_t cj = cur.j; AlignState cst = cur.st;
        bool reached_start = false;
        for (;;) {
            Step bt = step(cn, cj, cst, nc, plt, pn);
            // the reference dies at this step (u32 wrap onto u32::MAX in a Score subtraction): what was emitted so far
            // stands, nothing after it is defined
            if (pn) { R.flags |= DF_REF_PANIC; break; }
            if (!bt.found) break;
            if (nc != 1 || plt) { R.flags |= DF_AMBIGUOUS; R.n_ambiguous_steps++; }
            if (cst == ST_M && (bt.st == ST_I || bt.st == ST_D)) { cn = bt.node; cj = bt.j; cst = bt.st; continue; }
            if (cst == ST_M) R.alignment.push_back({cn, (uint32_t)cj - 1});
            else if (cst == ST_I) R.alignment.push_back({NONE32, (uint32_t)cj - 1});
            else R.alignment.push_back({cn, NONE32});
            // start-quirk certificate: this step used an out-edge of (bt.node, 0, M) with
            // sym(bt.node) == q[0]; dfa.rs:146-167 can suppress exactly those edges.
            if (bt.st == ST_M && bt.j == 0 && bt.node != g.start && cst != ST_D &&
                g.is_symbol_equal(bt.node, q[0]))
                R.flags |= DF_START_QUIRK;
            if (bt.node == g.start) { reached_start = true; break; }
            cn = bt.node; cj = bt.j; cst = bt.st;
        }
        if (!reached_start) R.flags |= DF_TRUNCATED;
        std::reverse(R.alignment.begin(), R.alignment.end());
    }


    // ---- two-piece model ---------------------------------------------------------------------------------------------
    void forward2(const uint8_t* q, size_t L, DenseResult& R) const {
        size_t rows = row_node.size(), P = L + 1;
        R.rows = rows; R.pitch = P;
        for (auto* pl : {&R.M, &R.I, &R.D, &R.I2, &R.D2}) pl->assign(rows * P, UNVISITED);
        const uint32_t x = costs.mismatch, oe = (uint32_t)costs.gap_open + costs.gap_extend, e1 = costs.gap_extend, e2 = costs.gap_extend2;
        std::vector<Score> H(P);
        for (size_t r = 0; r < rows; ++r) {
            uint32_t v = row_node[r];
            Score* Mv = &R.M[r * P]; Score* Iv = &R.I[r * P]; Score* Dv = &R.D[r * P]; Score* I2v = &R.I2[r * P]; Score* D2v = &R.D2[r * P];
            const bool is_end = v == g.end;
            for (size_t j = 0; j < P; ++j) {
                Score pm = UNVISITED, pd = UNVISITED, pd2 = UNVISITED, pml = UNVISITED;
                for (uint32_t p : g.pred[v]) {
                    pm = std::min(pm, R.M[rank[p] * P + j]);
                    pd = std::min(pd, R.D[rank[p] * P + j]);
                    pd2 = std::min(pd2, R.D2[rank[p] * P + j]);
                    if (j > 0) pml = std::min(pml, R.M[rank[p] * P + j - 1]);
                }
                Score d = sat_add(pd, e1);
                if (!is_end && open_d(v, q, L, j)) d = std::min(d, sat_add(pm, oe));   // a deletion never opens INTO the end row
                Dv[j] = d;
                D2v[j] = std::min(sat_add(pd, e2), sat_add(pd2, e2));
                Score diag = UNVISITED;
                if (is_end) diag = pm;                                                 // M[u][j] -> M[end][j], cost 0
                else if (j > 0) diag = sat_add(pml, mm(v, q, j - 1) ? x : 0);
                H[j] = std::min(diag, std::min(d, D2v[j]));
                if (v == g.start && j == 0) H[j] = 0;
            }
            Iv[0] = UNVISITED; I2v[0] = UNVISITED;
            if (!is_end)
                for (size_t j = 0; j < L; ++j) {
                    Score a = open_i(v, q, L, j) ? sat_add(H[j], oe) : UNVISITED;
                    Iv[j + 1] = std::min(sat_add(Iv[j], e1), a);
                    I2v[j + 1] = std::min(sat_add(Iv[j], e2), sat_add(I2v[j], e2));
                }
            for (size_t j = 0; j < P; ++j) Mv[j] = std::min(H[j], std::min(Iv[j], I2v[j]));
        }
        R.score = R.M[rank[g.end] * P + L];
    }

    // the reference's two-piece traceback rule on the five planes, every test of a step evaluated for the certificate
    void traceback2(const uint8_t* q, size_t L, DenseResult& R) const {
        const size_t P = R.pitch;
        const uint32_t x = costs.mismatch, o1 = costs.gap_open, e1 = costs.gap_extend, e2 = costs.gap_extend2;
        auto PL = [&](AlignState st) -> const std::vector<Score>& {
            return st == ST_M ? R.M : st == ST_I ? R.I : st == ST_D ? R.D : st == ST_I2 ? R.I2 : R.D2;
        };
        auto S = [&](uint32_t node, size_t j, AlignState st) { return PL(st)[rank[node] * P + j]; };
        R.alignment.clear();
        if (L == 0) return;
        if (L == 1) {
            R.flags |= DF_SHORT_QUERY;
            if (g.is_symbol_equal(g.end, q[0])) { R.alignment.push_back({g.end, 0}); return; }
        }
        struct Step { uint32_t node; size_t j; AlignState st; bool found; };
        auto step = [&](uint32_t v, size_t j, AlignState st, uint32_t& n_cand, bool& phantom_lt, bool& panic) -> Step {
            Step first{0, 0, ST_M, false};
            n_cand = 0; phantom_lt = false; panic = false;
            auto sub = [&](Score a, uint32_t b) { uint32_t r = a - b; if (r == UNVISITED) panic = true; return r; };
            auto cand = [&](uint32_t n2, size_t j2, AlignState s2) { if (!first.found) first = {n2, j2, s2, true}; n_cand++; };
            Score cs = S(v, j, st);
            if (cs == UNVISITED) return first;
            if (st == ST_M) {
                if (j > 0) {
                    bool moe = g.is_symbol_equal(v, q[j - 1]) || v == g.end;
                    size_t pj = (v == g.end) ? j : j - 1;
                    Score target = (moe || g.pred[v].empty()) ? cs : sub(cs, x);
                    for (uint32_t p : g.pred[v]) if (S(p, pj, ST_M) == target) cand(p, pj, ST_M);
                }
                for (AlignState gs : {ST_D, ST_D2, ST_I, ST_I2}) if (S(v, j, gs) == cs) cand(v, j, gs);
            } else if (st == ST_D) {
                Score t_open = sub(sub(cs, o1), e1), t_ext = sub(cs, e1);
                bool real_open = (v != g.end) && open_d(v, q, L, j);
                for (uint32_t p : g.pred[v]) {
                    Score ps = S(p, j, ST_M);
                    if (ps == t_open) cand(p, j, ST_M);
                    else if (!real_open && ps < t_open) phantom_lt = true;
                }
                for (uint32_t p : g.pred[v]) if (S(p, j, ST_D) == t_ext) cand(p, j, ST_D);
            } else if (st == ST_D2) {
                Score t = sub(cs, e2);
                for (uint32_t p : g.pred[v]) if (S(p, j, ST_D) == t) cand(p, j, ST_D);
                for (uint32_t p : g.pred[v]) if (S(p, j, ST_D2) == t) cand(p, j, ST_D2);
            } else if (st == ST_I) {
                if (j > 0) {
                    Score t_open = sub(sub(cs, o1), e1), t_ext = sub(cs, e1);
                    Score ps = S(v, j - 1, ST_M);
                    if (ps == t_open) cand(v, j - 1, ST_M);
                    else if (!open_i(v, q, L, j - 1) && ps < t_open) phantom_lt = true;
                    if (S(v, j - 1, ST_I) == t_ext) cand(v, j - 1, ST_I);
                }
            } else {
                if (j > 0) {
                    Score t = sub(cs, e2);
                    if (S(v, j - 1, ST_I) == t) cand(v, j - 1, ST_I);
                    if (S(v, j - 1, ST_I2) == t) cand(v, j - 1, ST_I2);
                }
            }
            return first;
        };
        uint32_t nc; bool plt, pn;
        Step cur = step(g.end, L, ST_M, nc, plt, pn);
        if (pn) { R.flags |= DF_REF_PANIC | DF_TRUNCATED; return; }
        if (cur.found && (nc != 1 || plt)) { R.flags |= DF_AMBIGUOUS; R.n_ambiguous_steps++; }
        if (!cur.found) {
            for (AlignState gs : {ST_I, ST_I2, ST_D, ST_D2}) {   // gap_affine_2piece.rs:972-978
                cur = step(g.end, L, gs, nc, plt, pn);
                if (pn) { R.flags |= DF_REF_PANIC | DF_TRUNCATED; return; }
                if (cur.found) break;
            }
            if (!cur.found) { R.flags |= DF_REF_PANIC; return; }
            R.flags |= DF_AMBIGUOUS;
        }
        uint32_t cn = cur.node; size_t cj = cur.j; AlignState cst = cur.st;
        bool reached_start = false;
        for (;;) {
            Step bt = step(cn, cj, cst, nc, plt, pn);
            if (pn) { R.flags |= DF_REF_PANIC; break; }
            if (!bt.found) break;
            if (nc != 1 || plt) { R.flags |= DF_AMBIGUOUS; R.n_ambiguous_steps++; }
            if (cst == ST_M && bt.st != ST_M) { cn = bt.node; cj = bt.j; cst = bt.st; continue; }
            if (cst == ST_M) R.alignment.push_back({cn, (uint32_t)cj - 1});
            else if (cst == ST_I || cst == ST_I2) R.alignment.push_back({NONE32, (uint32_t)cj - 1});
            else R.alignment.push_back({cn, NONE32});
            if (bt.st == ST_M && bt.j == 0 && bt.node != g.start && cst != ST_D && cst != ST_D2 && g.is_symbol_equal(bt.node, q[0]))
                R.flags |= DF_START_QUIRK;
            if (bt.node == g.start) { reached_start = true; break; }
            cn = bt.node; cj = bt.j; cst = bt.st;
        }
        if (!reached_start) R.flags |= DF_TRUNCATED;
        std::reverse(R.alignment.begin(), R.alignment.end());
    }

    DenseResult align(const uint8_t* q, size_t L, bool keep_planes = false) const {
        DenseResult R;
        forward(q, L, R);
        traceback(q, L, R);
        if (!keep_planes) for (auto* pl : {&R.M, &R.I, &R.D, &R.I2, &R.D2}) { pl->clear(); pl->shrink_to_fit(); }
        return R;
    }
};

}  // namespace poa_oracle
