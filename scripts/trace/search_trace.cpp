// Analysis tool (not shipped, not a test): runs the product's host build of the replay (poa_exact.hpp, ExactSearch::run)
// and logs every pop and every push, so that schedules for the wave kernel can be priced before they are built.
#include <cstdint>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

struct Touch { uint32_t pop, kind, row, a, b; };
static std::vector<Touch>* g_touch = nullptr;
static uint32_t g_pop = 0;
#define EX_TRACE_CELL(row, off, st, wr) do { if (g_touch) g_touch->push_back(Touch{g_pop, (uint32_t)(wr), (uint32_t)(row), (uint32_t)(off), (uint32_t)(st)}); } while (0)
#define EX_TRACE_REACH(row, lo, hi, wr) do { if (g_touch) g_touch->push_back(Touch{g_pop, 2u + (uint32_t)(wr), (uint32_t)(row), (uint32_t)(lo), (uint32_t)(hi)}); } while (0)
#include "../../include/poasta_amd.h"
#include "../../poasta_amd/csrc/poa_exact.hpp"
#include "../../poasta_amd/csrc/poa_graph.hpp"

using namespace poa_amd;

extern "C" {
// pops: [cap_pops][8] = f, st, g, row, j, outcome (0 expanded, 1 stale, 2 pruned, 3 end), dfa_visited, first push index
// pushes: [cap_push][5] = f, st, g, row, j
// out: n_pops, n_pushes, score, status
int search_trace(uint32_t n, uint32_t start, uint32_t end, const uint8_t* symbol, const uint32_t* succ_off,
                 const uint32_t* succ, const uint32_t* pred_off, const uint32_t* pred, uint8_t x, uint8_t o, uint8_t e,
                 int heuristic, int prune, const uint8_t* seq, uint32_t len, uint32_t* pops, uint64_t cap_pops,
                 uint32_t* pushes, uint64_t cap_push, uint64_t* out, uint32_t* row_info /* [n][4]: n_succ, n_bubbles, dist_min, dist_max */,
                 uint32_t* touches /* [cap_touch][5] pop, kind (0 cell read, 1 cell write, 2 reached-range read, 3 mark), row, a, b */, uint64_t cap_touch) {
    std::vector<Touch> tv;
    g_touch = touches ? &tv : nullptr;
    FlatGraph g;
    std::string err;
    int rc = build_flat_graph(n, start, end, symbol, succ_off, succ, pred_off, pred, g, err);
    if (rc != POA_OK) return rc;
    rc = build_bubble_index(g, err);
    if (rc != POA_OK) return rc;
    std::vector<uint8_t> row_sym(g.n);
    for (uint32_t r = 0; r < g.n; ++r) row_sym[r] = g.rows[r].sym;
    ExactGraph G{g.n, g.start_row, g.end_row, row_sym.data(), g.succ_row_off.data(), g.succ_rows.data(),
                 g.dist_min.data(), g.dist_max.data(), g.exit_idx.data(), g.n_exit, g.nbm_off.data(), g.nbm.data(),
                 g.node_row.data(), g.sp_to_end.data()};
    if (row_info) for (uint32_t r = 0; r < g.n; ++r) {
        row_info[4 * r] = g.succ_row_off[r + 1] - g.succ_row_off[r];
        row_info[4 * r + 1] = g.nbm_off[r + 1] - g.nbm_off[r];
        row_info[4 * r + 2] = g.dist_min[r]; row_info[4 * r + 3] = g.dist_max[r];
    }
    const uint32_t pitch = ((len + 1 + 63) / 64) * 64, wpn = (len + 1 + 63) / 64, swpn = (wpn + 63) / 64;
    std::vector<uint32_t> T((size_t)3 * n * pitch, EX_INF);
    std::vector<uint64_t> reached((size_t)g.n_exit * wpn + 1, 0), rsum((size_t)g.n_exit * swpn + 1, 0);
    const uint32_t n_prio = (n + len + 2) * std::max<uint32_t>(x, (uint32_t)o + e) + 2 * ((uint32_t)o + (n + len) * e) + 64;
    std::vector<uint32_t> head((size_t)3 * n_prio, EX_NIL);
    std::vector<ExQEntry> pool((size_t)4 * n * (len + 1) + 1024);
    std::vector<ExStackEntry> stack(n + len + 8);
    ExactWork W{T.data(), g.n, pitch, reached.data(), rsum.data(), wpn, swpn, head.data(), n_prio,
                pool.data(), (uint32_t)pool.size(), stack.data(), (uint32_t)stack.size()};
    ExactCosts EC{x, o, e, (uint32_t)heuristic, (uint32_t)prune, 0, 0, 0, 0, 0, 0};
    ExactSearch S(G, W, seq, len, EC);
    ExactResult R{EX_OK, EX_INF, 0, 0, 0, G.end_row, len};
    S.push_initial_states();
    uint32_t end_score = EX_INF;
    bool found = false;
    uint64_t np = 0, nq = 0;
    std::vector<uint32_t> push_state(pool.size(), 0);
    auto log_pushes = [&](uint32_t from) {
        for (uint32_t k = from; k < S.pool_top; ++k) {
            // state of a pushed entry: find which head list starts with it (it was just pushed: it is a head, or was overtaken
            // by a later push of the same (prio, state) within this expansion)
            const ExQEntry& q = pool[k];
            uint32_t st = 3;
            for (uint32_t s = 0; s < 3 && st == 3; ++s) {
                const uint32_t pr = q.score + S.h(q.row, q.offset, s);
                if (pr >= n_prio) continue;
                for (uint32_t c = head[3 * (uint64_t)pr + s]; c != EX_NIL && c >= from; c = pool[c].next) if (c == k) { st = s; break; }
            }
            if (nq < cap_push) {
                uint32_t* p = pushes + 5 * nq;
                p[0] = q.score + S.h(q.row, q.offset, st); p[1] = st; p[2] = q.score; p[3] = q.row; p[4] = q.offset;
            }
            nq++;
        }
    };
    log_pushes(0);
    while (!found && !S.err) {
        uint32_t score, row, off, st;
        const uint32_t f = S.layer_min;
        if (!S.pop_state(score, row, off, st)) { S.err = EX_PANIC; break; }
        const uint32_t from = S.pool_top;
        const uint32_t nv0 = S.num_visited;
        g_pop = (uint32_t)np;
        uint32_t sk = S.inspect_skip(score, row, off, st);
        if (sk == 2) S.num_pruned += 1;
        uint32_t outcome = sk;
        if (!sk && !S.err) {
            found = S.process_popped(score, row, off, st, R, end_score);
            if (found) outcome = 3;
        }
        if (np < cap_pops) {
            uint32_t* p = pops + 8 * np;
            p[0] = f; p[1] = st; p[2] = score; p[3] = row; p[4] = off; p[5] = outcome; p[6] = S.num_visited - nv0; p[7] = (uint32_t)nq;
        }
        np++;
        log_pushes(from);
    }
    out[0] = np; out[1] = nq; out[2] = end_score; out[3] = S.err ? S.err : (found ? EX_OK : EX_PANIC);
    out[4] = tv.size();
    for (size_t k = 0; k < tv.size() && k < cap_touch; ++k) { uint32_t* t = touches + 5 * k; t[0] = tv[k].pop; t[1] = tv[k].kind; t[2] = tv[k].row; t[3] = tv[k].a; t[4] = tv[k].b; }
    g_touch = nullptr;
    return 0;
}
}
