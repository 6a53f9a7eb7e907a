// Test harness (CPU): compiles the PRODUCT's exact-replay search (poasta_amd/csrc/poa_exact.hpp) and
// graph preprocessing (poa_graph.cpp) for the host so that they can be diffed against the oracle
// without a GPU.  Not shipped; built by tests/test_exact_replay.py.
#include <cstdint>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/poasta_amd.h"
#include "../../poasta_amd/csrc/poa_exact.hpp"
#include "../../poasta_amd/csrc/poa_graph.hpp"

using namespace poa_amd;

static uint32_t g_batch = 0, g_win = 0, g_par_lanes = 0, g_par_rmax = 4, g_par_fast = 1, g_rec = 0;   // g_par_rmax == 0: the flat schedule (run_flat)

extern "C" {

// 0: linked-list queue (ExactSearch::run); n > 0: bucket queue stepped in batches of n (ExactSearch::run_buckets)
void exact_host_set_batch(uint32_t n) { g_batch = n; }
void exact_host_set_window(uint32_t w) { g_win = w; }  // 0: a window that always suffices
// 1: the instantiation of the wave kernels whose one-round-trip path reads the per-row records (EX_AS_NO_SPEC | EX_AS_REC_LDS), run_buckets
void exact_host_set_records(uint32_t on) { g_rec = on; }
// lanes > 0: ExactSearch::run_parallel(lanes, rmax) — the step schedule of poa_psearch.hpp (entries of a stack expanded at once)
void exact_host_set_parallel(uint32_t lanes, uint32_t rmax, uint32_t use_fast) { g_par_lanes = lanes; g_par_rmax = rmax; g_par_fast = use_fast; }

// returns status (EX_*), or negative POA_ERR_* for graph errors.
// planes (optional): M/I/D as [node][len+1]; out[0..3] = score, num_queued, num_visited, num_pruned
int exact_host_run(uint32_t n, uint32_t start, uint32_t end, const uint8_t* symbol, const uint32_t* succ_off,
                   const uint32_t* succ, const uint32_t* pred_off, const uint32_t* pred, uint8_t x, uint8_t o, uint8_t e,
                   int heuristic, int prune, const uint8_t* seq, uint32_t len, uint32_t* out, uint32_t* pm, uint32_t* pi,
                   uint32_t* pd, const uint32_t* span /* null = Global; else {1, qfe_kind, qfe_val, gfb_kind, gfe_kind, gfe_val}; out[4..5] = end node, offset */) {
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
                 g.node_row.data(), g.sp_to_end.data(), g.row_rec.empty() ? nullptr : g.row_rec.data()};
    const uint32_t pitch = ((len + 1 + 63) / 64) * 64, wpn = (len + 1 + 63) / 64, swpn = (wpn + 63) / 64;  // as the engine lays it out
    std::vector<uint32_t> T((size_t)3 * n * pitch, EX_INF);  // tiled: ex_cell_index
    std::vector<uint64_t> reached((size_t)g.n_exit * wpn + 1, 0), rsum((size_t)g.n_exit * swpn + 1, 0);
    const uint32_t n_prio = (n + len + 2) * std::max<uint32_t>(x, (uint32_t)o + e) + 2 * ((uint32_t)o + (n + len) * e) + 64;
    std::vector<uint32_t> head((size_t)3 * n_prio, EX_NIL);
    std::vector<ExQEntry> pool((size_t)4 * n * (len + 1) + 1024);
    std::vector<ExStackEntry> stack(n + len + 8);
    ExactWork W{T.data(), g.n, pitch, reached.data(), rsum.data(), wpn, swpn, head.data(), n_prio,
                pool.data(), (uint32_t)pool.size(), stack.data(), (uint32_t)stack.size()};
    // bucket queue of the wave search (batch > 0): a small ring so that the window logic is exercised
    std::vector<uint32_t> bq_desc;
    std::vector<ExU4> bq_chunks;
    if (g_batch || g_par_lanes) {
        uint32_t win = 64;
        const uint32_t need = 2 * (std::max<uint32_t>(x, (uint32_t)o + e) + o + (uint32_t)(g.n + 2) * e) + 8;
        while (win < need) win *= 2;
        if (g_win) win = g_win;
        bq_desc.assign((size_t)3 * win, BQ_EMPTY);
        bq_chunks.resize((size_t)BQ_CHUNK * (pool.size() / 32 + 3 * win + 64));
        W.bq_desc = bq_desc.data(); W.bq_win = win; W.bq_chunks = bq_chunks.data(); W.bq_chunk_cap = (uint32_t)(bq_chunks.size() / BQ_CHUNK);
    }
    ExactCosts EC{x, o, e, (uint32_t)heuristic, (uint32_t)prune, 0, 0, 0, 0, 0, 0};
    if (span && span[0]) { EC.ends_free = 1; EC.qfe_kind = span[1]; EC.qfe_val = span[2]; EC.gfb_kind = span[3]; EC.gfe_kind = span[4]; EC.gfe_val = span[5]; }
    if (g_rec && g_batch && !g_par_lanes) {
        ExactSearchT<EX_AS_NO_SPEC | EX_AS_REC_LDS> S2(G, W, seq, len, EC);
        const ExactResult R2 = S2.run_buckets(g_batch);
#if defined(POA_EXACT_DIAG)
        if (getenv("EXH_VERBOSE")) {
            const char* why[8] = {"stale/pruned on the fast path", "several successors", "end row / misc", "bubble shape", "Match special", "probe undecided", "fast expand", "fast greedy walk"};
            for (int r = 0; r < 8; ++r) fprintf(stderr, "  %-32s M %8u  D %8u  I %8u\n", why[r], S2.diag[r][0], S2.diag[r][1], S2.diag[r][2]);
        }
#endif
        out[0] = R2.score; out[1] = R2.num_queued; out[2] = R2.num_visited; out[3] = R2.num_pruned;
        if (span) { out[4] = g.rows[R2.end_row].node; out[5] = R2.end_off; }
        if (pm) {
            for (uint32_t v = 0; v < n; ++v) {
                const uint32_t r = g.node_row[v];
                for (uint32_t j = 0; j <= len; ++j) {
                    pm[(size_t)v * (len + 1) + j] = T[ex_cell_index(r, j, EX_ST_M, g.n, pitch)];
                    pi[(size_t)v * (len + 1) + j] = T[ex_cell_index(r, j, EX_ST_I, g.n, pitch)];
                    pd[(size_t)v * (len + 1) + j] = T[ex_cell_index(r, j, EX_ST_D, g.n, pitch)];
                }
            }
        }
        return (int)R2.status;
    }
    ExactSearch S(G, W, seq, len, EC);
    ExactResult R = g_par_lanes ? (g_par_rmax ? S.run_parallel(g_par_lanes, g_par_rmax, g_par_fast != 0) : S.run_flat(g_par_lanes, g_par_fast != 0, g_par_fast == 2)) : g_batch ? S.run_buckets(g_batch) : S.run();
    if (getenv("EXH_VERBOSE")) fprintf(stderr, "chunks used %u of %u, status %u, fast-path tests %u, queued %u\n", S.bq_chunk_top, W.bq_chunk_cap, R.status, S.n_fast, R.num_queued);
    if (getenv("EXH_VERBOSE") && g_par_lanes) {
        fprintf(stderr, "  lanes committed per step:"); for (int k = 0; k < 65; ++k) if (S.par_hist[k]) fprintf(stderr, " %d:%u", k, S.par_hist[k]); fprintf(stderr, "\n");
        fprintf(stderr, "  conflict cut at lane:"); for (int k = 0; k < 65; ++k) if (S.par_hist_conf[k]) fprintf(stderr, " %d:%u", k, S.par_hist_conf[k]); fprintf(stderr, "\n");
        fprintf(stderr, "  entries offered per step:"); for (int k = 0; k < 65; ++k) if (S.par_hist_nb[k]) fprintf(stderr, " %d:%u", k, S.par_hist_nb[k]); fprintf(stderr, "\n");
    }
    if (getenv("EXH_VERBOSE") && g_par_lanes)
        fprintf(stderr, "parallel schedule: %u steps (%u sequential), %u entries in log mode; cuts: %u complex, %u conflict, %u leftover; visited %u pruned %u\n",
                S.par_steps, S.par_seq, S.par_entries, S.par_cut_complex, S.par_cut_conflict, S.par_cut_leftover, R.num_visited, R.num_pruned);
#if defined(POA_EXACT_DIAG)
    if (getenv("EXH_VERBOSE")) {
        fprintf(stderr, "  probe neighbour below t at distance 1/2/3-4/5-8/>8/none: %u %u %u %u %u %u; above: %u %u %u %u %u %u\n", S.pd_hist[0][1], S.pd_hist[0][2], S.pd_hist[0][4], S.pd_hist[0][5], S.pd_hist[0][6], S.pd_hist[0][7],
                S.pd_hist[1][1], S.pd_hist[1][2], S.pd_hist[1][4], S.pd_hist[1][5], S.pd_hist[1][6], S.pd_hist[1][7]);
    }
    if (getenv("EXH_VERBOSE") && g_par_lanes) {
        fprintf(stderr, "  why complex: read cells %u, mark ranges %u, cell writes %u, marks %u, pushes %u, pending %u, dfa stack %u, prio %u, shape %u\n", S.why_complex[1], S.why_complex[2],
                S.why_complex[3], S.why_complex[4], S.why_complex[5], S.why_complex[6], S.why_complex[7], S.why_complex[9], S.why_complex[10]);
    }
    if (getenv("EXH_VERBOSE")) {
        const char* why[8] = {"stale/pruned on the fast path", "several successors", "end row / misc", "bubble shape", "Match special", "probe undecided", "fast expand", "fast greedy walk"};
        for (int r = 0; r < 8; ++r) fprintf(stderr, "  %-32s M %8u  D %8u  I %8u\n", why[r], S.diag[r][0], S.diag[r][1], S.diag[r][2]);
    }
#endif
    out[0] = R.score; out[1] = R.num_queued; out[2] = R.num_visited; out[3] = R.num_pruned;
    if (span) { out[4] = g.rows[R.end_row].node; out[5] = R.end_off; }
    if (pm) {
        for (uint32_t v = 0; v < n; ++v) {
            const uint32_t r = g.node_row[v];
            for (uint32_t j = 0; j <= len; ++j) {
                pm[(size_t)v * (len + 1) + j] = T[ex_cell_index(r, j, EX_ST_M, g.n, pitch)];
                pi[(size_t)v * (len + 1) + j] = T[ex_cell_index(r, j, EX_ST_I, g.n, pitch)];
                pd[(size_t)v * (len + 1) + j] = T[ex_cell_index(r, j, EX_ST_D, g.n, pitch)];
            }
        }
    }
    return (int)R.status;
}

// Two-piece affine model (gap_affine_2piece.rs): the same search object instantiated with EX_AS_TWO_PIECE — generic code,
// linked-list queue, five plain planes.  costs = (mismatch, open1, extend1, open2, extend2); planes (optional): M, I1, D1, I2, D2
// as [node][len+1]; out[0..5] = score, num_queued, num_visited, num_pruned, end node, end offset.
int exact_host_run2(uint32_t n, uint32_t start, uint32_t end, const uint8_t* symbol, const uint32_t* succ_off,
                    const uint32_t* succ, const uint32_t* pred_off, const uint32_t* pred, const uint8_t* costs5,
                    int heuristic, int prune, const uint8_t* seq, uint32_t len, uint32_t* out, uint32_t* const* planes,
                    const uint32_t* span) {
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
                 g.node_row.data(), g.sp_to_end.data(), nullptr};
    const uint32_t x = costs5[0], o1 = costs5[1], e1 = costs5[2], o2 = costs5[3], e2 = costs5[4];
    const uint32_t pitch = ((len + 1 + 63) / 64) * 64, wpn = (len + 1 + 63) / 64, swpn = (wpn + 63) / 64;
    std::vector<uint32_t> T((size_t)5 * n * pitch, EX_INF);
    std::vector<uint64_t> reached((size_t)g.n_exit * wpn + 1, 0), rsum((size_t)g.n_exit * swpn + 1, 0);
    const uint32_t om = std::max(o1, o2);
    const uint32_t n_prio = (n + len + 2) * std::max<uint32_t>(x, om + e1) + 2 * (om + (n + len) * e1) + 64;
    std::vector<uint32_t> head((size_t)5 * n_prio, EX_NIL);
    std::vector<ExQEntry> pool((size_t)64 * n * (len + 1) + 1024);   // (a cell is queued again each time its score drops: often, under this model's inadmissible heuristic)
    std::vector<ExStackEntry> stack(n + len + 8);
    ExactWork W{T.data(), g.n, pitch, reached.data(), rsum.data(), wpn, swpn, head.data(), n_prio,
                pool.data(), (uint32_t)pool.size(), stack.data(), (uint32_t)stack.size()};
    ExactCosts EC{x, o1, e1, (uint32_t)heuristic, (uint32_t)prune, 0, 0, 0, 0, 0, 0, o2, e2};
    if (span && span[0]) { EC.ends_free = 1; EC.qfe_kind = span[1]; EC.qfe_val = span[2]; EC.gfb_kind = span[3]; EC.gfe_kind = span[4]; EC.gfe_val = span[5]; }
    ExactSearchT<EX_AS_NO_SPEC | EX_AS_TWO_PIECE> S(G, W, seq, len, EC);
    const ExactResult R = S.run();
    if (getenv("EXH_VERBOSE")) fprintf(stderr, "two-piece: status %u, pool %u of %zu, n_prio %u, layer_min %u + %u, queued %u\n", R.status, S.pool_top, pool.size(), n_prio, S.layer_min, S.n_layers, R.num_queued);
    out[0] = R.score; out[1] = R.num_queued; out[2] = R.num_visited; out[3] = R.num_pruned;
    out[4] = g.rows[R.end_row].node; out[5] = R.end_off;
    if (planes) {
        for (int pl = 0; pl < 5; ++pl)
            for (uint32_t v = 0; v < n; ++v)
                for (uint32_t j = 0; j <= len; ++j) planes[pl][(size_t)v * (len + 1) + j] = T[((size_t)g.node_row[v] * pitch + j) * 5 + pl];
    }
    return (int)R.status;
}

// bubble index of the product side, by NODE: dist (min,max), is_exit, node_bubble_map (exit node, min, max)
int exact_host_bubbles(uint32_t n, uint32_t start, uint32_t end, const uint8_t* symbol, const uint32_t* succ_off,
                       const uint32_t* succ, const uint32_t* pred_off, const uint32_t* pred, uint32_t* dmin, uint32_t* dmax,
                       uint8_t* is_exit, uint32_t* nbm_off, uint32_t* nbm, uint32_t cap) {
    FlatGraph g;
    std::string err;
    int rc = build_flat_graph(n, start, end, symbol, succ_off, succ, pred_off, pred, g, err);
    if (rc != POA_OK) return rc;
    rc = build_bubble_index(g, err);
    if (rc != POA_OK) return rc;
    uint32_t k = 0;
    for (uint32_t v = 0; v < n; ++v) {
        const uint32_t r = g.node_row[v];
        dmin[v] = g.dist_min[r]; dmax[v] = g.dist_max[r]; is_exit[v] = g.is_exit[r];
        nbm_off[v] = k;
        for (uint32_t i = g.nbm_off[r]; i < g.nbm_off[r + 1]; ++i) {
            if (k < cap) { nbm[3 * k] = g.rows[g.nbm[i].exit_row].node; nbm[3 * k + 1] = g.nbm[i].min_dist; nbm[3 * k + 2] = g.nbm[i].max_dist; }
            k++;
        }
    }
    nbm_off[n] = k;
    return 0;
}
}
