/* poasta_amd — C ABI of the MI355X (gfx950) gap-affine POA alignment engine.
 *
 * Drop-in boundary: everything behind `poasta::aligner::PoastaAligner::{align,
 * align_with_existing_bubbles, align_no_pruning}` (/root/reference/src/aligner/mod.rs:69-145),
 * i.e. `astar_alignment` (src/aligner/astar.rs:108-226) + the score-based backtrace
 * (src/aligner/scoring/gap_affine.rs:550-657, :804-915), for the gap-affine cost model
 * (`GapAffine`, gap_affine.rs:20-30) in Global mode with query offsets of type u32
 * (the only instantiation the reference binaries use: src/bin/poasta.rs:214, src/bin/lasagna.rs:125).
 *
 * The host keeps the graph (`POAGraph`), builds it, mutates it and does all I/O.  It hands the
 * library a flattened view of the `AlignableRefGraph` trait (src/graphs/mod.rs:23-53) and a batch of
 * queries (the `lasagna align` shape, src/bin/lasagna.rs:184-276: N reads x 1 immutable graph).
 *
 * Plain C: pointers and sizes only.  All functions return POA_OK (0) or a negative POA_ERR_* code;
 * nothing aborts or throws across this boundary.  INTEGRATION.md shows the Rust `extern "C"` block
 * and the ~30-line shim inside `PoastaAligner::align_internal`.
 */
#ifndef POASTA_AMD_H
#define POASTA_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define POA_NONE 0xFFFFFFFFu /* AlignedPair{rpos|qpos: None} (src/aligner/alignment.rs:4-13) */
#define POA_SCORE_UNVISITED 0xFFFFFFFFu /* Score::Unvisited (src/aligner/scoring/mod.rs:64-70) */

/* return codes */
#define POA_OK 0
#define POA_ERR_INVALID_ARG (-1)
#define POA_ERR_NOT_A_DAG (-2)      /* cycle, unreachable node, start with predecessors, end with successors */
#define POA_ERR_NO_DEVICE (-3)      /* no usable gfx950 device: the HIP path never falls back to the CPU */
#define POA_ERR_HIP (-4)            /* a HIP runtime call failed; see poa_last_error() */
#define POA_ERR_CAPACITY (-5)       /* pair_capacity too small; pair_off[n] holds the needed total */
#define POA_ERR_OUT_OF_MEMORY (-6)
#define POA_ERR_UNSUPPORTED (-7)

/* per-query flags (poa_align_batch `flags`).
 * Scores are exact whenever START_QUIRK and SHORT_QUERY are clear.  The alignment is the
 * reference's, bit for bit, whenever the word is 0: the optimal alignment is then unique in the
 * reference's own alignment graph, so every search order yields it. */
#define POA_FLAG_AMBIGUOUS 0x01u    /* >1 co-optimal alignment: the reference's pick depends on its search order */
#define POA_FLAG_START_QUIRK 0x02u  /* path uses an edge the reference's offset-0 special case may hide (dfa.rs:146-167) */
#define POA_FLAG_REF_PANIC 0x04u    /* the reference would panic on this input (u32 wrap in mod.rs:144-152) */
#define POA_FLAG_SHORT_QUERY 0x08u  /* len <= 1: reference special cases (gap_affine.rs:808-824) */
#define POA_FLAG_TRUNCATED 0x10u    /* backtrace ended before the start node (as the reference's would) */
#define POA_FLAG_EMPTY_GRAPH 0x20u  /* no real nodes: PoastaAligner::align shortcut (mod.rs:124-142), score 4*len */
#define POA_FLAG_EXACT_OVERFLOW 0x40u /* exact replay ran out of workspace: the dense result (and its flags) was kept */

/* modes (poa_config_t.mode) */
#define POA_MODE_DENSE 0u   /* dense planes + traceback: scores exact, alignment certified-or-flagged */
#define POA_MODE_EXACT 1u   /* dense pass, then a replay of the reference's own A* search (same pop order, greedy
                               extension and pruning) for EVERY query: alignments bit-identical incl. tie-breaks */
#define POA_MODE_HYBRID 2u  /* dense pass, then the exact replay only for queries whose dense flags are non-zero */
#define POA_HEURISTIC_DIJKSTRA 0u /* AffineDijkstra   (src/aligner/config.rs:49)  */
#define POA_HEURISTIC_MINGAP 1u   /* AffineMinGapCost (src/aligner/config.rs:104), the default of both reference CLIs */

/* GapAffine (gap_affine.rs:20-24).  NB the Rust constructor order is (mismatch, extend, open). */
typedef struct poa_costs {
    uint8_t mismatch;
    uint8_t gap_open;
    uint8_t gap_extend;
    uint8_t reserved;
} poa_costs_t;

/* GapAffine2Piece (src/aligner/scoring/gap_affine_2piece.rs:19-33; the reference's constructor order is (mismatch, extend1,
 * open1, extend2, open2) and it asserts extend1 >= extend2).  A gap opens in the first piece (open1 + extend1) and may move to
 * the second at extend2 per step; open2 enters only the reference's heuristic / pruning arithmetic, never the DP
 * (gap_affine_2piece.rs:362-368, :402-408), so the dense pass does not read it. */
typedef struct poa_costs2 {
    uint8_t mismatch;
    uint8_t gap_open1;
    uint8_t gap_extend1;
    uint8_t gap_open2;
    uint8_t gap_extend2;
    uint8_t wide_planes;      /* 1: u32 planes whatever the bound on the score allows (A/B) */
    uint8_t reserved[2];
} poa_costs2_t;

/* Which reference configuration the exact replay emulates (only the replay depends on it: heuristic and pruning
 * fix the reference's search order, not its optimum).  Zero-initialised == dense mode. */
/* std::ops::Bound<usize> of AlignmentType::EndsFree (scoring/mod.rs:50-62) */
typedef struct poa_bound {
    uint32_t kind;            /* POA_BOUND_* */
    uint32_t value;
} poa_bound_t;
#define POA_BOUND_UNBOUNDED 0u
#define POA_BOUND_INCLUDED 1u
#define POA_BOUND_EXCLUDED 2u
#define POA_SPAN_GLOBAL 0u    /* AlignmentType::Global */
#define POA_SPAN_ENDS_FREE 1u /* AlignmentType::EndsFree{..}: what the reference returns is defined by its SEARCH
                                 (initial states gap_affine.rs:136-183, is_end :185-248: with unbounded ends the first
                                 popped Match state past offset 0 ends it), so the engine replays that search for every
                                 query whatever `mode` says; POA_FLAG_TRUNCATED then only says that the alignment does
                                 not begin at the start node */

/* Overrides of the choices the engine makes per call (plane layout, forward / traceback / replay kernel and their shapes) — for
 * A/B measurements and for the tests that run every parity case under every variant.  poa_config_t.tune[k] == 0: the engine
 * decides; else the entry holds (value + 1).  The library reads no environment variable: the Python binding fills this block from
 * POA_<NAME> variables (poasta_amd/_lib.py: tune_from_env), a host that wants a variant for one call sets it in that call's config. */
enum {
    POA_TUNE_PLANES = 0,      /* 32: u32 planes */
    POA_TUNE_COMPACT,         /* 0: three planes instead of the compact layout */
    POA_TUNE_PACKED,          /* 0: scalar-arithmetic forward kernel */
    POA_TUNE_RELATIVE,        /* 0 / 1: relative encoding off / on */
    POA_TUNE_PX,              /* 0: adjacent-pairs packed kernel instead of pairs-across-quads */
    POA_TUNE_MF,              /* 0 / 1 / 2: at most that many flag pairs beside the score */
    POA_TUNE_MW,              /* 0: no multi-wave pipeline */
    POA_TUNE_PXMW,            /* 0 / 1: 1024-column multi-wave kernel off / on */
    POA_TUNE_FWD_QUADS,
    POA_TUNE_FUSE_TB,
    POA_TUNE_TB_GROUP,        /* lanes per traceback walk: 8 / 16 / 32 / 64 */
    POA_TUNE_TB_DEPTH,        /* speculative steps per round trip */
    POA_TUNE_EXACT_IMPL,      /* replay kernel: 1 one search per lane, 2 wave per query (default), 3 flat parallel steps */
    POA_TUNE_EXACT_LANES,
    POA_TUNE_EXACT_LDS,       /* 0: graph tables read from global memory */
    POA_TUNE_WS_LANES,
    POA_TUNE_WS_GROUP,
    POA_TUNE_WS_WAVES,
    POA_TUNE_WS_RING_GLOBAL,
    POA_TUNE_WS_STATIC,
    POA_TUNE_WS_CHUNK_CAP,
    POA_TUNE_WS_PROF,         /* per-phase cycle counts of the replay kernel, printed by poa_batch_stats */
    POA_TUNE_PS_LANES,
    POA_TUNE_PS_LEAN,         /* 0: flat kernel with the generic code in log mode */
    POA_TUNE_TIMING,          /* poa_align_batch: host-side timing printed to stderr */
    POA_TUNE_WS_ADAPT,        /* wave replay: entries tested in the step after an expansion (0: always WS_LANES) */
    POA_TUNE_WS_REC,          /* 0: the wave replay's one-round-trip path reads the graph arrays instead of the per-row records */
    POA_TUNE_COUNT = 32
};

typedef struct poa_config {
    uint32_t mode;            /* POA_MODE_* */
    uint32_t heuristic;       /* POA_HEURISTIC_* (replay only) */
    uint32_t pruning;         /* 1: align / align_with_existing_bubbles; 0: align_no_pruning (mod.rs:81-90) */
    float queue_entries_per_cell; /* replay queue pool, entries per (row x column) cell; 0 = default 0.25 */
    uint32_t flags;           /* POA_CFG_* */
    uint32_t span;            /* POA_SPAN_* ; the four bounds are read only for POA_SPAN_ENDS_FREE */
    poa_bound_t qry_free_begin;   /* carried, never read by the reference's 1-piece path (gap_affine.rs:146) */
    poa_bound_t qry_free_end;
    poa_bound_t graph_free_begin;
    poa_bound_t graph_free_end;
    uint32_t tune[POA_TUNE_COUNT]; /* overrides of the engine's own choices, read once per call; see POA_TUNE_* */
} poa_config_t;
#define POA_CFG_FULL_PLANES 1u /* keep all three score planes in memory (needed by poa_batch_fetch_planes); the default
                                  u16 layout stores M, a 4-bit code per cell instead of I, and only the D rows read back */

/* AlignedPair (alignment.rs:4-13): rpos = node index of the host graph, qpos = 0-based query position */
typedef struct poa_aln_pair {
    uint32_t rpos;
    uint32_t qpos;
} poa_aln_pair_t;

/* Counters returned per call — the analogue of AstarResult::{num_queued,num_visited,num_pruned}
 * (astar.rs:86-89) for a dense pass, plus HIP-event timings taken on the launch stream. */
typedef struct poa_stats {
    uint64_t cells;            /* sum over queries of rows * (len + 1) */
    uint64_t bases;            /* sum of query lengths */
    uint64_t plane_bytes;      /* bytes of M/I/D score planes written */
    uint32_t n_queries;
    uint32_t n_chunks;         /* query chunks processed (workspace reuse) */
    uint32_t n_forward_launches;
    uint32_t n_flagged;        /* queries with flags != 0 */
    float ms_forward;          /* sum of forward-kernel durations (HIP events) */
    float ms_traceback;        /* sum of traceback + compaction kernel durations */
    float ms_h2d;              /* query upload (poa_align_batch only) */
    float ms_d2h;              /* result download (poa_align_batch / poa_batch_fetch) */
    float ms_exact;            /* exact-replay search kernels (+ their traceback), 0 in dense mode */
    uint32_t n_exact;          /* queries whose result came from the exact replay (last run) */
    float ms_total;            /* first launch -> last kernel end, summed over runs */
    uint32_t n_runs;           /* poa_batch_run calls covered by the ms_* sums */
} poa_stats_t;

typedef struct poa_graph poa_graph_t;
typedef struct poa_batch poa_batch_t;

/* Threading: a poa_graph_t is immutable after poa_graph_create and may be shared by batches on several threads (its bubble
 * index, needed by exact / hybrid runs only, is built once under a lock).  A poa_batch_t belongs to one thread at a time.
 * poa_align_batch* are re-entrant on distinct output buffers.  poa_last_error() is thread-local. */

/* ---- library --------------------------------------------------------------------------- */
const char* poa_version(void);
const char* poa_last_error(void);     /* thread-local description of the last failure */
int poa_device_count(void);           /* number of visible HIP devices (0 if none) */

/* ---- graph: the flattened AlignableRefGraph --------------------------------------------- */
/* Built once per graph by iterating the trait in the host (graphs/mod.rs:23-53):
 *   n          node_count_with_start_and_end()
 *   start,end  start_node(), end_node()
 *   symbol[n]  get_symbol_char() as u8 ('#' start, '$' end: graphs/poa.rs:102-103)
 *   succ_off[n+1], succ[]   successors(v) in ITERATION ORDER
 *   pred_off[n+1], pred[]   predecessors(v) in ITERATION ORDER (decides traceback ties,
 *                           gap_affine.rs:591,:617,:627)
 * The end node equals every query symbol (POAGraph::is_symbol_equal, graphs/poa.rs:463-465).
 * The library copies everything; the caller keeps ownership of its arrays. */
int poa_graph_create(uint32_t n, uint32_t start, uint32_t end, const uint8_t* symbol,
                     const uint32_t* succ_off, const uint32_t* succ,
                     const uint32_t* pred_off, const uint32_t* pred, poa_graph_t** out);
void poa_graph_destroy(poa_graph_t* g);
/* Refresh a graph handle after the host graph changed (POAGraph::add_alignment_with_weights + post_process,
 * src/graphs/poa.rs:171-363: nodes appended, edges added, start / end edges re-wired): same arguments as poa_graph_create, the
 * handle stays the same object.  The row tables are re-flattened in place (O(N + E) on the host — microseconds at the sizes of a
 * sequential POA build, under 1 % of a read's alignment call: scripts/sequential_poa_latency.sh); batches created from the
 * handle before the call keep the tables they copied and must not be run again. */
int poa_graph_update(poa_graph_t* g, uint32_t n_nodes_with_start_end, uint32_t start, uint32_t end, const uint8_t* symbol,
                     const uint32_t* succ_off, const uint32_t* succ, const uint32_t* pred_off, const uint32_t* pred);

uint32_t poa_graph_rows(const poa_graph_t* g);           /* == n */
/* row (topological rank used for the score planes) of every node; rank[n] */
int poa_graph_node_rows(const poa_graph_t* g, uint32_t* rank);

/* ---- one-shot batch alignment (host buffers in, host buffers out) ------------------------ */
/* Replaces, for a batch of queries against one graph, what the reference does per query in
 * `align_sequence` (src/bin/lasagna.rs:112-138) / `perform_alignment` (src/bin/poasta.rs:214).
 *   qseq,qoff[n+1]  concatenated queries
 *   score[n]        AstarResult::score
 *   pairs, pair_off[n+1], pair_capacity   AstarResult::alignment of query i =
 *                   pairs[pair_off[i] .. pair_off[i+1]); capacity sum(len_i + n_nodes) always suffices
 *   flags[n]        POA_FLAG_* (may be NULL)
 *   stats           may be NULL
 *   device          HIP device ordinal */
int poa_align_batch(const poa_graph_t* g, const poa_costs_t* costs, uint32_t n_queries,
                    const uint8_t* qseq, const uint64_t* qoff, uint32_t* score,
                    poa_aln_pair_t* pairs, uint64_t* pair_off, uint64_t pair_capacity,
                    uint32_t* flags, poa_stats_t* stats, int device);
/* same with a mode (cfg NULL == dense) */
int poa_align_batch_ex(const poa_graph_t* g, const poa_costs_t* costs, const poa_config_t* cfg, uint32_t n_queries,
                       const uint8_t* qseq, const uint64_t* qoff, uint32_t* score,
                       poa_aln_pair_t* pairs, uint64_t* pair_off, uint64_t pair_capacity,
                       uint32_t* flags, poa_stats_t* stats, int device);
/* Two-piece affine model, Global, dense pass: replaces `PoastaAligner::new(Affine2PieceDijkstra(costs),
 * AlignmentType::Global).align_no_pruning(graph, seq)` per query — src/aligner/config.rs:160-213, the cost model of
 * `poasta align -g 6,24 -e 2,1` (src/bin/poasta.rs:319-445).  Same buffers as poa_align_batch.  Scores are the optimum of the
 * reference's two-piece alignment graph (what its search returns in Dijkstra order without pruning); flags == 0 certifies
 * the alignment as the one the reference's backtrace rule forces (gap_affine_2piece.rs:639-794).  Returns
 * POA_ERR_INVALID_ARG where the reference's constructor panics (extend1 < extend2). */
int poa_align_batch_2piece(const poa_graph_t* g, const poa_costs2_t* costs, uint32_t n_queries,
                           const uint8_t* qseq, const uint64_t* qoff, uint32_t* score,
                           poa_aln_pair_t* pairs, uint64_t* pair_off, uint64_t pair_capacity,
                           uint32_t* flags, poa_stats_t* stats, int device);
/* Two-piece model with a mode (cfg NULL or mode DENSE == poa_align_batch_2piece).  Mode EXACT (HYBRID is taken as EXACT: under
 * this model the dense certificate does not cover the cases where the reference's search is not optimal) replays the
 * reference's own search per query — replaces `PoastaAligner::new(Affine2PieceMinGapCost(costs) | Affine2PieceDijkstra(costs),
 * aln_type).align(graph, seq)` (src/aligner/config.rs:160-272, astar.rs:124-226 over scoring/gap_affine_2piece.rs: five
 * states, stacks popped M, D1, D2, I1, I2 (:1069-1097), gap_cost / heuristic (:99-127, heuristic.rs:70-102), pruning with the
 * second-piece branches of bubbles/reached.rs:84-124,:165-186, ends-free begin / end rules :179-290) — what
 * `poasta align -g 6,24 -e 2,1` runs (src/bin/poasta.rs:319-445).  cfg->heuristic / pruning / span / bounds as in
 * poa_align_batch_ex.  score[n] is the score the reference's search returns (which may exceed the dense optimum), pairs
 * its backtrace (gap_affine_2piece.rs:639-794, :944-1043); flags: POA_FLAG_REF_PANIC, POA_FLAG_TRUNCATED,
 * POA_FLAG_EXACT_OVERFLOW (queue pool: raise cfg->queue_entries_per_cell).  search_counters (may be NULL): 4 words per query —
 * num_queued, num_visited, num_pruned (AstarResult, astar.rs:228) and the queue entries that were live at once. */
int poa_align_batch_2piece_ex(const poa_graph_t* g, const poa_costs2_t* costs, const poa_config_t* cfg, uint32_t n_queries,
                              const uint8_t* qseq, const uint64_t* qoff, uint32_t* score,
                              poa_aln_pair_t* pairs, uint64_t* pair_off, uint64_t pair_capacity,
                              uint32_t* flags, poa_stats_t* stats, uint32_t* search_counters, int device);
/* debugging / parity: the five score planes M, I1, D1, I2, D2 of ONE query, rows x (len + 1) each, row = topological rank
 * (poa_graph_node_rows) */
int poa_planes_2piece(const poa_graph_t* g, const poa_costs2_t* costs, const uint8_t* seq, uint32_t len,
                      uint32_t* m, uint32_t* i1, uint32_t* d1, uint32_t* i2, uint32_t* d2, int device);

/* poa_align_batch parks its plane workspace (tens of GB; hipMalloc/hipFree of it cost seconds) per device for the next
 * call; this returns that memory to the driver.  (src/bin/lasagna.rs has no counterpart: its tables live on the heap.) */
void poa_release_cache(void);

/* ---- resident batch (queries and results stay in HBM; used by the multi-GPU driver) ------ */
/* poa_batch_create uploads graph + queries to `device` and sizes the score-plane workspace
 * (workspace_bytes = 0: pick from free memory).  poa_batch_run launches forward + traceback on
 * `stream` (a hipStream_t, NULL = default stream) and returns without synchronising.
 * poa_batch_fetch synchronises the stream and copies results to the host. */
int poa_batch_create(const poa_graph_t* g, int device, uint32_t n_queries, const uint8_t* qseq,
                     const uint64_t* qoff, uint64_t workspace_bytes, poa_batch_t** out);
int poa_batch_run(poa_batch_t* b, const poa_costs_t* costs, void* stream);
/* same with a mode: cfg NULL == dense */
int poa_batch_run_ex(poa_batch_t* b, const poa_costs_t* costs, const poa_config_t* cfg, void* stream);
int poa_batch_fetch(poa_batch_t* b, uint32_t* score, poa_aln_pair_t* pairs, uint64_t* pair_off,
                    uint64_t pair_capacity, uint32_t* flags, poa_stats_t* stats);
/* synchronise the stream and return the HIP-event timings accumulated over every poa_batch_run
 * since the last poa_batch_stats / poa_batch_fetch call (no device->host result copy) */
int poa_batch_stats(poa_batch_t* b, poa_stats_t* stats);
/* device pointers of the results of the last run (valid until the next run / destroy):
 * score u32[n], flags u32[n], pair_off u64[n+1], pairs poa_aln_pair_t[pair_off[n]] */
int poa_batch_device_results(poa_batch_t* b, void** score, void** flags, void** pair_off, void** pairs);
/* AstarResult::{num_queued, num_visited, num_pruned} (src/aligner/astar.rs:81-90) of the queries whose result came from the
 * replayed search in the last exact / hybrid run, plus the number of steps the wave search took: out[4 * n], one
 * {num_queued, num_visited, num_pruned, steps} per query (zeros for queries that were not replayed). */
int poa_batch_fetch_search_counters(poa_batch_t* b, uint32_t* out);
/* how the dense pass of the last run stored its score planes: POA_LAYOUT_* bits.  Chosen per run from a bound on the optimal
 * score (u32 Score values of the reference, src/aligner/scoring/mod.rs:64-70, are kept verbatim unless a narrower
 * encoding is provably exact for everything the result depends on). */
#define POA_LAYOUT_U16 1u       /* 2-byte cells (bound <= 65534) */
#define POA_LAYOUT_COMPACT 2u   /* M plane + 4 flag bits per cell + the D rows that are read back */
#define POA_LAYOUT_RELATIVE 4u  /* cells hold score - e * (shortest-path depth of the row - column): scores beyond u16 */
int poa_batch_last_layout(poa_batch_t* b, uint32_t* layout);
/* debugging / parity: copy the M, I, D score planes of query i (rows x (len+1), row = topological
 * rank, see poa_graph_node_rows) — only valid if the query's chunk was the last one run */
int poa_batch_fetch_planes(poa_batch_t* b, uint32_t query, uint32_t* m, uint32_t* i, uint32_t* d);
void poa_batch_destroy(poa_batch_t* b);

#ifdef __cplusplus
}
#endif
#endif /* POASTA_AMD_H */
