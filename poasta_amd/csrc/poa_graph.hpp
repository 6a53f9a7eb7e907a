// Host-side flattening of the reference's `AlignableRefGraph` view (src/graphs/mod.rs:23-53)
// into the row-ordered tables the gfx950 kernels walk.  Product code (not the oracle).
//
// Row order: a topological order that keeps chains contiguous (ready-stack Kahn), so that for
// most rows the only predecessor is the previous row and its M/D values are still in registers.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace poa_amd {

enum : uint8_t {
    ROW_START = 1,         // the start sentinel (src/graphs/poa.rs:102)
    ROW_END = 2,           // the end sentinel  (src/graphs/poa.rs:103)
    ROW_OPENI_ALWAYS = 4,  // has an edge to end, or >= 2 distinct child symbols
                           // (insertion-open rule, src/aligner/scoring/gap_affine.rs:360-366,:413-421)
    ROW_OPENI_NEVER = 8,   // no successors at all (only the end row)
    ROW_CHAIN = 16,        // exactly one predecessor and it is the previous row
    ROW_STORE_D = 32,      // some successor reads this row's D from memory (it is not a chain row right below): keep the D row
    ROW_FAR_PRED = 64,     // some predecessor lies more than ROW_NEAR rows back (multi-wave kernel: beyond the hand-over ring)
    ROW_SAME_PREDS = 128,  // same predecessor set as the previous row (sibling nodes of a bubble / an MSA column): the
                           // predecessor minima of the previous row can be reused
};
constexpr uint32_t ROW_NEAR = 32;

// Layout of the table the replayed search fills (and the traceback of that pass reads): [row][offset][state] — the three states
// of a cell side by side (12 bytes), rows of `pitch` cells.  The replay is bound by instruction issue, not by memory (DESIGN.md
// §4): an index costs a multiply-add and a shift-add, the state is an immediate offset of the access, and the cells of a
// relaxation (same row or the row below, offsets j / j + 1) share their address arithmetic.  Rounds 1-2 kept 8 x 8-cell tiles —
// the reference's blocks, gap_affine.rs:435-446 — for their locality; their index arithmetic cost a tenth of the kernel (0.884 ->
// 0.785 s on configs[1]; plain [state][row][offset] planes: 0.80 s).
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint64_t ex_cell_index(uint32_t row, uint32_t off, uint32_t st, uint32_t n_rows, uint32_t pitch) {
    return ((uint64_t)row * pitch + off) * 3u + st;
}
// the same index in 32-bit arithmetic, for tables of fewer than 2^32 elements (the replay kernels check that)
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint32_t ex_cell_index32(uint32_t row, uint32_t off, uint32_t st, uint32_t n_rows, uint32_t pitch) {
    return (row * pitch + off) * 3u + st;
}

// Compact layout of one query, in 2-byte elements: [M: rows * pitch][4 flag bits per cell: rows * pitch / 4][kept D rows:
// n_store_d * pitch] (pitch is a multiple of 64, so every part starts 16-byte aligned)
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint64_t compact_plane_elems(uint32_t rows, uint32_t pitch, uint32_t n_store_d) {
    const uint64_t rp = (uint64_t)rows * pitch;
    return rp + rp / 4 + (uint64_t)n_store_d * pitch;
}

struct RowMeta {  // 16 bytes, one per row
    uint32_t node;        // node index in the host graph (rpos of AlignedPair)
    uint32_t pred_begin;  // first entry in pred_rows
    uint32_t pred_count;
    uint8_t sym;
    uint8_t child_sym;    // common symbol of the non-end children (valid unless ALWAYS/NEVER)
    uint8_t flags;
    uint8_t sym_idx;      // low nibble: index of sym in "ACGT" (8: other); high nibble: same for child_sym (4: none / open always);
                          // (sym_idx & 0x88) == 0: both masks of the row come from the one-strip kernel's LDS tables
};
static_assert(sizeof(RowMeta) == 16, "RowMeta layout");

struct FlatGraph {
    uint32_t n = 0, start = 0, end = 0;
    uint32_t n_real = 0;                 // nodes other than start/end
    std::vector<uint8_t> symbol;
    std::vector<uint32_t> succ_off, succ, pred_off, pred;  // trait iteration order (node ids)
    std::vector<uint32_t> node_row;      // node -> row
    std::vector<RowMeta> rows;           // row -> metadata
    std::vector<uint32_t> pred_rows;     // predecessors as rows, trait order preserved
    uint32_t start_row = 0, end_row = 0;
    uint32_t max_indegree = 0;
    uint32_t min_path_nodes = 0;         // real nodes on the shortest start -> end path
    std::vector<uint32_t> sp_to_end;     // row -> edges on the shortest path to the end row (0xFFFFFFFF: none)
    // Depth potential for the relative u16 encoding: row_depth[r] = nodes on the shortest start -> r path.
    // A cell is stored as  score - e * row_depth[r] + e * column  (>= 0: reaching row r costs at least e per node that is not
    // matched to a query symbol), under which every move keeps a non-negative cost: the deletion extension becomes
    // e * pred_k, the (mis)match x * [differs] + e * pred_k, the insertion extension 2e, where
    // pred_k[edge p -> r] = 1 + row_depth[p] - row_depth[r] >= 0 (0 along every shortest path, in particular for chain rows).
    // Compact plane layout: only rows flagged ROW_STORE_D keep their D row; those rows are stored back to back, row r at
    // slot d_slot[r] (0xFFFFFFFF: not kept); pred_dslot[k] = d_slot[pred_rows[k]] saves the dependent lookup.  Both empty
    // (and n_store_d = n): the D rows are stored by row, for graphs that keep more than half of them.
    std::vector<uint32_t> d_slot, pred_dslot;
    uint32_t n_store_d = 0;
    std::vector<uint32_t> row_depth;     // [n]
    std::vector<uint32_t> pred_k;        // [pred_rows.size()]

    // ---- exact-replay mode only: the reference's per-graph preprocessing --------------------
    // successors as rows, trait order preserved (DFA / expand_all iterate them in this order)
    std::vector<uint32_t> succ_row_off, succ_rows;
    // BubbleIndex (src/bubbles/index.rs:33-45), indexed by ROW
    std::vector<uint32_t> dist_min, dist_max;     // dist_to_end (min, max)
    std::vector<uint8_t> is_exit;                 // bubble_exit[node].is_exit()
    std::vector<uint32_t> exit_idx;               // row -> index among the exit rows (0xFFFFFFFF: not an exit)
    uint32_t n_exit = 0;
    struct NodeBubble { uint32_t exit_row, min_dist, max_dist; };
    std::vector<uint32_t> nbm_off;                // [n+1]
    std::vector<NodeBubble> nbm;                  // node_bubble_map flattened, per-node order preserved
    // Everything the replay's step reads of a row, in one 32-byte record (graphs of fewer than 65535 rows): successors, the
    // (up to two) bubbles it lies in besides the one it exits — exit row, its index among the exits, the distances to it —
    // the row's own distances to the end (what the min-gap heuristic reads, heuristic.rs:70-102, already minus one) and the
    // symbols.  flags says which parts hold the whole truth; a row that has more takes the generic code.
    struct RowRec {
        uint16_t c0, c1;             // successor rows in trait order (0xFFFF: none)
        uint16_t e0, e1;             // exit rows of the bubbles (node_bubble_map order; 0xFFFF: none)
        uint16_t x0, x1;             // their indices among the exit rows
        uint8_t d0min, d0max, d1min, d1max;
        uint16_t dmin, dmax;         // max(dist_to_end - 1, 0), (min, max)
        uint16_t mde0, mde1;         // max(dist_min[exit] - 1, 0)
        uint8_t sym, sym0, sym1, flags;
        uint32_t pad;
    };
    enum : uint8_t { RR_SUCC_OK = 1,   // one or two successors, neither the end row; c0 / c1 / sym0 / sym1 say all
                     RR_HAS_C1 = 2,
                     RR_PROBE_OK = 4,  // at most two such bubbles, distances below 256 and at most two apart
                     RR_END = 8 };
    std::vector<RowRec> row_rec;                  // empty: the graph has too many rows for 16-bit fields
    bool bubbles_built = false;
};

// BubbleIndex::new (src/bubbles/index.rs:51-156) over SuperbubbleFinder (src/bubbles/finder.rs:15-178)
// and rev_postorder_nodes (src/graphs/tools.rs:5-37).  Needed only by the exact-replay mode: it steers
// the reference's heuristic and pruning and therefore which cells its search visits.
int build_bubble_index(FlatGraph& g, std::string& err);
static_assert(sizeof(FlatGraph::RowRec) == 32, "RowRec layout");

// Returns POA_OK or POA_ERR_*; `err` receives a description.
int build_flat_graph(uint32_t n, uint32_t start, uint32_t end, const uint8_t* symbol,
                     const uint32_t* succ_off, const uint32_t* succ, const uint32_t* pred_off,
                     const uint32_t* pred, FlatGraph& out, std::string& err);

}  // namespace poa_amd
