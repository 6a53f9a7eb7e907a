// C ABI of the gfx950 POA alignment engine (include/poasta_amd.h).  Product code.
// Host side: graph flattening, query upload, chunked score-plane workspace, kernel launches on a
// caller-supplied HIP stream, result compaction and download.  No CPU alignment path exists here:
// without a HIP device every entry point that computes returns POA_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <new>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/poasta_amd.h"
#include "poa_graph.hpp"
#include "poa_exact_kernel.hpp"
#include "poa_wsearch.hpp"
#include "poa_fsearch.hpp"

// poa_config_t.tune: what a call overrides (the library reads no environment variable)
struct TuneView {
    int val[POA_TUNE_COUNT]; bool set[POA_TUNE_COUNT];
    explicit TuneView(const poa_config_t* cfg) { for (int k = 0; k < POA_TUNE_COUNT; ++k) { set[k] = cfg && cfg->tune[k] != 0; val[k] = set[k] ? (int)cfg->tune[k] - 1 : 0; } }
    const int* ptr(int k) const { return set[k] ? &val[k] : nullptr; }
};
#include "poa_kernels.hpp"
#include "poa_forward_packed.hpp"
#include "poa_forward_px.hpp"
#include "poa_twopiece.hpp"

using namespace poa_amd;

namespace {
thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return fail(POA_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));           \
    } while (0)

// Device buffers come from a small per-device cache of released blocks: a host that calls poa_align_batch once per
// batch of reads would otherwise spend more time in hipMalloc / hipFree (measured 12 ms per call for the ~15 buffers of
// a config-2 batch) than in the kernels.  poa_release_cache() returns everything to the driver.
struct BufCache {
    static constexpr size_t kMaxCachedBytes = 8ull << 30;  // per device, not counting the plane workspace
    std::mutex mu;
    std::multimap<size_t, void*> free_blocks[16];
    size_t cached_bytes[16] = {};
    void* take(int dev, size_t need, size_t* got) {
        if (dev < 0 || dev >= 16) return nullptr;
        std::lock_guard<std::mutex> lk(mu);
        auto it = free_blocks[dev].lower_bound(need);
        if (it == free_blocks[dev].end() || it->first > 2 * need + (1u << 20)) return nullptr;
        void* p = it->second;
        *got = it->first;
        cached_bytes[dev] -= it->first;
        free_blocks[dev].erase(it);
        return p;
    }
    bool give(int dev, void* p, size_t bytes) {
        if (dev < 0 || dev >= 16) return false;
        std::lock_guard<std::mutex> lk(mu);
        if (cached_bytes[dev] + bytes > kMaxCachedBytes) return false;
        free_blocks[dev].emplace(bytes, p);
        cached_bytes[dev] += bytes;
        return true;
    }
    void release_all() {
        std::lock_guard<std::mutex> lk(mu);
        for (int d = 0; d < 16; ++d) {
            if (free_blocks[d].empty()) continue;
            (void)hipSetDevice(d);
            for (auto& kv : free_blocks[d]) (void)hipFree(kv.second);
            free_blocks[d].clear();
            cached_bytes[d] = 0;
        }
    }
};
static BufCache g_buf_cache;

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    size_t cap_bytes = 0;
    int dev = -1;
    ~DevBuf() { drop(); }
    void drop() {
        if (!p) return;
        if (!g_buf_cache.give(dev, p, cap_bytes)) { (void)hipSetDevice(dev); (void)hipFree(p); }
        p = nullptr; cap_bytes = 0;
    }
    hipError_t alloc(size_t count) {
        drop();
        n = count;
        if (count == 0) return hipSuccess;
        if (hipGetDevice(&dev) != hipSuccess) dev = -1;
        const size_t need = (count * sizeof(T) + 255) & ~(size_t)255;
        if (void* c = g_buf_cache.take(dev, need, &cap_bytes)) { p = (T*)c; return hipSuccess; }
        cap_bytes = need;
        return hipMalloc((void**)&p, need);
    }
};
}  // namespace

// One cached plane workspace per device.  hipMalloc / hipFree of tens of GB cost seconds (measured: 0.8-6 s per
// poa_align_batch on config 2 against 10 ms of kernels), and a host that calls poa_align_batch once per batch of reads
// must not pay that every time: a released workspace is parked here and handed to the next batch on that device.
struct PlaneWorkspace {
    uint32_t* p = nullptr;
    size_t bytes = 0;
    int device = -1;
    bool acquire(int dev, size_t need, std::string& err);
    void release();
    ~PlaneWorkspace() { release(); }
};
namespace {
struct WsCache {
    std::mutex mu;
    void* p[16] = {};
    size_t bytes[16] = {};
} g_ws_cache;
}  // namespace
bool PlaneWorkspace::acquire(int dev, size_t need, std::string& err) {
    release();
    device = dev;
    if (dev >= 0 && dev < 16) {
        std::lock_guard<std::mutex> lk(g_ws_cache.mu);
        if (g_ws_cache.p[dev] && g_ws_cache.bytes[dev] >= need) {
            p = (uint32_t*)g_ws_cache.p[dev]; bytes = g_ws_cache.bytes[dev];
            g_ws_cache.p[dev] = nullptr; g_ws_cache.bytes[dev] = 0;
            return true;
        }
        if (g_ws_cache.p[dev]) {  // too small: give it back before asking for a bigger one
            (void)hipFree(g_ws_cache.p[dev]);
            g_ws_cache.p[dev] = nullptr; g_ws_cache.bytes[dev] = 0;
        }
    }
    const hipError_t e = hipMalloc((void**)&p, need);
    if (e != hipSuccess) { p = nullptr; err = hipGetErrorString(e); return false; }
    bytes = need;
    return true;
}
void PlaneWorkspace::release() {
    if (!p) return;
    void* victim = p;
    if (device >= 0 && device < 16) {
        std::lock_guard<std::mutex> lk(g_ws_cache.mu);
        if (!g_ws_cache.p[device] || g_ws_cache.bytes[device] < bytes) {
            victim = g_ws_cache.p[device];
            g_ws_cache.p[device] = p; g_ws_cache.bytes[device] = bytes;
        }
    }
    if (victim) { (void)hipSetDevice(device); (void)hipFree(victim); }
    p = nullptr; bytes = 0;
}

struct poa_graph {
    FlatGraph g;
    std::mutex bubble_mu;   // the bubble index (exact / hybrid mode only) is built on first use; batches on other threads may share the handle
};

struct poa_batch {
    const poa_graph* graph = nullptr;
    int device = 0;
    uint32_t n_queries = 0;
    uint64_t total_bases = 0, total_cells = 0, plane_bytes_total = 0;
    std::vector<uint64_t> h_qoff;
    std::vector<uint32_t> h_pitch;
    std::vector<uint64_t> h_scratch_off;
    struct Chunk { uint32_t first, count; };
    // How the queries share the plane workspace.  plan[0]: 4-byte elements (u32 planes, exact replay);
    // plan[1]: 2-byte elements, three full planes (POA_CFG_FULL_PLANES with u16 scores) — twice the queries per chunk when
    // the batch needs chunks; plan[2]: the compact layout at its real size (compact_plane_elems, ~2.7 bytes per cell).
    struct Plan {
        std::vector<Chunk> chunks;
        std::vector<uint64_t> off;   // per query: offset of its planes in ELEMENTS of the layout's type
        uint32_t max_chunk = 0;
        DevBuf<uint64_t> d_off;
    };
    Plan plan[3];
    bool plan16_same = true;         // plan[1], plan[2] not built: everything fits in one chunk anyway
    int active_plan = 0;             // plan of the last run
    const Plan& cur() const { return plan[active_plan]; }
    uint64_t max_len = 0;
    bool narrow = false;  // last run used u16 planes
    bool compact = false; // last run used the compact layout (no I plane, partial D)
    int cols_per_lane = 16;

    DevBuf<RowMeta> d_rows;
    DevBuf<uint32_t> d_pred_rows;
    DevBuf<uint32_t> d_row_depth, d_pred_k;   // depth potential of the relative u16 encoding (FlatGraph::row_depth / pred_k)
    DevBuf<uint32_t> d_dslot, d_pred_dslot;   // compact layout: slots of the kept D rows (FlatGraph::d_slot / pred_dslot)
    bool relative = false;                    // last run stored scores relative to that potential
    bool dense_narrow = false, dense_compact = false, dense_relative = false;   // layout of the last run's dense pass (poa_batch_last_layout)
    DevBuf<uint8_t> d_qseq;
    DevBuf<uint64_t> d_qoff, d_scratch_off, d_pair_off;
    DevBuf<uint32_t> d_pitch, d_carry, d_score, d_flags, d_npairs;
    PlaneWorkspace d_planes;
    DevBuf<uint2> d_scratch, d_pairs;
    // exact-replay mode (allocated on first use)
    DevBuf<uint32_t> d_succ_off, d_succ_rows, d_dist_min, d_dist_max, d_nbm_off, d_ex_status, d_exit_idx, d_ex_head, d_node_row, d_sp_to_end, d_ex_end;
    DevBuf<FlatGraph::NodeBubble> d_nbm;
    DevBuf<uint8_t> d_ex_sym;
    DevBuf<uint64_t> d_ex_reached, d_ex_rsum;
    DevBuf<ExQEntry> d_ex_pool;
    DevBuf<ExStackEntry> d_ex_stack;
    uint32_t ex_n_prio = 0, ex_pool_cap = 0, ex_stack_cap = 0, ex_wpn = 0, ex_swpn = 0;
    uint32_t ex_win = 64;              // wave search: priorities in the descriptor ring (power of two)
    uint32_t ex_skew = 0;              // max over nodes of dist_to_end max - min: bounds how far the min-gap heuristic can grow along a greedy extension
    DevBuf<uint32_t> d_ex_order;         // wave search, persistent scheduling: [0] work counter, [1..] query order
    DevBuf<uint32_t> d_pipeline_error;   // FwdParams::pipeline_error
    DevBuf<unsigned long long> d_ex_prof;
    DevBuf<uint32_t> d_ex_counters;    // wave search: num_queued, num_visited, num_pruned, steps per query
    DevBuf<uint32_t> d_ex_logs;        // parallel-step search (poa_fsearch.hpp): the lanes' push logs, per resident wave
    DevBuf<FlatGraph::RowRec> d_ex_rec; // per-row records of its lean step (FlatGraph::row_rec)
    bool exact_ready = false;
    bool prof_on = false;              // the last run asked for the replay kernel's cycle counts (POA_TUNE_WS_PROF)
    uint32_t last_mode = 0;

    // one event set per run since the last stats call: [begin, (fwd_end, tb_end) per chunk..., end]
    std::vector<std::vector<hipEvent_t>> runs;
    std::vector<std::vector<hipEvent_t>> free_sets;
    bool ran = false;
    hipStream_t last_stream = nullptr;
    float ms_h2d = 0.f;

    ~poa_batch() {
        for (auto& r : runs) for (auto e : r) (void)hipEventDestroy(e);
        for (auto& r : free_sets) for (auto e : r) (void)hipEventDestroy(e);
    }
};

// sums the HIP-event timings of every run recorded since the last call; the stream must be idle.
// a wave of the multi-wave pipeline gave up waiting for its neighbour (mw_wait_gt): the planes are not to be trusted
static int check_pipeline_error(poa_batch* b) {
    uint32_t w = 0;
    HIP_TRY(hipMemcpy(&w, b->d_pipeline_error.p, 4, hipMemcpyDeviceToHost));
    if (w) {
        (void)hipMemset(b->d_pipeline_error.p, 0, 4);
        return fail(POA_ERR_HIP, "multi-wave forward pipeline timed out waiting for a neighbouring strip: results discarded");
    }
    return POA_OK;
}

static void collect_stats(poa_batch* b, poa_stats_t* stats) {
    const uint32_t n = b->n_queries;
    const uint32_t keep_flagged = stats->n_flagged;
    stats->cells = b->total_cells; stats->bases = b->total_bases; stats->plane_bytes = b->plane_bytes_total;
    stats->n_queries = n; stats->n_chunks = (uint32_t)b->cur().chunks.size();
    stats->n_flagged = keep_flagged;
    stats->ms_h2d = b->ms_h2d; stats->ms_d2h = 0.f;
    float fwd = 0.f, tb = 0.f, ex = 0.f, total = 0.f;
    uint32_t launches = 0;
    for (auto& events : b->runs) {
        if (n) {
            size_t ev = 1;
            hipEvent_t prev = events[0];
            const size_t n_chunks = (events.size() - 2) / 3;
            for (size_t c = 0; c < n_chunks; ++c) {
                float a = 0.f, t2 = 0.f, t3 = 0.f;
                (void)hipEventElapsedTime(&a, prev, events[ev]);
                (void)hipEventElapsedTime(&t2, events[ev], events[ev + 1]);
                (void)hipEventElapsedTime(&t3, events[ev + 1], events[ev + 2]);
                fwd += a; tb += t2; ex += t3;
                prev = events[ev + 2];
                ev += 3;
                launches++;
            }
            float tail = 0.f, tot = 0.f;
            (void)hipEventElapsedTime(&tail, prev, events[ev]);
            tb += tail;
            (void)hipEventElapsedTime(&tot, events[0], events[ev]);
            total += tot;
        }
    }
    stats->n_runs = (uint32_t)b->runs.size();
    stats->n_forward_launches = launches;
    stats->ms_forward = fwd; stats->ms_traceback = tb; stats->ms_exact = ex; stats->ms_total = total;
    for (auto& r : b->runs) b->free_sets.push_back(std::move(r));
    b->runs.clear();
}

// the 1024-column multi-wave kernel when the chunk fills the chip with one wave per 1024 columns (else 512-column strips of
// the adjacent-pairs kernel give twice the waves)
// waves per workgroup of a multi-wave launch: all strips at once when they fit, else the strips spread evenly over the
// groups that run one after the other (16 + 4 strips would leave twelve waves idle in the second group)
static uint32_t mw_waves(uint32_t strips) {
    const uint32_t groups = (strips + MW_MAX_WAVES - 1) / MW_MAX_WAVES;
    return (strips + groups - 1) / groups;
}

static bool pxmw_ok(const TuneView& T, uint32_t count, uint32_t max_pitch) {
    if (const int* v = T.ptr(POA_TUNE_PXMW)) return (*v) != 0;
    return (uint64_t)count * ((max_pitch + 1023) / 1024) >= 1024;
}

extern "C" {

const char* poa_version(void) { return "poasta_amd 0.1 (gfx950)"; }
const char* poa_last_error(void) { return g_err.c_str(); }

int poa_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int poa_graph_create(uint32_t n, uint32_t start, uint32_t end, const uint8_t* symbol, const uint32_t* succ_off,
                     const uint32_t* succ, const uint32_t* pred_off, const uint32_t* pred, poa_graph_t** out) {
    if (!out) return fail(POA_ERR_INVALID_ARG, "poa_graph_create: out is null");
    *out = nullptr;
    std::unique_ptr<poa_graph> h(new (std::nothrow) poa_graph);
    if (!h) return fail(POA_ERR_OUT_OF_MEMORY, "poa_graph_create: host allocation failed");
    std::string err;
    int rc;
    try {
        rc = build_flat_graph(n, start, end, symbol, succ_off, succ, pred_off, pred, h->g, err);
    } catch (const std::bad_alloc&) {
        return fail(POA_ERR_OUT_OF_MEMORY, "poa_graph_create: host allocation failed");
    }
    if (rc != POA_OK) return fail(rc, err);
    *out = h.release();
    return POA_OK;
}

void poa_graph_destroy(poa_graph_t* g) { delete g; }
uint32_t poa_graph_rows(const poa_graph_t* g) { return g ? g->g.n : 0; }
int poa_graph_update(poa_graph_t* g, uint32_t n, uint32_t start, uint32_t end, const uint8_t* symbol, const uint32_t* succ_off,
                     const uint32_t* succ, const uint32_t* pred_off, const uint32_t* pred) {
    if (!g) return fail(POA_ERR_INVALID_ARG, "poa_graph_update: null graph");
    std::string err;
    int rc;
    FlatGraph ng;
    try {
        rc = build_flat_graph(n, start, end, symbol, succ_off, succ, pred_off, pred, ng, err);
    } catch (const std::bad_alloc&) {
        return fail(POA_ERR_OUT_OF_MEMORY, "poa_graph_update: host allocation failed");
    }
    if (rc != POA_OK) return fail(rc, err);   // (the handle keeps the graph it had)
    std::lock_guard<std::mutex> lk(g->bubble_mu);
    g->g = std::move(ng);
    return POA_OK;
}

int poa_graph_node_rows(const poa_graph_t* g, uint32_t* rank) {
    if (!g || !rank) return fail(POA_ERR_INVALID_ARG, "poa_graph_node_rows: null argument");
    std::memcpy(rank, g->g.node_row.data(), g->g.n * sizeof(uint32_t));
    return POA_OK;
}

int poa_batch_create(const poa_graph_t* g, int device, uint32_t n_queries, const uint8_t* qseq, const uint64_t* qoff,
                     uint64_t workspace_bytes, poa_batch_t** out) {
    if (!out) return fail(POA_ERR_INVALID_ARG, "poa_batch_create: out is null");
    *out = nullptr;
    if (!g || !qoff || (n_queries && qoff[n_queries] && !qseq))
        return fail(POA_ERR_INVALID_ARG, "poa_batch_create: null argument");
    for (uint32_t i = 0; i < n_queries; ++i) {
        if (qoff[i + 1] < qoff[i]) return fail(POA_ERR_INVALID_ARG, "poa_batch_create: qoff not monotone");
        if (qoff[i + 1] - qoff[i] > 0x7FFFFFF0ull) return fail(POA_ERR_UNSUPPORTED, "query longer than 2^31");
    }
    int ndev = poa_device_count();
    if (ndev <= 0) return fail(POA_ERR_NO_DEVICE, "no HIP device visible: the gfx950 path has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(POA_ERR_INVALID_ARG, "poa_batch_create: device ordinal out of range");
    HIP_TRY(hipSetDevice(device));

    std::unique_ptr<poa_batch> b(new (std::nothrow) poa_batch);
    if (!b) return fail(POA_ERR_OUT_OF_MEMORY, "host allocation failed");
    const FlatGraph& fg = g->g;
    b->graph = g; b->device = device; b->n_queries = n_queries;
    const uint32_t rows = fg.n;
    try {
        b->h_qoff.assign(qoff, qoff + n_queries + 1);
        b->h_pitch.resize(n_queries);
        b->h_scratch_off.resize((size_t)n_queries + 1);
    } catch (const std::bad_alloc&) { return fail(POA_ERR_OUT_OF_MEMORY, "host allocation failed"); }

    uint64_t scratch_total = 0, max_len = 0;
    std::vector<uint64_t> q_plane_elems(n_queries);
    for (uint32_t i = 0; i < n_queries; ++i) {
        const uint64_t L = qoff[i + 1] - qoff[i];
        max_len = std::max(max_len, L);
        const uint32_t pitch = (uint32_t)(((L + 1 + 63) / 64) * 64);
        b->h_pitch[i] = pitch;
        q_plane_elems[i] = 3ull * rows * pitch;
        b->h_scratch_off[i] = scratch_total;
        scratch_total += L + rows;
        b->max_len = std::max<uint64_t>(b->max_len, L);
        b->total_bases += L;
        b->total_cells += (uint64_t)rows * (L + 1);
        b->plane_bytes_total += q_plane_elems[i] * 4;
    }
    b->h_scratch_off[n_queries] = scratch_total;

    // workspace: as many queries' planes as fit; chunks reuse it.
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    const uint64_t fixed = scratch_total * 16 + (uint64_t)n_queries * 64 + qoff[n_queries] + (64ull << 20);
    uint64_t ws = workspace_bytes;
    if (ws == 0) {
        const uint64_t avail = free_b > fixed ? (uint64_t)((free_b - fixed) * 0.85) : 0;
        ws = std::min<uint64_t>(b->plane_bytes_total, avail);
    }
    uint64_t biggest = 0;
    for (uint32_t i = 0; i < n_queries; ++i) biggest = std::max(biggest, q_plane_elems[i] * 4);
    if (ws < biggest) {
        if (workspace_bytes == 0 || workspace_bytes < biggest) {
            if (biggest + fixed > free_b)
                return fail(POA_ERR_OUT_OF_MEMORY, "score planes of the largest query do not fit in device memory");
            ws = biggest;
        }
    }
    // Greedy chunking, once per layout — with the chunks evened out: the forward kernels that take a workgroup per query
    // finish with their most loaded CU, so 2 000 queries that fit 527 at a time run as 4 x 500, not 3 x 527 + 419
    // (measured on configs[4]: 434 -> 320 ms).
    auto make_plan = [&](poa_batch::Plan& pl, uint64_t elem_bytes, bool compact_size) {
        auto need_of = [&](uint32_t i) {
            return (compact_size ? compact_plane_elems(rows, b->h_pitch[i], fg.n_store_d) : q_plane_elems[i]) * elem_bytes;
        };
        auto greedy = [&](uint64_t budget, std::vector<poa_batch::Chunk>& chunks, std::vector<uint64_t>& off) {
            chunks.clear();
            off.assign(n_queries, 0);
            uint32_t first = 0;
            uint64_t used = 0;
            for (uint32_t i = 0; i < n_queries; ++i) {
                const uint64_t need = need_of(i);
                if (used + need > budget && i > first) {
                    chunks.push_back({first, i - first});
                    first = i; used = 0;
                }
                off[i] = used / elem_bytes;
                used += need;
            }
            if (n_queries > first) chunks.push_back({first, n_queries - first});
        };
        greedy(ws, pl.chunks, pl.off);
        if (pl.chunks.size() > 1) {
            uint64_t total = 0;
            for (uint32_t i = 0; i < n_queries; ++i) total += need_of(i);
            const uint64_t target = (total + pl.chunks.size() - 1) / pl.chunks.size();
            std::vector<poa_batch::Chunk> c2;
            std::vector<uint64_t> o2;
            for (double slack : {1.0, 1.02, 1.05, 1.1}) {
                const uint64_t budget = std::min<uint64_t>(ws, (uint64_t)((double)target * slack));
                greedy(budget, c2, o2);
                if (c2.size() == pl.chunks.size()) { pl.chunks = c2; pl.off = o2; break; }
            }
        }
        for (auto& c : pl.chunks) pl.max_chunk = std::max(pl.max_chunk, c.count);
    };
    make_plan(b->plan[0], 4, false);
    b->plan16_same = b->plan[0].chunks.size() <= 1;
    if (!b->plan16_same) { make_plan(b->plan[1], 2, false); make_plan(b->plan[2], 2, true); }
    const uint32_t max_chunk_any = std::max(b->plan[0].max_chunk, std::max(b->plan[1].max_chunk, b->plan[2].max_chunk));
    b->cols_per_lane = 16;

    // device buffers
    HIP_TRY(b->d_rows.alloc(fg.rows.size()));
    HIP_TRY(b->d_pred_rows.alloc(std::max<size_t>(fg.pred_rows.size(), 1)));
    HIP_TRY(b->d_pred_k.alloc(std::max<size_t>(fg.pred_k.size(), 1)));
    HIP_TRY(b->d_row_depth.alloc(std::max<size_t>(fg.row_depth.size(), 1)));
    HIP_TRY(b->d_dslot.alloc(std::max<size_t>(fg.d_slot.size(), 1)));
    HIP_TRY(b->d_pred_dslot.alloc(std::max<size_t>(fg.pred_dslot.size(), 1)));
    HIP_TRY(b->d_qseq.alloc(std::max<uint64_t>(qoff[n_queries], 1)));
    HIP_TRY(b->d_qoff.alloc((size_t)n_queries + 1));
    HIP_TRY(b->d_pitch.alloc(std::max<uint32_t>(n_queries, 1)));
    for (int k = 0; k < (b->plan16_same ? 1 : 3); ++k) HIP_TRY(b->plan[k].d_off.alloc(std::max<uint32_t>(n_queries, 1)));
    HIP_TRY(b->d_scratch_off.alloc((size_t)n_queries + 1));
    HIP_TRY(b->d_pair_off.alloc((size_t)n_queries + 1));
    HIP_TRY(b->d_score.alloc(std::max<uint32_t>(n_queries, 1)));
    HIP_TRY(b->d_flags.alloc(std::max<uint32_t>(n_queries, 1)));
    HIP_TRY(b->d_npairs.alloc(std::max<uint32_t>(n_queries, 1)));
    HIP_TRY(b->d_scratch.alloc(std::max<uint64_t>(scratch_total, 1)));
    HIP_TRY(b->d_pairs.alloc(std::max<uint64_t>(scratch_total, 1)));
    HIP_TRY(b->d_carry.alloc(std::max<uint64_t>(2ull * max_chunk_any * rows, 1)));
    HIP_TRY(b->d_pipeline_error.alloc(1));
    HIP_TRY(hipMemset(b->d_pipeline_error.p, 0, 4));
    if (n_queries) {
        std::string werr;
        if (!b->d_planes.acquire(device, ws + 256, werr)) return fail(POA_ERR_OUT_OF_MEMORY, "score-plane workspace: " + werr);
    }

    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, nullptr));
    HIP_TRY(hipMemcpy(b->d_rows.p, fg.rows.data(), fg.rows.size() * sizeof(RowMeta), hipMemcpyHostToDevice));
    if (!fg.pred_rows.empty())
        HIP_TRY(hipMemcpy(b->d_pred_rows.p, fg.pred_rows.data(), fg.pred_rows.size() * 4, hipMemcpyHostToDevice));
    if (!fg.pred_k.empty())
        HIP_TRY(hipMemcpy(b->d_pred_k.p, fg.pred_k.data(), fg.pred_k.size() * 4, hipMemcpyHostToDevice));
    if (!fg.row_depth.empty())
        HIP_TRY(hipMemcpy(b->d_row_depth.p, fg.row_depth.data(), fg.row_depth.size() * 4, hipMemcpyHostToDevice));
    if (!fg.d_slot.empty())
        HIP_TRY(hipMemcpy(b->d_dslot.p, fg.d_slot.data(), fg.d_slot.size() * 4, hipMemcpyHostToDevice));
    if (!fg.pred_dslot.empty())
        HIP_TRY(hipMemcpy(b->d_pred_dslot.p, fg.pred_dslot.data(), fg.pred_dslot.size() * 4, hipMemcpyHostToDevice));
    if (qoff[n_queries]) HIP_TRY(hipMemcpy(b->d_qseq.p, qseq, qoff[n_queries], hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b->d_qoff.p, qoff, ((size_t)n_queries + 1) * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b->d_scratch_off.p, b->h_scratch_off.data(), ((size_t)n_queries + 1) * 8, hipMemcpyHostToDevice));
    if (n_queries) {
        HIP_TRY(hipMemcpy(b->d_pitch.p, b->h_pitch.data(), (size_t)n_queries * 4, hipMemcpyHostToDevice));
        for (int k = 0; k < (b->plan16_same ? 1 : 3); ++k)
            HIP_TRY(hipMemcpy(b->plan[k].d_off.p, b->plan[k].off.data(), (size_t)n_queries * 8, hipMemcpyHostToDevice));
    }
    HIP_TRY(hipEventRecord(e1, nullptr));
    HIP_TRY(hipEventSynchronize(e1));
    (void)hipEventElapsedTime(&b->ms_h2d, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);

    *out = b.release();
    return POA_OK;
}

static int prepare_exact(poa_batch* b, const poa_costs_t* costs, const poa_config_t* cfg) {
    const FlatGraph& fg = b->graph->g;
    std::string err;
    int rc;
    {
        std::lock_guard<std::mutex> lk(const_cast<poa_graph*>(b->graph)->bubble_mu);
        rc = build_bubble_index(const_cast<FlatGraph&>(fg), err);
    }
    if (rc != POA_OK) return fail(rc, err);
    const uint32_t n = fg.n;
    {
        // spread of the path lengths to the end: bounds how far the min-gap heuristic can grow along a greedy extension
        // (heuristic.rs:70-102), hence the priority range the wave search keeps live at once (its descriptor ring)
        uint32_t skew = 0;
        for (uint32_t r = 0; r < n; ++r) skew = std::max<uint32_t>(skew, fg.dist_max[r] - std::min(fg.dist_min[r], fg.dist_max[r]));
        b->ex_skew = skew;
    }
    const uint32_t maxc = std::max<uint32_t>(costs->mismatch, (uint32_t)costs->gap_open + costs->gap_extend);
    const uint64_t n_prio64 = ((uint64_t)n + b->max_len + 2) * maxc + costs->gap_open + ((uint64_t)n + b->max_len) * costs->gap_extend + 64;
    if (n_prio64 > (1ull << 26)) return fail(POA_ERR_UNSUPPORTED, "exact replay: priority range too large for this graph / query size");
    if (3ull * n * (((uint64_t)b->max_len + 64) & ~63ull) >= (1ull << 32))
        return fail(POA_ERR_UNSUPPORTED, "exact replay: the visited table of one query exceeds 2^32 cells");
    if ((uint64_t)fg.n_exit * ((b->max_len + 64) / 64) >= (1ull << 32))   // (the reached sets are indexed in 32-bit arithmetic)
        return fail(POA_ERR_UNSUPPORTED, "exact replay: reached sets of one query exceed 2^32 words");
    const float f = (cfg && cfg->queue_entries_per_cell > 0.f) ? cfg->queue_entries_per_cell : 0.25f;
    const uint64_t pool64 = std::max<uint64_t>(256, (uint64_t)(f * (double)n * (double)(b->max_len + 1)));
    if (pool64 > 0xFFFFFFF0ull) return fail(POA_ERR_UNSUPPORTED, "exact replay: queue pool too large");
    // wave search: the descriptor ring covers `win` priorities (see the launch), and every live stack holds a chunk of its own
    uint32_t win = 64;
    {
        const uint64_t win_need = (uint64_t)maxc + costs->gap_open + ((uint64_t)b->ex_skew + 3) * costs->gap_extend + 8;
        while (win < win_need && win < (1u << 24)) win *= 2;
    }
    const uint64_t pool_ws = pool64 + (uint64_t)BQ_CHUNK * (3ull * win + 8);
    if (pool_ws > 0xFFFFFFF0ull) return fail(POA_ERR_UNSUPPORTED, "exact replay: queue pool too large");
    const uint32_t n_prio = (uint32_t)n_prio64, pool_cap = (uint32_t)pool_ws;
    b->ex_win = win;
    const uint32_t stack_cap = (uint32_t)(n + b->max_len + 8), wpn = (uint32_t)((b->max_len + 1 + 63) / 64), swpn = (wpn + 63) / 64;
    if (b->exact_ready && b->ex_n_prio >= n_prio && b->ex_pool_cap >= pool_cap) return POA_OK;
    // the workspace below is re-allocated: released blocks go to the buffer cache, where another batch may pick them up,
    // so nothing of an earlier run on this batch may still be in flight
    if (b->ran) HIP_TRY(hipStreamSynchronize(b->last_stream));
    const uint64_t slots = b->plan[0].max_chunk;
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    const uint64_t need = slots * ((uint64_t)fg.n_exit * (wpn + swpn) * 8 + 3ull * n_prio * 4 + (uint64_t)stack_cap * 12 + (uint64_t)pool_cap * 16);
    if (need + (256ull << 20) > free_b + b->d_ex_pool.n * 16 + b->d_ex_head.n * 4 + b->d_ex_reached.n * 8)
        return fail(POA_ERR_OUT_OF_MEMORY, "exact replay workspace does not fit: create the batch with a smaller workspace_bytes (fewer queries per chunk) or lower queue_entries_per_cell");
    if (!b->exact_ready) {
        HIP_TRY(b->d_succ_off.alloc(fg.succ_row_off.size()));
        HIP_TRY(b->d_succ_rows.alloc(std::max<size_t>(fg.succ_rows.size(), 1)));
        HIP_TRY(b->d_dist_min.alloc(n)); HIP_TRY(b->d_dist_max.alloc(n)); HIP_TRY(b->d_exit_idx.alloc(n));
        HIP_TRY(b->d_node_row.alloc(n)); HIP_TRY(b->d_sp_to_end.alloc(n));
        HIP_TRY(b->d_ex_end.alloc(2 * (size_t)std::max<uint32_t>(b->n_queries, 1)));
        HIP_TRY(hipMemcpy(b->d_node_row.p, fg.node_row.data(), n * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(b->d_sp_to_end.p, fg.sp_to_end.data(), n * 4, hipMemcpyHostToDevice));
        HIP_TRY(b->d_ex_sym.alloc((size_t)n + 4));
        {
            std::vector<uint8_t> row_sym((size_t)n + 4, 0);
            for (uint32_t r = 0; r < n; ++r) row_sym[r] = fg.rows[r].sym;
            HIP_TRY(hipMemcpy(b->d_ex_sym.p, row_sym.data(), row_sym.size(), hipMemcpyHostToDevice));
        }
        HIP_TRY(b->d_nbm_off.alloc(n + 1)); HIP_TRY(b->d_nbm.alloc(std::max<size_t>(fg.nbm.size(), 1)));
        if (!fg.row_rec.empty()) {
            HIP_TRY(b->d_ex_rec.alloc(fg.row_rec.size()));
            HIP_TRY(hipMemcpy(b->d_ex_rec.p, fg.row_rec.data(), fg.row_rec.size() * sizeof(FlatGraph::RowRec), hipMemcpyHostToDevice));
        }
        HIP_TRY(b->d_ex_status.alloc(std::max<uint32_t>(b->n_queries, 1)));
        HIP_TRY(b->d_ex_counters.alloc(4 * (size_t)std::max<uint32_t>(b->n_queries, 1)));
        HIP_TRY(b->d_ex_prof.alloc(8 * (size_t)std::max<uint32_t>(b->n_queries, 1)));   // (diagnostics; once, not per chunk while a kernel may write it)
        HIP_TRY(hipMemcpy(b->d_succ_off.p, fg.succ_row_off.data(), fg.succ_row_off.size() * 4, hipMemcpyHostToDevice));
        if (!fg.succ_rows.empty()) HIP_TRY(hipMemcpy(b->d_succ_rows.p, fg.succ_rows.data(), fg.succ_rows.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(b->d_dist_min.p, fg.dist_min.data(), n * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(b->d_dist_max.p, fg.dist_max.data(), n * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(b->d_exit_idx.p, fg.exit_idx.data(), n * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(b->d_nbm_off.p, fg.nbm_off.data(), (n + 1) * 4, hipMemcpyHostToDevice));
        if (!fg.nbm.empty()) HIP_TRY(hipMemcpy(b->d_nbm.p, fg.nbm.data(), fg.nbm.size() * sizeof(FlatGraph::NodeBubble), hipMemcpyHostToDevice));
    }
    HIP_TRY(b->d_ex_reached.alloc(std::max<uint64_t>(slots * fg.n_exit * wpn, 1)));
    HIP_TRY(b->d_ex_rsum.alloc(std::max<uint64_t>(slots * fg.n_exit * swpn, 1)));
    HIP_TRY(b->d_ex_head.alloc(slots * 3 * n_prio));
    HIP_TRY(b->d_ex_stack.alloc(slots * stack_cap));
    {
        hipError_t e = b->d_ex_pool.alloc(slots * pool_cap);
        if (e != hipSuccess) return fail(POA_ERR_OUT_OF_MEMORY, std::string("exact replay queue pool: ") + hipGetErrorString(e));
    }
    b->ex_n_prio = n_prio; b->ex_pool_cap = pool_cap; b->ex_stack_cap = stack_cap; b->ex_wpn = wpn; b->ex_swpn = swpn;
    b->exact_ready = true;
    return POA_OK;
}

int poa_batch_run(poa_batch_t* b, const poa_costs_t* costs, void* stream_v) { return poa_batch_run_ex(b, costs, nullptr, stream_v); }

int poa_batch_run_ex(poa_batch_t* b, const poa_costs_t* costs, const poa_config_t* cfg, void* stream_v) {
    if (!b || !costs) return fail(POA_ERR_INVALID_ARG, "poa_batch_run: null argument");
    const TuneView T(cfg);   // what this call overrides, read once
    b->prof_on = T.ptr(POA_TUNE_WS_PROF) != nullptr;
    uint32_t mode = cfg ? cfg->mode : POA_MODE_DENSE;
    if (mode > POA_MODE_HYBRID) return fail(POA_ERR_INVALID_ARG, "poa_batch_run_ex: unknown mode");
    if (cfg && cfg->span > POA_SPAN_ENDS_FREE) return fail(POA_ERR_INVALID_ARG, "poa_batch_run_ex: unknown alignment span");
    const bool ends_free = cfg && cfg->span == POA_SPAN_ENDS_FREE;
    if (ends_free) {
        if (cfg->qry_free_end.kind > POA_BOUND_EXCLUDED || cfg->graph_free_begin.kind > POA_BOUND_EXCLUDED ||
            cfg->graph_free_end.kind > POA_BOUND_EXCLUDED || cfg->qry_free_begin.kind > POA_BOUND_EXCLUDED)
            return fail(POA_ERR_INVALID_ARG, "poa_batch_run_ex: unknown bound kind");
        mode = POA_MODE_EXACT;  // an ends-free result is defined by the reference's search: replay it for every query
    }
    if (cfg && cfg->heuristic > POA_HEURISTIC_MINGAP) return fail(POA_ERR_INVALID_ARG, "poa_batch_run_ex: unknown heuristic");
    hipStream_t stream = (hipStream_t)stream_v;
    HIP_TRY(hipSetDevice(b->device));
    const FlatGraph& fg = b->graph->g;
    if (mode != POA_MODE_DENSE && b->n_queries) {
        int rc = prepare_exact(b, costs, cfg);
        if (rc != POA_OK) return rc;
    }
    b->last_mode = mode;
    b->last_stream = stream;
    if (b->runs.size() >= 256) return fail(POA_ERR_UNSUPPORTED, "poa_batch_run: call poa_batch_stats/fetch at least every 256 runs");
    // u16 planes whenever every value that can matter fits.  u16 arithmetic saturates at 0xFFFF = INF, so every stored
    // value is min(true value, 0xFFFF); costs are non-negative, hence every cell on an optimal path — and every
    // predecessor candidate the traceback can accept — has a value <= the final score, and the final score is at most
    //     ub = [o + e*L] + [o + e*(nodes on the shortest start->end path)]      (insert the query, delete that path).
    // ub <= 65534  =>  everything the result depends on is exact in u16.  (The cruder bound (rows + L + 2) * max(x, o+e)
    // on ANY finite value is always >= ub.)  POA_PLANES=32 forces u32 (debug / A-B).
    const uint64_t ub = (b->max_len ? (uint64_t)costs->gap_open + (uint64_t)costs->gap_extend * b->max_len : 0) +
                        (fg.min_path_nodes ? (uint64_t)costs->gap_open + (uint64_t)costs->gap_extend * fg.min_path_nodes : 0);
    bool narrow = ub <= 65534;
    if (const int* pv = T.ptr(POA_TUNE_PLANES)) { if ((*pv) == 32) narrow = false; }
    // compact layout (u16 only): 4-bit codes instead of the I plane, D rows only where they are read back.
    // POA_CFG_FULL_PLANES (or POA_COMPACT=0) keeps all three planes, e.g. for poa_batch_fetch_planes.
    const bool want_full = cfg && (cfg->flags & POA_CFG_FULL_PLANES);
    bool compact = narrow && !want_full;
    if (const int* cv = T.ptr(POA_TUNE_COMPACT)) { if ((*cv) == 0) compact = false; }
    bool packed = true;  // packed-u16 arithmetic kernel for the compact layout (POA_PACKED=0: scalar u32 arithmetic)
    if (const int* pv2 = T.ptr(POA_TUNE_PACKED)) packed = (*pv2) != 0;
    // Scores beyond u16 (a read against a much longer graph, Global): store every cell relative to the depth potential
    // e * (row_depth - column) (FlatGraph::row_depth).  All moves keep non-negative costs there, so the saturation argument
    // above holds for the relative values, and the largest one on an optimal path is the end cell's:
    //     S* - e * (shortest path nodes - L)  <=  ub - e * min_path_nodes + e * L  =  2 * (o + e * L).
    // Only the pairs-across-quads kernels (compact layout) implement it; POA_RELATIVE=0 keeps u32 planes, =1 forces it
    // wherever the bound allows (A-B against the absolute encodings).
    const uint64_t rel_ub = 2 * ((uint64_t)costs->gap_open + (uint64_t)costs->gap_extend * b->max_len);
    bool relative = !narrow && rel_ub <= 65534 && !want_full && packed && !T.ptr(POA_TUNE_PLANES) && !T.ptr(POA_TUNE_COMPACT);
    if (const int* rv = T.ptr(POA_TUNE_RELATIVE)) {
        if ((*rv) == 0) relative = false;
        else relative = rel_ub <= 65534 && !want_full && packed;
    }
    if (relative) { narrow = true; compact = true; }
    b->relative = relative;
    b->dense_narrow = narrow; b->dense_compact = compact; b->dense_relative = relative;
    b->narrow = narrow;
    b->compact = compact;
    // 2-byte elements let twice the queries share the workspace; the exact replay needs the u32 plan
    b->active_plan = (narrow && mode == POA_MODE_DENSE && !b->plan16_same) ? (compact ? 2 : 1) : 0;
    const poa_batch::Plan& PL = b->cur();
    std::vector<hipEvent_t> events;
    const size_t n_events = 2 + 3 * PL.chunks.size();
    for (size_t k = 0; k < b->free_sets.size(); ++k) {
        if (b->free_sets[k].size() == n_events) {
            events = std::move(b->free_sets[k]);
            b->free_sets.erase(b->free_sets.begin() + (long)k);
            break;
        }
    }
    if (events.empty()) {
        events.resize(n_events);
        for (auto& e : events) HIP_TRY(hipEventCreate(&e));
    }
    b->runs.push_back(events);
    HIP_TRY(hipEventRecord(events[0], stream));
    if (b->n_queries == 0) {
        HIP_TRY(hipMemsetAsync(b->d_pair_off.p, 0, 8, stream));
        HIP_TRY(hipEventRecord(events[1], stream));
        b->ran = true;
        return POA_OK;
    }
    uint32_t spec_depth = 12;  // traceback speculation depth (lanes per round) of the full-plane layouts; the compact one: see the launch below
    if (const int* sv = T.ptr(POA_TUNE_TB_DEPTH)) { const int v = (*sv); if (v >= 1 && v <= 64) spec_depth = (uint32_t)v; }
    int tb_group = 16;   // POA_TB_GROUP override (lanes per walk); default: chosen per chunk at the launch below
    if (const int* gv = T.ptr(POA_TUNE_TB_GROUP)) { const int v = (*gv); if (v == 8 || v == 16 || v == 32 || v == 64) tb_group = v; }
    bool fuse_tb = false;  // measured slower (16.0 vs 13.4 ms): tracing waves hold slots without HBM traffic. trace each query in the epilogue of its forward wave (POA_FUSE_TB=0: separate launch)
    if (const int* fv = T.ptr(POA_TUNE_FUSE_TB)) fuse_tb = (*fv) != 0;
    int quads_override = 0;
    if (const int* ov = T.ptr(POA_TUNE_FWD_QUADS)) quads_override = (*ov);  // tuning override
    size_t ev = 1;
    for (const auto& ch : PL.chunks) {
        TbParams tp;
        tp.rows = b->d_rows.p; tp.pred_rows = b->d_pred_rows.p; tp.n_rows = fg.n;
        tp.start_row = fg.start_row; tp.end_row = fg.end_row;
        tp.first_query = ch.first; tp.n_queries = ch.count;
        tp.qseq = b->d_qseq.p; tp.qoff = b->d_qoff.p; tp.pitch = b->d_pitch.p; tp.plane_off = PL.d_off.p;
        tp.planes = b->d_planes.p; tp.scratch_off = b->d_scratch_off.p; tp.scratch = b->d_scratch.p;
        tp.score = b->d_score.p; tp.flags = b->d_flags.p; tp.n_pairs = b->d_npairs.p;
        tp.cost_x = costs->mismatch; tp.cost_o = costs->gap_open; tp.cost_e = costs->gap_extend;
        tp.spec_depth = spec_depth;
        tp.exact_pass = 0; tp.ex_status = nullptr; tp.ex_end = nullptr; tp.code_fmt = 0;
        tp.row_depth = relative ? b->d_row_depth.p : nullptr;
        tp.d_slot = fg.d_slot.empty() ? nullptr : b->d_dslot.p; tp.pred_dslot = fg.d_slot.empty() ? nullptr : b->d_pred_dslot.p;
        FwdParams fp;
        fp.rows = b->d_rows.p; fp.pred_rows = b->d_pred_rows.p; fp.n_rows = fg.n;
        fp.first_query = ch.first; fp.n_queries = ch.count;
        fp.qseq = b->d_qseq.p; fp.qoff = b->d_qoff.p; fp.pitch = b->d_pitch.p; fp.plane_off = PL.d_off.p;
        fp.planes = b->d_planes.p; fp.strip_carry = b->d_carry.p;
        fp.cost_x = costs->mismatch; fp.cost_oe = (uint32_t)costs->gap_open + costs->gap_extend; fp.cost_e = costs->gap_extend;
        // the same recurrences under the depth potential: deletions lose the e the potential already charges per row,
        // insertions pay it twice, every predecessor edge adds e * pred_k
        fp.cost_de = relative ? 0u : fp.cost_e;
        fp.cost_doe = relative ? (uint32_t)costs->gap_open : fp.cost_oe;
        fp.cost_ie = relative ? 2u * fp.cost_e : fp.cost_e;
        fp.cost_ioe = relative ? fp.cost_oe + fp.cost_e : fp.cost_oe;
        fp.pred_k = relative ? b->d_pred_k.p : nullptr;
        fp.d_slot = fg.d_slot.empty() ? nullptr : b->d_dslot.p; fp.pred_dslot = fg.d_slot.empty() ? nullptr : b->d_pred_dslot.p;
        fp.pipeline_error = b->d_pipeline_error.p;
        const uint32_t blocks = (ch.count + 3) / 4;
#define LAUNCH_FWD(QQ, TT)                                                                                              \
    do {                                                                                                               \
        if (fuse_tb) hipLaunchKernelGGL((poa_forward_kernel<QQ, TT, true, false>), dim3(blocks), dim3(256), 0, stream, fp, tp);  \
        else hipLaunchKernelGGL((poa_forward_kernel<QQ, TT, false, false>), dim3(blocks), dim3(256), 0, stream, fp, tp);         \
    } while (0)
        uint32_t max_pitch = 0;
        for (uint32_t i = ch.first; i < ch.first + ch.count; ++i) max_pitch = std::max(max_pitch, b->h_pitch[i]);
        // strip width: as narrow as the widest plane row of the chunk allows, at most 1024 columns
        if (narrow) {
            uint32_t quads = max_pitch <= 512 ? 1 : 2;  // 512 columns per quad (8 x u16 per lane)
            if (quads_override == 1 || quads_override == 2) quads = (uint32_t)quads_override;
            if (relative) quads = 2;  // the relative encoding lives in the pairs-across-quads kernels only
            if (compact && packed) {
                // queries longer than one strip: one workgroup per query, its strips pipelined over the waves (MW)
                bool mw = max_pitch > 512 * quads;
                if (const int* mv = T.ptr(POA_TUNE_MW)) mw = mw && ((*mv) != 0 || relative);
                bool px = max_pitch <= 1024 && quads == 2 && (!fuse_tb || relative);  // one strip of up to 1024 columns: the pairs-across-quads kernel
                if (const int* xv = T.ptr(POA_TUNE_PX)) px = (px && (*xv) != 0) || (px && relative);
                if (px) {
                    // scores below 0x3FFF (same bound as for u16, one power lower): two flags ride in the stored M value
                    // (below 0x0FFF: all four; POA_MF = 0 / 1 / 2 caps the variant for A-B runs)
                    int mf = relative ? 0 : (ub <= 4094 ? 2 : (ub <= 16382 ? 1 : 0));
                    if (const int* fv2 = T.ptr(POA_TUNE_MF)) mf = std::min(mf, std::max(0, (*fv2)));
                    tp.code_fmt = mf == 2 ? 3u : (mf == 1 ? 2u : 1u);
                    if (mf == 2) hipLaunchKernelGGL(poa_forward_px_kernel<2>, dim3(blocks), dim3(256), 0, stream, fp);
                    else if (mf == 1) hipLaunchKernelGGL(poa_forward_px_kernel<1>, dim3(blocks), dim3(256), 0, stream, fp);
                    else hipLaunchKernelGGL(poa_forward_px_kernel<0>, dim3(blocks), dim3(256), 0, stream, fp);
                } else if (mw && (relative || pxmw_ok(T, ch.count, max_pitch))) {
                    // pairs-across-quads mapping, 1024-column strips pipelined over the waves of a workgroup
                    tp.code_fmt = 1;
                    const uint32_t waves = mw_waves((max_pitch + 1023) / 1024);
                    hipLaunchKernelGGL(poa_forward_pxmw_kernel, dim3(ch.count), dim3(64 * waves), 0, stream, fp);
                } else if (mw) {
                    // narrow strips (more waves) until the chunk alone fills the chip
                    if (!quads_override) quads = ((uint64_t)ch.count * ((max_pitch + 1023) / 1024) >= 8192) ? 2 : 1;
                    const uint32_t strips = (max_pitch + 512 * quads - 1) / (512 * quads);
                    const uint32_t waves = mw_waves(strips);
                    if (quads == 1) hipLaunchKernelGGL((poa_forward_packed_kernel<1, false, true>), dim3(ch.count), dim3(64 * waves), 0, stream, fp, tp);
                    else hipLaunchKernelGGL((poa_forward_packed_kernel<2, false, true>), dim3(ch.count), dim3(64 * waves), 0, stream, fp, tp);
                } else if (quads == 1) {
                    if (fuse_tb) hipLaunchKernelGGL((poa_forward_packed_kernel<1, true, false>), dim3(blocks), dim3(256), 0, stream, fp, tp);
                    else hipLaunchKernelGGL((poa_forward_packed_kernel<1, false, false>), dim3(blocks), dim3(256), 0, stream, fp, tp);
                } else {
                    if (fuse_tb) hipLaunchKernelGGL((poa_forward_packed_kernel<2, true, false>), dim3(blocks), dim3(256), 0, stream, fp, tp);
                    else hipLaunchKernelGGL((poa_forward_packed_kernel<2, false, false>), dim3(blocks), dim3(256), 0, stream, fp, tp);
                }
            } else if (compact) {
                if (quads == 1) hipLaunchKernelGGL((poa_forward_kernel<1, uint16_t, false, true>), dim3(blocks), dim3(256), 0, stream, fp, tp);
                else hipLaunchKernelGGL((poa_forward_kernel<2, uint16_t, false, true>), dim3(blocks), dim3(256), 0, stream, fp, tp);
            } else if (quads == 1) LAUNCH_FWD(1, uint16_t);
            else LAUNCH_FWD(2, uint16_t);
        } else {
            uint32_t quads = max_pitch <= 256 ? 1 : (max_pitch <= 512 ? 2 : 4);  // 256 columns per quad (4 x u32 per lane)
            if (quads_override == 1 || quads_override == 2 || quads_override == 4) quads = (uint32_t)quads_override;
            bool mw = max_pitch > 1024;  // longer than the widest strip: pipeline the strips over the waves of a workgroup
            if (const int* mv = T.ptr(POA_TUNE_MW)) mw = mw && (*mv) != 0;
            if (mw) {
                // 512-column strips (up to 16 waves per query) or 1024-column ones (up to 10): a query whose strips do not all
                // fit one workgroup runs as several groups one after the other, and few long queries are latency bound per
                // row (a 1024-column row costs ~1.4x a 512-column one), so take the variant with the least groups x row cost;
                // the waves per workgroup are balanced over the groups
                const uint32_t s2 = (max_pitch + 511) / 512, s4 = (max_pitch + 1023) / 1024;
                const uint32_t g2 = (s2 + MW_MAX_WAVES - 1) / MW_MAX_WAVES, g4 = (s4 + 9) / 10;
                bool wide = 14 * g4 < 10 * g2 || (uint64_t)ch.count * s4 >= 8192;
                if (quads_override == 4) wide = true;
                if (quads_override == 2) wide = false;
                if (wide) {
                    const uint32_t waves = (s4 + g4 - 1) / g4;
                    hipLaunchKernelGGL((poa_forward_kernel<4, uint32_t, false, false, true>), dim3(ch.count), dim3(64 * waves), 0, stream, fp, tp);
                } else {
                    const uint32_t waves = (s2 + g2 - 1) / g2;
                    hipLaunchKernelGGL((poa_forward_kernel<2, uint32_t, false, false, true>), dim3(ch.count), dim3(64 * waves), 0, stream, fp, tp);
                }
            } else if (quads == 1) LAUNCH_FWD(1, uint32_t);
            else if (quads == 2) LAUNCH_FWD(2, uint32_t);
            else LAUNCH_FWD(4, uint32_t);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(events[ev++], stream));

        if (!(fuse_tb && !relative && max_pitch <= 1024 && (!compact || packed))) {
            // Lanes per walk: as many as keep the launch within ~6 000 waves (what the chip holds at this kernel's occupancy
            // with room to spare), and a speculation depth to match.  Measured on config 2 after the insertion-run speculation
            // (ms per batch; 64 lanes x depth 32 / 32 x 32 / 16 x 16): 1 024 walks 0.44 / 0.57 / 0.74, 4 096 walks 0.68 / 0.77 /
            // 0.84, 10 000 walks 1.22 / 0.97 / 1.05, 20 000 walks 2.22 / 1.79 / 1.35.
            int tbg = ch.count <= 6144 ? 64 : (ch.count <= 12288 ? 32 : 16);
            if (T.ptr(POA_TUNE_TB_GROUP)) tbg = tb_group;
            TbParams tpg = tp;
            if (!T.ptr(POA_TUNE_TB_DEPTH)) tpg.spec_depth = tbg == 16 ? 16u : (compact ? 32u : spec_depth);
            if (compact && tbg == 16) {
                hipLaunchKernelGGL((poa_traceback_kernel<uint16_t, true, 16>), dim3((ch.count + 15) / 16), dim3(256), 0, stream, tpg);
            } else if (compact && tbg == 32) {
                hipLaunchKernelGGL((poa_traceback_kernel<uint16_t, true, 32>), dim3((ch.count + 7) / 8), dim3(256), 0, stream, tpg);
            } else if (compact && tbg == 8) {
                hipLaunchKernelGGL((poa_traceback_kernel<uint16_t, true, 8>), dim3((ch.count + 31) / 32), dim3(256), 0, stream, tpg);
            }
            else if (compact) hipLaunchKernelGGL((poa_traceback_kernel<uint16_t, true>), dim3((ch.count + 3) / 4), dim3(256), 0, stream, tpg);
            else if (narrow) hipLaunchKernelGGL((poa_traceback_kernel<uint16_t, false>), dim3((ch.count + 3) / 4), dim3(256), 0, stream, tp);
            else hipLaunchKernelGGL((poa_traceback_kernel<uint32_t, false>), dim3((ch.count + 3) / 4), dim3(256), 0, stream, tp);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipEventRecord(events[ev++], stream));

        if (mode != POA_MODE_DENSE) {
            // exact replay of the reference's search on the (re-initialised, u32) planes of this chunk
            const uint32_t hybrid = mode == POA_MODE_HYBRID ? 1u : 0u;
            HIP_TRY(hipMemsetAsync(b->d_ex_status.p + ch.first, 0xFF, (size_t)ch.count * 4, stream));
            HIP_TRY(hipMemsetAsync(b->d_ex_counters.p + 4 * (size_t)ch.first, 0, (size_t)ch.count * 16, stream));
            hipLaunchKernelGGL(poa_fill_planes_kernel, dim3(64, ch.count), dim3(256), 0, stream, b->d_planes.p, PL.d_off.p,
                               b->d_pitch.p, fg.n, ch.first, hybrid, b->d_flags.p);
            HIP_TRY(hipGetLastError());
            if (fg.n_exit) {
                HIP_TRY(hipMemsetAsync(b->d_ex_reached.p, 0, (size_t)ch.count * fg.n_exit * b->ex_wpn * 8, stream));
                HIP_TRY(hipMemsetAsync(b->d_ex_rsum.p, 0, (size_t)ch.count * fg.n_exit * b->ex_swpn * 8, stream));
            }
            HIP_TRY(hipMemsetAsync(b->d_ex_head.p, 0xFF, (size_t)ch.count * 3 * b->ex_n_prio * 4, stream));
            ExactParams ep;
            ep.G = ExactGraph{fg.n, fg.start_row, fg.end_row, b->d_ex_sym.p, b->d_succ_off.p, b->d_succ_rows.p, b->d_dist_min.p,
                              b->d_dist_max.p, b->d_exit_idx.p, fg.n_exit, b->d_nbm_off.p, b->d_nbm.p, b->d_node_row.p, b->d_sp_to_end.p,
                              fg.row_rec.empty() ? nullptr : b->d_ex_rec.p};
            ep.first_query = ch.first; ep.n_queries = ch.count; ep.hybrid = hybrid; ep.dense_flags = b->d_flags.p;
            ep.qseq = b->d_qseq.p; ep.qoff = b->d_qoff.p; ep.pitch = b->d_pitch.p; ep.plane_off = PL.d_off.p;
            ep.planes = b->d_planes.p;
            ep.reached = b->d_ex_reached.p; ep.rsum = b->d_ex_rsum.p; ep.wpn = b->ex_wpn; ep.swpn = b->ex_swpn;
            ep.head = b->d_ex_head.p; ep.n_prio = b->ex_n_prio; ep.pool = b->d_ex_pool.p; ep.pool_cap = b->ex_pool_cap;
            ep.stack = b->d_ex_stack.p; ep.stack_cap = b->ex_stack_cap;
            ep.C = ExactCosts{costs->mismatch, costs->gap_open, costs->gap_extend, cfg ? cfg->heuristic : POA_HEURISTIC_MINGAP,
                              cfg ? cfg->pruning : 1u, 0, 0, 0, 0, 0, 0};
            if (ends_free) {
                ep.C.ends_free = 1;
                ep.C.qfe_kind = cfg->qry_free_end.kind; ep.C.qfe_val = cfg->qry_free_end.value;
                ep.C.gfb_kind = cfg->graph_free_begin.kind;
                ep.C.gfe_kind = cfg->graph_free_end.kind; ep.C.gfe_val = cfg->graph_free_end.value;
            }
            ep.status = b->d_ex_status.p;
            ep.end_cell = b->d_ex_end.p;
            ep.n_succ = (uint32_t)fg.succ_rows.size(); ep.n_nbm = (uint32_t)fg.nbm.size();
            // Wave-per-query search (poa_wsearch.hpp) unless its descriptor ring cannot hold the priorities that are live
            // at once: one pop pushes at most max(x, o+e) above its own priority plus what the heuristic can jump along
            // one edge and a state change (o), and the greedy extension walks that jump once more.
            // live priority range: a pop at priority f pushes at most max(x, o+e) + o + e * (spread + 3) above f, where
            // spread = max over nodes of dist_max - dist_min (the heuristic's gap length grows by at most that along any
            // greedy extension: heuristic.rs:70-102)
            const uint32_t win = b->ex_win;
            int lds_cap = 64 * 1024;
            (void)hipDeviceGetAttribute(&lds_cap, hipDeviceAttributeMaxSharedMemoryPerBlock, b->device);
            const uint32_t graph_lds = exact_lds_bytes(fg.n, ep.n_succ, ep.n_nbm);
            const int* impl = T.ptr(POA_TUNE_EXACT_IMPL);   // 1 lane, 2 wave, 3 flat
            const bool wave_search = !(impl && *impl == 1) && win <= b->ex_n_prio;
            // "flat": the parallel-step kernel (poa_fsearch.hpp) — bit-identical like the others, not the fastest yet
            const bool par_search = wave_search && impl && *impl == 3;
            if (par_search) {
                // the next entries in pop order expanded at once, one per lane, several queries per wave (poa_fsearch.hpp)
                FSearchParams pp;
                pp.E = ep;
                pp.chunks = reinterpret_cast<ExU4*>(b->d_ex_pool.p);
                pp.chunk_cap = b->ex_pool_cap / BQ_CHUNK;
                if (const int* cv = T.ptr(POA_TUNE_WS_CHUNK_CAP)) { const int v = (*cv); if (v >= 1 && (uint32_t)v < pp.chunk_cap) pp.chunk_cap = (uint32_t)v; }
                pp.win = win;
                pp.counters = b->d_ex_counters.p;
                pp.max_lanes = 63;   // (capped at the group's lanes below)
                if (const int* lv = T.ptr(POA_TUNE_PS_LANES)) { const int v = (*lv); if (v >= 1 && v <= 63) pp.max_lanes = (uint32_t)v; }
                pp.prof = nullptr;
                if (T.ptr(POA_TUNE_WS_PROF)) pp.prof = b->d_ex_prof.p;
                // Lanes per query: a step commits a dozen lanes on the benchmark's reads, and the kernel waits for memory more than it
                // issues — so four searches share a wave (16 lanes each), all stepping through one instruction stream.
                pp.lean = (!fg.row_rec.empty() && !ends_free) ? 1u : 0u;
                if (const int* lv = T.ptr(POA_TUNE_PS_LEAN)) pp.lean = (pp.lean && (*lv) != 0) ? 1u : 0u;
                const uint32_t group = pp.lean ? 8u : 16u;   // (the group sizes poa_fsearch.hpp is built for)
                const uint32_t qpw = 64 / group;
                pp.group = group;
                if (pp.max_lanes > group) pp.max_lanes = group;
                // LDS of a block: the staged graph (shared by its waves) + per wave the descriptor rings of its queries and the
                // logs / read sets / conflict tables of the step.  As many waves per block as fit 160 KB, at most 8 (two per SIMD:
                // the kernel keeps a lane's search state in ~250 registers).
                const uint64_t lds_budget = std::min<uint64_t>((uint64_t)lds_cap, 160u * 1024u);
                const uint64_t ring_b = ((uint64_t)qpw * 3 * win * 4 + 15) & ~15ull;
                // the lean step reads one 32-byte record per row: those are staged (the arrays of the generic code, which it
                // falls back to for a row in a thousand, stay in global memory); without records the generic code runs in log mode
                const uint64_t rec_b = pp.lean ? ((sizeof(FlatGraph::RowRec) * (uint64_t)fg.n + 15) & ~15ull) : 0;
                const uint64_t stage_b = pp.lean ? rec_b : graph_lds;
                bool ring_lds = ring_b + ps_lds_bytes() <= lds_budget && !T.ptr(POA_TUNE_WS_RING_GLOBAL);
                uint64_t per_wave = (ring_lds ? ring_b : 0) + ps_lds_bytes();
                bool stage = stage_b + per_wave <= lds_budget;
                if (const int* gv = T.ptr(POA_TUNE_EXACT_LDS)) stage = stage && (*gv) != 0;
                uint32_t wpb = (uint32_t)std::min<uint64_t>(8, (lds_budget - (stage ? stage_b : 0)) / per_wave);
                if (const int* wv = T.ptr(POA_TUNE_WS_WAVES)) { const int v = (*wv); if (v >= 1 && (uint32_t)v <= wpb) wpb = (uint32_t)v; }
                if (wpb < 1) return fail(POA_ERR_UNSUPPORTED, "exact replay: the step's logs do not fit the LDS");
                pp.graph_lds = (stage && !pp.lean) ? graph_lds : 0;
                pp.rec_lds = (stage && pp.lean) ? (uint32_t)rec_b : 0;
                pp.waves_per_block = wpb;
                pp.ring_global = ring_lds ? nullptr : b->d_ex_head.p;   // [slots * 3 * ex_n_prio] holds slots * 3 * win
                const uint32_t lds_bytes = pp.graph_lds + pp.rec_lds + (uint32_t)(wpb * per_wave);
                const void* kfn = pp.lean ? reinterpret_cast<const void*>(poa_fsearch_lean_kernel) : reinterpret_cast<const void*>(poa_fsearch_kernel);
                if (lds_bytes > 48u * 1024u) HIP_TRY(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
                const uint32_t per_block = wpb * qpw;
                uint32_t n_blocks = (ch.count + per_block - 1) / per_block;
                pp.work_counter = nullptr; pp.order = nullptr;
                int per_cu = 1, cus = 256;
                (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, b->device);
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, (int)(64 * wpb), lds_bytes) != hipSuccess || per_cu < 1) per_cu = 1;
                const uint32_t resident = (uint32_t)cus * (uint32_t)per_cu;
                if (n_blocks > resident && !T.ptr(POA_TUNE_WS_STATIC)) {
                    // persistent waves, longest expected search first (see the wave search below)
                    std::vector<uint32_t> sc(ch.count), ord(ch.count);
                    HIP_TRY(hipMemcpyAsync(sc.data(), b->d_score.p + ch.first, (size_t)ch.count * 4, hipMemcpyDeviceToHost, stream));
                    HIP_TRY(hipStreamSynchronize(stream));
                    for (uint32_t i = 0; i < ch.count; ++i) ord[i] = i;
                    std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t c2) { return sc[a] > sc[c2]; });
                    HIP_TRY(b->d_ex_order.alloc(ch.count + 1));
                    HIP_TRY(hipMemcpyAsync(b->d_ex_order.p + 1, ord.data(), (size_t)ch.count * 4, hipMemcpyHostToDevice, stream));
                    HIP_TRY(hipMemsetAsync(b->d_ex_order.p, 0, 4, stream));
                    HIP_TRY(hipStreamSynchronize(stream));   // `ord` is a host temporary
                    pp.work_counter = b->d_ex_order.p; pp.order = b->d_ex_order.p + 1;
                    n_blocks = resident;
                }
                HIP_TRY(b->d_ex_logs.alloc((size_t)n_blocks * wpb * ps_scratch_words()));
                pp.scratch = b->d_ex_logs.p;
                if (pp.lean) hipLaunchKernelGGL(poa_fsearch_lean_kernel, dim3(n_blocks), dim3(64 * wpb), lds_bytes, stream, pp);
                else hipLaunchKernelGGL(poa_fsearch_kernel, dim3(n_blocks), dim3(64 * wpb), lds_bytes, stream, pp);
            } else if (wave_search) {
                WSearchParams wp;
                wp.E = ep;
                wp.chunks = reinterpret_cast<ExU4*>(b->d_ex_pool.p);
                wp.chunk_cap = b->ex_pool_cap / BQ_CHUNK;
                if (const int* cv = T.ptr(POA_TUNE_WS_CHUNK_CAP)) { const int v = (*cv); if (v >= 1 && (uint32_t)v < wp.chunk_cap) wp.chunk_cap = (uint32_t)v; }
                wp.win = win;
                wp.counters = b->d_ex_counters.p;
                wp.max_lanes = 32;   // entries tested per step at most: runs of stale / pruned entries are short (1.85 pops per step)
                if (const int* lv = T.ptr(POA_TUNE_WS_LANES)) { const int v = (*lv); if (v >= 1 && v <= 63) wp.max_lanes = (uint32_t)v; }
                wp.adapt_lanes = 4;   // after an expansion the top of the stack is fresh: few entries tested; a run that used up its lanes widens
                if (const int* av = T.ptr(POA_TUNE_WS_ADAPT)) { const int v = (*av); if (v >= 0 && v <= 63) wp.adapt_lanes = (uint32_t)v; }
                wp.prof = nullptr;
                if (T.ptr(POA_TUNE_WS_PROF)) wp.prof = b->d_ex_prof.p;  // per-phase cycle counts of the wave search (diagnostics)
                // Lanes per query.  One query per wave (64) with persistent scheduling is the default.  Several queries per wave
                // (POA_WS_GROUP = 32 / 16 / 8 lanes each, all through one instruction stream) issue fewer instructions in
                // all but lose: the iteration takes the union of the groups' code paths and every wave waits for its slowest
                // query (measured on config 2, 10 000 queries: 1.09 s at 64 persistent, 1.83 s at 64 static, 1.49 / 1.85 /
                // 2.53 s at 32 / 16 / 8).  Waves per block: as many as share one staged copy of the graph, at most 16.
                uint32_t group = 64;
                if (const int* gv = T.ptr(POA_TUNE_WS_GROUP)) { const int v = (*gv); if (v == 8 || v == 16 || v == 32 || v == 64) group = (uint32_t)v; }
                uint32_t wpb = 16;
                if (const int* wv = T.ptr(POA_TUNE_WS_WAVES)) { const int v = (*wv); if (v >= 1 && v <= 16) wpb = (uint32_t)v; }
                const uint64_t lds_budget = std::min<uint64_t>((uint64_t)lds_cap, 80u * 1024u);
                bool ring_lds = false, stage = false;
                // per-row records of the one-round-trip path (one query per wave: a block of 16 waves has the CU to itself)
                uint32_t rec_lds = 0;
                for (;;) {
                    const uint64_t rb = (uint64_t)wpb * (64 / group) * win * 12;
                    ring_lds = rb <= lds_budget && !T.ptr(POA_TUNE_WS_RING_GLOBAL);
                    stage = graph_lds + (ring_lds ? rb : 0) <= lds_budget;
                    if (const int* gv = T.ptr(POA_TUNE_EXACT_LDS)) stage = stage && (*gv) != 0;
                    if (group == 64 || (ring_lds && stage)) break;
                    // several queries per wave need everything in LDS: fewer waves per block first, then fewer queries per wave
                    if (wpb > 2 && !T.ptr(POA_TUNE_WS_WAVES)) wpb /= 2; else { group *= 2; if (!T.ptr(POA_TUNE_WS_WAVES)) wpb = 16; }
                }
                const uint64_t ring_bytes = ring_lds ? (uint64_t)wpb * (64 / group) * win * 12 : 0;
                if (group == 64 && stage && ring_lds && !fg.row_rec.empty() && !(T.ptr(POA_TUNE_WS_REC) && *T.ptr(POA_TUNE_WS_REC) == 0)) {
                    const uint64_t rb = (fg.row_rec.size() * sizeof(FlatGraph::RowRec) + 15) & ~15ull;
                    if (graph_lds + rb + ring_bytes <= std::min<uint64_t>((uint64_t)lds_cap, 150u * 1024u)) rec_lds = (uint32_t)rb;
                }
                wp.rec_lds = rec_lds;
                wp.graph_lds = stage ? graph_lds : 0;
                wp.waves_per_block = wpb;
                wp.group = group;
                wp.ring_global = nullptr;
                if (!ring_lds) wp.ring_global = b->d_ex_head.p;  // [slots * 3 * ex_n_prio] holds slots * 3 * win
                const uint32_t lds_bytes = wp.graph_lds + wp.rec_lds + (uint32_t)ring_bytes;
                const bool lds_all = wp.graph_lds && !wp.ring_global;   // (else the kernel whose search reads through generic pointers)
                const void* kfn = group != 64 ? reinterpret_cast<const void*>(poa_wsearch_groups_kernel)
                                              : (lds_all ? reinterpret_cast<const void*>(poa_wsearch_kernel) : reinterpret_cast<const void*>(poa_wsearch_global_kernel));
                if (lds_bytes > 48u * 1024u)
                    HIP_TRY(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
                const uint32_t per_block = wpb * (64 / group);
                uint32_t n_blocks = (ch.count + per_block - 1) / per_block;
                wp.work_counter = nullptr; wp.order = nullptr;
                if (group == 64 && !T.ptr(POA_TUNE_WS_STATIC)) {
                    // persistent waves: as many blocks as are resident at once; queries handed out longest-expected-search
                    // first (by the dense pass's score: more edits, more buckets).  The sort needs the scores on the host:
                    // one small copy behind the dense pass of this chunk.
                    int per_cu = 1, cus = 256;
                    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, b->device);
                    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, (int)(64 * wpb), lds_bytes) != hipSuccess || per_cu < 1) per_cu = 1;
                    const uint32_t resident = (uint32_t)cus * (uint32_t)per_cu;
                    if (n_blocks > resident) {
                        std::vector<uint32_t> sc(ch.count), ord(ch.count);
                        HIP_TRY(hipMemcpyAsync(sc.data(), b->d_score.p + ch.first, (size_t)ch.count * 4, hipMemcpyDeviceToHost, stream));
                        HIP_TRY(hipStreamSynchronize(stream));
                        for (uint32_t i = 0; i < ch.count; ++i) ord[i] = i;
                        std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t c2) { return sc[a] > sc[c2]; });
                        HIP_TRY(b->d_ex_order.alloc(ch.count + 1));
                        HIP_TRY(hipMemcpyAsync(b->d_ex_order.p + 1, ord.data(), (size_t)ch.count * 4, hipMemcpyHostToDevice, stream));
                        HIP_TRY(hipMemsetAsync(b->d_ex_order.p, 0, 4, stream));
                        HIP_TRY(hipStreamSynchronize(stream));   // `ord` is a host temporary
                        wp.work_counter = b->d_ex_order.p; wp.order = b->d_ex_order.p + 1;
                        n_blocks = resident;
                    }
                }
                if (group != 64) hipLaunchKernelGGL(poa_wsearch_groups_kernel, dim3(n_blocks), dim3(64 * wpb), lds_bytes, stream, wp);
                else if (lds_all) hipLaunchKernelGGL(poa_wsearch_kernel, dim3(n_blocks), dim3(64 * wpb), lds_bytes, stream, wp);
                else hipLaunchKernelGGL(poa_wsearch_global_kernel, dim3(n_blocks), dim3(64 * wpb), lds_bytes, stream, wp);
            } else {
            // active lanes per wave: one sequential search per lane.  Few lanes = little divergence but many
            // waves; enough waves to fill the chip (~16 per CU) first, then more lanes per wave.
            uint32_t lanes = (ch.count + 4095) / 4096;
            if (lanes > 64) lanes = 64;
            if (const int* lv = T.ptr(POA_TUNE_EXACT_LANES)) { const int v = (*lv); if (v >= 1 && v <= 64) lanes = (uint32_t)v; }
            ep.lanes_per_wave = lanes;
            // graph arrays in LDS when they fit beside three other blocks of the same CU (160 KB per CU)
            uint32_t lds_bytes = exact_lds_bytes(fg.n, ep.n_succ, ep.n_nbm);
            bool lds_graph = lds_bytes <= 160u * 1024u / 3u;
            if (const int* gv = T.ptr(POA_TUNE_EXACT_LDS)) lds_graph = lds_graph && (*gv) != 0;
            if (lds_graph && lds_bytes > 48u * 1024u)
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(poa_exact_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            ep.lds_graph = lds_graph ? 1u : 0u;
            if (lds_graph && !T.ptr(POA_TUNE_EXACT_LANES)) {
                // all blocks resident at once (no tail round): LDS bounds the blocks per CU
                int cus = 256;
                (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, b->device);
                const uint32_t per_cu = std::max(1u, std::min(8u, 160u * 1024u / std::max(lds_bytes, 1u)));
                const uint32_t cap_waves = (uint32_t)cus * per_cu * (EXACT_BLOCK / 64);
                lanes = std::min(64u, std::max(lanes, (ch.count + cap_waves - 1) / cap_waves));
                ep.lanes_per_wave = lanes;
            }
            const uint32_t per_block = lanes * (EXACT_BLOCK / 64);
            hipLaunchKernelGGL(poa_exact_kernel, dim3((ch.count + per_block - 1) / per_block), dim3(EXACT_BLOCK), lds_graph ? lds_bytes : 0, stream, ep);
            }
            HIP_TRY(hipGetLastError());
            tp.exact_pass = 1; tp.ex_status = b->d_ex_status.p; tp.ex_end = b->d_ex_end.p; tp.row_depth = nullptr;
            hipLaunchKernelGGL((poa_traceback_kernel<uint32_t, false>), dim3((ch.count + 3) / 4), dim3(256), 0, stream, tp);
            HIP_TRY(hipGetLastError());
            b->narrow = false; b->compact = false; b->relative = false;  // the planes now hold the replayed u32 table
        }
        HIP_TRY(hipEventRecord(events[ev++], stream));
    }
    hipLaunchKernelGGL(poa_scan_kernel, dim3(1), dim3(1024), 0, stream, b->d_npairs.p, b->d_pair_off.p, b->n_queries);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(poa_compact_kernel, dim3((b->n_queries + 3) / 4), dim3(256), 0, stream, b->d_scratch.p,
                       b->d_scratch_off.p, b->d_npairs.p, b->d_pair_off.p, b->d_pairs.p, b->n_queries);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(events[ev], stream));
    b->ran = true;
    return POA_OK;
}

int poa_batch_fetch(poa_batch_t* b, uint32_t* score, poa_aln_pair_t* pairs, uint64_t* pair_off, uint64_t pair_capacity,
                    uint32_t* flags, poa_stats_t* stats) {
    if (!b) return fail(POA_ERR_INVALID_ARG, "poa_batch_fetch: null batch");
    if (!b->ran) return fail(POA_ERR_INVALID_ARG, "poa_batch_fetch: poa_batch_run has not been called");
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipStreamSynchronize(b->last_stream));
    {
        int prc = check_pipeline_error(b);
        if (prc != POA_OK) return prc;
    }
    const uint32_t n = b->n_queries;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, nullptr));
    std::vector<uint64_t> off_local;
    uint64_t* off = pair_off;
    if (!off) { off_local.resize((size_t)n + 1); off = off_local.data(); }
    HIP_TRY(hipMemcpy(off, b->d_pair_off.p, ((size_t)n + 1) * 8, hipMemcpyDeviceToHost));
    int rc = POA_OK;
    uint32_t n_exact = 0;
    if (n) {
        if (score) HIP_TRY(hipMemcpy(score, b->d_score.p, (size_t)n * 4, hipMemcpyDeviceToHost));
        std::vector<uint32_t> fl_local;
        uint32_t* fl = flags;
        if (!fl && stats) { fl_local.resize(n); fl = fl_local.data(); }
        if (fl) HIP_TRY(hipMemcpy(fl, b->d_flags.p, (size_t)n * 4, hipMemcpyDeviceToHost));
        if (pairs) {
            if (off[n] > pair_capacity) rc = fail(POA_ERR_CAPACITY, "pair_capacity too small; pair_off[n] holds the needed total");
            else if (off[n]) HIP_TRY(hipMemcpy(pairs, b->d_pairs.p, off[n] * sizeof(poa_aln_pair_t), hipMemcpyDeviceToHost));
        }
        if (stats && fl) {
            uint32_t nf = 0;
            for (uint32_t i = 0; i < n; ++i) nf += fl[i] != 0;
            stats->n_flagged = nf;
        }
        if (stats && b->last_mode != POA_MODE_DENSE) {
            std::vector<uint32_t> stt(n);
            HIP_TRY(hipMemcpy(stt.data(), b->d_ex_status.p, (size_t)n * 4, hipMemcpyDeviceToHost));
            uint32_t ne = 0;
            for (uint32_t i = 0; i < n; ++i) ne += stt[i] == 0;
            n_exact = ne;
        }
    } else if (stats) {
        stats->n_flagged = 0;
    }
    HIP_TRY(hipEventRecord(e1, nullptr));
    HIP_TRY(hipEventSynchronize(e1));
    float ms_d2h = 0.f;
    (void)hipEventElapsedTime(&ms_d2h, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (stats) {
        collect_stats(b, stats);
        stats->ms_d2h = ms_d2h;
        stats->n_exact = n_exact;
    }
    return rc;
}

int poa_batch_stats(poa_batch_t* b, poa_stats_t* stats) {
    if (!b || !stats) return fail(POA_ERR_INVALID_ARG, "poa_batch_stats: null argument");
    HIP_TRY(hipSetDevice(b->device));
    if (b->ran) HIP_TRY(hipStreamSynchronize(b->last_stream));
    std::memset(stats, 0, sizeof(*stats));
    collect_stats(b, stats);
    return b->ran ? check_pipeline_error(b) : POA_OK;
}

int poa_batch_fetch_search_counters(poa_batch_t* b, uint32_t* out) {
    if (!b || !out) return fail(POA_ERR_INVALID_ARG, "poa_batch_fetch_search_counters: null argument");
    if (!b->ran || b->last_mode == POA_MODE_DENSE || !b->d_ex_counters.p)
        return fail(POA_ERR_INVALID_ARG, "poa_batch_fetch_search_counters: the last run was not an exact / hybrid run");
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipStreamSynchronize(b->last_stream));
    if (b->n_queries) HIP_TRY(hipMemcpy(out, b->d_ex_counters.p, (size_t)b->n_queries * 16, hipMemcpyDeviceToHost));
    if (b->prof_on && b->d_ex_prof.p && b->n_queries) {
        std::vector<unsigned long long> pr(8 * (size_t)b->n_queries);
        HIP_TRY(hipMemcpy(pr.data(), b->d_ex_prof.p, pr.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long sum[8] = {0};
        for (size_t i = 0; i < pr.size(); ++i) sum[i & 7] += pr[i];
        // wave search: queue+entries, parallel test, drop, fast expand, generic, (counts: fast, generic); parallel-step search: queue+entries,
        // log-mode phase, conflict test, commit, sequential code, pushes, (counts: sequential steps, lanes committed)
        fprintf(stderr, "[ws prof] cycles / counts: %llu %llu %llu %llu %llu %llu %llu %llu\n", sum[0], sum[1], sum[2], sum[3], sum[4], sum[5], sum[6], sum[7]);
    }
    return POA_OK;
}

int poa_batch_device_results(poa_batch_t* b, void** score, void** flags, void** pair_off, void** pairs) {
    if (!b) return fail(POA_ERR_INVALID_ARG, "poa_batch_device_results: null batch");
    // the results of the last run: wait for it and report a multi-wave pipeline that gave up (as poa_batch_fetch / _stats do), so
    // that a caller gathering straight from HBM (poasta_amd/dist.py) never ships the output of a failed run
    if (b->ran) {
        HIP_TRY(hipSetDevice(b->device));
        HIP_TRY(hipStreamSynchronize(b->last_stream));
        const int prc = check_pipeline_error(b);
        if (prc != POA_OK) return prc;
    }
    if (score) *score = b->d_score.p;
    if (flags) *flags = b->d_flags.p;
    if (pair_off) *pair_off = b->d_pair_off.p;
    if (pairs) *pairs = b->d_pairs.p;
    return POA_OK;
}

int poa_batch_last_layout(poa_batch_t* b, uint32_t* layout) {
    if (!b || !layout) return fail(POA_ERR_INVALID_ARG, "poa_batch_last_layout: null argument");
    if (!b->ran) return fail(POA_ERR_INVALID_ARG, "poa_batch_last_layout: poa_batch_run has not been called");
    *layout = (b->dense_narrow ? POA_LAYOUT_U16 : 0u) | (b->dense_compact ? POA_LAYOUT_COMPACT : 0u) | (b->dense_relative ? POA_LAYOUT_RELATIVE : 0u);
    return POA_OK;
}

int poa_batch_fetch_planes(poa_batch_t* b, uint32_t query, uint32_t* m, uint32_t* i, uint32_t* d) {
    if (!b || !m || !i || !d) return fail(POA_ERR_INVALID_ARG, "poa_batch_fetch_planes: null argument");
    if (!b->ran || query >= b->n_queries) return fail(POA_ERR_INVALID_ARG, "poa_batch_fetch_planes: bad query / not run");
    if (b->compact) return fail(POA_ERR_UNSUPPORTED, "poa_batch_fetch_planes: the last run used the compact layout; run with POA_CFG_FULL_PLANES");
    if (b->last_mode != POA_MODE_DENSE) return fail(POA_ERR_UNSUPPORTED, "poa_batch_fetch_planes: after an exact / hybrid run the workspace holds the replayed search's tiled table");
    const auto& last = b->cur().chunks.back();
    if (query < last.first || query >= last.first + last.count)
        return fail(POA_ERR_INVALID_ARG, "poa_batch_fetch_planes: the query's planes were overwritten by a later chunk");
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipStreamSynchronize(b->last_stream));
    const uint32_t rows = b->graph->g.n, pitch = b->h_pitch[query];
    const uint32_t cols = (uint32_t)(b->h_qoff[query + 1] - b->h_qoff[query]) + 1;
    const uint64_t RP = (uint64_t)rows * pitch;
    uint32_t* dst[3] = {m, i, d};
    if (!b->narrow) {
        for (int k = 0; k < 3; ++k) {
            const uint32_t* src = b->d_planes.p + b->cur().off[query] + k * RP;
            HIP_TRY(hipMemcpy2D(dst[k], (size_t)cols * 4, src, (size_t)pitch * 4, (size_t)cols * 4, rows, hipMemcpyDeviceToHost));
        }
    } else {
        std::vector<uint16_t> tmp((size_t)rows * cols);
        const uint16_t* base = reinterpret_cast<const uint16_t*>(b->d_planes.p) + b->cur().off[query];
        for (int k = 0; k < 3; ++k) {
            HIP_TRY(hipMemcpy2D(tmp.data(), (size_t)cols * 2, base + k * RP, (size_t)pitch * 2, (size_t)cols * 2, rows, hipMemcpyDeviceToHost));
            for (size_t t = 0; t < tmp.size(); ++t) dst[k][t] = tmp[t] == 0xFFFFu ? 0xFFFFFFFFu : tmp[t];
        }
    }
    return POA_OK;
}

void poa_batch_destroy(poa_batch_t* b) {
    if (!b) return;
    (void)hipSetDevice(b->device);
    if (b->ran) (void)hipStreamSynchronize(b->last_stream);  // the plane workspace may be handed to another batch next
    delete b;
}

int poa_align_batch(const poa_graph_t* g, const poa_costs_t* costs, uint32_t n_queries, const uint8_t* qseq,
                    const uint64_t* qoff, uint32_t* score, poa_aln_pair_t* pairs, uint64_t* pair_off,
                    uint64_t pair_capacity, uint32_t* flags, poa_stats_t* stats, int device) {
    return poa_align_batch_ex(g, costs, nullptr, n_queries, qseq, qoff, score, pairs, pair_off, pair_capacity, flags, stats, device);
}

int poa_align_batch_ex(const poa_graph_t* g, const poa_costs_t* costs, const poa_config_t* cfg, uint32_t n_queries,
                       const uint8_t* qseq, const uint64_t* qoff, uint32_t* score, poa_aln_pair_t* pairs, uint64_t* pair_off,
                       uint64_t pair_capacity, uint32_t* flags, poa_stats_t* stats, int device) {
    if (!g || !costs || !qoff) return fail(POA_ERR_INVALID_ARG, "poa_align_batch: null argument");
    const TuneView T(cfg);
    if (stats) std::memset(stats, 0, sizeof(*stats));
    // PoastaAligner::align, empty-graph shortcut (src/aligner/mod.rs:124-142): score 4*len, no pairs
    if (g->g.n_real == 0) {
        for (uint32_t i = 0; i < n_queries; ++i) {
            const uint64_t L = qoff[i + 1] - qoff[i];
            if (score) score[i] = (uint32_t)(L * 4);
            if (flags) flags[i] = POA_FLAG_EMPTY_GRAPH;
            if (pair_off) pair_off[i] = 0;
        }
        if (pair_off) pair_off[n_queries] = 0;
        if (stats) { stats->n_queries = n_queries; stats->n_flagged = n_queries; }
        return POA_OK;
    }
    // size the workspace for the layout this run will use (2-byte elements when the dense pass can run in u16)
    uint64_t ws_hint = 0;
    {
        uint64_t max_len = 0, elems = 0;
        for (uint32_t i = 0; i < n_queries; ++i) {
            const uint64_t L = qoff[i + 1] - qoff[i];
            max_len = std::max(max_len, L);
            elems += 3ull * g->g.n * (((L + 1 + 63) / 64) * 64);
        }
        const uint64_t ub = (max_len ? (uint64_t)costs->gap_open + (uint64_t)costs->gap_extend * max_len : 0) +
                            (g->g.min_path_nodes ? (uint64_t)costs->gap_open + (uint64_t)costs->gap_extend * g->g.min_path_nodes : 0);
        const bool dense = !cfg || (cfg->mode == POA_MODE_DENSE && cfg->span == POA_SPAN_GLOBAL);
        if (dense && ub <= 65534 && !T.ptr(POA_TUNE_PLANES)) {
            size_t free_b = 0, total_b = 0;
            if (hipSetDevice(device) == hipSuccess && hipMemGetInfo(&free_b, &total_b) == hipSuccess && elems * 2 < free_b / 2)
                ws_hint = elems * 2;
        }
    }
    const bool timing = T.ptr(POA_TUNE_TIMING);
    auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    poa_batch_t* b = nullptr;
    int rc = poa_batch_create(g, device, n_queries, qseq, qoff, ws_hint, &b);
    if (rc != POA_OK) return rc;
    const double t1 = now();
    rc = poa_batch_run_ex(b, costs, cfg, nullptr);
    const double t2 = now();
    if (rc == POA_OK) rc = poa_batch_fetch(b, score, pairs, pair_off, pair_capacity, flags, stats);
    // replay out of queue space (POA_FLAG_EXACT_OVERFLOW keeps the dense result): before settling for that, run again with
    // a queue pool 8x, then 64x the size (unless the caller fixed a tiny budget on purpose: queue_entries_per_cell < 1e-3)
    if (rc == POA_OK && cfg && cfg->mode != POA_MODE_DENSE && flags && !(cfg->queue_entries_per_cell > 0.f && cfg->queue_entries_per_cell < 1e-3f)) {
        poa_config_t c2 = *cfg;
        float f = cfg->queue_entries_per_cell > 0.f ? cfg->queue_entries_per_cell : 0.25f;
        for (int attempt = 0; attempt < 2; ++attempt) {
            bool over = false;
            for (uint32_t i = 0; i < n_queries && !over; ++i) over = (flags[i] & POA_FLAG_EXACT_OVERFLOW) != 0;
            if (!over) break;
            f *= 8.f;
            c2.queue_entries_per_cell = f;
            rc = poa_batch_run_ex(b, costs, &c2, nullptr);
            if (rc == POA_ERR_OUT_OF_MEMORY) { rc = POA_OK; break; }   // the larger pool does not fit: keep what the first run gave
            if (rc == POA_OK) rc = poa_batch_fetch(b, score, pairs, pair_off, pair_capacity, flags, stats);
            if (rc != POA_OK) break;
        }
    }
    const double t3 = now();
    poa_batch_destroy(b);
    if (timing) std::fprintf(stderr, "poa_align_batch: create %.2f ms, launch %.2f ms, fetch (sync + copies) %.2f ms, destroy %.2f ms\n", t1 - t0, t2 - t1, t3 - t2, now() - t3);
    return rc;
}

void poa_release_cache(void) {
    g_buf_cache.release_all();
    std::lock_guard<std::mutex> lk(g_ws_cache.mu);
    for (int d = 0; d < 16; ++d) {
        if (!g_ws_cache.p[d]) continue;
        (void)hipSetDevice(d);
        (void)hipFree(g_ws_cache.p[d]);
        g_ws_cache.p[d] = nullptr; g_ws_cache.bytes[d] = 0;
    }
}

}  // extern "C"


// ---- two-piece affine model (poa_twopiece.hpp) ---------------------------------------------------------------------------
namespace {
// planes_out: null, or five host pointers (M, I1, D1, I2, D2) receiving the planes of query 0, rows x (len + 1)
int run_two_piece(const poa_graph_t* g, const poa_costs2_t* costs, const poa_config_t* cfg, uint32_t n_queries, const uint8_t* qseq,
                  const uint64_t* qoff, uint32_t* score, poa_aln_pair_t* pairs, uint64_t* pair_off, uint64_t pair_capacity, uint32_t* flags,
                  poa_stats_t* stats, int device, uint32_t* const* planes_out, uint32_t* counters_out) {
    if (!g || !costs || (n_queries && (!qseq || !qoff))) return fail(POA_ERR_INVALID_ARG, "poa_align_batch_2piece: null argument");
    // mode: DENSE = the dense Global pass; EXACT / HYBRID = the replay of the reference's search for EVERY query (under this
    // model a dense flag of 0 certifies the alignment only where the reference's search is optimal, and it need not be:
    // DESIGN.md §6a — so there is no cheaper hybrid)
    const bool exact = cfg && cfg->mode != POA_MODE_DENSE;
    if (cfg && cfg->mode > POA_MODE_HYBRID) return fail(POA_ERR_INVALID_ARG, "poa_align_batch_2piece_ex: unknown mode");
    if (cfg && cfg->span > POA_SPAN_ENDS_FREE) return fail(POA_ERR_INVALID_ARG, "poa_align_batch_2piece_ex: unknown alignment span");
    const bool ends_free = cfg && cfg->span == POA_SPAN_ENDS_FREE;
    if (ends_free && !exact) return fail(POA_ERR_UNSUPPORTED, "two-piece model: ends-free alignment needs the exact replay (mode EXACT)");
    const TuneView T(cfg);
    if (costs->gap_extend1 < costs->gap_extend2)
        return fail(POA_ERR_INVALID_ARG, "gap_extend1 must be greater than or equal to gap_extend2 for two-piece model");
    if (poa_device_count() <= 0) return fail(POA_ERR_NO_DEVICE, "no HIP device: the engine has no CPU alignment path");
    HIP_TRY(hipSetDevice(device));
    const FlatGraph& fg = g->g;
    if (stats) std::memset(stats, 0, sizeof(*stats));
    if (pair_off) pair_off[0] = 0;
    if (n_queries == 0) return POA_OK;
    uint64_t max_len = 0, cells = 0;
    for (uint32_t i = 0; i < n_queries; ++i) {
        if (qoff[i + 1] < qoff[i]) return fail(POA_ERR_INVALID_ARG, "qoff must be non-decreasing");
        max_len = std::max<uint64_t>(max_len, qoff[i + 1] - qoff[i]);
        cells += (uint64_t)fg.n * (qoff[i + 1] - qoff[i] + 1);
    }
    if (fg.n_real == 0) {  // empty graph: mod.rs:124-142
        for (uint32_t i = 0; i < n_queries; ++i) {
            if (score) score[i] = (uint32_t)(4 * (qoff[i + 1] - qoff[i]));
            if (flags) flags[i] = POA_FLAG_EMPTY_GRAPH;
            if (pair_off) pair_off[i + 1] = 0;
        }
        if (stats) { stats->n_queries = n_queries; stats->n_flagged = n_queries; }
        return POA_OK;
    }
    const uint32_t pitch = (uint32_t)((max_len + 64) & ~63ull);
    const uint64_t per_query = 5ull * fg.n * pitch;   // plane elements
    if (per_query >= (1ull << 34)) return fail(POA_ERR_UNSUPPORTED, "two-piece pass: planes of one query too large");
    if (exact && per_query >= (1ull << 32)) return fail(POA_ERR_UNSUPPORTED, "two-piece replay: the visited table of one query exceeds 2^32 cells");
    // u16 planes under the bound of the one-piece pass (poa_batch_run_ex) taken with the first piece's costs: a gap never
    // costs more than its first-piece price, so [o1 + e1 L] + [o1 + e1 (shortest path)] bounds the optimum here too
    const uint64_t ub = (max_len ? (uint64_t)costs->gap_open1 + (uint64_t)costs->gap_extend1 * max_len : 0) +
                        (fg.min_path_nodes ? (uint64_t)costs->gap_open1 + (uint64_t)costs->gap_extend1 * fg.min_path_nodes : 0);
    bool narrow = ub <= 65534;
    if (costs->wide_planes || exact) narrow = false;   // (the replayed table is u32)
    const uint64_t elem = narrow ? 2 : 4;
    // replay workspace per query slot (prepare_exact's sizes with five stacks per priority; the pool holds the entries live
    // at once — popped slots are reused — at cfg->queue_entries_per_cell entries per cell, default 0.5)
    uint32_t x_n_prio = 0, x_pool_cap = 0, x_stack_cap = 0, x_wpn = 0, x_swpn = 0;
    uint64_t x_bytes = 0;
    if (exact) {
        std::string err;
        int brc;
        {
            std::lock_guard<std::mutex> lk(const_cast<poa_graph*>(g)->bubble_mu);
            brc = build_bubble_index(const_cast<FlatGraph&>(fg), err);
        }
        if (brc != POA_OK) return fail(brc, err);
        const uint32_t om = std::max<uint32_t>(costs->gap_open1, costs->gap_open2);
        const uint32_t maxc = std::max<uint32_t>(costs->mismatch, om + costs->gap_extend1);
        const uint64_t n_prio64 = ((uint64_t)fg.n + max_len + 2) * maxc + 2ull * (om + ((uint64_t)fg.n + max_len) * costs->gap_extend1) + 64;
        if (n_prio64 > (1ull << 26)) return fail(POA_ERR_UNSUPPORTED, "two-piece replay: priority range too large for this graph / query size");
        const float f = (cfg->queue_entries_per_cell > 0.f) ? cfg->queue_entries_per_cell : 0.5f;
        const uint64_t pool64 = std::max<uint64_t>(1024, (uint64_t)(f * (double)fg.n * (double)(max_len + 1)));
        if (pool64 > 0xFFFFFFF0ull) return fail(POA_ERR_UNSUPPORTED, "two-piece replay: queue pool too large");
        x_n_prio = (uint32_t)n_prio64; x_pool_cap = (uint32_t)pool64; x_stack_cap = (uint32_t)(fg.n + max_len + 8);
        x_wpn = (uint32_t)((max_len + 1 + 63) / 64); x_swpn = (x_wpn + 63) / 64;
        x_bytes = (uint64_t)fg.n_exit * (x_wpn + x_swpn) * 8 + 5ull * x_n_prio * 4 + (uint64_t)x_stack_cap * 12 + (uint64_t)x_pool_cap * 16;
    }
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    // (the replay is bound by the latency of its dependent loads: as many searches in flight as the memory holds)
    const uint64_t budget = exact ? (uint64_t)(free_b * 0.85) : std::min<uint64_t>((uint64_t)(free_b * 0.6), 64ull << 30);
    uint32_t chunk = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(n_queries, budget / (per_query * elem + x_bytes + 1)));
    const uint32_t stride = (uint32_t)std::min<uint64_t>(fg.n + max_len + 1, 0xFFFFFFFFull);
    DevBuf<RowMeta> d_rows; DevBuf<uint32_t> d_pred, d_planes, d_score, d_flags, d_np; DevBuf<uint8_t> d_q; DevBuf<uint64_t> d_qoff;
    DevBuf<poa_aln_pair_t> d_scratch;
    HIP_TRY(d_rows.alloc(fg.rows.size())); HIP_TRY(d_pred.alloc(std::max<size_t>(fg.pred_rows.size(), 1)));
    HIP_TRY(d_q.alloc(std::max<uint64_t>(qoff[n_queries], 1))); HIP_TRY(d_qoff.alloc(n_queries + 1));
    HIP_TRY(d_score.alloc(n_queries)); HIP_TRY(d_flags.alloc(n_queries)); HIP_TRY(d_np.alloc(n_queries));
    if (d_planes.alloc((size_t)((chunk * per_query * elem + 3) / 4)) != hipSuccess) return fail(POA_ERR_OUT_OF_MEMORY, "two-piece pass: plane workspace");
    HIP_TRY(d_scratch.alloc((size_t)chunk * stride));
    // the replay's view of the graph and its workspace
    DevBuf<uint32_t> x_succ_off, x_succ, x_dmin, x_dmax, x_exit, x_nbm_off, x_node_row, x_sp, x_head, x_status, x_end, x_counters;
    DevBuf<uint8_t> x_sym; DevBuf<FlatGraph::NodeBubble> x_nbm; DevBuf<uint64_t> x_reached, x_rsum;
    DevBuf<ExQEntry> x_pool; DevBuf<ExStackEntry> x_stack;
    TwoPieceExact X;
    if (exact) {
        const uint32_t n = fg.n;
        HIP_TRY(x_succ_off.alloc(fg.succ_row_off.size())); HIP_TRY(x_succ.alloc(std::max<size_t>(fg.succ_rows.size(), 1)));
        HIP_TRY(x_dmin.alloc(n)); HIP_TRY(x_dmax.alloc(n)); HIP_TRY(x_exit.alloc(n)); HIP_TRY(x_nbm_off.alloc(n + 1));
        HIP_TRY(x_nbm.alloc(std::max<size_t>(fg.nbm.size(), 1))); HIP_TRY(x_node_row.alloc(n)); HIP_TRY(x_sp.alloc(n)); HIP_TRY(x_sym.alloc((size_t)n + 4));
        std::vector<uint8_t> row_sym((size_t)n + 4, 0);
        for (uint32_t r = 0; r < n; ++r) row_sym[r] = fg.rows[r].sym;
        HIP_TRY(hipMemcpy(x_sym.p, row_sym.data(), row_sym.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(x_succ_off.p, fg.succ_row_off.data(), fg.succ_row_off.size() * 4, hipMemcpyHostToDevice));
        if (!fg.succ_rows.empty()) HIP_TRY(hipMemcpy(x_succ.p, fg.succ_rows.data(), fg.succ_rows.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(x_dmin.p, fg.dist_min.data(), n * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(x_dmax.p, fg.dist_max.data(), n * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(x_exit.p, fg.exit_idx.data(), n * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(x_nbm_off.p, fg.nbm_off.data(), ((size_t)n + 1) * 4, hipMemcpyHostToDevice));
        if (!fg.nbm.empty()) HIP_TRY(hipMemcpy(x_nbm.p, fg.nbm.data(), fg.nbm.size() * sizeof(FlatGraph::NodeBubble), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(x_node_row.p, fg.node_row.data(), n * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(x_sp.p, fg.sp_to_end.data(), n * 4, hipMemcpyHostToDevice));
        HIP_TRY(x_reached.alloc(std::max<uint64_t>((uint64_t)chunk * fg.n_exit * x_wpn, 1)));
        HIP_TRY(x_rsum.alloc(std::max<uint64_t>((uint64_t)chunk * fg.n_exit * x_swpn, 1)));
        HIP_TRY(x_head.alloc((size_t)chunk * 5 * x_n_prio)); HIP_TRY(x_stack.alloc((size_t)chunk * x_stack_cap));
        if (x_pool.alloc((size_t)chunk * x_pool_cap) != hipSuccess) return fail(POA_ERR_OUT_OF_MEMORY, "two-piece replay: queue pool");
        HIP_TRY(x_status.alloc(n_queries)); HIP_TRY(x_end.alloc(2 * (size_t)n_queries)); HIP_TRY(x_counters.alloc(4 * (size_t)n_queries));
        X.G = ExactGraph{n, fg.start_row, fg.end_row, x_sym.p, x_succ_off.p, x_succ.p, x_dmin.p, x_dmax.p, x_exit.p, fg.n_exit,
                         x_nbm_off.p, x_nbm.p, x_node_row.p, x_sp.p, nullptr};
        X.C = ExactCosts{costs->mismatch, costs->gap_open1, costs->gap_extend1, cfg->heuristic, cfg->pruning, 0, 0, 0, 0, 0, 0,
                         costs->gap_open2, costs->gap_extend2};
        if (ends_free) {
            X.C.ends_free = 1;
            X.C.qfe_kind = cfg->qry_free_end.kind; X.C.qfe_val = cfg->qry_free_end.value;
            X.C.gfb_kind = cfg->graph_free_begin.kind;
            X.C.gfe_kind = cfg->graph_free_end.kind; X.C.gfe_val = cfg->graph_free_end.value;
        }
        X.reached = x_reached.p; X.rsum = x_rsum.p; X.wpn = x_wpn; X.swpn = x_swpn;
        X.head = x_head.p; X.n_prio = x_n_prio; X.pool = x_pool.p; X.pool_cap = x_pool_cap;
        X.stack = x_stack.p; X.stack_cap = x_stack_cap;
        X.status = x_status.p; X.end_cell = x_end.p; X.counters = x_counters.p;
    }
    HIP_TRY(hipMemcpy(d_rows.p, fg.rows.data(), fg.rows.size() * sizeof(RowMeta), hipMemcpyHostToDevice));
    if (!fg.pred_rows.empty()) HIP_TRY(hipMemcpy(d_pred.p, fg.pred_rows.data(), fg.pred_rows.size() * 4, hipMemcpyHostToDevice));
    if (qoff[n_queries]) HIP_TRY(hipMemcpy(d_q.p, qseq, qoff[n_queries], hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_qoff.p, qoff, ((size_t)n_queries + 1) * 8, hipMemcpyHostToDevice));
    TwoPieceParams P;
    P.rows = d_rows.p; P.pred_rows = d_pred.p; P.n_rows = fg.n; P.start_row = fg.start_row; P.end_row = fg.end_row;
    P.qseq = d_q.p; P.qoff = d_qoff.p; P.pitch = pitch; P.planes = d_planes.p;
    P.x = costs->mismatch; P.o1 = costs->gap_open1; P.e1 = costs->gap_extend1; P.e2 = costs->gap_extend2; P.oe = (uint32_t)costs->gap_open1 + costs->gap_extend1;
    P.score = d_score.p; P.flags = d_flags.p; P.n_pairs = d_np.p; P.scratch = d_scratch.p; P.scratch_stride = stride;
    P.exact_pass = exact ? 1u : 0u; P.ex_status = x_status.p; P.ex_end = x_end.p;
    struct Events {   // destroyed on every way out
        hipEvent_t e[3] = {nullptr, nullptr, nullptr};
        ~Events() { for (auto ev : e) if (ev) (void)hipEventDestroy(ev); }
    } evs;
    for (auto& ev : evs.e) HIP_TRY(hipEventCreate(&ev));
    hipEvent_t &e0 = evs.e[0], &e1 = evs.e[1], &e2 = evs.e[2];
    float ms_f = 0.f, ms_t = 0.f;
    std::vector<uint32_t> h_np(n_queries), h_flags(n_queries);
    std::vector<poa_aln_pair_t> h_scratch((size_t)chunk * stride);
    uint64_t at = 0;
    uint32_t n_chunks = 0;
    int rc = POA_OK;
    for (uint32_t first = 0; first < n_queries; first += chunk) {
        const uint32_t cnt = std::min(chunk, n_queries - first);
        P.first_query = first; P.n_queries = cnt;
        HIP_TRY(hipEventRecord(e0, nullptr));
        if (exact) {
            // the search fills its table from nothing: unvisited planes, empty stacks, empty reached sets
            HIP_TRY(hipMemsetAsync(d_planes.p, 0xFF, (size_t)cnt * per_query * 4, nullptr));
            HIP_TRY(hipMemsetAsync(x_head.p, 0xFF, (size_t)cnt * 5 * x_n_prio * 4, nullptr));
            if (fg.n_exit) {
                HIP_TRY(hipMemsetAsync(x_reached.p, 0, (size_t)cnt * fg.n_exit * x_wpn * 8, nullptr));
                HIP_TRY(hipMemsetAsync(x_rsum.p, 0, (size_t)cnt * fg.n_exit * x_swpn * 8, nullptr));
            }
            // active lanes per wave: enough waves to fill the chip first (one divergent search per lane), then more lanes
            uint32_t lanes = (cnt + 8191) / 8192;
            if (lanes > 64) lanes = 64;
            if (const int* lv = T.ptr(POA_TUNE_EXACT_LANES)) { const int v = (*lv); if (v >= 1 && v <= 64) lanes = (uint32_t)v; }
            X.lanes_per_wave = lanes;
            // graph arrays in LDS when they fit (one copy per block of 16 waves)
            X.n_succ = (uint32_t)fg.succ_rows.size(); X.n_nbm = (uint32_t)fg.nbm.size();
            const uint32_t lds_bytes = exact_lds_bytes(fg.n, X.n_succ, X.n_nbm);
            bool lds_graph = lds_bytes <= 150u * 1024u;
            if (const int* gv = T.ptr(POA_TUNE_EXACT_LDS)) lds_graph = lds_graph && (*gv) != 0;
            if (lds_graph && lds_bytes > 48u * 1024u)
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(poa2_exact_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            X.lds_graph = lds_graph ? 1u : 0u;
            const uint32_t per_block = lanes * (EXACT2_BLOCK / 64);
            hipLaunchKernelGGL(poa2_exact_kernel, dim3((cnt + per_block - 1) / per_block), dim3(EXACT2_BLOCK), lds_graph ? lds_bytes : 0, nullptr, P, X);
        }
        // previous row in registers for up to 1024 (u16: two passes of 512) / 1024 (u32: four passes of 256) columns
        else if (narrow) hipLaunchKernelGGL((poa2_forward_kernel<uint16_t, 2>), dim3(cnt), dim3(64), 0, nullptr, P);
        else hipLaunchKernelGGL((poa2_forward_kernel<uint32_t, 4>), dim3(cnt), dim3(64), 0, nullptr, P);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(e1, nullptr));
        if (narrow) hipLaunchKernelGGL(poa2_traceback_kernel<uint16_t>, dim3((cnt + 63) / 64), dim3(64), 0, nullptr, P);
        else hipLaunchKernelGGL(poa2_traceback_kernel<uint32_t>, dim3((cnt + 63) / 64), dim3(64), 0, nullptr, P);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(e2, nullptr));
        HIP_TRY(hipEventSynchronize(e2));
        float a = 0, b2 = 0;
        (void)hipEventElapsedTime(&a, e0, e1); (void)hipEventElapsedTime(&b2, e1, e2);
        ms_f += a; ms_t += b2; n_chunks++;
        HIP_TRY(hipMemcpy(h_np.data() + first, d_np.p + first, (size_t)cnt * 4, hipMemcpyDeviceToHost));
        if (pairs || pair_off) {
            HIP_TRY(hipMemcpy(h_scratch.data(), d_scratch.p, (size_t)cnt * stride * sizeof(poa_aln_pair_t), hipMemcpyDeviceToHost));
            for (uint32_t i = 0; i < cnt; ++i) {
                const uint32_t np = h_np[first + i];
                if (np > stride) return fail(POA_ERR_HIP, "two-piece traceback: walk longer than rows + length");
                if (pairs) {
                    if (at + np > pair_capacity) rc = fail(POA_ERR_CAPACITY, "pair_capacity too small");
                    else std::memcpy(pairs + at, h_scratch.data() + (size_t)i * stride + (stride - np), (size_t)np * sizeof(poa_aln_pair_t));
                }
                at += np;
                if (pair_off) pair_off[first + i + 1] = at;
            }
        }
        if (planes_out && first == 0) {
            const uint32_t L0 = (uint32_t)(qoff[1] - qoff[0]);
            std::vector<uint32_t> row(pitch);
            std::vector<uint16_t> row16(pitch);
            for (int pl = 0; pl < 5; ++pl)
                for (uint32_t r = 0; r < fg.n; ++r) {
                    const uint64_t at_el = ((uint64_t)pl * fg.n + r) * pitch;
                    if (narrow) {
                        HIP_TRY(hipMemcpy(row16.data(), reinterpret_cast<const uint16_t*>(d_planes.p) + at_el, (size_t)pitch * 2, hipMemcpyDeviceToHost));
                        for (uint32_t c = 0; c < pitch; ++c) row[c] = row16[c] == 0xFFFFu ? 0xFFFFFFFFu : row16[c];
                    } else {
                        HIP_TRY(hipMemcpy(row.data(), d_planes.p + at_el, (size_t)pitch * 4, hipMemcpyDeviceToHost));
                    }
                    std::memcpy(planes_out[pl] + (size_t)r * (L0 + 1), row.data(), ((size_t)L0 + 1) * 4);
                }
        }
    }
    if (score) HIP_TRY(hipMemcpy(score, d_score.p, (size_t)n_queries * 4, hipMemcpyDeviceToHost));
    if (exact && counters_out) HIP_TRY(hipMemcpy(counters_out, x_counters.p, (size_t)n_queries * 16, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(h_flags.data(), d_flags.p, (size_t)n_queries * 4, hipMemcpyDeviceToHost));
    if (flags) std::memcpy(flags, h_flags.data(), (size_t)n_queries * 4);
    if (stats) {
        stats->cells = cells; stats->bases = qoff[n_queries]; stats->plane_bytes = cells * 5 * elem; stats->n_queries = n_queries;
        stats->n_chunks = n_chunks; stats->n_forward_launches = n_chunks; stats->ms_forward = ms_f; stats->ms_traceback = ms_t;
        uint32_t nf = 0;
        for (uint32_t i = 0; i < n_queries; ++i) nf += h_flags[i] != 0;
        stats->n_flagged = nf;
        if (exact) { stats->n_exact = n_queries; stats->ms_exact = ms_f; stats->ms_forward = 0; }
    }
    return rc;
}
}  // namespace

int poa_align_batch_2piece(const poa_graph_t* g, const poa_costs2_t* costs, uint32_t n_queries, const uint8_t* qseq,
                           const uint64_t* qoff, uint32_t* score, poa_aln_pair_t* pairs, uint64_t* pair_off,
                           uint64_t pair_capacity, uint32_t* flags, poa_stats_t* stats, int device) {
    return run_two_piece(g, costs, nullptr, n_queries, qseq, qoff, score, pairs, pair_off, pair_capacity, flags, stats, device, nullptr, nullptr);
}

int poa_align_batch_2piece_ex(const poa_graph_t* g, const poa_costs2_t* costs, const poa_config_t* cfg, uint32_t n_queries,
                              const uint8_t* qseq, const uint64_t* qoff, uint32_t* score, poa_aln_pair_t* pairs, uint64_t* pair_off,
                              uint64_t pair_capacity, uint32_t* flags, poa_stats_t* stats, uint32_t* search_counters, int device) {
    return run_two_piece(g, costs, cfg, n_queries, qseq, qoff, score, pairs, pair_off, pair_capacity, flags, stats, device, nullptr, search_counters);
}

int poa_planes_2piece(const poa_graph_t* g, const poa_costs2_t* costs, const uint8_t* seq, uint32_t len, uint32_t* m,
                      uint32_t* i1, uint32_t* d1, uint32_t* i2, uint32_t* d2, int device) {
    if (!m || !i1 || !d1 || !i2 || !d2) return fail(POA_ERR_INVALID_ARG, "poa_planes_2piece: null plane");
    const uint64_t qoff[2] = {0, len};
    uint32_t* out[5] = {m, i1, d1, i2, d2};
    uint32_t sc = 0, fl = 0;
    return run_two_piece(g, costs, nullptr, 1, seq, qoff, &sc, nullptr, nullptr, 0, &fl, nullptr, device, out, nullptr);
}
