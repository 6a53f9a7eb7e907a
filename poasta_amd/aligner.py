"""Host-side mirror of the reference's aligner entry point, running on the gfx950 engine.

Mirrors `poasta::aligner::PoastaAligner` (/root/reference/src/aligner/mod.rs:40-146) and the types a
caller touches: `GapAffine` (scoring/gap_affine.rs:20-30), the `AlignmentConfig` bindings
`AffineMinGapCost` / `AffineDijkstra` (config.rs:49,:104), `AlignmentType` (scoring/mod.rs:50-62),
`AstarResult` (astar.rs:81-90) and `AlignedPair` (alignment.rs:4-13).  Same names, same argument
meaning; errors are exceptions where the reference panics.

What differs, by design: a call computes the full M/I/D score planes on the GPU instead of searching
them with A*; `AstarResult.flags` reports when the reference's own search order could have picked
a different co-optimal alignment (0 = provably the reference's alignment).
`align_batch` is the data-parallel shape of `lasagna align` (src/bin/lasagna.rs:184-276).
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from .graph import FlatGraph, pack_queries


class GapAffine:
    """`GapAffine::new(cost_mismatch, cost_gap_extend, cost_gap_open)` — NB the argument order
    (scoring/gap_affine.rs:27)."""

    def __init__(self, cost_mismatch, cost_gap_extend, cost_gap_open):
        for v in (cost_mismatch, cost_gap_extend, cost_gap_open):
            if not 0 <= int(v) <= 255:
                raise ValueError("GapAffine costs are u8")
        self.cost_mismatch, self.cost_gap_extend, self.cost_gap_open = int(cost_mismatch), int(cost_gap_extend), int(cost_gap_open)

    def mismatch(self):
        return self.cost_mismatch

    def gap_open(self):
        return self.cost_gap_open

    def gap_extend(self):
        return self.cost_gap_extend

    def gap_cost(self, in_gap_state, length):
        """gap_affine.rs:68-80; in_gap_state False == AlignState::Match."""
        if length == 0:
            return 0
        return (0 if in_gap_state else self.cost_gap_open) + length * self.cost_gap_extend

    def _c(self):
        return _lib.PoaCosts(self.cost_mismatch, self.cost_gap_open, self.cost_gap_extend, 0)


class GapAffine2Piece:
    """`GapAffine2Piece::new(cost_mismatch, cost_gap_extend1, cost_gap_open1, cost_gap_extend2, cost_gap_open2)` — the
    reference's argument order (gap_affine_2piece.rs:28-33); it panics unless extend1 >= extend2."""

    def __init__(self, cost_mismatch, cost_gap_extend1, cost_gap_open1, cost_gap_extend2, cost_gap_open2):
        if cost_gap_extend1 < cost_gap_extend2:
            raise ValueError("gap_extend1 must be greater than or equal to gap_extend2 for two-piece model")
        self.cost_mismatch, self.cost_gap_extend1, self.cost_gap_open1 = cost_mismatch, cost_gap_extend1, cost_gap_open1
        self.cost_gap_extend2, self.cost_gap_open2 = cost_gap_extend2, cost_gap_open2

    def mismatch(self):
        return self.cost_mismatch

    def gap_open(self):
        return self.cost_gap_open1

    def gap_extend(self):
        return self.cost_gap_extend1

    def gap_open2(self):
        return self.cost_gap_open2

    def gap_extend2(self):
        return self.cost_gap_extend2

    def breakpoint(self):
        """gap_affine_2piece.rs:36-66"""
        if self.cost_gap_extend1 == self.cost_gap_extend2:
            return (1 << 64) - 1 if self.cost_gap_open1 <= self.cost_gap_open2 else 0
        den = self.cost_gap_extend1 - self.cost_gap_extend2
        if self.cost_gap_open2 >= self.cost_gap_open1:
            return (self.cost_gap_open2 - self.cost_gap_open1) // den
        return (self.cost_gap_open1 - self.cost_gap_open2 + den - 1) // den

    def _c(self):
        return _lib.PoaCosts2(self.cost_mismatch, self.cost_gap_open1, self.cost_gap_extend1, self.cost_gap_open2, self.cost_gap_extend2,
                              1 if os.environ.get("POA_PLANES") == "32" else 0)


class Affine2PieceDijkstra:
    """config.rs:160-213.  The engine's two-piece pass is the dense one (Global): the optimum of the reference's two-piece
    alignment graph = what this configuration's search returns without pruning (`align_no_pruning`)."""
    heuristic = _lib.HEURISTIC_DIJKSTRA
    two_piece = True

    def __init__(self, costs):
        self.costs = costs


class Affine2PieceMinGapCost:
    """config.rs:215-272 — what `poasta align -g o1,o2 -e e1,e2` constructs (src/bin/poasta.rs:319-445).  Its results are
    those of the reference's SEARCH (min-gap heuristic over a gap_cost that charges the open cost again inside a gap,
    pruning): run it with `PoastaAligner(config, aln_type, mode="exact")`."""
    heuristic = _lib.HEURISTIC_MINGAP
    two_piece = True

    def __init__(self, costs):
        self.costs = costs


class Bound:
    """std::ops::Bound<usize> as the reference's AlignmentType::EndsFree uses it."""
    Unbounded = (_lib.BOUND_UNBOUNDED, 0)

    @staticmethod
    def Included(n):
        return (_lib.BOUND_INCLUDED, int(n))

    @staticmethod
    def Excluded(n):
        return (_lib.BOUND_EXCLUDED, int(n))


class EndsFree:
    """AlignmentType::EndsFree {qry_free_begin, qry_free_end, graph_free_begin, graph_free_end} (scoring/mod.rs:56-61).
    What the reference returns for it is defined by its search (gap_affine.rs:136-248), so the engine replays that
    search for every query (exact mode is implied)."""

    def __init__(self, qry_free_begin=Bound.Unbounded, qry_free_end=Bound.Unbounded, graph_free_begin=Bound.Unbounded,
                 graph_free_end=Bound.Unbounded):
        self.bounds = (qry_free_begin, qry_free_end, graph_free_begin, graph_free_end)


class AlignmentType:
    """scoring/mod.rs:50-62: `AlignmentType.Global` (what `lasagna` hard-codes, src/bin/lasagna.rs:256) or
    `AlignmentType.EndsFree(...)`."""
    Global = "global"
    EndsFree = EndsFree


class AffineMinGapCost:
    """config.rs:104 — the default config of both reference CLIs.  The heuristic only fixes the
    reference's search order: the dense GPU pass has none, the exact replay emulates it."""
    heuristic = _lib.HEURISTIC_MINGAP

    def __init__(self, costs):
        self.costs = costs


class AffineDijkstra(AffineMinGapCost):
    """config.rs:49."""
    heuristic = _lib.HEURISTIC_DIJKSTRA


MODES = {"dense": _lib.MODE_DENSE, "exact": _lib.MODE_EXACT, "hybrid": _lib.MODE_HYBRID}


def make_config(mode="dense", heuristic=_lib.HEURISTIC_MINGAP, pruning=True, queue_entries_per_cell=0.0, full_planes=False,
                aln_type=AlignmentType.Global, **tune):
    """poa_config_t: `mode` "dense" | "exact" (replay the reference's A* for every query: bit-identical
    tie-breaks) | "hybrid" (replay only the queries the dense pass could not certify); `aln_type` Global or EndsFree(...)."""
    cfg = _lib.PoaConfig(MODES[mode] if isinstance(mode, str) else int(mode), int(heuristic), 1 if pruning else 0,
                         float(queue_entries_per_cell), _lib.CFG_FULL_PLANES if full_planes else 0)
    if isinstance(aln_type, EndsFree):
        cfg.span = _lib.SPAN_ENDS_FREE
        for name, (kind, value) in zip(("qry_free_begin", "qry_free_end", "graph_free_begin", "graph_free_end"), aln_type.bounds):
            setattr(cfg, name, _lib.PoaBound(kind, value))
    _lib.tune_from_env(cfg, **tune)   # kernel / layout overrides: POA_<NAME> variables, keyword arguments (poa_config_t.tune)
    return cfg


class AlignedPair:
    __slots__ = ("rpos", "qpos")

    def __init__(self, rpos, qpos):
        self.rpos, self.qpos = rpos, qpos

    def is_aligned(self):
        return self.rpos is not None and self.qpos is not None

    def is_indel(self):
        return not self.is_aligned()

    def __eq__(self, o):
        return (self.rpos, self.qpos) == (o.rpos, o.qpos)

    def __repr__(self):
        return "AlignedPair(rpos=%r, qpos=%r)" % (self.rpos, self.qpos)


class AstarResult:
    """astar.rs:81-90.  `num_*` search counters have no meaning for a dense pass and are 0;
    `cells` = rows x (len + 1) computed for this query."""

    def __init__(self, score, alignment, flags=0, cells=0):
        self.score, self.alignment, self.flags, self.cells = score, alignment, flags, cells
        self.num_queued = self.num_visited = self.num_pruned = 0

    def pairs(self):
        return [(p.rpos, p.qpos) for p in self.alignment]


class BatchResult:
    """Struct-of-arrays result of `align_batch`."""

    def __init__(self, score, pairs, pair_off, flags, stats):
        self.score, self.pairs, self.pair_off, self.flags, self.stats = score, pairs, pair_off, flags, stats

    def __len__(self):
        return len(self.score)

    def alignment(self, i):
        """[(rpos|None, qpos|None), ...] of query i."""
        a = self.pairs[int(self.pair_off[i]):int(self.pair_off[i + 1])]
        return [(None if r == _lib.POA_NONE else int(r), None if q == _lib.POA_NONE else int(q)) for r, q in a.tolist()]

    def raw_alignment(self, i):
        return [tuple(x) for x in self.pairs[int(self.pair_off[i]):int(self.pair_off[i + 1])].tolist()]

    def result(self, i):
        return AstarResult(int(self.score[i]), [AlignedPair(r, q) for r, q in self.alignment(i)], int(self.flags[i]))


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class DeviceGraph:
    """Owns a `poa_graph_t` (the flattened AlignableRefGraph)."""

    def __init__(self, graph):
        if not isinstance(graph, FlatGraph):
            graph = FlatGraph.from_dict(graph)
        self.graph = graph
        h = C.c_void_p()
        _lib.check(_lib.lib().poa_graph_create(graph.n, graph.start, graph.end, _p(graph.symbol), _p(graph.succ_off),
                                               _p(graph.succ), _p(graph.pred_off), _p(graph.pred), C.byref(h)))
        self.handle = h

    def node_rows(self):
        r = np.zeros(self.graph.n, np.uint32)
        _lib.check(_lib.lib().poa_graph_node_rows(self.handle, _p(r)))
        return r

    def __del__(self):
        try:
            if self.handle:
                _lib.lib().poa_graph_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def _device_graph(graph):
    if isinstance(graph, DeviceGraph):
        return graph
    dg = getattr(graph, "_device_graph", None)
    if dg is None:
        dg = DeviceGraph(graph)
        try:
            graph._device_graph = dg
        except Exception:
            pass
    return dg


class ResidentBatch:
    """Queries + results resident in HBM (`poa_batch_*`): create once, run many times."""

    def __init__(self, graph, qseq, qoff, device=0, workspace_bytes=0):
        self.dg = _device_graph(graph)
        self.qseq = np.ascontiguousarray(qseq, np.uint8)
        self.qoff = np.ascontiguousarray(qoff, np.uint64)
        self.n = len(self.qoff) - 1
        h = C.c_void_p()
        _lib.check(_lib.lib().poa_batch_create(self.dg.handle, device, self.n, _p(self.qseq), _p(self.qoff),
                                               int(workspace_bytes), C.byref(h)))
        self.handle = h
        self.pair_capacity = int(self.qoff[-1]) + self.n * self.dg.graph.n

    def run(self, costs, stream=None, config=None):
        c = costs._c()
        if config is None:
            config = _lib.tune_from_env()   # (a dense-mode config carrying the overrides, if any are set)
        if config is None:
            _lib.check(_lib.lib().poa_batch_run(self.handle, C.byref(c), C.c_void_p(stream or 0)))
        else:
            _lib.check(_lib.lib().poa_batch_run_ex(self.handle, C.byref(c), C.byref(config), C.c_void_p(stream or 0)))

    def fetch(self, want_pairs=True, pinned=False, copy=False):
        """Synchronise and copy the results to the host.  pinned=True: the destination buffers are page-locked (allocated
        once per batch through torch, if importable), which lets the device->host copy run at PCIe speed instead of through
        the driver's staging of pageable memory.  NB the arrays of a pinned fetch ALIAS those per-batch buffers: the next
        pinned fetch of this batch overwrites them — pass copy=True (or copy what you keep) to get arrays of your own."""
        n = self.n
        bufs = self._host_buffers(want_pairs, pinned)
        score, flags, pair_off, pairs = bufs
        st = _lib.PoaStats()
        _lib.check(_lib.lib().poa_batch_fetch(self.handle, _p(score), _p(pairs), _p(pair_off), self.pair_capacity,
                                              _p(flags), C.byref(st)))
        if want_pairs:
            pairs = pairs[:int(pair_off[n])]
        if copy and pinned:
            score, pairs, pair_off, flags = score.copy(), pairs.copy(), pair_off.copy(), flags.copy()
        return BatchResult(score, pairs, pair_off, flags, st.as_dict())

    def _host_buffers(self, want_pairs, pinned):
        n = self.n
        if pinned:
            cached = getattr(self, "_pinned", None)
            if cached is None:
                try:
                    import torch
                    mk = lambda shape, dt: torch.empty(shape, dtype=dt, pin_memory=True).numpy()
                    cached = (mk((n,), torch.int32).view(np.uint32), mk((n,), torch.int32).view(np.uint32),
                              mk((n + 1,), torch.int64).view(np.uint64), mk((max(self.pair_capacity, 1), 2), torch.int32).view(np.uint32))
                except Exception:
                    cached = False
                self._pinned = cached
            if cached:
                return cached[0], cached[1], cached[2], (cached[3] if want_pairs else None)
        score, flags = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        pair_off = np.zeros(n + 1, np.uint64)
        pairs = np.zeros((max(self.pair_capacity, 1), 2), np.uint32) if want_pairs else None
        return score, flags, pair_off, pairs

    def stats(self):
        """Synchronise and return HIP-event timings summed over the runs since the last call."""
        st = _lib.PoaStats()
        _lib.check(_lib.lib().poa_batch_stats(self.handle, C.byref(st)))
        return st.as_dict()

    def search_counters(self):
        """AstarResult::{num_queued, num_visited, num_pruned} + wave-search steps, one row per query (exact / hybrid runs)."""
        out = np.zeros((self.n, 4), np.uint32)
        _lib.check(_lib.lib().poa_batch_fetch_search_counters(self.handle, _p(out)))
        return out

    def layout(self):
        """How the dense pass of the last run stored its planes: subset of {"u16", "compact", "relative"} (empty: u32 planes)."""
        v = C.c_uint32(0)
        _lib.check(_lib.lib().poa_batch_last_layout(self.handle, C.byref(v)))
        return {name for bit, name in ((1, "u16"), (2, "compact"), (4, "relative")) if v.value & bit}

    def device_results(self):
        ptrs = [C.c_void_p() for _ in range(4)]
        _lib.check(_lib.lib().poa_batch_device_results(self.handle, *[C.byref(p) for p in ptrs]))
        return dict(zip(("score", "flags", "pair_off", "pairs"), [p.value for p in ptrs]))

    def planes(self, query):
        rows = self.dg.graph.n
        cols = int(self.qoff[query + 1] - self.qoff[query]) + 1
        m, i, d = (np.zeros((rows, cols), np.uint32) for _ in range(3))
        _lib.check(_lib.lib().poa_batch_fetch_planes(self.handle, query, _p(m), _p(i), _p(d)))
        return m, i, d

    def close(self):
        if getattr(self, "handle", None):
            _lib.lib().poa_batch_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PoastaAligner:
    """`PoastaAligner::new(config, aln_type)` (mod.rs:53)."""

    def __init__(self, config, aln_type=AlignmentType.Global, device=0, mode="dense", queue_entries_per_cell=0.0):
        if aln_type != AlignmentType.Global and not isinstance(aln_type, EndsFree):
            raise ValueError("aln_type must be AlignmentType.Global or AlignmentType.EndsFree(...)")
        self.config, self.aln_type, self.device = config, aln_type, device
        self.mode, self.queue_entries_per_cell = mode, queue_entries_per_cell

    # -- the three reference entry points; all run the same dense pass ------------------------
    def align(self, ref_graph, seq, pruning=True):
        """mod.rs:114-145."""
        return self.align_batch(ref_graph, [seq], pruning=pruning).result(0)

    def align_with_existing_bubbles(self, ref_graph, seq, existing_bubbles=None):
        """mod.rs:69-79.  The bubble index only steers the reference's pruning; unused here."""
        return self.align(ref_graph, seq)

    def align_no_pruning(self, ref_graph, seq):
        """mod.rs:81-90 (matters only for the exact replay: no pruning changes which cells the reference visits)."""
        return self.align(ref_graph, seq, pruning=False)

    def planes_2piece(self, ref_graph, seq):
        """M, I1, D1, I2, D2 of one query under the two-piece model, rows = topological rank (parity tests)."""
        dg = _device_graph(ref_graph)
        s = np.ascontiguousarray(np.frombuffer(seq, np.uint8) if isinstance(seq, (bytes, bytearray)) else seq, np.uint8)
        shape = (dg.graph.n, len(s) + 1)
        out = [np.zeros(shape, np.uint32) for _ in range(5)]
        c = self.config.costs._c()
        _lib.check(_lib.lib().poa_planes_2piece(dg.handle, C.byref(c), _p(s), len(s), *[_p(a) for a in out], self.device))
        return out

    # -- batch shape (lasagna.rs:246-268) ------------------------------------------------------
    def align_batch(self, ref_graph, seqs=None, qseq=None, qoff=None, want_pairs=True, pruning=True):
        dg = _device_graph(ref_graph)
        if seqs is not None:
            qseq, qoff = pack_queries(seqs)
        qseq = np.ascontiguousarray(qseq, np.uint8)
        qoff = np.ascontiguousarray(qoff, np.uint64)
        n = len(qoff) - 1
        score, flags = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        pair_off = np.zeros(n + 1, np.uint64)
        cap = int(qoff[-1]) + n * dg.graph.n
        pairs = np.zeros((max(cap, 1), 2), np.uint32) if want_pairs else None
        st = _lib.PoaStats()
        c = self.config.costs._c()
        if getattr(self.config, "two_piece", False):
            if self.mode == "dense":
                if self.aln_type != AlignmentType.Global:
                    raise ValueError("two-piece model: ends-free alignment runs as the exact replay (mode='exact')")
                _lib.check(_lib.lib().poa_align_batch_2piece(dg.handle, C.byref(c), n, _p(qseq), _p(qoff), _p(score), _p(pairs),
                                                             _p(pair_off), cap, _p(flags), C.byref(st), self.device))
                counters = None
            else:
                # the reference's own search, five states (gap_affine_2piece.rs): scores and alignments are what it returns
                cfg = make_config(self.mode, self.config.heuristic, pruning, self.queue_entries_per_cell, aln_type=self.aln_type)
                counters = np.zeros((n, 4), np.uint32)
                _lib.check(_lib.lib().poa_align_batch_2piece_ex(dg.handle, C.byref(c), C.byref(cfg), n, _p(qseq), _p(qoff), _p(score),
                                                                _p(pairs), _p(pair_off), cap, _p(flags), C.byref(st), _p(counters),
                                                                self.device))
            if want_pairs:
                pairs = pairs[:int(pair_off[n])]
            res = BatchResult(score, pairs, pair_off, flags, st.as_dict())
            res.search_counters = counters   # num_queued, num_visited, num_pruned, live queue entries (exact mode)
            return res
        cfg = make_config(self.mode, self.config.heuristic, pruning, self.queue_entries_per_cell, aln_type=self.aln_type)
        _lib.check(_lib.lib().poa_align_batch_ex(dg.handle, C.byref(c), C.byref(cfg), n, _p(qseq), _p(qoff), _p(score),
                                                 _p(pairs), _p(pair_off), cap, _p(flags), C.byref(st), self.device))
        if want_pairs:
            pairs = pairs[:int(pair_off[n])]
        return BatchResult(score, pairs, pair_off, flags, st.as_dict())
