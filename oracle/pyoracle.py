"""ORACLE — TEST INFRASTRUCTURE ONLY (ctypes view of oracle/libpoa_oracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (poasta_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpoa_oracle.so")
NONE = 0xFFFFFFFF
UNVISITED = 0xFFFFFFFF
ST_M, ST_D, ST_I = 0, 1, 2
H_DIJKSTRA, H_MINGAP = 0, 1

DF_AMBIGUOUS, DF_START_QUIRK, DF_REF_PANIC, DF_SHORT_QUERY, DF_TRUNCATED = 1, 2, 4, 8, 16


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cpp", ".hpp"))]
    if force or not os.path.exists(_LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "libpoa_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        vp, u8p, u32p, u64p, i32p = C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_int32)
        L.oracle_last_error.restype = C.c_char_p
        L.oracle_graph_from_csr.restype = vp
        L.oracle_graph_from_csr.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp, C.c_int]
        L.oracle_graph_new_poa.restype = vp
        L.oracle_graph_mock.restype = vp
        L.oracle_graph_mock.argtypes = [C.c_int]
        L.oracle_graph_mock_edges.restype = vp
        L.oracle_graph_mock_edges.argtypes = [C.c_uint32, vp, C.c_uint32, vp]
        L.oracle_graph_free.argtypes = [vp]
        L.oracle_graph_set_symbols.argtypes = [vp, vp]
        for f in ("oracle_graph_n", "oracle_graph_start", "oracle_graph_end", "oracle_graph_n_edges", "oracle_poa_n_sequences"):
            getattr(L, f).restype = C.c_uint32
            getattr(L, f).argtypes = [vp]
        L.oracle_poa_seq_start.restype = C.c_uint32
        L.oracle_poa_seq_start.argtypes = [vp, C.c_uint32]
        L.oracle_poa_aligned_nodes.restype = C.c_uint32
        L.oracle_poa_aligned_nodes.argtypes = [vp, C.c_uint32, vp, C.c_uint32]
        L.oracle_graph_export.argtypes = [vp] * 7
        L.oracle_poa_add_alignment.argtypes = [vp, C.c_char_p, vp, C.c_uint64, vp, C.c_int64]
        L.oracle_gap_cost.restype = C.c_uint64
        L.oracle_gap_cost.argtypes = [C.c_uint8, C.c_uint8, C.c_uint8, C.c_int, C.c_uint64]
        L.oracle_rev_postorder.restype = C.c_uint32
        L.oracle_rev_postorder.argtypes = [vp, vp]
        L.oracle_superbubbles.argtypes = [vp, vp]
        L.oracle_bubble_index.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_uint32, vp, vp]
        L.oracle_heuristic_h.restype = C.c_uint64
        L.oracle_heuristic_h.argtypes = [vp, C.c_uint8, C.c_uint8, C.c_uint8, C.c_int, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int]
        L.oracle_dfa_first_event.argtypes = [vp, vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, vp]
        L.oracle_queue_new.restype = vp
        L.oracle_queue_free.argtypes = [vp]
        L.oracle_queue_push.argtypes = [vp, C.c_uint32, C.c_uint64]
        L.oracle_queue_pop.restype = C.c_int64
        L.oracle_queue_pop.argtypes = [vp]
        for f in ("oracle_queue_layers", "oracle_queue_layer_min"):
            getattr(L, f).restype = C.c_uint64
            getattr(L, f).argtypes = [vp]
        L.oracle_queue_layer_len.restype = C.c_uint64
        L.oracle_queue_layer_len.argtypes = [vp, C.c_uint64]
        L.oracle_astar_align.argtypes = [vp, C.c_uint8, C.c_uint8, C.c_uint8, C.c_int, C.c_int, vp, C.c_uint64, vp, vp, C.c_uint64, vp, vp]
        L.oracle_astar_batch.argtypes = [vp, C.c_uint8, C.c_uint8, C.c_uint8, C.c_int, C.c_int, C.c_uint32, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int]
        L.oracle_dense_align.argtypes = [vp, C.c_uint8, C.c_uint8, C.c_uint8, vp, C.c_uint64, vp, vp, C.c_uint64, vp, vp, vp, vp, vp]
        L.oracle_dense_batch.argtypes = [vp, C.c_uint8, C.c_uint8, C.c_uint8, C.c_uint32, vp, vp, vp, vp, vp, vp, vp, C.c_int]
        L.oracle_set_alignment_type.argtypes = [vp]
        L.oracle_set_two_piece.argtypes = [C.c_int, C.c_uint8, C.c_uint8]
        L.oracle_breakpoint.restype = C.c_uint64
        L.oracle_breakpoint.argtypes = [C.c_uint8] * 5
        L.oracle_dense_planes2.argtypes = [vp, C.c_uint8, C.c_uint8, C.c_uint8, vp, C.c_uint64, vp, vp]
        L.oracle_poa_to_fasta.restype = C.c_int64
        L.oracle_poa_to_fasta.argtypes = [vp, C.c_char_p, C.c_uint64]
        L.oracle_poa_from_msa.restype = vp
        L.oracle_poa_from_msa.argtypes = [C.c_char_p, C.c_char_p]
        L.oracle_is_end.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int]
        _lib = L
    return _lib


UNBOUNDED, INCLUDED, EXCLUDED = 0, 1, 2


class alignment_type:
    """Context manager: `with alignment_type(ends_free(...)):` — every search inside runs with that AlignmentType
    (scoring/mod.rs:50-62); Global outside.  A bound is UNBOUNDED or (INCLUDED | EXCLUDED, n)."""

    def __init__(self, spec):
        self.spec = spec

    def __enter__(self):
        lib().oracle_set_alignment_type(_p(self.spec))
        return self

    def __exit__(self, *exc):
        lib().oracle_set_alignment_type(None)
        return False


class two_piece:
    """Context manager: every search / dense pass inside runs the two-piece affine model
    GapAffine2Piece::new(mismatch, extend1, open1, extend2, open2) (gap_affine_2piece.rs:19-33); the Costs(mismatch, open, extend)
    passed to the calls are the first piece.  The reference asserts extend1 >= extend2 in the constructor: checked by the caller."""

    def __init__(self, gap_open2, gap_extend2):
        self.o2, self.e2 = gap_open2, gap_extend2

    def __enter__(self):
        lib().oracle_set_two_piece(1, self.o2, self.e2)
        return self

    def __exit__(self, *exc):
        lib().oracle_set_two_piece(0, 0, 0)
        return False


def breakpoint2(mismatch, extend1, open1, extend2, open2):
    """GapAffine2Piece::breakpoint (gap_affine_2piece.rs:36-66); argument order = the reference's constructor."""
    return int(lib().oracle_breakpoint(mismatch, open1, extend1, open2, extend2))


ST_D2, ST_I2 = 3, 4


def ends_free(qry_free_begin=UNBOUNDED, qry_free_end=UNBOUNDED, graph_free_begin=UNBOUNDED, graph_free_end=UNBOUNDED):
    spec = [1]
    for b in (qry_free_begin, qry_free_end, graph_free_begin, graph_free_end):
        spec += [b, 0] if isinstance(b, int) else [b[0], b[1]]
    return np.array(spec, np.uint64)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _seq(s):
    if isinstance(s, (bytes, bytearray)):
        return np.frombuffer(bytes(s), dtype=np.uint8)
    if isinstance(s, str):
        return np.frombuffer(s.encode(), dtype=np.uint8)
    return np.ascontiguousarray(s, dtype=np.uint8)


class Costs:
    """GapAffine; NB the reference constructor order is (mismatch, extend, open) — gap_affine.rs:27."""

    def __init__(self, mismatch=4, gap_open=6, gap_extend=2):
        self.mismatch, self.gap_open, self.gap_extend = mismatch, gap_open, gap_extend

    def t(self):
        return (self.mismatch, self.gap_open, self.gap_extend)


class OracleGraph:
    def __init__(self, handle):
        if not handle:
            raise RuntimeError(lib().oracle_last_error().decode())
        self.h = handle

    def __del__(self):
        try:
            if self.h:
                lib().oracle_graph_free(self.h)
                self.h = None
        except Exception:
            pass

    # -- constructors --
    @classmethod
    def new_poa(cls):
        return cls(lib().oracle_graph_new_poa())

    @classmethod
    def mock(cls, which):
        return cls(lib().oracle_graph_mock(which))

    @classmethod
    def mock_edges(cls, n, edges, symbols=None):
        e = np.ascontiguousarray(edges, dtype=np.uint32).reshape(-1, 2)
        s = _seq(symbols) if symbols is not None else None
        return cls(lib().oracle_graph_mock_edges(n, _p(e), len(e), _p(s)))

    @classmethod
    def from_csr(cls, csr, end_matches_all=True):
        """csr: dict with n,start,end,symbol,succ_off,succ,pred_off,pred (trait iteration order)."""
        a = {k: np.ascontiguousarray(csr[k], dtype=(np.uint8 if k == "symbol" else np.uint32))
             for k in ("symbol", "succ_off", "succ", "pred_off", "pred")}
        return cls(lib().oracle_graph_from_csr(int(csr["n"]), int(csr["start"]), int(csr["end"]), _p(a["symbol"]),
                                               _p(a["succ_off"]), _p(a["succ"]), _p(a["pred_off"]), _p(a["pred"]),
                                               1 if end_matches_all else 0))

    @classmethod
    def from_fasta_msa(cls, records):
        """load_graph_from_fasta_msa (src/io/graph.rs:36-103); records = [(name, gapped row), ...]."""
        names = "\n".join(n for n, _ in records).encode()
        rows = "\n".join(r for _, r in records).encode()
        return cls(lib().oracle_poa_from_msa(names, rows))

    def to_fasta(self):
        """poa_graph_to_fasta (src/io/fasta.rs:69-156) as text."""
        k = lib().oracle_poa_to_fasta(self.h, None, 0)
        if k < 0:
            raise RuntimeError(lib().oracle_last_error().decode())
        buf = C.create_string_buffer(int(k) + 1)
        lib().oracle_poa_to_fasta(self.h, buf, int(k) + 1)
        return buf.value.decode()

    # -- accessors --
    @property
    def n(self):
        return lib().oracle_graph_n(self.h)

    @property
    def start(self):
        return lib().oracle_graph_start(self.h)

    @property
    def end(self):
        return lib().oracle_graph_end(self.h)

    def set_symbols(self, symbols):
        s = _seq(symbols)
        assert len(s) == self.n
        lib().oracle_graph_set_symbols(self.h, _p(s))

    def export_csr(self):
        n, ne = self.n, lib().oracle_graph_n_edges(self.h)
        sym = np.zeros(n, np.uint8)
        so, po = np.zeros(n + 1, np.uint32), np.zeros(n + 1, np.uint32)
        s, p = np.zeros(max(ne, 1), np.uint32), np.zeros(max(ne, 1), np.uint32)
        rank = np.zeros(n, np.uint32)
        lib().oracle_graph_export(self.h, _p(sym), _p(so), _p(s), _p(po), _p(p), _p(rank))
        return dict(n=n, start=self.start, end=self.end, symbol=sym, succ_off=so, succ=s[:ne], pred_off=po,
                    pred=p[:ne], rank=rank)

    def add_alignment(self, name, seq, alignment=None):
        """POAGraph::add_alignment_with_weights (weights ignored: they do not affect topology)."""
        s = _seq(seq)
        if alignment is None:
            rc = lib().oracle_poa_add_alignment(self.h, name.encode(), _p(s), len(s), None, -1)
        else:
            a = np.ascontiguousarray(alignment, dtype=np.uint32).reshape(-1, 2)
            rc = lib().oracle_poa_add_alignment(self.h, name.encode(), _p(s), len(s), _p(a), len(a))
        if rc != 0:
            raise RuntimeError("InvalidAlignment" if rc == 1 else lib().oracle_last_error().decode())

    def seq_start_nodes(self):
        """Sequence(name, start_node) of every added sequence (poa.rs:21, :313-316)."""
        return [lib().oracle_poa_seq_start(self.h, i) for i in range(lib().oracle_poa_n_sequences(self.h))]

    def rev_postorder(self):
        out = np.zeros(self.n, np.uint32)
        k = lib().oracle_rev_postorder(self.h, _p(out))
        return out[:k].tolist()

    def superbubbles(self):
        out = np.zeros(2 * self.n, np.uint32)
        k = lib().oracle_superbubbles(self.h, _p(out))
        if k < 0:
            raise RuntimeError(lib().oracle_last_error().decode())
        return [tuple(x) for x in out[:2 * k].reshape(-1, 2).tolist()]

    def bubble_index(self):
        n = self.n
        cap = 64 * n + 64
        dmin, dmax = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
        off = np.zeros(n + 1, np.uint32)
        ex, mn, mx = np.zeros(cap, np.uint32), np.zeros(cap, np.uint64), np.zeros(cap, np.uint64)
        ent, exi = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
        k = lib().oracle_bubble_index(self.h, _p(dmin), _p(dmax), _p(off), _p(ex), _p(mn), _p(mx), cap, _p(ent), _p(exi))
        if k < 0:
            raise RuntimeError(lib().oracle_last_error().decode())
        nbm = [[(int(ex[i]), int(mn[i]), int(mx[i])) for i in range(off[v], off[v + 1])] for v in range(n)]
        return dict(dist_to_end=list(zip(dmin.tolist(), dmax.tolist())), node_bubble_map=nbm,
                    is_entrance=ent.astype(bool).tolist(), is_exit=exi.astype(bool).tolist())

    def heuristic_h(self, costs, heuristic, seq_len, node, offset, state):
        return lib().oracle_heuristic_h(self.h, *costs.t(), heuristic, seq_len, node, offset, state)

    def dfa_first_event(self, seq, node, offset, force_prune=False):
        s = _seq(seq)
        out = np.zeros(7, np.uint64)
        rc = lib().oracle_dfa_first_event(self.h, _p(s), len(s), node, offset, 1 if force_prune else 0, _p(out))
        if rc != 0:
            raise RuntimeError(lib().oracle_last_error().decode())
        kind = {0: None, 1: "RefGraphEnd", 2: "QueryEnd", 3: "Mismatch"}[int(out[0])]
        return dict(kind=kind, parent=(int(out[1]), int(out[2])), child=(int(out[3]), int(out[4])),
                    num_visited=int(out[5]), num_pruned=int(out[6]))

    def is_end(self, seq_len, node, offset, state=0):
        return bool(lib().oracle_is_end(self.h, seq_len, node, offset, state))

    # -- alignment --
    def astar_align(self, seq, costs=None, heuristic=H_MINGAP, prune=True):
        costs = costs or Costs()
        s = _seq(seq)
        cap = self.n + len(s) + 8
        pairs = np.zeros((cap, 2), np.uint32)
        score, npairs, counters = C.c_uint32(), C.c_uint64(), np.zeros(3, np.uint64)
        rc = lib().oracle_astar_align(self.h, *costs.t(), heuristic, 1 if prune else 0, _p(s), len(s), C.byref(score),
                                      _p(pairs), cap, C.byref(npairs), _p(counters))
        if rc == 1:
            raise RefPanic(lib().oracle_last_error().decode())
        if rc != 0:
            raise RuntimeError("oracle_astar_align rc=%d %s" % (rc, lib().oracle_last_error().decode()))
        return dict(score=score.value, alignment=[tuple(x) for x in pairs[:npairs.value].tolist()],
                    num_queued=int(counters[0]), num_visited=int(counters[1]), num_pruned=int(counters[2]))

    def dense_align(self, seq, costs=None, planes=False):
        costs = costs or Costs()
        s = _seq(seq)
        cap = self.n + len(s) + 8
        pairs = np.zeros((cap, 2), np.uint32)
        score, npairs, flags = C.c_uint32(), C.c_uint64(), C.c_uint32()
        pm = pi = pd = None
        if planes:
            shape = (self.n, len(s) + 1)
            pm, pi, pd = (np.zeros(shape, np.uint32) for _ in range(3))
        rc = lib().oracle_dense_align(self.h, *costs.t(), _p(s), len(s), C.byref(score), _p(pairs), cap, C.byref(npairs),
                                      C.byref(flags), _p(pm), _p(pi), _p(pd))
        if rc != 0:
            raise RuntimeError("oracle_dense_align rc=%d %s" % (rc, lib().oracle_last_error().decode()))
        out = dict(score=score.value, alignment=[tuple(x) for x in pairs[:npairs.value].tolist()], flags=flags.value)
        if planes:
            out.update(M=pm, I=pi, D=pd)
        return out

    def dense_planes2(self, seq, costs):
        """I2, D2 of the two-piece model (inside `with two_piece(...)`)."""
        s = _seq(seq)
        shape = (self.n, len(s) + 1)
        pi2, pd2 = np.zeros(shape, np.uint32), np.zeros(shape, np.uint32)
        if lib().oracle_dense_planes2(self.h, *costs.t(), _p(s), len(s), _p(pi2), _p(pd2)) != 0:
            raise RuntimeError(lib().oracle_last_error().decode())
        return pi2, pd2

    def _batch_bufs(self, qseq, qoff, want_pairs):
        qseq = np.ascontiguousarray(qseq, np.uint8)
        qoff = np.ascontiguousarray(qoff, np.uint64)
        nq = len(qoff) - 1
        lens = (qoff[1:] - qoff[:-1]).astype(np.uint64)
        pair_off = np.zeros(nq + 1, np.uint64)
        pair_off[1:] = np.cumsum(lens + np.uint64(self.n + 2))
        pairs = np.zeros((int(pair_off[-1]), 2), np.uint32) if want_pairs else None
        return qseq, qoff, nq, pair_off, pairs

    def astar_batch(self, qseq, qoff, costs=None, heuristic=H_MINGAP, prune=True, threads=1, want_pairs=True,
                    want_counters=False):
        costs = costs or Costs()
        qseq, qoff, nq, pair_off, pairs = self._batch_bufs(qseq, qoff, want_pairs)
        scores, npairs, status = np.zeros(nq, np.uint32), np.zeros(nq, np.uint64), np.zeros(nq, np.int32)
        counters = np.zeros((nq, 3), np.uint64) if want_counters else None
        rc = lib().oracle_astar_batch(self.h, *costs.t(), heuristic, 1 if prune else 0, nq, _p(qseq), _p(qoff), _p(scores),
                                      _p(pairs), _p(pair_off), _p(npairs), _p(counters), _p(status), threads)
        if rc != 0:
            raise RuntimeError(lib().oracle_last_error().decode())
        return dict(score=scores, pairs=pairs, pair_off=pair_off, n_pairs=npairs, status=status, counters=counters)

    def dense_batch(self, qseq, qoff, costs=None, threads=1, want_pairs=True):
        costs = costs or Costs()
        qseq, qoff, nq, pair_off, pairs = self._batch_bufs(qseq, qoff, want_pairs)
        scores, npairs, flags = np.zeros(nq, np.uint32), np.zeros(nq, np.uint64), np.zeros(nq, np.uint32)
        rc = lib().oracle_dense_batch(self.h, *costs.t(), nq, _p(qseq), _p(qoff), _p(scores), _p(pairs), _p(pair_off),
                                      _p(npairs), _p(flags), threads)
        if rc != 0:
            raise RuntimeError(lib().oracle_last_error().decode())
        return dict(score=scores, pairs=pairs, pair_off=pair_off, n_pairs=npairs, flags=flags)


def read_fasta(path):
    """[(name, sequence)] of a FASTA file (multi-line records joined; name = first word of the header)."""
    recs = []
    with open(path) as f:
        for line in f:
            line = line.rstrip("\n")
            if line.startswith(">"):
                recs.append([line[1:].split()[0] if line[1:].split() else "", ""])
            elif recs:
                recs[-1][1] += line.strip()
    return [(n, s) for n, s in recs]


def sequential_poa(records, costs=None, heuristic=H_MINGAP, prune=True, graph=None, alignments=None):
    """`poasta align` (src/bin/poasta.rs:163-236): the first read seeds the graph, every other read is aligned
    (Global) and added with add_alignment_with_weights.  Returns (graph, [score per aligned read]); `alignments`, a list,
    receives (name, seq, alignment-or-None) per read."""
    costs = costs or Costs()
    g = graph or OracleGraph.new_poa()
    scores = []
    for name, seq in records:
        if g.n == 2:  # graph.is_empty()
            g.add_alignment(name, seq, None)
            if alignments is not None:
                alignments.append((name, seq, None))
        else:
            r = g.astar_align(seq, costs, heuristic, prune)
            scores.append(r["score"])
            g.add_alignment(name, seq, r["alignment"])
            if alignments is not None:
                alignments.append((name, seq, r["alignment"]))
    return g, scores


class RefPanic(RuntimeError):
    """The reference would panic on this input (message names the panic site)."""


def batch_alignment(res, i):
    """i-th alignment of a batch result as a list of (rpos, qpos) tuples."""
    o, k = int(res["pair_off"][i]), int(res["n_pairs"][i])
    return [tuple(x) for x in res["pairs"][o:o + k].tolist()]


class LayeredQueue:
    def __init__(self):
        self.q = lib().oracle_queue_new()

    def __del__(self):
        try:
            lib().oracle_queue_free(self.q)
        except Exception:
            pass

    def queue(self, value, priority):
        lib().oracle_queue_push(self.q, value, priority)

    def pop(self):
        v = lib().oracle_queue_pop(self.q)
        return None if v < 0 else v

    @property
    def n_layers(self):
        return lib().oracle_queue_layers(self.q)

    @property
    def layer_min(self):
        return lib().oracle_queue_layer_min(self.q)

    def layer_len(self, ix):
        return lib().oracle_queue_layer_len(self.q, ix)


def gap_cost(costs, state, length):
    return lib().oracle_gap_cost(*costs.t(), state, length)
