"""Flattened view of the reference's `AlignableRefGraph` trait (src/graphs/mod.rs:23-53).

The aligner only ever sees a graph through that trait: node symbols, start/end sentinels and the
predecessor / successor iterators.  `FlatGraph` is the CSR hand-over format of the C ABI
(`include/poasta_amd.h: poa_graph_create`): adjacency is stored in TRAIT ITERATION ORDER, which for
the reference's petgraph-backed `POAGraph` is "most recently added edge first"
(src/graphs/poa.rs:433-440).  The order is semantic: it decides traceback ties
(src/aligner/scoring/gap_affine.rs:591,:617,:627).
"""
import numpy as np

START_SYMBOL = ord("#")  # src/graphs/poa.rs:102
END_SYMBOL = ord("$")    # src/graphs/poa.rs:103


class FlatGraph:
    """CSR graph: n nodes incl. start/end; succ/pred in trait iteration order."""

    def __init__(self, n, start, end, symbol, succ_off, succ, pred_off, pred):
        self.n = int(n)
        self.start = int(start)
        self.end = int(end)
        self.symbol = np.ascontiguousarray(symbol, dtype=np.uint8)
        self.succ_off = np.ascontiguousarray(succ_off, dtype=np.uint32)
        self.succ = np.ascontiguousarray(succ, dtype=np.uint32)
        self.pred_off = np.ascontiguousarray(pred_off, dtype=np.uint32)
        self.pred = np.ascontiguousarray(pred, dtype=np.uint32)
        assert len(self.symbol) == self.n and len(self.succ_off) == self.n + 1 and len(self.pred_off) == self.n + 1
        assert self.succ_off[-1] == len(self.succ) and self.pred_off[-1] == len(self.pred)

    @property
    def n_edges(self):
        return len(self.succ)

    def successors(self, v):
        return self.succ[self.succ_off[v]:self.succ_off[v + 1]]

    def predecessors(self, v):
        return self.pred[self.pred_off[v]:self.pred_off[v + 1]]

    def as_dict(self):
        return dict(n=self.n, start=self.start, end=self.end, symbol=self.symbol, succ_off=self.succ_off,
                    succ=self.succ, pred_off=self.pred_off, pred=self.pred)

    @classmethod
    def from_dict(cls, d):
        return cls(d["n"], d["start"], d["end"], d["symbol"], d["succ_off"], d["succ"], d["pred_off"], d["pred"])

    def save(self, path):
        np.savez_compressed(path, n=self.n, start=self.start, end=self.end, symbol=self.symbol, succ_off=self.succ_off,
                            succ=self.succ, pred_off=self.pred_off, pred=self.pred)

    @classmethod
    def load(cls, path):
        z = np.load(path, allow_pickle=False)
        return cls(int(z["n"]), int(z["start"]), int(z["end"]), z["symbol"], z["succ_off"], z["succ"], z["pred_off"], z["pred"])


class GraphBuilder:
    """Edge-by-edge builder with the reference's adjacency semantics.

    Mirrors what the host does when it builds a `POAGraph`: node 0 is the start sentinel '#', node 1
    the end sentinel '$' (src/graphs/poa.rs:100-112); `add_edge` keeps an existing edge where it is
    (poa.rs:118-134) and otherwise makes the new edge the FIRST one its endpoints' iterators yield;
    `finish()` does what `post_process` does (poa.rs:323-363): nodes without predecessors get an edge
    from start, nodes without successors get an edge to end, both scanned in ascending node index.
    """

    def __init__(self):
        self.symbol = [START_SYMBOL, END_SYMBOL]
        self.succ = [[], []]
        self.pred = [[], []]
        self.start, self.end = 0, 1

    def add_node(self, sym):
        self.symbol.append(int(sym))
        self.succ.append([])
        self.pred.append([])
        return len(self.symbol) - 1

    def add_edge(self, s, t):
        if t in self.succ[s]:
            return
        self.succ[s].insert(0, t)
        self.pred[t].insert(0, s)

    def add_path(self, seq):
        """Add a chain of new nodes for `seq` (poa.rs:136-169); returns the node ids."""
        ids = []
        for c in seq:
            v = self.add_node(c)
            if ids:
                self.add_edge(ids[-1], v)
            ids.append(v)
        return ids

    def finish(self):
        n = len(self.symbol)
        for v in list(self.succ[self.start]):
            self.pred[v].remove(self.start)
        self.succ[self.start] = []
        for v in list(self.pred[self.end]):
            self.succ[v].remove(self.end)
        self.pred[self.end] = []
        for v in range(n):
            if v not in (self.start, self.end) and not self.pred[v]:
                self.succ[self.start].insert(0, v)
                self.pred[v].insert(0, self.start)
        for v in range(n):
            if v not in (self.start, self.end) and not self.succ[v]:
                self.succ[v].insert(0, self.end)
                self.pred[self.end].insert(0, v)
        succ_off = np.zeros(n + 1, np.uint32)
        pred_off = np.zeros(n + 1, np.uint32)
        succ_off[1:] = np.cumsum([len(a) for a in self.succ])
        pred_off[1:] = np.cumsum([len(a) for a in self.pred])
        succ = np.array([t for a in self.succ for t in a], dtype=np.uint32)
        pred = np.array([t for a in self.pred for t in a], dtype=np.uint32)
        return FlatGraph(n, self.start, self.end, np.array(self.symbol, np.uint8), succ_off, succ, pred_off, pred)


def pack_queries(seqs):
    """Concatenate sequences -> (qseq u8[total], qoff u64[n+1]), the batch layout of the C ABI."""
    arrs = [np.frombuffer(s, dtype=np.uint8) if isinstance(s, (bytes, bytearray)) else
            (np.frombuffer(s.encode(), dtype=np.uint8) if isinstance(s, str) else np.asarray(s, dtype=np.uint8))
            for s in seqs]
    qoff = np.zeros(len(arrs) + 1, np.uint64)
    if arrs:
        qoff[1:] = np.cumsum([len(a) for a in arrs])
    qseq = np.concatenate(arrs).astype(np.uint8) if arrs and qoff[-1] > 0 else np.zeros(0, np.uint8)
    return np.ascontiguousarray(qseq), qoff
