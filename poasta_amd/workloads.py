"""Seeded synthetic workloads for BASELINE.json's configs.

All generators are deterministic functions of their seed (numpy PCG64).  Graphs are produced in
trait-iteration order through `GraphBuilder` ("newest edge first", as the reference host would
produce them); queries are source-to-sink walks with substitution / insertion / deletion errors.
"""
import numpy as np

from .graph import FlatGraph, GraphBuilder, pack_queries

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def _rng(seed):
    return np.random.Generator(np.random.PCG64(int(seed)))


def _other_base(rng, base):
    """A uniformly random base different from `base` (array ok)."""
    base = np.asarray(base, dtype=np.uint8)
    idx = np.searchsorted(ACGT, base)
    shift = rng.integers(1, 4, size=base.shape)
    return ACGT[(idx + shift) % 4]


def mutate(rng, seq, p_sub, p_ins, p_del):
    """Per-base errors: substitution, insertion (a random base BEFORE the base), deletion."""
    seq = np.asarray(seq, dtype=np.uint8)
    r = rng.random(len(seq))
    out = seq.copy()
    sub = r < p_sub
    if sub.any():
        out[sub] = _other_base(rng, seq[sub])
    ins = (r >= p_sub) & (r < p_sub + p_ins)
    dele = (r >= p_sub + p_ins) & (r < p_sub + p_ins + p_del)
    counts = np.ones(len(seq), dtype=np.int64)
    counts[ins] = 2
    counts[dele] = 0
    res = np.repeat(out, counts)
    if ins.any():
        # first copy of every doubled base becomes the inserted random base
        starts = np.cumsum(counts) - counts
        pos = starts[ins]
        res[pos] = ACGT[rng.integers(0, 4, size=len(pos))]
    return res


def fit_length(rng, seq, length):
    """Trim or pad (random bases) to exactly `length`."""
    if len(seq) >= length:
        return seq[:length]
    pad = ACGT[rng.integers(0, 4, size=length - len(seq))]
    return np.concatenate([seq, pad])


# ---------------------------------------------------------------------------------------------
# config 2 / 3: 1000-node linear-ish POA, queries x 1 kbp
class LinearishPOA:
    """Backbone of `n_backbone` random bases + `n_snp` one-node SNP bubbles + `n_ins` two-node
    insertion branches (900 + 50 + 25*2 = 1000 real nodes)."""

    def __init__(self, n_backbone=900, n_snp=50, n_ins=25, seed=1):
        rng = _rng(seed)
        self.backbone = ACGT[rng.integers(0, 4, size=n_backbone)]
        # SNPs need both neighbours; insertion branches hang between p and p+1
        pos = rng.permutation(np.arange(1, n_backbone - 1))
        self.snp_pos = np.sort(pos[:n_snp])
        self.ins_pos = np.sort(pos[n_snp:n_snp + n_ins])
        self.snp_base = _other_base(rng, self.backbone[self.snp_pos])
        self.ins_bases = ACGT[rng.integers(0, 4, size=(n_ins, 2))]
        b = GraphBuilder()
        ids = b.add_path(self.backbone)
        self.backbone_ids = np.array(ids)
        self.snp_ids = []
        for p, base in zip(self.snp_pos, self.snp_base):
            v = b.add_node(base)
            b.add_edge(ids[p - 1], v)
            b.add_edge(v, ids[p + 1])
            self.snp_ids.append(v)
        self.ins_ids = []
        for p, bases in zip(self.ins_pos, self.ins_bases):
            v1, v2 = b.add_node(bases[0]), b.add_node(bases[1])
            b.add_edge(ids[p], v1)
            b.add_edge(v1, v2)
            b.add_edge(v2, ids[p + 1])
            self.ins_ids.append((v1, v2))
        self.graph = b.finish()

    def walk(self, rng):
        """Uniform random source-to-sink walk (each bubble is a fair coin)."""
        seq = self.backbone.copy()
        take = rng.random(len(self.snp_pos)) < 0.5
        seq[self.snp_pos[take]] = self.snp_base[take]
        take_i = rng.random(len(self.ins_pos)) < 0.5
        if take_i.any():
            p = self.ins_pos[take_i]
            seq = np.insert(seq, np.repeat(p + 1, 2), self.ins_bases[take_i].reshape(-1))
        return seq

    def queries(self, n, length=1000, seed=2, first=0, p_sub=0.02, p_ins=0.01, p_del=0.01):
        """Queries first .. first+n-1; query i depends only on (seed, i) so shards agree."""
        out = []
        for i in range(first, first + n):
            rng = _rng((seed << 32) + i)
            s = mutate(rng, self.walk(rng), p_sub, p_ins, p_del)
            out.append(fit_length(rng, s, length) if length else s)
        return out


def config2(n_queries=10000, length=1000, first=0, graph_seed=1, query_seed=2):
    poa = LinearishPOA(seed=graph_seed)
    qs = poa.queries(n_queries, length=length, seed=query_seed, first=first)
    return poa.graph, pack_queries(qs)


def scaled_linearish(n_backbone, n_snp, n_ins, n_queries, length, graph_seed=1, query_seed=2, first=0, **err):
    """Same family as config 2 at another size (parity-test shapes)."""
    poa = LinearishPOA(n_backbone, n_snp, n_ins, seed=graph_seed)
    qs = poa.queries(n_queries, length=length, seed=query_seed, first=first, **err)
    return poa.graph, pack_queries(qs)


# ---------------------------------------------------------------------------------------------
# config 5: deep bubble-rich graph (layers of `width` nodes, mean in-degree `indeg`)
class LayeredPOA:
    def __init__(self, n_layers=5000, width=4, indeg=4, seed=5):
        rng = _rng(seed)
        self.n_layers, self.width = n_layers, width
        b = GraphBuilder()
        self.sym = ACGT[rng.integers(0, 4, size=(n_layers, width))]
        self.ids = np.zeros((n_layers, width), dtype=np.int64)
        for l in range(n_layers):
            for w in range(width):
                self.ids[l, w] = b.add_node(self.sym[l, w])
        k = min(indeg, width)
        self.preds = np.zeros((n_layers, width, k), dtype=np.int64)
        for l in range(1, n_layers):
            for w in range(width):
                ps = rng.permutation(width)[:k]
                self.preds[l, w] = ps
                for p in ps:
                    b.add_edge(int(self.ids[l - 1, p]), int(self.ids[l, w]))
        # make sure every node of layer l-1 has a successor (otherwise it would link to end)
        for l in range(1, n_layers):
            used = set(self.preds[l].reshape(-1).tolist())
            for p in range(width):
                if p not in used:
                    b.add_edge(int(self.ids[l - 1, p]), int(self.ids[l, int(rng.integers(0, width))]))
        self.builder_succ = [list(a) for a in b.succ]
        self.graph = b.finish()

    def walk(self, rng):
        g = self.graph
        v = int(g.successors(g.start)[int(rng.integers(0, len(g.successors(g.start))))])
        out = []
        while v != g.end:
            out.append(g.symbol[v])
            s = g.successors(v)
            v = int(s[int(rng.integers(0, len(s)))])
        return np.array(out, dtype=np.uint8)

    def queries(self, n, length=5000, seed=6, first=0, p_err=0.03):
        out = []
        for i in range(first, first + n):
            rng = _rng((seed << 32) + i)
            s = mutate(rng, self.walk(rng), p_err / 3, p_err / 3, p_err / 3)
            out.append(fit_length(rng, s, length) if length else s)
        return out


def config5(n_queries=2000, n_layers=5000, width=4, indeg=4, length=5000, first=0):
    poa = LayeredPOA(n_layers, width, indeg)
    return poa.graph, pack_queries(poa.queries(n_queries, length=length, first=first))


# ---------------------------------------------------------------------------------------------
# config 4: pangenome-style POA imported from a columnar MSA
def msa_to_graph(rows):
    """The reference's MSA import rule (src/io/graph.rs:36-103): per column one node per distinct
    symbol, '-' skipped, each row threads an edge from its previous node; rows in order."""
    rows = [np.frombuffer(r, dtype=np.uint8) if isinstance(r, (bytes, bytearray)) else np.asarray(r, np.uint8) for r in rows]
    ncol = len(rows[0])
    b = GraphBuilder()
    col_nodes = [dict() for _ in range(ncol)]
    for r in rows:
        prev = None
        for c in range(ncol):
            s = int(r[c])
            if s == ord("-"):
                continue
            v = col_nodes[c].get(s)
            if v is None:
                v = b.add_node(s)
                col_nodes[c][s] = v
            if prev is not None:
                b.add_edge(prev, v)
            prev = v
    return b.finish()


class PangenomePOA:
    def __init__(self, ref_len=50000, n_hap=32, p_snp=0.001, p_indel=0.0002, max_indel=50, seed=4):
        rng = _rng(seed)
        ref = ACGT[rng.integers(0, 4, size=ref_len)]
        # build haplotypes as edit scripts against the reference, then a columnar MSA
        ins_at = {}  # ref position -> max inserted length over haplotypes (columns to reserve)
        haps = []
        for _ in range(n_hap):
            r = rng.random(ref_len)
            snp = r < p_snp
            indel = (r >= p_snp) & (r < p_snp + p_indel)
            is_ins = rng.random(ref_len) < 0.5
            lens = rng.integers(1, max_indel + 1, size=ref_len)
            h = dict(snp=np.flatnonzero(snp), snp_base=_other_base(rng, ref[snp]), ins={}, dele=[])
            for p in np.flatnonzero(indel):
                if is_ins[p]:
                    h["ins"][int(p)] = ACGT[rng.integers(0, 4, size=int(lens[p]))]
                    ins_at[int(p)] = max(ins_at.get(int(p), 0), int(lens[p]))
                else:
                    h["dele"].append((int(p), int(min(lens[p], ref_len - p))))
            haps.append(h)
        # column layout: for each ref position p: [inserted columns before p] + [p]
        extra = np.zeros(ref_len + 1, dtype=np.int64)
        for p, k in ins_at.items():
            extra[p] = k
        col_of = np.arange(ref_len) + np.cumsum(extra[:ref_len])
        ncol = int(ref_len + extra[:ref_len].sum())
        self.rows = []
        self.hap_seqs = []
        for h in haps:
            row = np.full(ncol, ord("-"), dtype=np.uint8)
            s = ref.copy()
            s[h["snp"]] = h["snp_base"]
            keep = np.ones(ref_len, dtype=bool)
            for p, k in h["dele"]:
                keep[p:p + k] = False
            row[col_of[keep]] = s[keep]
            for p, bases in h["ins"].items():
                c0 = col_of[p] - extra[p]
                row[c0:c0 + len(bases)] = bases
            self.rows.append(row)
            self.hap_seqs.append(row[row != ord("-")])
        self.graph = msa_to_graph(self.rows)

    def queries(self, n, length=10000, seed=7, first=0, p_err=0.05):
        out = []
        for i in range(first, first + n):
            rng = _rng((seed << 32) + i)
            h = self.hap_seqs[int(rng.integers(0, len(self.hap_seqs)))]
            a = int(rng.integers(0, max(1, len(h) - length)))
            s = mutate(rng, h[a:a + length], 0.3 * p_err, 0.3 * p_err, 0.4 * p_err)
            out.append(fit_length(rng, s, length) if length else s)
        return out


def config4(n_queries=5000, ref_len=50000, n_hap=32, length=10000, first=0):
    poa = PangenomePOA(ref_len, n_hap)
    return poa.graph, pack_queries(poa.queries(n_queries, length=length, first=first))


# ---------------------------------------------------------------------------------------------
def random_dag(seed, n_nodes=12, p_edge=0.25, alphabet=b"ACGT"):
    """Small random DAG (property tests): nodes 2..n+1 in topological id order, random forward
    edges added in random order so that iteration order is arbitrary."""
    rng = _rng(seed)
    alpha = np.frombuffer(alphabet, dtype=np.uint8)
    b = GraphBuilder()
    ids = [b.add_node(alpha[int(rng.integers(0, len(alpha)))]) for _ in range(n_nodes)]
    edges = [(i, j) for i in range(n_nodes) for j in range(i + 1, n_nodes)
             if (j == i + 1 and rng.random() < 0.7) or rng.random() < p_edge / max(1, (j - i))]
    for k in rng.permutation(len(edges)):
        i, j = edges[int(k)]
        b.add_edge(ids[i], ids[j])
    return b.finish()


def random_walk_query(rng, g, p_err=0.15, alphabet=b"ACGT"):
    alpha = np.frombuffer(alphabet, dtype=np.uint8)
    s = g.successors(g.start)
    v = int(s[int(rng.integers(0, len(s)))])
    out = []
    while v != g.end:
        out.append(g.symbol[v])
        s = g.successors(v)
        v = int(s[int(rng.integers(0, len(s)))])
    seq = np.array(out, dtype=np.uint8)
    r = rng.random(len(seq))
    res = []
    for c, x in zip(seq, r):
        if x < p_err / 3:
            res.append(alpha[int(rng.integers(0, len(alpha)))])
        elif x < 2 * p_err / 3:
            res.append(alpha[int(rng.integers(0, len(alpha)))])
            res.append(c)
        elif x < p_err:
            continue
        else:
            res.append(c)
    return np.array(res, dtype=np.uint8)
