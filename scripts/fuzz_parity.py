#!/usr/bin/env python3
"""Randomised differential run (GPU box): the HIP path against the oracle on random graphs, costs and query lengths —
dense mode vs the oracle's dense restatement (scores, alignments, flags), exact mode vs the oracle's A*.
Test infrastructure; prints one JSON line; non-zero exit on the first difference."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O
from poasta_amd import aligner as E, workloads as W
from poasta_amd.graph import pack_queries

ap = argparse.ArgumentParser()
ap.add_argument("--seeds", type=int, default=200)
ap.add_argument("--first", type=int, default=0)
ap.add_argument("--seconds", type=float, default=240.0)
ap.add_argument("--verbose", action="store_true")
args = ap.parse_args()
t0 = time.time()
n_dense = n_exact = n_exact2 = n_cases = 0
for seed in range(args.first, args.first + args.seeds):
    if time.time() - t0 > args.seconds:
        break
    rng = np.random.Generator(np.random.PCG64(77000 + seed))
    kind = seed % 8
    span = None
    if args.verbose:
        print("seed", seed, "kind", kind, file=sys.stderr, flush=True)
    if kind == 0:
        alpha = b"AC" if seed % 8 else b"ACGT"
        g = W.random_dag(seed, n_nodes=int(rng.integers(3, 40)), p_edge=float(rng.uniform(0.1, 0.4)), alphabet=alpha)
        qs = [W.random_walk_query(rng, g, float(rng.uniform(0.05, 0.5)), alpha) for _ in range(12)]
    elif kind == 1:
        nb = int(rng.integers(200, 1100))
        L = int(rng.integers(max(64, nb - 300), nb + 150))
        g, (qseq, qoff) = W.scaled_linearish(nb, int(nb * 0.05), int(nb * 0.025), 6, L, graph_seed=seed, query_seed=seed + 1,
                                             p_sub=float(rng.uniform(0.01, 0.1)), p_ins=0.02, p_del=0.02)
        qs = [qseq[int(qoff[i]):int(qoff[i + 1])] for i in range(6)]
        qs += [qs[0][:int(rng.integers(1, len(qs[0])))], qs[1][:int(rng.integers(1, len(qs[1])))]]
    elif kind == 2:
        poa = W.LayeredPOA(n_layers=int(rng.integers(20, 400)), width=int(rng.integers(2, 5)), indeg=int(rng.integers(1, 4)), seed=seed)
        g = poa.graph
        qs = poa.queries(6, length=0, seed=seed + 3)
    elif kind == 3:
        pg = W.PangenomePOA(ref_len=int(rng.integers(100, 900)), n_hap=int(rng.integers(2, 8)), p_snp=0.02, p_indel=0.01, max_indel=6, seed=seed)
        g = pg.graph
        qs = pg.queries(6, length=int(rng.integers(50, 700)), seed=seed + 5)
    elif kind == 4:
        # long queries: several strips, the multi-wave pipeline (mixed with short ones in the same launch)
        nb = int(rng.integers(150, 500))
        L = int(rng.integers(1100, 3200))
        g, (qseq, qoff) = W.scaled_linearish(nb, int(nb * 0.05), int(nb * 0.025), 4, L, graph_seed=seed, query_seed=seed + 1,
                                             p_sub=0.03, p_ins=0.02, p_del=0.02)
        qs = [qseq[int(qoff[i]):int(qoff[i + 1])] for i in range(4)]
        qs += [qs[0][:int(rng.integers(1, 700))], qs[1][:int(rng.integers(1025, L))]]
    elif kind == 5:
        # scores beyond u16: u32 planes
        nb = int(rng.integers(300, 900))
        g, (qseq, qoff) = W.scaled_linearish(nb, int(nb * 0.05), int(nb * 0.025), 4, int(rng.integers(100, 1400)), graph_seed=seed,
                                             query_seed=seed + 1, p_sub=0.05, p_ins=0.02, p_del=0.02)
        qs = [qseq[int(qoff[i]):int(qoff[i + 1])] for i in range(4)]
    elif kind == 7:
        # a read against a much longer graph, Global: scores beyond u16 in 2-byte cells (relative encoding) or u32 planes,
        # as the bounds decide; bypass edges and inserted branches make the depth potential's edge terms non-zero
        nb = int(rng.integers(30000, 42000))
        from poasta_amd.graph import GraphBuilder
        b = GraphBuilder()
        acgt = np.frombuffer(b"ACGT", np.uint8)
        sym = acgt[rng.integers(0, 4, nb)]
        ids = b.add_path(sym)
        for _ in range(int(rng.integers(0, 6))):
            a = int(rng.integers(0, nb - 500)); b.add_edge(ids[a], ids[a + 2 + int(rng.integers(0, 300))])
        for _ in range(int(rng.integers(0, 30))):
            a = int(rng.integers(0, nb - 200)); skip = int(rng.integers(1, 30)); prev = ids[a]
            for c in acgt[rng.integers(0, 4, skip + int(rng.integers(1, 60)))]:
                v = b.add_node(int(c)); b.add_edge(prev, v); prev = v
            b.add_edge(prev, ids[a + skip + 1])
        for i in rng.choice(np.arange(1, nb - 1), 200, replace=False):
            v = b.add_node(int(acgt[rng.integers(0, 4)])); b.add_edge(ids[i - 1], v); b.add_edge(v, ids[i + 1])
        g = b.finish()
        qs = []
        for _ in range(4):
            Lq = int(rng.integers(30, 2600)); a = int(rng.integers(0, nb - Lq))
            qs.append(W.mutate(rng, sym[a:a + Lq].copy(), 0.05, 0.03, 0.03))
    else:
        # ends-free spans (replayed): random bounds
        alpha = b"ACGT"
        g = W.random_dag(seed, n_nodes=int(rng.integers(3, 30)), p_edge=float(rng.uniform(0.1, 0.4)), alphabet=alpha)
        qs = [W.random_walk_query(rng, g, float(rng.uniform(0.05, 0.5)), alpha) for _ in range(10)]
        def bound():
            k = int(rng.integers(0, 3))
            # (Excluded(0) can never be satisfied: the reference's search then never ends — nothing to compare)
            return (k, int(rng.integers(1 if k == 2 else 0, 5))) if k else (0, 0)
        span = dict(qry_free_end=bound(), graph_free_begin=bound(), graph_free_end=bound())
    costs = (int(rng.integers(1, 10)), int(rng.integers(0, 13)), int(rng.integers(1, 5)))
    if kind == 5:
        costs = (int(rng.integers(100, 256)), int(rng.integers(100, 256)), int(rng.integers(60, 256)))
    qseq, qoff = pack_queries(qs)
    og = O.OracleGraph.from_csr(g.as_dict())
    oc = O.Costs(*costs)
    m, o, e = costs
    al = E.PoastaAligner(E.AffineMinGapCost(E.GapAffine(m, e, o)))
    res = al.align_batch(g, qseq=qseq, qoff=qoff) if span is None else None
    D = og.dense_batch(qseq, qoff, oc, threads=8) if span is None else None
    for i in range(len(qs) if span is None else 0):
        ok = int(res.score[i]) == int(D["score"][i]) and res.raw_alignment(i) == O.batch_alignment(D, i) and int(res.flags[i]) == int(D["flags"][i])
        if not ok:
            print(json.dumps(dict(fail="dense", seed=seed, kind=kind, query=i, costs=costs, gpu=[int(res.score[i]), int(res.flags[i])],
                                  oracle=[int(D["score"][i]), int(D["flags"][i])])))
            sys.exit(1)
        n_dense += 1
    if g.n * max(len(q) for q in qs) < 400000 and kind != 5:
        heur, prune = (O.H_MINGAP, True) if seed % 3 else (O.H_DIJKSTRA, seed % 2 == 0)
        cfgc = E.AffineMinGapCost if heur == O.H_MINGAP else E.AffineDijkstra
        if span is None:
            ax = E.PoastaAligner(cfgc(E.GapAffine(m, e, o)), mode="hybrid" if seed % 5 == 0 else "exact", queue_entries_per_cell=12.0)
            rx = ax.align_batch(g, qseq=qseq, qoff=qoff, pruning=prune)
            A = og.astar_batch(qseq, qoff, oc, heur, prune, threads=8)
        else:
            B = E.Bound
            conv = lambda b: B.Unbounded if b[0] == 0 else (B.Included(b[1]) if b[0] == 1 else B.Excluded(b[1]))
            at = E.AlignmentType.EndsFree(qry_free_end=conv(span["qry_free_end"]), graph_free_begin=conv(span["graph_free_begin"]),
                                          graph_free_end=conv(span["graph_free_end"]))
            ax = E.PoastaAligner(cfgc(E.GapAffine(m, e, o)), aln_type=at, queue_entries_per_cell=12.0)
            rx = ax.align_batch(g, qseq=qseq, qoff=qoff, pruning=prune)
            ob = lambda b: b if b[0] else 0
            with O.alignment_type(O.ends_free(0, ob(span["qry_free_end"]), ob(span["graph_free_begin"]), ob(span["graph_free_end"]))):
                A = og.astar_batch(qseq, qoff, oc, heur, prune, threads=8)
        for i in range(len(qs)):
            if int(rx.flags[i]) & 0x40:
                continue  # replay workspace overflow: flagged, the dense result was kept — nothing to compare
            if A["status"][i] != 0:
                ok = bool(int(rx.flags[i]) & 4)
            else:
                ok = int(rx.flags[i]) & ~0x10 == 0 and int(rx.score[i]) == int(A["score"][i]) and rx.raw_alignment(i) == O.batch_alignment(A, i)
            if not ok:
                print(json.dumps(dict(fail="exact", seed=seed, kind=kind, query=i, costs=costs, heur=heur, prune=prune,
                                      gpu=[int(rx.score[i]), int(rx.flags[i])], oracle=[int(A["status"][i]), int(A["score"][i])])))
                sys.exit(1)
            n_exact += 1
    if g.n * max(len(q) for q in qs) < 150000 and kind != 5:
        # two-piece model: the replay of the five-state search (poa_align_batch_2piece_ex), random second piece
        e2, o2 = int(rng.integers(0, e + 1)), int(rng.integers(0, 30))
        heur, prune = (O.H_MINGAP, True) if seed % 2 else (O.H_DIJKSTRA, seed % 4 == 0)
        cfg2 = E.Affine2PieceMinGapCost if heur == O.H_MINGAP else E.Affine2PieceDijkstra
        if span is None:
            at, ospec = E.AlignmentType.Global, None
        else:
            B = E.Bound
            conv = lambda b: B.Unbounded if b[0] == 0 else (B.Included(b[1]) if b[0] == 1 else B.Excluded(b[1]))
            at = E.AlignmentType.EndsFree(qry_free_end=conv(span["qry_free_end"]), graph_free_begin=conv(span["graph_free_begin"]),
                                          graph_free_end=conv(span["graph_free_end"]))
            ob = lambda b: b if b[0] else 0
            ospec = O.ends_free(0, ob(span["qry_free_end"]), ob(span["graph_free_begin"]), ob(span["graph_free_end"]))
        r2 = E.PoastaAligner(cfg2(E.GapAffine2Piece(m, e, o, e2, o2)), aln_type=at, mode="exact", queue_entries_per_cell=16.0).align_batch(
            g, qseq=qseq, qoff=qoff, pruning=prune)
        with O.two_piece(o2, e2):
            if ospec is None:
                A2 = og.astar_batch(qseq, qoff, oc, heur, prune, threads=8, want_counters=True)
            else:
                with O.alignment_type(ospec):
                    A2 = og.astar_batch(qseq, qoff, oc, heur, prune, threads=8, want_counters=True)
        for i in range(len(qs)):
            if int(r2.flags[i]) & 0x40:
                continue
            if A2["status"][i] != 0:
                ok = bool(int(r2.flags[i]) & (4 | 0x10))
            else:
                ok = (int(r2.flags[i]) & ~0x10 == 0 and int(r2.score[i]) == int(A2["score"][i]) and r2.raw_alignment(i) == O.batch_alignment(A2, i)
                      and r2.search_counters[i, :3].tolist() == [int(x) for x in A2["counters"][i]])
            if not ok:
                print(json.dumps(dict(fail="two-piece exact", seed=seed, kind=kind, query=i, costs=costs + (o2, e2), heur=heur, prune=prune,
                                      gpu=[int(r2.score[i]), int(r2.flags[i])], oracle=[int(A2["status"][i]), int(A2["score"][i])])))
                sys.exit(1)
            n_exact2 += 1
    n_cases += 1
    if n_cases % 100 == 0:  # keeps a long run visibly alive
        print(json.dumps(dict(progress=n_cases, seconds=round(time.time() - t0, 1))), flush=True)
print(json.dumps(dict(ok=True, graphs=n_cases, dense_queries=n_dense, exact_queries=n_exact, two_piece_exact_queries=n_exact2, seconds=round(time.time() - t0, 1))))
