"""Pop / push statistics of the replayed search on config-2 queries (analysis tool)."""
import ctypes as C, os, sys, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from poasta_amd import workloads as W

ROOT = os.path.dirname(os.path.abspath(__file__))
X = C.CDLL(os.path.join(ROOT, "libsearch_trace.so"))
vp = C.c_void_p
X.search_trace.argtypes = [C.c_uint32] * 3 + [vp] * 5 + [C.c_uint8] * 3 + [C.c_int, C.c_int, vp, C.c_uint32, vp, C.c_uint64, vp, C.c_uint64, vp, vp, vp, C.c_uint64]
_p = lambda a: a.ctypes.data_as(vp)

def trace(g, q, costs=(4, 6, 2), heur=1, prune=1, touches=False):
    q = np.ascontiguousarray(q, np.uint8)
    cap = 2_000_000
    pops = np.zeros((cap, 8), np.uint32); pushes = np.zeros((cap, 5), np.uint32); out = np.zeros(5, np.uint64)
    tcap = 20_000_000 if touches else 0
    tch = np.zeros((max(tcap, 1), 5), np.uint32)
    info = np.zeros((g.n, 4), np.uint32)
    rc = X.search_trace(g.n, g.start, g.end, _p(g.symbol), _p(g.succ_off), _p(g.succ), _p(g.pred_off), _p(g.pred), *costs, heur, prune,
                        _p(q), len(q), _p(pops), cap, _p(pushes), cap, _p(out), _p(info), _p(tch) if touches else None, tcap)
    assert rc == 0
    if touches:
        return pops[:out[0]], pushes[:out[1]], out, info, tch[:out[4]]
    return pops[:out[0]], pushes[:out[1]], out, info

if __name__ == "__main__":
    nq = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    length = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    poa = W.LinearishPOA(seed=1)
    qs = poa.queries(nq, length=length, seed=2)
    for q in qs:
        pops, pushes, out, info = trace(poa.graph, q)
        f, st, g, row, j, oc, dv = (pops[:, k].astype(np.int64) for k in range(7))
        print(f"L={len(q)} score={out[2]} pops={len(pops)} pushes={len(pushes)} buckets={f.max()-f.min()+1} dfa_visited={dv.sum()}")
        for s, nm in enumerate("MDI"):
            m = st == s
            print(f"  {nm}: pops {m.sum():6d}  expanded {(m&(oc==0)).sum():6d} stale {(m&(oc==1)).sum():6d} pruned {(m&(oc==2)).sum():6d}")
        # runs: consecutive expanded pops of the same state where each is the child (same row, j+1 for I; succ row, same j for D) of the previous
        runs = collections.Counter()
        k = 0; n = len(pops)
        while k < n:
            if oc[k] != 0: k += 1; continue
            s = st[k]; r = 1
            while k + r < n and st[k + r] == s and oc[k + r] == 0 and f[k + r] == f[k] and (
                (s == 2 and row[k + r] == row[k + r - 1] and j[k + r] == j[k + r - 1] + 1) or
                (s == 1 and j[k + r] == j[k + r - 1] and row[k + r] != row[k + r - 1])): r += 1
            runs[(int(s), min(r, 64))] += 1
            k += r
        for s, nm in enumerate("MDI"):
            tot = sum(c for (ss, r), c in runs.items() if ss == s); cells = sum(c * r for (ss, r), c in runs.items() if ss == s)
            print(f"  {nm} runs: {tot} covering {cells} expanded pops; length histogram:", sorted((r, c) for (ss, r), c in runs.items() if ss == s)[:12], "...")
        # steps if: leading stale/pruned popped with the first expanded (now), and a whole run per step
        steps_now = (oc == 0).sum() + 0
        steps_run = sum(runs.values())
        print(f"  expanded pops (= steps today, lower bound) {steps_now}; steps with whole runs {steps_run}")
        print(f"  dfa matches per expanded M pop: {dv[(st==0)&(oc==0)].sum() / max(1,((st==0)&(oc==0)).sum()):.2f}")
        # entries per bucket
        fb = collections.Counter(f.tolist())
        print(f"  pops per bucket: mean {np.mean(list(fb.values())):.1f} max {max(fb.values())}")

def dump(qi=0, lo=20000, n=80):
    poa = W.LinearishPOA(seed=1)
    q = poa.queries(qi + 1, length=1000, seed=2)[qi]
    pops, pushes, out, info = trace(poa.graph, q)
    for k in range(lo, lo + n):
        f, st, g, row, j, oc, dv, pi = pops[k].tolist()
        pe = pops[k + 1][7] if k + 1 < len(pops) else len(pushes)
        ps = " ".join(f"{'MDI'[p[1]]}({p[3]},{p[4]})g{p[2]}@{p[0]}" for p in pushes[pi:pe].tolist())
        print(f"{k:6d} f={f} {'MDI'[st]}({row},{j}) g={g} {['EXP','stale','PRUNED','END'][oc]} dfa={dv} nsucc={info[row][0]} -> {ps}")
