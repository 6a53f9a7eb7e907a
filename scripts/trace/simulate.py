"""Prices step schedules for the wave replay on a logged search (analysis tool): how many pops a step can take when the
top entries of the current stack are expanded by different lanes at once and the step is cut at the first conflict."""
import sys, os, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import analyze as A
from poasta_amd import workloads as W

def simulate(pops, pushes, touches, max_lanes=64, overlay=False, gran="cell", max_desc=8, verbose=False):
    n = len(pops)
    f, st, g, row, j, oc = (pops[:, k].astype(np.int64) for k in range(6))
    first_push = pops[:, 7].astype(np.int64)
    # pusher of every pop
    key_of_push = {}
    pe = np.append(first_push[1:], len(pushes))
    # pushes logged before pop 0 (initial states) have pusher -1
    pusher_of_push = np.full(len(pushes), -1, np.int64)
    for k in range(n):
        pusher_of_push[first_push[k]:pe[k]] = k
    # NOTE first_push[k] is the push count BEFORE the pushes of pop k were logged? (p[7] = nq before log_pushes) yes
    for i, p in enumerate(pushes.tolist()):
        key_of_push[(p[1], p[3], p[4], p[2])] = pusher_of_push[i]
    pusher = np.array([key_of_push.get((int(st[k]), int(row[k]), int(j[k]), int(g[k])), -1) for k in range(n)])
    # footprints per pop
    tp = touches[:, 0].astype(np.int64)
    order = np.argsort(tp, kind="stable")
    tt = touches[order]
    bounds = np.searchsorted(tt[:, 0], np.arange(n + 1))
    def foot(k):
        rc, wc, rr, wm = set(), set(), [], set()
        rows_r, rows_w = set(), set()
        for t in tt[bounds[k]:bounds[k + 1]].tolist():
            _, kind, r, a, b = t
            if kind == 0: rc.add((r, a, b)); rows_r.add(r)
            elif kind == 1: wc.add((r, a, b)); rows_w.add(r)
            elif kind == 2: rr.append((r, a, b)); rows_r.add(r)
            else: wm.add((r, a)); rows_w.add(r)
        return rc, wc, rr, wm, rows_r, rows_w
    steps = 0; s = 0
    lanes_hist = collections.Counter(); cut_why = collections.Counter()
    pops_in_steps = 0
    while s < n:
        S = (f[s], st[s])
        # lanes: list of (list of pop indices)
        W_cells, W_marks, W_rows = set(), set(), set()
        k = s; lanes = 0; why = "end"
        while k < n and lanes < max_lanes:
            if (f[k], st[k]) != S:
                why = "stack"; break
            if pusher[k] >= s:
                why = "pushed_in_step"; break
            # this root and (overlay) its immediate descendants
            grp = [k]; m = k + 1
            while m < n and pusher[m] >= s: grp.append(m); m += 1
            has_desc = len(grp) > 1
            if has_desc and (not overlay or len(grp) - 1 > max_desc):
                if not overlay:
                    # the root itself can still be taken; the step ends after it
                    grp = [k]
                else:
                    why = "desc_cap"
                    if lanes == 0: grp = [k]  # sequential fallback takes the root alone
                    else: break
            # conflict of this lane with earlier lanes of the step
            conflict = False
            fr = [foot(q) for q in grp]
            for rc, wc, rr, wm, rows_r, rows_w in fr:
                if gran == "row":
                    if rows_r & W_rows: conflict = True
                elif gran == "cell":
                    if rc & W_cells: conflict = True
                    for (r, lo, hi) in rr:
                        for (mr, mo) in W_marks:
                            if mr == r and lo <= mo <= hi: conflict = True
                if conflict: break
            if conflict and lanes > 0:
                why = "conflict"; break
            for rc, wc, rr, wm, rows_r, rows_w in fr:
                W_cells |= wc; W_marks |= wm; W_rows |= rows_w
            lanes += 1; k = grp[-1] + 1
            if oc[grp[0]] == 3: why = "found"; break
            if has_desc and not overlay:
                why = "desc"; break
        steps += 1
        lanes_hist[lanes] += 1; cut_why[why] += 1
        pops_in_steps += k - s
        s = k
    return steps, n / steps, lanes_hist, cut_why

if __name__ == "__main__":
    nq = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    poa = W.LinearishPOA(seed=1)
    qs = poa.queries(nq, length=1000, seed=2)
    for q in qs:
        pops, pushes, out, info, touches = A.trace(poa.graph, q, touches=True)
        print(f"pops {len(pops)} touches {len(touches)}")
        for overlay in (False, True):
            for gran in ("row", "cell"):
                for ml in (16, 64):
                    steps, pps, lh, cw = simulate(pops, pushes, touches, max_lanes=ml, overlay=overlay, gran=gran)
                    print(f"  overlay={overlay!s:5} gran={gran:4} lanes<={ml:2}: steps {steps:6d}  pops/step {pps:5.1f}  cuts {dict(cw)}")
