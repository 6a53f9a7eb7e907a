#!/bin/bash
# Round-3 closing evidence (GPU box): the replay kernels after the scratch-memory fix (timings + rocprof stats + PMC of the default
# wave kernel), the two-piece replay (timing, kernel stats, PMC), the bench line.  Outputs under gpurun_out/r03f/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out/r03f; rm -rf $O; mkdir -p $O
cd $ROOT
T="timeout -k 10 200 python scripts/exact_timing.py --queries 10000 --mode hybrid --check 0 --reps 2"
$T > $O/wave_timing.log 2>&1; echo "wave rc=$?"
$T --length 0 > $O/wave_unpadded_timing.log 2>&1; echo "wave unpadded rc=$?"
POA_EXACT_IMPL=flat POA_PS_LEAN=1 $T > $O/flat_lean_timing.log 2>&1; echo "flat lean rc=$?"
POA_EXACT_IMPL=flat POA_PS_LEAN=0 $T > $O/flat_generic_timing.log 2>&1; echo "flat generic rc=$?"
timeout -k 10 300 python scripts/two_piece_exact_timing.py --queries 6000 --check 16 --reps 1 > $O/two_piece_exact_timing.log 2>&1; echo "2piece exact rc=$?"
timeout -k 10 300 python scripts/two_piece_exact_timing.py --queries 2000 --heuristic dijkstra --check 8 --reps 1 > $O/two_piece_exact_dijkstra_timing.log 2>&1; echo "2piece exact dijkstra rc=$?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/wave_trace -- python3 $ROOT/scripts/exact_timing.py --queries 10000 --mode hybrid --check 0 --reps 1 > $O/wave_trace.log 2>&1; echo "wave trace rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tp_trace -- python3 $ROOT/scripts/two_piece_exact_timing.py --queries 2000 --reps 1 > $O/tp_trace.log 2>&1; echo "2piece trace rc=$?"
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_IFETCH SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $C --output-format csv -d $O/wave_pmc_$N -- python3 $ROOT/scripts/exact_timing.py --queries 10000 --mode hybrid --check 0 --reps 1 > $O/wave_pmc_$N.log 2>&1; echo "wave pmc $N rc=$?"
  rocprofv3 --pmc $C --output-format csv -d $O/tp_pmc_$N -- python3 $ROOT/scripts/two_piece_exact_timing.py --queries 2000 --reps 1 > $O/tp_pmc_$N.log 2>&1; echo "2piece pmc $N rc=$?"
done
python3 - <<PY
import csv, glob, collections, json
for tag, pat in (("wave", "wsearch"), ("tp", "poa2_exact")):
    tot = collections.defaultdict(float); nd = collections.Counter()
    for f in glob.glob("$O/%s_pmc_*/*/*counter_collection.csv" % tag):
        for r in csv.DictReader(open(f)):
            if pat in r.get("Kernel_Name", ""):
                tot[r["Counter_Name"]] += float(r["Counter_Value"]); nd[r["Counter_Name"]] += 1
    tot["dispatches_per_counter"] = max(nd.values()) if nd else 0
    json.dump(tot, open("$O/%s_pmc_summary.json" % tag, "w"), indent=1, sort_keys=True)
    for f in glob.glob("$O/%s_trace/*/*kernel_stats.csv" % tag):
        open("$O/%s_kernel_stats.csv" % tag, "w").write(open(f).read())
PY
find $O -name "*.csv" -size +1000k -delete; find $O -name "*.db" -delete
cd $ROOT && timeout -k 10 300 python bench.py > $O/bench_r03_final.json 2> $O/bench.err; echo "bench rc=$?"
