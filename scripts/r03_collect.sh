#!/bin/bash
# Round-3 evidence run (GPU box): micro-benchmark, sequential POA latency, two-piece profile, dense forward profile.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out/r03; mkdir -p $O
cd $ROOT
timeout -k 10 120 profiles/microbench/valu_issue > $O/valu_issue_r03.json 2> $O/valu_issue.err; echo "microbench rc=$?"
timeout -k 10 300 python scripts/sequential_poa_latency.py > $O/sequential_poa_latency.json 2> $O/seq.err; echo "seq latency rc=$?"
timeout -k 10 200 python bench.py --model 2piece --steps 2 --warmup 1 > $O/bench_2piece.json 2> $O/b2.err; echo "bench 2piece rc=$?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tp_trace -- python3 $ROOT/bench.py --model 2piece --steps 1 --warmup 1 > $O/tp_trace.log 2>&1; echo "2piece trace rc=$?"
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $C --output-format csv -d $O/tp_pmc_$N -- python3 $ROOT/bench.py --model 2piece --steps 1 --warmup 1 > $O/tp_pmc_$N.log 2>&1; echo "2piece pmc $N rc=$?"
done
python3 - <<PY
import csv, glob, json, collections
out = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$O/tp_pmc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "poa2_" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            out[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
res = {k: dict(v) for k, v in out.items()}
for k in res:
    res[k]["dispatches_per_counter"] = max(c for (kk, _), c in n.items() if kk == k)
    if "WRITE_SIZE" in res[k] and "FETCH_SIZE" in res[k]:
        res[k]["hbm_bytes"] = res[k]["WRITE_SIZE"] * 1024.0 + res[k]["FETCH_SIZE"] * 1024.0 * 2.0   # KiB; gfx950 tallies 128-B fetches at 64 B
json.dump(res, open("$O/two_piece_pmc_summary.json", "w"), indent=1, sort_keys=True)
for f in glob.glob("$O/tp_trace/*/*kernel_stats.csv"):
    open("$O/two_piece_kernel_stats.csv", "w").write(open(f).read())
PY
find $O -name "*.csv" -size +1000k -delete; find $O -name "*.db" -delete
cd $ROOT && bash profiles/run_rocprof.sh r03 > $O/run_rocprof.log 2>&1; echo "dense rocprof rc=$?"
