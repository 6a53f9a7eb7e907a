#!/bin/bash
# PMC counters of the multi-wave forward kernel on a configs[3] sample (GPU box): instructions per wave and row
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_pxmw
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $C --output-format csv -d $OUT/$N -- python3 $ROOT/scripts/config_throughput.py --config 4 --queries 64 --runs 1 > $OUT/$N.log 2>&1
done
python3 - <<PY
import csv, glob, json, collections
out = collections.defaultdict(dict)
for f in glob.glob("$OUT/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "pxmw" in r["Kernel_Name"]:
            out["pxmw"][r["Counter_Name"]] = out["pxmw"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
k = out["pxmw"]
rows = 56101
k["per_wave_row"] = {n: round(v / (k["SQ_WAVES"] * rows), 2) for n, v in k.items() if n.startswith("SQ_INSTS")}
json.dump(out, open("$OUT/summary.json", "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
PY
find $OUT -name "*.csv" -size +1000k -delete; find $OUT -name "*.db" -delete
