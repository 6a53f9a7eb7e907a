#!/bin/bash
# PMC counters of the replay kernel (GPU box): instructions and wait cycles per step.
# usage: scripts/pmc_exact.sh [queries] [tag]     outputs under gpurun_out/pmc_exact_<tag>/   (environment passes through: POA_PS_GROUP, POA_EXACT_IMPL ...)
set -u
Q=${1:-512}
TAG=${2:-run}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_exact_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_IFETCH SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $C --output-format csv -d $OUT/$N -- python3 $ROOT/scripts/exact_timing.py --queries $Q --mode hybrid --check 0 --reps 1 > $OUT/$N.log 2>&1
  echo "pmc $N rc=$?"
done
python3 - <<PY
import csv, glob, collections, json
tot = collections.defaultdict(float)
for f in glob.glob("$OUT/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "search" in r.get("Kernel_Name", ""):
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
print(json.dumps(tot, indent=1))
json.dump(tot, open("$OUT/summary.json", "w"), indent=1)
PY
find $OUT -name "*.csv" -size +500k -delete
