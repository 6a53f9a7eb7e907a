#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_tlb
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
grep -i "UTCL\|TLB\|LATENCY" $OUT/counters.txt | head -40 > $OUT/counters_tlb.txt
for C in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $C --output-format csv -d $OUT/$N -- python3 $ROOT/scripts/exact_timing.py --queries 2048 --mode exact --check 0 --reps 1 > $OUT/$N.log 2>&1
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
PY
find $OUT -name "*.csv" -size +500k -delete
