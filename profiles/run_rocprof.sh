#!/bin/bash
# Profiling recipe (run on the GPU box via gpurun): kernel trace + separate PMC passes of `bench.py`.
# usage: profiles/run_rocprof.sh <tag>      outputs under gpurun_out/prof_<tag>/ ; copy the summaries into profiles/<tag>/
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-extras > $OUT/trace.log 2>&1
echo "trace rc=$?" >> $OUT/trace.log
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-extras > $OUT/pmc_$N.log 2>&1
  echo "pmc $N rc=$?" >> $OUT/trace.log
done
# summaries: kernel stats (trace) and counters per kernel (one launch each: --steps 1 --warmup 0)
python3 - <<PY
import csv, glob, json, collections, sys
sys.path.insert(0, "$ROOT")
from bench import forward_kernel_source_hash
out = collections.defaultdict(dict)
for f in glob.glob("$OUT/pmc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        out[r["Kernel_Name"]][r["Counter_Name"]] = out[r["Kernel_Name"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
json.dump(out, open("$OUT/pmc_summary.json", "w"), indent=1, sort_keys=True)
for f in glob.glob("$OUT/trace/*/*kernel_stats.csv"):
    open("$OUT/kernel_stats.csv", "w").write(open(f).read())
kname, k = next(((n, v) for n, v in out.items() if "poa_forward_px_kernel" in n), (None, None))
if k and "WRITE_SIZE" in k and "FETCH_SIZE" in k:
    w, fch = k["WRITE_SIZE"] * 1024.0, k["FETCH_SIZE"] * 1024.0 * 2.0   # KiB -> bytes; gfx950 tallies 128-B fetches at 64 B
    kc = {"workload": "config2", "queries": 10000, "kernel": kname.split("(")[0].replace("void ", ""),
          "hbm_bytes_per_launch": int(w + fch), "write_bytes": int(w), "fetch_bytes_corrected": int(fch),
          "sq_insts_valu_per_launch": k.get("SQ_INSTS_VALU"), "sq_insts_salu_per_launch": k.get("SQ_INSTS_SALU"),
          "sq_wave_cycles": k.get("SQ_WAVE_CYCLES"), "sq_wait_inst_any": k.get("SQ_WAIT_INST_ANY"), "sq_waves": k.get("SQ_WAVES"),
          "csrc_sha16": forward_kernel_source_hash()[0], "hashed_files": forward_kernel_source_hash()[1],
          "source": "profiles/run_rocprof.sh $TAG: rocprofv3 --pmc in separate passes, one launch each (bench.py --steps 1 --warmup 0)",
          "note": "WRITE_SIZE / FETCH_SIZE in KiB -> bytes; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B)"}
    json.dump(kc, open("$OUT/kernel_counters.json", "w"), indent=1)
PY
find $OUT -name "*.csv" -size +1000k -delete
find $OUT -name "*.db" -delete
tail -3 $OUT/trace.log
