#!/bin/bash
# Profiling recipe (run on the GPU box via gpurun): kernel trace + separate PMC passes.
# usage: profiles/run_rocprof.sh <tag>      outputs under gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 > $OUT/trace.log 2>&1
echo "trace rc=$?" >> $OUT/trace.log
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 $ROOT/bench.py --steps 1 --warmup 0 --cpu-sample 0 > $OUT/pmc_$N.log 2>&1
  echo "pmc $N rc=$?" >> $OUT/trace.log
done
# keep only small summaries
find $OUT -name "*.csv" -size +2000k -delete
ls -R $OUT | head -50 >> $OUT/trace.log
