set -u
ROOT=${GRAFT_REPO_ROOT}
OUT=$ROOT/gpurun_out/prof_exact
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM GRBM_GUI_ACTIVE" "WRITE_SIZE" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 $ROOT/scripts/exact_timing.py --queries 4096 --mode exact --check 0 > $OUT/pmc_$N.log 2>&1
done
python3 - <<'PY'
import csv,glob,collections,os
agg=collections.defaultdict(float)
for f in glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/prof_exact/pmc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","")
        if "exact_kernel" in k: agg[r["Counter_Name"]]+=float(r["Counter_Value"])
print({k:"%.4g"%v for k,v in sorted(agg.items())})
PY
