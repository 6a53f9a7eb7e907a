#!/bin/bash
# The GPU parity and replay tests under every kernel-selection override (run on the GPU box):
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash scripts/variant_sweep.sh'
# tests/test_relative_encoding.py asserts which cell encoding a run picked, so it joins only the overrides that leave that choice alone.
for v in "POA_PLANES=32" "POA_COMPACT=0" "POA_PACKED=0" "POA_RELATIVE=1" "POA_RELATIVE=0"; do
  env $v timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_exact_replay.py -m gpu -q -x -k "not randomised" > gpurun_out/t_var.log 2>&1
  echo "$v: $(tail -n 1 gpurun_out/t_var.log)"
done
for v in "POA_PX=0" "POA_MW=0" "POA_PXMW=1" "POA_PXMW=0" "POA_MF=0" "POA_MF=1" \
         "POA_TB_GROUP=64" "POA_TB_GROUP=16" "POA_TB_DEPTH=1" "POA_EXACT_LDS=0" "POA_FWD_QUADS=1" "POA_FWD_QUADS=2"; do
  env $v timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_exact_replay.py tests/test_relative_encoding.py -m gpu -q -x -k "not randomised" > gpurun_out/t_var.log 2>&1
  echo "$v: $(tail -n 1 gpurun_out/t_var.log)"
done
