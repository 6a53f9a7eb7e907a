#!/bin/bash
# Replay throughput (hybrid, config 2) against the lanes tested per step and the waves per block (GPU box).
q=${1:-10000}
for l in 63 16 8 4; do
  for w in 16 8; do
    POA_WS_LANES=$l POA_WS_WAVES=$w python scripts/exact_timing.py --queries $q --mode hybrid --check 0 --reps 1 2>/dev/null > gpurun_out/sweep_tmp.json
    python - <<PY
import json
d = json.load(open("gpurun_out/sweep_tmp.json"))
print("queries", d["queries"], "lanes", $l, "waves/block", $w, "ms_exact", round(d["ms_exact"], 1), "Gcells/s", d["gcells_per_s"], "steps_mean", d.get("steps_mean"))
PY
  done
done
