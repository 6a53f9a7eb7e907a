#!/bin/bash
# Step latency of the wave-per-query replay: lanes tested per step x waves per block (GPU box).
q=${1:-512}
for l in 63 16 4 1; do
  for w in 16 4; do
    POA_WS_LANES=$l POA_WS_WAVES=$w python scripts/exact_timing.py --queries $q --mode exact --check 0 --reps 1 2>/dev/null > gpurun_out/sweep_tmp.json
    python - <<PY
import json
d = json.load(open("gpurun_out/sweep_tmp.json"))
print("queries", d["queries"], "lanes", $l, "waves/block", $w, "ms_exact", round(d["ms_exact"], 1), "steps_mean", d.get("steps_mean"), "us/step(longest)", d.get("us_per_step_longest"))
PY
  done
done
