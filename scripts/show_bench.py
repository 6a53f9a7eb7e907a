#!/usr/bin/env python3
"""Print the headline numbers of a bench.py JSON line (helper for GPU sessions)."""
import json
import sys

for path in sys.argv[1:]:
    for l in open(path):
        if l.startswith("{"):
            j = json.loads(l)
            r = j["roofline"]
            print(path, j["value"], "Gcells/s step", j["ms_per_step"], "ms fwd", r["avg_launch_ms"], "ms frac", r["frac"],
                  "tb", r["traceback_ms_per_step"], "chk", j["config"]["score_checksum"], j["config"]["flagged_queries"],
                  "cpu", (j.get("cpu_baseline") or {}).get("value"))
