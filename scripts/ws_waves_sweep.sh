#!/bin/bash
# waves per block / lanes sweep of the wave replay kernel (GPU box)
for w in 8 12 16; do for l in 16 32 63; do echo "waves $w lanes $l"; POA_WS_WAVES=$w POA_WS_LANES=$l timeout -k 10 100 python scripts/exact_timing.py --queries 10000 --mode hybrid --check 0 --reps 1 2>&1 | tail -n 1 | cut -c1-130 || exit 1; done; done
