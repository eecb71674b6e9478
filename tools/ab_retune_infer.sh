#!/bin/bash
# re-measure the per-key launch configurations of the config-2 step in the step (after a kernel change that shifts the trade-offs)
O=gpurun_out
timeout -k 10 400 python tools/ab_overlay.py $O/rt_small.json small infer > $O/rt_small.log 2>&1; grep -v "^M" $O/rt_small.log | tail -n 1
ALDM_TUNED_PATCH=$O/rt_small.json timeout -k 10 400 python tools/ab_overlay.py $O/rt_halo.json halo infer > $O/rt_halo.log 2>&1; grep -v "^M" $O/rt_halo.log | tail -n 1
