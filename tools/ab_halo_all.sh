#!/bin/bash
# the in-step A/B of the halo tiles, every workload, on ONE box (gpurun_out/abh_*)
O=gpurun_out
timeout -k 10 400 python tools/ab_overlay.py $O/abh_infer.json halo infer > $O/abh_infer.log 2>&1; grep -v "^M" $O/abh_infer.log | tail -n 1
ALDM_TUNED_PATCH=$O/abh_infer.json timeout -k 10 400 python tools/ab_overlay.py $O/abh_train.json halo train > $O/abh_train.log 2>&1; grep -v "^M" $O/abh_train.log | tail -n 1
export ALDM_AB_BATCH=1
timeout -k 10 300 python tools/ab_overlay.py $O/abh_b1h.json halo infer > $O/abh_b1h.log 2>&1; grep -v "^M" $O/abh_b1h.log | tail -n 1
