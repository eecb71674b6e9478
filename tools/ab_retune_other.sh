#!/bin/bash
# the same re-measurement for the training step and generate_audio.py's batch-1 step
O=gpurun_out
timeout -k 10 500 python tools/ab_overlay.py $O/rt_train.json all train > $O/rt_train.log 2>&1; grep -v "^M" $O/rt_train.log | tail -n 1
export ALDM_AB_BATCH=1
timeout -k 10 400 python tools/ab_overlay.py $O/rt_b1.json small infer > $O/rt_b1.log 2>&1; grep -v "^M" $O/rt_b1.log | tail -n 1
