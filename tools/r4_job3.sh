#!/bin/bash
set -e
OUT=$PWD/gpurun_out
python -m pytest tests/test_gpu_ops.py tests/test_gpu_train_ops.py tests/test_gpu_clap_text.py -x -q > $OUT/r4_t4.log 2>&1 || (tail -40 $OUT/r4_t4.log; exit 1)
tail -2 $OUT/r4_t4.log
python tools/bench_attn.py 2>&1 | grep attention | tee $OUT/r4_attn_new2.txt
python tools/step_table.py > $OUT/r4_step_attn.txt 2>&1
tail -1 $OUT/r4_step_attn.txt
