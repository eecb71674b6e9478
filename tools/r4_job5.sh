#!/bin/bash
set -e
OUT=$PWD/gpurun_out
python -m pytest tests/test_gpu_ops.py -x -q -k "attn_block" > $OUT/r4_t5.log 2>&1 || (tail -40 $OUT/r4_t5.log; exit 1)
tail -12 $OUT/r4_t5.log
python -m pytest tests/test_gpu_cold_determinism.py tests/test_gpu_unet.py tests/test_gpu_engine.py -x -q > $OUT/r4_t6.log 2>&1 || (tail -40 $OUT/r4_t6.log; exit 1)
tail -8 $OUT/r4_t6.log
python tools/step_table.py > $OUT/r4_step_b256.txt 2>&1
tail -1 $OUT/r4_step_b256.txt
grep "attn_block256\|N1152" $OUT/r4_step_b256.txt | head -8
