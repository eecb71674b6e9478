#!/bin/bash
set -e
OUT=$PWD/gpurun_out
python -m pytest tests/test_gpu_ops.py -x -q -k "hifigan" > $OUT/r4_t7.log 2>&1 || (tail -40 $OUT/r4_t7.log; exit 1)
tail -3 $OUT/r4_t7.log
python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_fullsize_tail.py -x -q > $OUT/r4_t8.log 2>&1 || (tail -40 $OUT/r4_t8.log; exit 1)
tail -14 $OUT/r4_t8.log
python tools/bench_tail.py > $OUT/r4_tail_new.txt 2>&1; head -32 $OUT/r4_tail_new.txt
ALDM_NO_HIFIGAN_PAIR=1 python tools/bench_tail.py 2>&1 | grep "^vocoder\|^vae" 
