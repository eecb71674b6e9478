#!/bin/bash
OUT=$PWD/gpurun_out
for v in $(ls variants | sed 's/libaldm_//; s/.so//'); do
  echo "== $v" | tee -a $OUT/r4_attn_variants.txt
  ALDM_LIB=$PWD/variants/libaldm_$v.so python tools/bench_attn.py 2>&1 | grep "prescaled=1" | grep "N1000 H8\|N252\|B16" | tee -a $OUT/r4_attn_variants.txt
done
