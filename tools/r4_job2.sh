#!/bin/bash
# round-4 measurement job 2: attention changes (tests + A/B timer), new tests, FETCH_SIZE with both XCD maps
set -e
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_ops.py tests/test_gpu_train_ops.py tests/test_gpu_cold_determinism.py tests/test_gpu_train_script.py -x -q > $OUT/r4_t3.log 2>&1 || (tail -40 $OUT/r4_t3.log; exit 1)
tail -3 $OUT/r4_t3.log
python tools/bench_attn.py > $OUT/r4_attn_new.txt 2>&1
ALDM_ATTN_CFG=162 python tools/bench_attn.py > $OUT/r4_attn_162.txt 2>&1 || true
cat $OUT/r4_attn_new.txt | grep attention; grep "N1000" $OUT/r4_attn_162.txt || true
PB="bench.py --no-cpu-baseline --no-train --no-trace --steps 3 --warmup 1"
for m in 1 2; do
  export ALDM_XCD_MAP=$m
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_r4_fetch_m$m -o f -- python3 $PB > $OUT/prof_r4_fetch_m$m.log 2>&1
  python3 tools/pmc_traffic.py $OUT/prof_r4_fetch_m$m/f_counter_collection.csv $OUT/prof_r4_fetch_m$m/f_counter_collection.csv > $OUT/r4_fetch_xmap$m.json
  rm -rf $OUT/prof_r4_fetch_m$m
  echo "[job] fetch pass xmap=$m done"
done
