#!/bin/bash
# round-4 measurement job 1: MFMA / VALU overlap microbenchmark, new tests, FETCH_SIZE with both XCD maps
set -e
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu_gap.hip -o /tmp/mfma_valu_gap 2>/dev/null
/tmp/mfma_valu_gap > $OUT/r4_micro_gap.txt 2>&1
echo "[job] micro done"
python -m pytest tests/test_gpu_cold_determinism.py tests/test_gpu_train_script.py -x -q > $OUT/r4_t2.log 2>&1 || (tail -30 $OUT/r4_t2.log; exit 1)
tail -3 $OUT/r4_t2.log
PB="bench.py --no-cpu-baseline --no-train --no-trace --steps 3 --warmup 1"
for m in 1 2; do
  export ALDM_XCD_MAP=$m
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_r4_fetch_m$m -o f -- python3 $PB > $OUT/prof_r4_fetch_m$m.log 2>&1
  python3 tools/pmc_traffic.py $OUT/prof_r4_fetch_m$m/f_counter_collection.csv $OUT/prof_r4_fetch_m$m/f_counter_collection.csv > $OUT/r4_fetch_xmap$m.json
  rm -rf $OUT/prof_r4_fetch_m$m
  echo "[job] fetch pass xmap=$m done"
done
