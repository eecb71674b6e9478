#!/bin/bash
# end-of-round check on the MI355X box: the whole -m gpu suite, then the driver's bench line
OUT=$PWD/gpurun_out
TAG=${1:-r4}
python -m pytest tests -x -q -m gpu > $OUT/${TAG}_gpu_all.log 2>&1; echo "pytest rc=$?"; grep "passed\|failed" $OUT/${TAG}_gpu_all.log | tail -2
python bench.py --steps 200 --warmup 10 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$?"; tail -2 $OUT/${TAG}_bench.err
