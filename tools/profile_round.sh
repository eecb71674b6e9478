#!/bin/bash
# rocprofv3 evidence for one state of the repository (run on the MI355X box from the repository root):
#   tools/profile_round.sh r02a      -> gpurun_out/<tag>_*   (copy the summaries you want judged into profiles/)
# Kernel trace and PMC passes are separate runs (counters never share a run with other trace domains).
set -e
TAG=${1:-r02x}
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
BENCH="bench.py --no-cpu-baseline --no-train --no-trace --steps 50 --warmup 5"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG} -o inf -- python3 $BENCH > $OUT/prof_${TAG}.log 2>&1
python3 tools/prof_summary.py $OUT/prof_${TAG}/inf_kernel_trace.csv --top 70 > $OUT/${TAG}_kernel_trace_summary.txt
python3 tools/prof_summary.py $OUT/prof_${TAG}/inf_kernel_trace.csv --top 90 --by-grid > $OUT/${TAG}_kernel_trace_by_grid.txt
cp $OUT/prof_${TAG}/inf_kernel_stats.csv $OUT/${TAG}_bench_kernel_stats.csv
echo "[profile] kernel trace done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_train -o trn -- python3 tools/train_step.py 10 > $OUT/prof_${TAG}_train.log 2>&1
python3 tools/prof_summary.py $OUT/prof_${TAG}_train/trn_kernel_trace.csv --top 70 > $OUT/${TAG}_train_kernel_trace_summary.txt
cp $OUT/prof_${TAG}_train/trn_kernel_stats.csv $OUT/${TAG}_train_kernel_stats.csv
echo "[profile] training trace done"
PB="bench.py --no-cpu-baseline --no-train --no-trace --steps 3 --warmup 1"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_${TAG}_fetch -o f -- python3 $PB > $OUT/prof_${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/prof_${TAG}_write -o w -- python3 $PB > $OUT/prof_${TAG}_write.log 2>&1
python3 tools/pmc_traffic.py $OUT/prof_${TAG}_fetch/f_counter_collection.csv $OUT/prof_${TAG}_write/w_counter_collection.csv > $OUT/${TAG}_pmc_traffic.json
echo "[profile] PMC traffic done"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/prof_${TAG}_sq -o s -- python3 $PB > $OUT/prof_${TAG}_sq.log 2>&1 || echo "[profile] SQ pass failed (counter names?)"
python3 tools/pmc_sq.py $OUT/prof_${TAG}_sq/s_counter_collection.csv attention_kernel igemm groupnorm pgemm attn_block64 ddim gn_silu > $OUT/${TAG}_pmc_sq.json || echo "[profile] pmc_sq.py failed"
# the raw traces are hundreds of MB (gpurun merges at most 64 MiB back): keep the summaries only
rm -rf $OUT/prof_${TAG} $OUT/prof_${TAG}_train $OUT/prof_${TAG}_fetch $OUT/prof_${TAG}_write $OUT/prof_${TAG}_sq
echo "[profile] done"
