#!/bin/bash
OUT=$PWD/gpurun_out
python tools/train_table.py > $OUT/r4_train_table.txt 2>&1
head -80 $OUT/r4_train_table.txt | tail -75
