"""Runs only the LoRA-training leg of bench.py (config 3) -- the program to put behind `rocprofv3 --kernel-trace --stats --`.
usage: python tools/train_step.py [steps] [batch] [rank]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rank = int(sys.argv[3]) if len(sys.argv) > 3 else 8
print(json.dumps(bench.bench_train(1, 0, steps=steps, warmup=4, batch=batch, rank_lora=rank)))
