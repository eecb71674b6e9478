"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid) average duration over the timed region.
usage: python tools/prof_summary.py gpurun_out/prof_x/.../*_kernel_trace.csv [--steps N] [--top 50]"""
import argparse
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("bool _Accum", "bf16")
    m = re.match(r"void (\w+<[^>]*>)", name)
    if m:
        return m.group(1)
    return name.split("(")[0][-60:]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--steps", type=int, default=0, help="divide totals by this many steps")
    ap.add_argument("--top", type=int, default=60)
    ap.add_argument("--by-grid", action="store_true")
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.csv)))
    agg = defaultdict(lambda: [0, 0])
    for r in rows:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        key = short(r["Kernel_Name"])
        if a.by_grid:
            key += f" grid={r['Grid_Size_X']}x{r.get('Grid_Size_Z', '1')} lds={r.get('LDS_Block_Size', '?')}"
        agg[key][0] += d
        agg[key][1] += 1
    tot = sum(v[0] for v in agg.values())
    div = a.steps or 1
    print(f"total kernel time {tot / 1e6:.3f} ms over {len(rows)} dispatches" + (f" ({tot / 1e6 / div:.3f} ms/step)" if a.steps else ""))
    for k, (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[: a.top]:
        print(f"{t / tot * 100:5.1f}%  {t / 1e3 / div:9.1f} us{'/step' if a.steps else ''}  {n / div:7.1f} calls  avg {t / n / 1e3:7.1f} us  {k}")


if __name__ == "__main__":
    main()
