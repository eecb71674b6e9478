"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: the last `--steps` replays of a periodic launch sequence.
usage: python tools/prof_db.py gpurun_out/prof_x/x_results.db --steps 10 [--by-grid] [--top 40]"""
import argparse
import re
import sqlite3
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    m = re.match(r"(?:void )?([\w:]+(?:<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:90]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--per-step", type=int, default=0, help="kernels per step (take the last steps*per_step dispatches)")
    ap.add_argument("--by-grid", action="store_true")
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    cur = sqlite3.connect(a.db).cursor()
    rows = cur.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels order by start").fetchall()
    if a.per_step:
        rows = rows[-a.steps * a.per_step:]
    agg = defaultdict(lambda: [0, 0])
    for name, s, e, gx, gy, gz, wx in rows:
        key = short(name)
        if a.by_grid:
            key += f" wg={gx // max(wx, 1)}x{gy}x{gz}"
        agg[key][0] += e - s
        agg[key][1] += 1
    tot = sum(v[0] for v in agg.values())
    span = rows[-1][2] - rows[0][1]
    print(f"{len(rows)} dispatches, kernel time {tot / 1e6:.3f} ms, wall span {span / 1e6:.3f} ms"
          f" ({tot / 1e6 / a.steps:.3f} / {span / 1e6 / a.steps:.3f} ms per step)")
    for k, (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[: a.top]:
        print(f"{t / tot * 100:5.1f}%  {t / 1e3 / a.steps:9.1f} us/step  {n / a.steps:7.1f} calls  avg {t / n / 1e3:7.1f} us  {k}")


if __name__ == "__main__":
    main()
