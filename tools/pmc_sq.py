"""Per-kernel SQ counters from a rocprofv3 --pmc pass (SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU
SQ_VALU_MFMA_BUSY_CYCLES): averages per launch, instructions per wave, and the VALU / MFMA pipe occupancy the guide's cycle
constants imply.  usage: python tools/pmc_sq.py <counter_collection.csv> [substring ...] > profiles/rNN_pmc_sq.json"""
import collections
import csv
import json
import re
import sys


def fam(name):
    name = name.replace("(anonymous namespace)::", "").replace("aldm_igemm_detail::", "")
    m = re.search(r"(\w+<[^>]*>)", name)
    return m.group(1) if m else name.split("(")[0][-50:]


def main():
    want = sys.argv[2:] or ["attention_kernel", "igemm", "groupnorm"]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    grid = {}
    for r in csv.DictReader(open(sys.argv[1])):
        k = fam(r["Kernel_Name"])
        if not any(w in k for w in want):
            continue
        k = f"{k} grid={r['Grid_Size']} wg={r['Workgroup_Size']}"
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        grid[k] = (int(r["VGPR_Count"]), int(r["LDS_Block_Size"]))
    out = {}
    for k, c in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_BUSY_CYCLES", [0]))):
        a = {n: sum(v) / len(v) for n, v in c.items()}
        waves = a.get("SQ_WAVES", 0) or 1
        out[k] = {"launches_sampled": len(next(iter(c.values()))), "vgprs": grid[k][0], "lds_bytes": grid[k][1],
                  "waves": round(waves), "valu_insts_per_wave": round(a.get("SQ_INSTS_VALU", 0) / waves, 1),
                  "mfma_insts_per_wave": round(a.get("SQ_INSTS_MFMA", 0) / waves, 1),
                  "sq_busy_cycles": round(a.get("SQ_BUSY_CYCLES", 0)),
                  "active_inst_valu_quadcycles": round(a.get("SQ_ACTIVE_INST_VALU", 0)),
                  "valu_mfma_busy_cycles": round(a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0))}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
