#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ directory (written by scripts/profile_round.sh on the
GPU box) into the small, committed files under profiles/:

  <tag>_kernel_stats.csv              rocprofv3 --stats of bench.py's headline step
  <tag>_kernel_stats_workload.csv     ... of scripts/profile_workload.py (the other kernels)
  <tag>_pmc_summary.csv               every counter, per kernel and grid: dispatches, mean, min, max
  <tag>_summary.json                  per kernel: average duration, HBM bytes per launch
                                      (FETCH_SIZE x 2 + WRITE_SIZE, the gfx950 correction of
                                      MI355X_MICROARCH.md, HBM section), and for the MFMA kernels
                                      flops, TFLOP/s and the MFMA busy fraction
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)
F64_MATRIX_PEAK_TF = 78.6   # MI355X datasheet; scripts/micro/mfma_f64_peak.hip sustains 47.4


def short(name):
    return name.split("(")[0].replace("void ", "")


def grid_of(r):
    """Total work-items of a dispatch (the counter files give only this; the kernel trace gives
    the three dimensions)."""
    return str(int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1))


for sub, dst in (("trace", "kernel_stats"), ("x_trace", "kernel_stats_workload")):
    ks = glob.glob(f"{src}/{sub}/*/*_kernel_stats.csv")
    if ks:
        shutil.copy(ks[0], f"profiles/{tag}_{dst}.csv")

# per-dispatch durations (ms) per (kernel, grid, workgroup)
dur = collections.defaultdict(list)
main_total = collections.Counter()      # headline step only: which kernel dominates it
for sub in ("trace", "x_trace"):
    for kt in glob.glob(f"{src}/{sub}/*/*_kernel_trace.csv"):
        for r in csv.DictReader(open(kt)):
            if "bmc::" not in r["Kernel_Name"]:
                continue
            key = (short(r["Kernel_Name"]), grid_of(r), r["Workgroup_Size_X"])
            ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            if sub == "trace":
                main_total[key] += ms
            dur[key].append(ms)
# the headline kernel's own durations: from the headline run only (the workload launches the
# same instantiation with other chain counts and lengths)
head = main_total.most_common(1)[0][0] if main_total else None
if head:
    dur[head] = []
    for kt in glob.glob(f"{src}/trace/*/*_kernel_trace.csv"):
        for r in csv.DictReader(open(kt)):
            if (short(r["Kernel_Name"]), grid_of(r), r["Workgroup_Size_X"]) == head:
                dur[head].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)

pmc = collections.defaultdict(list)
for name in ("pmc_fetch", "pmc_write", "pmc_sq", "x_pmc_fetch", "x_pmc_write", "x_pmc_mfma"):
    for f in glob.glob(f"{src}/{name}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "bmc::" not in r["Kernel_Name"]:
                continue
            k3 = (short(r["Kernel_Name"]), r["Grid_Size"], r["Counter_Name"])
            # the headline kernel's counters come from the headline passes only
            if head and k3[:2] == head[:2] and name.startswith("x_"):
                continue
            pmc[k3].append(float(r["Counter_Value"]))

with open(f"profiles/{tag}_pmc_summary.csv", "w", newline="") as fo:
    w = csv.writer(fo)
    w.writerow(["kernel", "grid_size", "counter", "dispatches", "mean", "min", "max"])
    for (k, g, c), v in sorted(pmc.items()):
        w.writerow([k, g, c, len(v), sum(v) / len(v), min(v), max(v)])

summary = {
    "tag": tag,
    "commands": ["python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra   (headline step)",
                 "python3 scripts/profile_workload.py   (every other kernel at the BASELINE sizes)"],
    "note": "FETCH_SIZE/WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half of the bytes of wide "
            "coalesced reads (MI355X_MICROARCH.md, HBM section) -> hbm_read_bytes = 2*FETCH*1024. "
            "Counters come from separate --pmc passes (never combined with another trace domain). "
            "MFMA: flops = SQ_INSTS_VALU_MFMA_MOPS_F64 * 512 per launch; mfma_busy = "
            "SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs * 256 CUs * GRBM_GUI_ACTIVE / 8 XCDs).",
    "headline": f"{head[0]} grid={head[1]}" if head else None,
    "kernels": {}}


def mean(v):
    return sum(v) / len(v) if v else None


for (k, g, wg), v in sorted(dur.items()):
    e = {"dispatches": len(v), "avg_ms": mean(v), "min_ms": min(v), "max_ms": max(v),
         "workgroup_size": int(wg)}
    f, wr = pmc.get((k, g, "FETCH_SIZE")), pmc.get((k, g, "WRITE_SIZE"))
    if f:
        e["hbm_read_bytes_per_launch"] = 2 * 1024 * mean(f)
    if wr:
        e["hbm_write_bytes_per_launch"] = 1024 * mean(wr)
    mops = pmc.get((k, g, "SQ_INSTS_VALU_MFMA_MOPS_F64"))
    if mops and mean(mops) > 0:
        flops = 512.0 * mean(mops)
        e["mfma_f64_flops_per_launch"] = flops
        e["mfma_f64_TFLOPs"] = flops / (e["avg_ms"] * 1e-3) / 1e12
        e["mfma_frac_of_78.6TF"] = e["mfma_f64_TFLOPs"] / F64_MATRIX_PEAK_TF
        busy, gui = pmc.get((k, g, "SQ_VALU_MFMA_BUSY_CYCLES")), pmc.get((k, g, "GRBM_GUI_ACTIVE"))
        if busy and gui and mean(gui) > 0:
            e["mfma_busy_frac"] = mean(busy) / (4 * 256 * mean(gui) / 8)
        e["mfma_instructions_per_launch"] = mean(pmc.get((k, g, "SQ_INSTS_MFMA"), [0]))
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_WAVES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY",
              "SQ_BUSY_CYCLES"):
        x = pmc.get((k, g, c))
        if x:
            e[c + "_per_launch"] = mean(x)
    summary["kernels"][f"{k} grid={g}"] = e
json.dump(summary, open(f"profiles/{tag}_summary.json", "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk in ("avg_ms", "dispatches",
      "hbm_read_bytes_per_launch", "mfma_f64_TFLOPs", "mfma_busy_frac")}
      for k, v in summary["kernels"].items()}, indent=1))
