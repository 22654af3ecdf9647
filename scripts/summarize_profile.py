#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ directory (written by scripts/profile_round.sh on the
GPU box) into the small, committed files under profiles/."""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)
ks = glob.glob(f"{src}/trace/*/*_kernel_stats.csv")[0]
shutil.copy(ks, f"profiles/{tag}_kernel_stats.csv")

xs = glob.glob(f"{src}/x_trace/*/*_kernel_stats.csv")
if xs:
    shutil.copy(xs[0], f"profiles/{tag}_kernel_stats_with_extras.csv")
# per-dispatch durations: the loop kernel from the main (single-chain) run, the residual
# kernel from the run that includes the extras
dur = collections.defaultdict(list)
for sub, want in (("trace", "gibbs_loop_kernel"), ("x_trace", "residual_rss_kernel")):
    for kt in glob.glob(f"{src}/{sub}/*/*_kernel_trace.csv"):
        for r in csv.DictReader(open(kt)):
            if want in r["Kernel_Name"]:
                key = (r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Grid_Size_X"],
                       r["Workgroup_Size_X"])
                dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
pmc = collections.defaultdict(list)
for f in glob.glob(f"{src}/x_pmc_fetch/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "residual_rss_kernel" in r["Kernel_Name"]:
            k = (r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Grid_Size"], r["Counter_Name"])
            pmc[k].append(float(r["Counter_Value"]))
for name in ("pmc_fetch", "pmc_write", "pmc_sq"):
    for f in glob.glob(f"{src}/{name}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Grid_Size"], r["Counter_Name"])
            pmc[k].append(float(r["Counter_Value"]))
with open(f"profiles/{tag}_pmc_summary.csv", "w", newline="") as fo:
    w = csv.writer(fo)
    w.writerow(["kernel", "grid_size", "counter", "dispatches", "mean", "min", "max"])
    for (k, g, c), v in sorted(pmc.items()):
        if "bmc::" in k:
            w.writerow([k, g, c, len(v), sum(v) / len(v), min(v), max(v)])
summary = {"tag": tag, "command": "python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra "
                                  "(loop kernel); bench.py --steps 2 --warmup 1 --no-cpu-baseline (extras)",
           "note": "FETCH_SIZE/WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half of the bytes of "
                   "wide coalesced reads (MI355X_MICROARCH.md, HBM section) -> hbm_read_bytes = 2*FETCH*1024",
           "kernels": {}}
for (k, g, wg), v in sorted(dur.items()):
    e = {"dispatches": len(v), "avg_ms": sum(v) / len(v), "min_ms": min(v), "max_ms": max(v),
         "workgroup_size": int(wg)}
    f = pmc.get((k, g, "FETCH_SIZE"))
    wr = pmc.get((k, g, "WRITE_SIZE"))
    if f:
        e["hbm_read_bytes_per_launch"] = 2 * 1024 * sum(f) / len(f)
    if wr:
        e["hbm_write_bytes_per_launch"] = 1024 * sum(wr) / len(wr)
    summary["kernels"][f"{k} grid={g}"] = e
json.dump(summary, open(f"profiles/{tag}_summary.json", "w"), indent=1)
print(json.dumps(summary, indent=1))
