"""Turns gpurun_out/prof_<workload>_<mode>/ (written by tools/profile.sh) into the summaries kept
under profiles/<round>/: <workload>_<mode>_kernel_stats.csv, _pmc.json and _bench.json."""
import csv
import glob
import json
import os
import shutil
import sys

workload, mode, rnd = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "r1")
src = f"gpurun_out/prof_{workload}_{mode}"
dst = f"profiles/{rnd}"
os.makedirs(dst, exist_ok=True)
key = "wavefront" if mode == "wavefront" else "megakernel"
# A counter pass runs one bench step.  A step of the megakernel can be two launches (a probe of
# a few samples that records tile costs, then the rest in cost order): counters are summed
# over the step's launches, so every figure is per step.
out = {}
kernels = []
for d in "abcd":
    # a pass directory can hold files of earlier runs (gpurun merges into the local copy): the newest counts
    for f in sorted(glob.glob(f"{src}/{d}/*/*_counter_collection.csv"), key=os.path.getmtime)[-1:]:
        for r in csv.DictReader(open(f)):
            if key in r["Kernel_Name"]:
                out[r["Counter_Name"]] = out.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                if r["Kernel_Name"] not in kernels:
                    kernels.append(r["Kernel_Name"])
out["kernels"] = kernels
if "SQ_THREAD_CYCLES_VALU" in out:
    out["thread_active_fraction"] = out["SQ_THREAD_CYCLES_VALU"] / (64 * out["SQ_ACTIVE_INST_VALU"])
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    out["hbm_bytes_per_launch"] = (2 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024
json.dump(out, open(f"{dst}/{workload}_{mode}_pmc.json", "w"), indent=1, sort_keys=True)
for f in sorted(glob.glob(f"{src}/trace/*/*_kernel_stats.csv"), key=os.path.getmtime)[-1:]:
    shutil.copy(f, f"{dst}/{workload}_{mode}_kernel_stats.csv")
for line in open(f"{src}/trace.log"):
    if line.startswith("{"):
        open(f"{dst}/{workload}_{mode}_bench.json", "w").write(line)
print(json.dumps(out, indent=1, sort_keys=True))
