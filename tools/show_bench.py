"""Prints the headline and the per-workload figures of a bench.py JSON line: tools/show_bench.py file.json"""
import json
import sys
d = json.load(open(sys.argv[1]))
r = d.get("roofline", {})
print("headline %s  %.1f Msamples/s  %.2f ms/step  frac %.4f  cpu %s" % (d["config"]["workload"][:3], d["value"], d["ms_per_step"],
      r.get("frac", 0), d.get("cpu_baseline", {}).get("value")))
for k, v in d.get("workloads", {}).items():
    print("%-14s %8.1f ms %8.1f Msamples/s frac %.4f" % (k, v["ms_per_step"], v["msamples_per_s"], v.get("frac", 0)))
