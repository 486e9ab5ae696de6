#!/usr/bin/env python3
"""Averages the counters of a rocprofv3 --pmc pass over the launches of the render kernel.
usage: collect_sq.py <counter_collection.csv> [out.json [workload spp_per_launch]]
With a workload name the record is merged into out.json under that key."""
import csv
import json
import sys
from collections import defaultdict

sums = defaultdict(float)
launches = set()
kernel = ""
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if "render_tiles" in r["Kernel_Name"]:
            sums[r["Counter_Name"]] += float(r["Counter_Value"])
            launches.add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
            kernel = r["Kernel_Name"]
out = {"kernel": kernel, "launches": len(launches), "per_launch": {k: v / max(1, len(launches)) for k, v in sorted(sums.items())}}
pl = out["per_launch"]
if "SQ_WAVE_CYCLES" in pl and "SQ_ACTIVE_INST_VALU" in pl:
    out["valu_active_per_wave_cycle"] = pl["SQ_ACTIVE_INST_VALU"] / pl["SQ_WAVE_CYCLES"]
if "SQ_BUSY_CYCLES" in pl and "SQ_ACTIVE_INST_VALU" in pl:
    out["valu_active_per_busy_cycle"] = pl["SQ_ACTIVE_INST_VALU"] / pl["SQ_BUSY_CYCLES"]
text = json.dumps(out, indent=1)
print(text)
if len(sys.argv) > 4:
    import os
    data = {}
    if os.path.exists(sys.argv[2]):
        data = json.load(open(sys.argv[2]))
    out["spp_per_launch"] = int(sys.argv[4])
    data[sys.argv[3]] = out
    open(sys.argv[2], "w").write(json.dumps(data, indent=1, sort_keys=True) + "\n")
elif len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text + "\n")
