"""Per-kernel sums of the rocprofv3 --pmc passes of tools/gpu_pmc2.sh -> profiles/r02_pmc_summary.json.
usage: python tools/pmc_summarize.py gpurun_out/pmc2 profiles/r02_pmc_summary.json"""
import collections
import csv
import glob
import json
import os
import sys

src, out = sys.argv[1], sys.argv[2]
res = collections.defaultdict(lambda: collections.defaultdict(lambda: {"launches": 0, "sum": 0.0, "ns": 0}))
for d in sorted(glob.glob(os.path.join(src, "*"))):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if "fh::" not in name and "copy" not in name.lower():
                continue
            short = name.split("(")[0].replace("void ", "")[:90]
            e = res[short][r["Counter_Name"]]
            e["sum"] += float(r["Counter_Value"])
            key = (r["Dispatch_Id"], r["Counter_Name"])
            if key not in seen:
                seen.add(key)
                e["launches"] += 1
                e["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
summary = {}
for k, ctrs in res.items():
    summary[k] = {c: {"launches": v["launches"], "total": v["sum"], "per_launch": v["sum"] / max(v["launches"], 1),
                      "avg_us": v["ns"] / max(v["launches"], 1) / 1e3} for c, v in ctrs.items()}
json.dump(summary, open(out, "w"), indent=1, sort_keys=True)
print("wrote", out, len(summary), "kernels")
