"""Condenses the rocprofv3 outputs of tools/profile_round.sh into one JSON: per yk_* kernel the average duration (kernel trace)
and the per-launch counter averages (PMC passes).  FETCH_SIZE / WRITE_SIZE are reported raw (KB as rocprofv3 prints them) and
as corrected bytes: x1024, and FETCH_SIZE x2 on gfx950 (guide: MI355X_MICROARCH.md, HBM/rocprofv3 section)."""
import csv, glob, json, os, sys
from collections import defaultdict

root = sys.argv[1]
out = {}

def short(name):
    name = name.split("(")[0].strip()
    name = name.split("<")[0].strip()              # template instantiations (yk_encode2_kernel<false, false>) under the kernel's name
    return name.split(" ")[-1]

for f in glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = short(r["Name"])
        if n.startswith("yk_"):
            out.setdefault(n, {})["avg_ns"] = float(r["AverageNs"]); out[n]["calls"] = int(r["Calls"])
for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            n = short(r["Kernel_Name"])
            if n.startswith("yk_"):
                acc[n][(r["Counter_Name"], r["Dispatch_Id"])].append(float(r["Counter_Value"]))
    for n, d in acc.items():
        per = defaultdict(list)
        for (cname, _disp), vals in d.items():
            per[cname].append(sum(vals))
        for cname, vals in per.items():
            out.setdefault(n, {})[cname] = sum(vals) / len(vals)
for n, d in out.items():
    if "FETCH_SIZE" in d: d["fetch_bytes_corrected"] = d["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in d: d["write_bytes_corrected"] = d["WRITE_SIZE"] * 1024
    if "fetch_bytes_corrected" in d and "write_bytes_corrected" in d:
        d["hbm_traffic_bytes"] = d["fetch_bytes_corrected"] + d["write_bytes_corrected"]
print(json.dumps(out, indent=1, sort_keys=True))
