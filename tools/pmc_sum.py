#!/usr/bin/env python3
"""Developer tool: per counter, the value of the LONGEST dispatch of the stage kernels in rocprofv3 --pmc CSV output.
usage: python tools/pmc_sum.py <dir> [<dir> ...] [--kernel substr] [--json out.json]"""
import csv, glob, json, os, sys

args = sys.argv[1:]
kern = "rt_stage"
out_json = None
dirs = []
while args:
    a = args.pop(0)
    if a == "--kernel": kern = args.pop(0)
    elif a == "--json": out_json = args.pop(0)
    else: dirs.append(a)
res = {}
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"]]
        if not rows:
            continue
        # longest dispatch
        dur = {}
        for r in rows:
            dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        best = max(dur, key=dur.get)
        for r in rows:
            if r["Dispatch_Id"] == best:
                res[r["Counter_Name"]] = res.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                res.setdefault("_kernel", r["Kernel_Name"][:60]); res["_ns_" + os.path.basename(os.path.normpath(d))] = dur[best]
                res["_vgpr"] = r["VGPR_Count"]; res["_scratch"] = r["Scratch_Size"]; res["_lds"] = r["LDS_Block_Size"]
for k in sorted(res):
    print(f"{k:32s} {res[k]}")
if out_json:
    json.dump(res, open(out_json, "w"), indent=1)
