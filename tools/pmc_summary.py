#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files: python3 tools/pmc_summary.py DIR [DIR...]"""
import collections, csv, glob, sys
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                agg[k]["dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for k, v in agg.items():
            if "k_" not in k:
                continue
            print(d, k, {c: round(sum(x) / len(x)) for c, x in v.items()})
