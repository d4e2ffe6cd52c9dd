#!/usr/bin/env python3
"""Counts the HIP API calls between the two hipDeviceSynchronize marker pairs of tools/static_trace.py in a rocprofv3
--hip-trace CSV (argv[1] = directory)."""
import csv, glob, sys, collections
path = glob.glob(sys.argv[1] + "/**/*hip_api_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if r["Function"] == "hipDeviceSynchronize"]
pairs = [i for i, j in zip(marks, marks[1:]) if j == i + 1]
assert len(pairs) >= 2, pairs
a, b = pairs[0] + 2, pairs[-1]
cnt = collections.Counter(r["Function"] for r in rows[a:b])
print("HIP API calls made by 10 vkr_render_static frames (5 modes, 1280x720) + vkr_wait:")
for k, v in sorted(cnt.items()):
    print("  %-32s %d" % (k, v))
bad = [k for k in cnt if any(w in k for w in ("Malloc", "Free", "EventCreate", "StreamCreate", "HostAlloc", "HostRegister"))]
print("allocating calls:", bad if bad else "none")
