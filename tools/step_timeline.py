#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV and prints the launch timeline of one steady-state window: kernel, duration, gap to the previous
kernel's end.   python tools/step_timeline.py <kernel_trace.csv> [anchor-kernel-substring] [occurrence] [count]"""
import csv
import sys

path = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "adam_flat_kernel"
occ = int(sys.argv[3]) if len(sys.argv) > 3 else -3
count = int(sys.argv[4]) if len(sys.argv) > 4 else 60
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
idx = [i for i, r in enumerate(rows) if anchor in r[2]]
if not idx:
    sys.exit("anchor kernel not found")
i0 = idx[occ] + 1
prev_end = rows[i0 - 1][1]
t0 = rows[i0][0]
tot_k = 0
for s, e, name in rows[i0:i0 + count]:
    short = name.replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", "")
    short = short.split("(")[0][:86]
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  {short}")
    tot_k += e - s
    prev_end = e
    if anchor in name:
        break
print(f"window {(prev_end - t0) / 1e3:.1f} us, kernels {tot_k / 1e3:.1f} us")
