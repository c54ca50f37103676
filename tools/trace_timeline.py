#!/usr/bin/env python3
"""The last frame of a rocprofv3 kernel trace as a timeline: every dispatch after the last accumulator clear, with its
start relative to the first, its duration and the idle gap before it.  usage: trace_timeline.py <kernel_trace.csv>"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# a frame begins at the queue-head clear before the first render kernel that follows a sum kernel
frames, cur = [], []
for r in rows:
    cur.append(r)
    if r["Kernel_Name"].startswith("rt_sum_split_kernel"):
        frames.append(cur)
        cur = []
if not frames:
    frames = [rows]
last = frames[-1]
t0 = int(last[0]["Start_Timestamp"])
prev_end = t0
busy = 0
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:7.1f} us  {r['Kernel_Name'][:60]}")
    busy += e - s
    prev_end = max(prev_end, e)
print(f"span {(prev_end - t0) / 1e3:.1f} us, kernels {busy / 1e3:.1f} us, idle {(prev_end - t0 - busy) / 1e3:.1f} us")
