#!/usr/bin/env python3
"""Timeline of the last frame in a rocprofv3 kernel trace (t_kernel_trace.csv): every dispatch's start relative to the
frame's first dispatch, its duration and the idle gap before it.   usage: tools/share_timeline.py <t_kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a frame starts with the clear of the accumulator: the last fillBuffer that is followed by a render kernel within 3 dispatches
starts = [i for i, r in enumerate(rows) if "fillBuffer" in r["Kernel_Name"] and any("rt_render" in q["Kernel_Name"] for q in rows[i + 1:i + 4])]
first = starts[-1] if starts else 0
t0 = int(rows[first]["Start_Timestamp"])
prev_end = t0
busy = 0
for r in rows[first:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:7.1f} us  {r['Kernel_Name'].replace('void ', '')[:70]}")
    busy += e - s
    prev_end = e
print(f"frame: {(prev_end - t0) / 1e3:.1f} us from first start to last end, {busy / 1e3:.1f} us inside kernels")
