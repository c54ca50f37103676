#!/usr/bin/env python3
"""Compile the HIP library with -Rpass-analysis=kernel-resource-usage and print one line per kernel."""
import re, subprocess, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytrace_amd import build
cmd = [build.HIPCC] + build.HIP_FLAGS + ["-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/rt_usage.so",
       os.path.join(build.CSRC, "rt_api.hip"), os.path.join(build.CSRC, "rt_gl_capture.cpp")]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", cur).replace("void ", "")
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z][\w ]*?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
filt = sys.argv[1] if len(sys.argv) > 1 else ""
for k, v in rows.items():
    if filt in k:
        print(f"{k:60s} vgpr={v.get('VGPRs')} agpr={v.get('AGPRs')} sgpr={v.get('TotalSGPRs')} scratch={v.get('ScratchSize')} occ={v.get('Occupancy')} lds={v.get('LDS Size')}")
