#!/usr/bin/env python3
"""VALU instructions per bench step of the headline kernel, from a counter summary of tools/pmc_quick.sh
(profiles/r2_c2_lbvh_pmc_sq.txt: per-dispatch averages over whole frames of two render dispatches each), as the JSON
bench.py reads (profiles/c2_issue.json), stamped with the commit and the hash of the kernel sources it was taken at.   usage: tools/issue_rate.py <summary.txt> > profiles/c2_issue.json"""
import datetime, json, os, re, subprocess, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytrace_amd.build import kernel_sources_sha  # noqa: E402

vals = {}
for line in open(sys.argv[1]):
    m = re.match(r"\s+(SQ_\w+)\s+([0-9.]+)", line)
    if m:
        vals[m.group(1)] = float(m.group(2))
per_frame = 2  # probe pass + main pass
out = {"rt_render_stored_kernel<lbvh4,lds,pad_ray>|487-sphere scene 1920x1080x256spp depth50": {
    "valu_insts_per_step": vals["SQ_INSTS_VALU"] * per_frame,
    "salu_insts_per_step": vals["SQ_INSTS_SALU"] * per_frame,
    "lds_insts_per_step": vals["SQ_INSTS_LDS"] * per_frame,
    "branch_insts_per_step": vals["SQ_INSTS_BRANCH"] * per_frame,
    "active_lanes_per_valu_inst": vals["SQ_THREAD_CYCLES_VALU"] / vals["SQ_ACTIVE_INST_VALU"],
    "issuing_fraction_of_wave_cycles": vals["SQ_ACTIVE_INST_ANY"] / vals["SQ_WAVE_CYCLES"],
    "source": "rocprofv3 --pmc SQ_* passes over tools/quick_c2.py (tools/pmc_quick.sh), wave-instructions of the two render dispatches of a frame"}}
try:
    commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=os.path.dirname(os.path.abspath(__file__))).decode().strip()
except Exception:
    commit = "?"
out["_meta"] = {"date": datetime.date.today().isoformat(), "commit": commit, "kernel_sources_sha": kernel_sources_sha()}
print(json.dumps(out, indent=1))
