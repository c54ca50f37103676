#!/usr/bin/env python3
"""HBM traffic per bench step from the two counter passes of tools/profile_round.sh.

usage: tools/pmc_traffic.py <run dir> <name>...      (name = c2_lbvh, c2_linear, c3, c4)

Each pass ran `bench.py --steps 1 --warmup 0`: one counted step, then the timed step.  The
timed step's dispatches are the ones after the last counted render / trace dispatch.  Per
step, bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: rocprofv3 reports both in KiB, and on
gfx950 FETCH_SIZE tallies 128-byte read requests at 64 bytes (an upper bound for narrow
reads).  Prints one JSON object keyed by "<kernel>|<workload>" for profiles/pmc_traffic.json.
"""
import csv, json, os, sys

def timed_step_rows(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    def counted(name):  # rt_render_kernel<search, rng, counted, noise>, rt_render_stored_kernel<search, counted, noise>, rt_trace_kernel<search, counted>
        if not name.startswith(("void rt_render_kernel<", "void rt_render_stored_kernel<", "void rt_trace_kernel<")):
            return False
        args = [a.strip() for a in name[name.index("<") + 1:name.index(">")].split(",")]
        return (args[-1] if name.startswith("void rt_trace_kernel<") else args[-2]) == "true"
    last_counted = max((i for i, r in enumerate(rows) if counted(r["Kernel_Name"])), default=-1)
    return rows[last_counted + 1:]

def main():
    run, names = sys.argv[1], sys.argv[2:]
    out = {}
    for name in names:
        d = os.path.join(run, "pmc_" + name)
        bench = json.loads(open(os.path.join(run, f"pmc_{name}.fetch.json")).read().strip().splitlines()[-1])
        fetch = timed_step_rows(os.path.join(d, "fetch_counter_collection.csv"), "FETCH_SIZE")
        write = timed_step_rows(os.path.join(d, "write_counter_collection.csv"), "WRITE_SIZE")
        f_kib = sum(float(r["Counter_Value"]) for r in fetch)
        w_kib = sum(float(r["Counter_Value"]) for r in write)
        per_kernel = {}
        for r in fetch:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            per_kernel.setdefault(k, [0.0, 0.0])[0] += float(r["Counter_Value"])
        for r in write:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            per_kernel.setdefault(k, [0.0, 0.0])[1] += float(r["Counter_Value"])
        key = bench["roofline"]["kernel"] + "|" + bench["config"]["workload"]
        out[key] = {
            "hbm_bytes_per_step": (2 * f_kib + w_kib) * 1024,
            "fetch_size_kib": f_kib, "write_size_kib": w_kib,
            "dispatches_per_step": len(fetch),
            "per_kernel_kib_fetch_write": {k: v for k, v in per_kernel.items()},
            "source": f"rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE / --pmc WRITE_SIZE, bench.py --steps 1 --warmup 0, timed step only",
        }
    import datetime, subprocess
    try:
        commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=os.path.dirname(os.path.abspath(__file__))).decode().strip()
    except Exception:
        commit = "?"
    out["_meta"] = {"date": datetime.date.today().isoformat(), "commit": commit,
                    "command": "tools/profile_round.sh (rocprofv3 --pmc passes of bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline)"}
    print(json.dumps(out, indent=1))

main()
