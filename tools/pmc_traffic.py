#!/usr/bin/env python3
"""HBM traffic per bench step from the two counter passes of tools/profile_round.sh.

usage: tools/pmc_traffic.py <run dir> <name>...      (name = c2_lbvh, c2_linear, c3, c4)

Each pass ran `bench.py --steps 1 --warmup 0`: one counted step, then the timed step.  The timed step's dispatches
are the last render / trace dispatch that is not a counted one, everything after it, and everything before it back to
the end of the counted step (its last dispatch is rt_sum_split_kernel, or the counted kernel itself).
Bytes per step = sum over kernels of (factor * FETCH_SIZE + WRITE_SIZE) * 1024: rocprofv3 reports both in KiB.  The
factor comes from tools/microbench/fetch_calib (known-byte kernels under the same counters on this GPU,
profiles/r2_fetch_calibration.txt): FETCH_SIZE tallies one fabric read request at 64 bytes whatever its size — a
streamed read (16 bytes per lane, or 32-byte pairs) makes one 128-byte request per line and is reported at half its
bytes (factor 2), a 64-byte node gather of four lanes makes one 64-byte request and is reported exactly (factor 1).
WRITE_SIZE is exact for every pattern tried.  The closest-hit kernel's reads are node and leaf gathers (factor 1) but
for its streamed ray input, 24 bytes per ray, of which the counter shows half: the other half is added from the ray
count.  Every other kernel gets factor 2, an upper bound for whatever narrow reads it makes.
Prints one JSON object keyed by "<kernel>|<workload>" for profiles/pmc_traffic.json.
"""
import csv, json, os, sys

MAIN = ("void rt_render_kernel<", "void rt_render_stored_kernel<", "void rt_trace_kernel<")


def counted(name):  # rt_render_kernel<search, rng, counted, noise>, rt_render_stored_kernel<search, counted, noise, owned>, rt_trace_kernel<search, counted>
    if not name.startswith(MAIN):
        return False
    args = [a.strip() for a in name[name.index("<") + 1:name.index(">")].split(",")]
    return args[2 if name.startswith("void rt_render_kernel<") else 1] == "true"


def timed_step_rows(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    last_counted = max((i for i, r in enumerate(rows) if counted(r["Kernel_Name"])), default=-1)
    start = last_counted + 1
    # the counted step ends with the sum of the samples its counted main pass stored
    if start < len(rows) and rows[start]["Kernel_Name"].startswith("rt_sum_split_kernel"):
        start += 1
    return rows[start:]


def read_factor(kernel):
    return 1.0 if kernel.startswith("rt_trace_kernel<") else 2.0


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
        requests = {}
        total = sum((read_factor(k) * v[0] + v[1]) * 1024 for k, v in per_kernel.items())
        rays = bench["config"].get("rays")
        if rays:  # the streamed half of the ray input that the counter does not show
            total += 12.0 * rays
            requests["rays_per_step"] = rays
        req_path = os.path.join(d, "req_counter_collection.csv")
        if os.path.exists(req_path):
            for counter, label in (("TCC_EA0_RDREQ_sum", "fabric_read_requests_per_step"), ("TCC_REQ_sum", "l2_requests_per_step"),
                                   ("TCC_HIT_sum", "l2_hits_per_step")):
                requests[label] = sum(float(r["Counter_Value"]) for r in timed_step_rows(req_path, counter))
        key = bench["roofline"]["kernel"] + "|" + bench["config"]["workload"]
        out[key] = {
            "hbm_bytes_per_step": total,
            "uncalibrated_2x_fetch_plus_write": (2 * f_kib + w_kib) * 1024,
            "read_factor_per_kernel": {k: read_factor(k) for k in per_kernel},
            "fetch_size_kib": f_kib, "write_size_kib": w_kib,
            "dispatches_per_step": len(fetch), **requests,
            "per_kernel_kib_fetch_write": {k: v for k, v in per_kernel.items()},
            "source": f"rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE / --pmc WRITE_SIZE, bench.py --steps 1 --warmup 0, timed step only",
        }
    import datetime, subprocess
    try:
        commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=os.path.dirname(os.path.abspath(__file__))).decode().strip()
    except Exception:
        commit = "?"
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from raytrace_amd.build import kernel_sources_sha
    out["_meta"] = {"date": datetime.date.today().isoformat(), "commit": commit, "kernel_sources_sha": kernel_sources_sha(),
                    "command": "tools/profile_round.sh (rocprofv3 --pmc passes of bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline)"}
    print(json.dumps(out, indent=1))

if __name__ == "__main__":
    main()
