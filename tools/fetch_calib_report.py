#!/usr/bin/env python3
"""Counter readings of tools/microbench/fetch_calib against the bytes each kernel is known to move.

usage: tools/fetch_calib_report.py <dir with rates.txt and p*_counter_collection.csv>
Per kernel: the last dispatch's counters (the first launch warms nothing up that matters, the buffers exceed every
cache), FETCH_SIZE and WRITE_SIZE in bytes (rocprofv3 gives KiB) over the known bytes, fabric requests per 128-byte line.
"""
import csv, glob, os, sys

def main():
    d = sys.argv[1]
    known, rate = {}, {}
    for line in open(os.path.join(d, "rates.txt")):
        f = line.split()
        if len(f) >= 7 and f[1] == "known_bytes":
            known[f[0]] = float(f[2])
            rate[f[0]] = (float(f[4]), float(f[6]))
    vals = {}
    for path in sorted(glob.glob(os.path.join(d, "p*_counter_collection.csv"))):
        last = {}
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            last[(k, r["Counter_Name"])] = (int(r["Dispatch_Id"]), float(r["Counter_Value"]))
        for (k, c), (_, v) in last.items():
            vals.setdefault(k, {})[c] = v
    for k in known:
        v = vals.get(k, {})
        kb = known[k]
        print(f"{k}: known {kb / 2**30:.3f} GiB, {rate[k][0]:.3f} ms, {rate[k][1]:.0f} GB/s of known bytes")
        if "FETCH_SIZE" in v:
            print(f"   FETCH_SIZE {v['FETCH_SIZE'] * 1024 / 2**30:.3f} GiB = {v['FETCH_SIZE'] * 1024 / kb:.3f} x known")
        if "WRITE_SIZE" in v:
            print(f"   WRITE_SIZE {v['WRITE_SIZE'] * 1024 / 2**30:.3f} GiB = {v['WRITE_SIZE'] * 1024 / kb:.3f} x known")
        for c in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"):
            if c in v:
                print(f"   {c} {v[c]:.0f} = {v[c] / (kb / 128):.3f} per 128-byte line of known bytes")

main()
