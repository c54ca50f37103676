#!/usr/bin/env python3
"""Static instruction budget of one kernel from the compiler's assembly with line tables.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
        --cuda-device-only -S -gline-tables-only -o rt_api.s raytrace_amd/csrc/rt_api.hip
  python tools/isa_budget.py rt_api.s _Z23rt_render_stored_kernelILi1ELb0EEv11render_args

Every instruction of the kernel is attributed to the source function whose body holds the line its `.loc` names (the
innermost inlined function), and counted by issue class: VALU (of which transcendental: rcp / rsq / sqrt / sin / cos /
exp / log, and 64-bit float), SALU (of which s_nop and s_waitcnt), LDS, VMEM (global / buffer / scratch), branch.
Functions are found in the sources by their definitions' first lines (a crude scan: enough for these files).
"""
import collections
import os
import re
import sys

DEF = re.compile(r"^\s*(?:template\s*<[^>]*>\s*)?(?:RT_DEV|__global__|__device__|static|inline|__forceinline__|__launch_bounds__\([^)]*\)|\s)*"
                 r"[A-Za-z_][\w:<>,\s\*&]*?\b([A-Za-z_]\w*)\s*\([^;]*$")


def function_ranges(path):
    """[(first line, last line, name)] of the function bodies in a source file, by brace depth."""
    out, stack, depth = [], [], 0
    pending = None
    for no, line in enumerate(open(path, errors="replace"), 1):
        code = line.split("//")[0]
        if depth <= 1 and pending is None:
            m = DEF.match(code)
            if m and m.group(1) not in ("if", "for", "while", "switch", "return", "sizeof", "defined"):
                pending = (no, m.group(1))
        for ch in code:
            if ch == "{":
                depth += 1
                if pending is not None:
                    stack.append((depth, pending))
                    pending = None
            elif ch == "}":
                if stack and stack[-1][0] == depth:
                    _, (first, name) = stack.pop()
                    out.append((first, no, name))
                depth -= 1
        if pending is not None and code.rstrip().endswith(";"):
            pending = None
    return out


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_cbranch") or op in ("s_branch", "s_setpc_b64", "s_endpgm"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return "vmem"
    return None


def main():
    asm, kernel = sys.argv[1], sys.argv[2]
    files, ranges = {}, {}
    cur = ("?", 0)
    counts = collections.defaultdict(lambda: collections.Counter())
    inside = False
    for line in open(asm):
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', line)
        if m:
            files[int(m.group(1))] = os.path.join(m.group(2), m.group(3))
            continue
        if line.startswith(kernel + ":"):
            inside = True
            continue
        if not inside:
            continue
        if line.startswith(".Lfunc_end"):
            break
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", line)
        if m:
            cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
            continue
        t = line.strip().split(" ")[0] if line.strip() else ""
        kind = classify(t)
        if kind is None:
            continue
        path, no = cur
        if path not in ranges:
            ranges[path] = function_ranges(path) if os.path.exists(path) else []
        best = None
        for first, last, name in ranges[path]:
            if first <= no <= last and (best is None or first >= best[0]):
                best = (first, last, name)
        where = (os.path.basename(path), best[2] if best else f"line {no}")
        c = counts[where]
        c[kind] += 1
        if kind == "valu" and re.match(r"v_(rcp|rsq|sqrt|sin|cos|exp|log)", t):
            c["trans"] += 1
        if kind == "valu" and "_f64" in t:
            c["f64"] += 1
        if t == "s_nop":
            c["nop"] += 1
        if t == "s_waitcnt":
            c["wait"] += 1
    total = collections.Counter()
    print(f"{'file':18s} {'function':28s} {'VALU':>6s} {'trans':>5s} {'f64':>5s} {'SALU':>6s} {'nop':>4s} {'wait':>4s} {'LDS':>5s} {'VMEM':>5s} {'branch':>6s}")
    for (f, fn), c in sorted(counts.items(), key=lambda kv: -kv[1]["valu"]):
        print(f"{f:18s} {fn:28s} {c['valu']:6d} {c['trans']:5d} {c['f64']:5d} {c['salu']:6d} {c['nop']:4d} {c['wait']:4d} {c['lds']:5d} {c['vmem']:5d} {c['branch']:6d}")
        total.update(c)
    print(f"{'':18s} {'total':28s} {total['valu']:6d} {total['trans']:5d} {total['f64']:5d} {total['salu']:6d} {total['nop']:4d} {total['wait']:4d} {total['lds']:5d} {total['vmem']:5d} {total['branch']:6d}")


if __name__ == "__main__":
    main()
