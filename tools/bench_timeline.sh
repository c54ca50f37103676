#!/bin/bash
# usage: tools/bench_timeline.sh <name> <bench.py arguments...>   (GPU box) — the kernels of the last frame of a bench
# run, one after the other with their durations and the gaps between them (tools/trace_timeline.py on a kernel trace)
name=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/r3/bt_$name
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$out" -o t -- python3 "$root/bench.py" "$@" --no-cpu-baseline --no-extras > "$out/bench.json" 2> "$out/bench.err" || { tail -n 3 "$out/bench.err"; exit 1; }
python3 "$root/tools/trace_timeline.py" "$(find "$out" -name '*kernel_trace.csv' | head -n 1)" | grep -v "trampoline\|fillBuffer\|init_lookback"
