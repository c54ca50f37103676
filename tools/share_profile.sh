#!/bin/bash
# Kernel times of one rank's share under rocprofv3: tools/share_profile.sh <out-name> <share_rate.py arguments...>
# (environment switches such as RT_SPLIT_BUDGET_MB are taken from the caller's environment)
name=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/r3/$name
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o p -- python3 "$root/tools/share_rate.py" "$@" > "$out/run.txt" 2>&1 || exit 1
f=$(find "$out" -name '*kernel_stats.csv' | head -n 1)
cp "$f" "$root/gpurun_out/r3/$name.kernel_stats.csv"
t=$(find "$out" -name '*kernel_trace.csv' | head -n 1)
python3 "$root/tools/trace_timeline.py" "$t" > "$root/gpurun_out/r3/$name.timeline.txt"
tail -n 3 "$out/run.txt"
head -n 8 "$f" | cut -c1-200
