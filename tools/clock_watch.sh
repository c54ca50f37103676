#!/bin/bash
# usage: tools/clock_watch.sh <tag> <bench args...>  -- samples rocm-smi clocks and power while bench.py runs (GPU box)
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/clk_$tag
mkdir -p $out
timeout -k 10 300 python3 $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu-baseline > $out/bench.json 2> $out/bench.err &
pid=$!
for i in $(seq 1 200); do
  kill -0 $pid 2>/dev/null || break
  { date +%s.%N; timeout 5 rocm-smi --showclocks --showpower --showuse 2>&1 | grep -E "sclk|mclk|fclk|Power|GPU use|busy"; } >> $out/smi.log
  sleep 0.3
done
wait $pid
echo rc=$?
tail -c 600 $out/bench.json
