#!/bin/bash
# usage: tools/share_env.sh <N> "<VAR=val ...>" ...   (GPU box) — tools/share_rate.py for one rank count, once per environment given
n=$1; shift
mkdir -p gpurun_out/sharenv
for e in "$@"; do
  tag=$(echo "$e" | tr ' =/' '___')
  env RT_TUNING=1 $e timeout -k 10 120 python3 tools/share_rate.py $n > gpurun_out/sharenv/$tag.log 2>&1 || { echo "$e failed"; tail -3 gpurun_out/sharenv/$tag.log; exit 1; }
  echo "$e: $(grep '^N=' gpurun_out/sharenv/$tag.log | awk '{print $6}' | tr '\n' ' ') ms"
done
