#!/bin/bash
# usage: tools/env_quick.sh "<VAR=val VAR2=val2>" ...   (GPU box) — tools/quick_c2.py --no-check once per environment given; best frame each
mkdir -p gpurun_out/envq
for e in "$@"; do
  tag=$(echo "$e" | tr ' =/' '___')
  env RT_TUNING=1 $e timeout -k 10 120 python3 tools/quick_c2.py --no-check --frames 4 > gpurun_out/envq/$tag.log 2>&1 || { echo "$e failed"; tail -3 gpurun_out/envq/$tag.log; exit 1; }
  echo "$e: $(grep '^c2' gpurun_out/envq/$tag.log | awk '{print $3}' | sort -n | head -1) ms kernel, $(grep '^c2' gpurun_out/envq/$tag.log | awk '{print $9}' | sort -n | tail -1) Mpaths/s best"
done
