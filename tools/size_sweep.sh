#!/bin/bash
# usage: tools/size_sweep.sh   (GPU box) — rate of the 487-sphere scene at several frame sizes and sample counts, and of the Perlin scene
cd "$(dirname "$0")/.."
show='import json,sys; j=json.loads(sys.stdin.read()); print(j["config"]["workload"], round(j["value"]), "Mpaths/s", round(j["ms_per_step"],2), "ms")'
for a in "1920 1080 256" "3840 2160 256" "1920 1080 64" "960 540 256" "2560 1440 256" "1920 1080 1024"; do
  set -- $a
  timeout -k 10 300 python3 bench.py --width $1 --height $2 --spp $3 --steps 2 --no-cpu-baseline | python3 -c "$show" || exit 1
done
timeout -k 10 200 python3 bench.py --config c3 --steps 2 --no-cpu-baseline | python3 -c "$show"
