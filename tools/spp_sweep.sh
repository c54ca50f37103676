#!/bin/bash
# usage: tools/spp_sweep.sh [spp ...]   (GPU box) — the C2 frame at several sample counts: Mpaths/s and ms per sample
for spp in ${@:-4 8 16 32 48 64 96 128 256 512 1024}; do
  timeout -k 10 200 python3 bench.py --spp $spp --steps 3 --no-cpu-baseline --no-extras > gpurun_out/spp_$spp.json 2>gpurun_out/spp_$spp.err || { echo "spp $spp failed"; tail -2 gpurun_out/spp_$spp.err; exit 1; }
  python3 - gpurun_out/spp_$spp.json $spp <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
spp = int(sys.argv[2])
print("spp", spp, "Mpaths/s", round(j["value"]), "ms", round(j["ms_per_step"], 3), "ms/spp", round(j["ms_per_step"] / spp, 4), j["roofline"]["kernel"], flush=True)
PY
done
