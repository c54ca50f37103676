#!/bin/bash
# usage: tools/env_sweep.sh <ENV_NAME> "<values>" <bench args...>   (GPU box) — one bench.py run per value of the variable
name=$1; vals=$2; shift 2
mkdir -p gpurun_out/sweep
for v in $vals; do
  env RT_TUNING=1 $name=$v timeout -k 10 200 python3 bench.py "$@" --no-cpu-baseline > gpurun_out/sweep/${name}_$v.json 2> gpurun_out/sweep/${name}_$v.err || { echo "$name=$v failed"; tail -3 gpurun_out/sweep/${name}_$v.err; exit 1; }
  python3 - gpurun_out/sweep/${name}_$v.json $name $v <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = j["counters_per_launch"]
print(sys.argv[2], sys.argv[3], "value", round(j["value"], 1), j["unit"], "kernel_ms", round(j["roofline"]["kernel_ms"], 2),
      "node steps %.1fM x %.1f lanes" % (c["trav_wave_steps"] / 1e6, c["trav_lane_steps"] / max(c["trav_wave_steps"], 1)),
      "leaf steps %.1fM x %.1f lanes" % (c["leaf_wave_steps"] / 1e6, c["leaf_lane_steps"] / max(c["leaf_wave_steps"], 1)),
      "path %.1fM x %.1f lanes" % (c["path_wave_phases"] / 1e6, c["path_lane_phases"] / max(c["path_wave_phases"], 1)),
      "node_tests %.2fG sphere_tests %.2fG" % (c["node_tests"] / 1e9, c["sphere_tests"] / 1e9))
PY
done
