#!/bin/bash
# usage: tools/variant_sweep.sh <tag> "<variant names>" <bench args...>   (GPU box) — one bench.py run per variant library
tag=$1; names=$2; shift 2
out=gpurun_out/var_$tag; mkdir -p $out
for n in $names; do
  RAYTRACE_AMD_LIB=$PWD/variants/$n.so timeout -k 10 200 python3 bench.py "$@" --no-cpu-baseline > $out/$n.json 2> $out/$n.err || { echo "$n failed"; tail -3 $out/$n.err; exit 1; }
  python3 - $out/$n.json $n <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = j["counters_per_launch"]
print(sys.argv[2], "value", round(j["value"], 1), j["unit"], "kernel_ms", round(j["roofline"]["kernel_ms"], 2), "frac", round(j["roofline"]["frac"], 4),
      "segments %.3fG node steps %.1fM path phases %.1fM ns/segment %.4f" % (c["segments"] / 1e9, c["trav_wave_steps"] / 1e6, c["path_wave_phases"] / 1e6,
                                                                  j["roofline"]["kernel_ms"] * 1e6 / max(c["segments"], 1)))
PY
done
