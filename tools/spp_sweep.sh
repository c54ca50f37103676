for spp in 64 128 256 512 1024 2048; do
  timeout -k 10 200 python3 bench.py --spp $spp --steps 2 --no-cpu-baseline > gpurun_out/spp_$spp.json 2>/dev/null || exit 1
  python3 - gpurun_out/spp_$spp.json $spp <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
spp=int(sys.argv[2])
print("spp", spp, "value", round(j["value"]), "kernel_ms", round(j["roofline"]["kernel_ms"],2), "ms/spp", round(j["roofline"]["kernel_ms"]/spp,4))
PY
done
