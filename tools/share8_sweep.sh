mkdir -p gpurun_out/r3
export RT_TUNING=1
A="--config c2 --ranks 8 --whole-ms 51.75"
(
echo "== default"; timeout -k 10 60 python tools/share_rate.py $A | grep "^N=" &&
for mp in 10 12 16; do echo "== RT_MIN_PARTS=$mp"; RT_MIN_PARTS=$mp timeout -k 10 60 python tools/share_rate.py $A | grep "^N=" || exit 1; done &&
for ps in 8 10 16; do echo "== RT_PROBE_SAMPLES=$ps"; RT_PROBE_SAMPLES=$ps timeout -k 10 60 python tools/share_rate.py $A | grep "^N=" || exit 1; done &&
for sw in 4 12; do echo "== RT_PROBE_SWEEPS=$sw"; RT_PROBE_SWEEPS=$sw timeout -k 10 60 python tools/share_rate.py $A | grep "^N=" || exit 1; done &&
for f in 128 256; do echo "== RT_SPLIT_FINENESS=$f"; RT_SPLIT_FINENESS=$f timeout -k 10 60 python tools/share_rate.py $A | grep "^N=" || exit 1; done
) > gpurun_out/r3/share8_sweep.txt 2>&1
cat gpurun_out/r3/share8_sweep.txt
