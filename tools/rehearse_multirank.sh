#!/bin/bash
# Rehearsal of bench.py's N > 1 flow on a one-GPU box: gloo backend, every rank on GPU 0.  Checks the flow, not the rate.
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/rehearse
run() {  # ranks, port, bench arguments
  local n=$1 port=$2; shift 2
  timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port "$port" \
    bench.py --gpus "$n" --backend gloo "$@" > "gpurun_out/rehearse/n${n}_$(echo "$@" | tr -c 'a-z0-9\n' '_').json" 2> "gpurun_out/rehearse/n${n}_$(echo "$@" | tr -c 'a-z0-9\n' '_').err"
}
run 2 29511 --steps 2 --warmup 1
run 2 29512 --steps 2 --warmup 1 --config c4 --rays 20000000
run 4 29513 --steps 1 --warmup 1 --config c5 --spp 256
# Eight ranks on one device: four runs of five passed; in the fifth all ranks sat inside gloo's reduce of the device
# tensors (stacks from RT_BENCH_WATCHDOG) - the rehearsal backend, not the library - so the script stops at four ranks.
cat gpurun_out/rehearse/*.json
