#!/bin/bash
# usage: tools/pmc_c4_sq.sh <tag> [rays]   (GPU box) — issue counters of the closest-hit kernel, one pass per group
set -u
tag=$1; rays=${2:-20000000}
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
groups=(
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
 "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS"
)
i=0
for g in "${groups[@]}"; do
  timeout -s KILL 180 rocprofv3 --pmc $g --output-format csv -d $out -o s$i -- python3 $GRAFT_REPO_ROOT/bench.py --config c4 --rays $rays --steps 1 --warmup 0 --no-cpu-baseline > $out/s$i.stdout 2> $out/s$i.stderr || { echo "group $i failed"; tail -3 $out/s$i.stderr; exit 1; }
  i=$((i+1))
done
python3 $GRAFT_REPO_ROOT/tools/pmc_report.py $out rt_trace_kernel | tee $out/summary.txt
grep -h counters_per_launch $out/s0.stdout | head -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print({k:v for k,v in j['counters_per_launch'].items() if v})"
