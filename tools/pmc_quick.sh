#!/bin/bash
# usage: tools/pmc_quick.sh <tag> [quick_c2.py args...]   (on the GPU box; one rocprofv3 --pmc pass per counter group
# over tools/quick_c2.py --no-check --frames 1).  Writes gpurun_out/pmc_<tag>/ and prints the per-kernel averages.
set -u
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
groups=(
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
 "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS"
)
i=0
for g in "${groups[@]}"; do
  timeout -s KILL 120 rocprofv3 --pmc $g --output-format csv -d $out -o g$i -- python3 $GRAFT_REPO_ROOT/tools/quick_c2.py --no-check --frames 1 "$@" > $out/g$i.stdout 2> $out/g$i.stderr || { echo "group $i failed"; tail -3 $out/g$i.stderr; exit 1; }
  i=$((i+1))
done
python3 $GRAFT_REPO_ROOT/tools/pmc_report.py $out rt_render_ | tee $out/summary.txt
