#!/bin/bash
# usage: tools/pmc.sh <tag> <bench args...>   (runs on the GPU box; one rocprofv3 --pmc pass per counter group)
# Writes gpurun_out/pmc_<tag>/<group>_counter_collection.csv
set -u
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
groups=(
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
 "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32"
 "GRBM_GUI_ACTIVE FETCH_SIZE"
 "WRITE_SIZE"
)
i=0
for g in "${groups[@]}"; do
  timeout -s KILL 180 rocprofv3 --pmc $g --output-format csv -d $out -o g$i -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu-baseline > $out/g$i.stdout 2> $out/g$i.stderr || { echo "group $i failed"; tail -3 $out/g$i.stderr; exit 1; }
  i=$((i+1))
done
ls $out | head -30
