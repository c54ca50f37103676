#!/bin/bash
# usage: tools/pmc_c4_mem.sh <tag> [rays]   (GPU box) — memory-pipeline counters of the closest-hit kernel, one pass per group
set -u
tag=$1; rays=${2:-20000000}
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
groups=(
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"
 "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCP_LATENCY_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum GRBM_GUI_ACTIVE"
)
i=0
for g in "${groups[@]}"; do
  timeout -s KILL 180 rocprofv3 --pmc $g --output-format csv -d $out -o m$i -- python3 $GRAFT_REPO_ROOT/bench.py --config c4 --rays $rays --steps 1 --warmup 0 --no-cpu-baseline > $out/m$i.stdout 2> $out/m$i.stderr || { echo "group $i failed"; tail -3 $out/m$i.stderr; exit 1; }
  i=$((i+1))
done
ls $out
