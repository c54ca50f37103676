#!/bin/bash
# usage: tools/fetch_calib.sh     (GPU box; build tools/microbench/fetch_calib with hipcc beforehand)
# Known-byte kernels under the FETCH_SIZE / WRITE_SIZE / fabric-request counters: one pass per counter group.
set -u
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/fetch_calib
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
bin=$root/tools/microbench/fetch_calib
timeout -k 10 120 $bin > $out/rates.txt 2> $out/rates.err || { echo "plain run failed"; tail -3 $out/rates.err; exit 1; }
i=0
for g in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  timeout -s KILL 120 rocprofv3 --pmc $g --output-format csv -d $out -o p$i -- $bin > $out/p$i.stdout 2> $out/p$i.stderr || { echo "pass $i failed"; tail -3 $out/p$i.stderr; exit 1; }
  i=$((i+1))
done
python3 $root/tools/fetch_calib_report.py $out > $out/report.txt && cat $out/report.txt
