#!/bin/bash
# usage: tools/profile_round.sh <tag> [all|bench|pmc]      (GPU box)
# Kernel-trace statistics of the default bench command and of the linear-scan one, bench lines
# of the other configurations, and the two HBM-traffic counter passes per workload.
set -u
tag=$1
what=${2:-all}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { echo "== $*"; "$@" || { echo "FAILED: $*"; exit 1; }; }
if [ $what != pmc ]; then
run timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_c2_lbvh -o t -- python3 $root/bench.py --no-extras > $out/bench_c2_lbvh.json 2> $out/bench_c2_lbvh.err
run timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_c2_linear -o t -- python3 $root/bench.py --accel linear --steps 3 --no-cpu-baseline --no-extras > $out/bench_c2_linear.json 2> $out/bench_c2_linear.err
run timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_c4 -o t -- python3 $root/bench.py --config c4 --steps 3 > $out/bench_c4.json 2> $out/bench_c4.err
run timeout -k 10 200 python3 $root/bench.py --config c1 --steps 20 > $out/bench_c1.json 2> $out/bench_c1.err
run timeout -k 10 300 python3 $root/bench.py --config c3 --steps 3 > $out/bench_c3.json 2> $out/bench_c3.err
run timeout -k 10 300 python3 $root/bench.py > $out/bench_default.json 2> $out/bench_default.err
run timeout -k 10 300 python3 $root/bench.py --config c5 --steps 2 --no-cpu-baseline > $out/bench_c5_1gpu.json 2> $out/bench_c5_1gpu.err
fi
[ $what = bench ] && exit 0
for w in "c2_lbvh --steps 1 --warmup 0" "c2_linear --accel linear --steps 1 --warmup 0" "c4 --config c4 --steps 1 --warmup 0" "c3 --config c3 --steps 1 --warmup 0"; do
  set -- $w; name=$1; shift
  run timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $out/pmc_$name -o fetch -- python3 $root/bench.py "$@" --no-cpu-baseline --no-extras > $out/pmc_$name.fetch.json 2> $out/pmc_$name.fetch.err
  run timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_$name -o write -- python3 $root/bench.py "$@" --no-cpu-baseline --no-extras > $out/pmc_$name.write.json 2> $out/pmc_$name.write.err
  # read requests of L2 to the fabric, and L2's own requests and hits (the closest-hit batch is bound by the first)
  run timeout -s KILL 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_HIT_sum --output-format csv -d $out/pmc_$name -o req -- python3 $root/bench.py "$@" --no-cpu-baseline --no-extras > $out/pmc_$name.req.json 2> $out/pmc_$name.req.err
done
python3 $root/tools/pmc_traffic.py $out c2_lbvh c2_linear c3 c4 > $out/pmc_traffic.json || echo "pmc_traffic.py failed"
ls -R $out | head -60
