#!/bin/bash
# usage: tools/collect_profiles.sh <tag>   (here, after tools/profile_round.sh r3prof + tools/pmc_quick.sh r3c2 ran on the
# GPU box): copies the bench lines, kernel statistics and counter summaries into profiles/ under the round's tag and
# stamps the two replayed JSON files with the commit and the kernel sources' hash.
set -e
tag=${1:-r3}
cd "$(dirname "$0")/.."
R=gpurun_out/${tag}prof
for c in default c1 c2_lbvh c2_linear c3 c4 c5_1gpu; do tail -n 1 $R/bench_$c.json > profiles/${tag}_bench_$c.json; done
cp "$(find $R/trace_c2_lbvh -name '*kernel_stats.csv' | head -n1)" profiles/${tag}_c2_lbvh_kernel_stats.csv
cp "$(find $R/trace_c2_linear -name '*kernel_stats.csv' | head -n1)" profiles/${tag}_c2_linear_kernel_stats.csv
cp "$(find $R/trace_c4 -name '*kernel_stats.csv' | head -n1)" profiles/${tag}_c4_kernel_stats.csv
cp gpurun_out/pmc_${tag}c2/summary.txt profiles/${tag}_c2_lbvh_pmc_sq.txt
python3 - "$R" <<'PY'
import json, subprocess, sys
commit = subprocess.check_output(['git', 'rev-parse', '--short', 'HEAD']).decode().strip()
d = json.load(open(sys.argv[1] + '/pmc_traffic.json'))
d['_meta']['commit'] = commit
json.dump(d, open('profiles/pmc_traffic.json', 'w'), indent=1)
print(d['_meta'])
PY
python3 tools/issue_rate.py profiles/${tag}_c2_lbvh_pmc_sq.txt > profiles/c2_issue.json
python3 -c "from raytrace_amd.build import kernel_sources_sha; print('sources now:', kernel_sources_sha())"
for c in default c2_lbvh c2_linear c3 c4 c1 c5_1gpu; do python3 -c "
import json,sys; b=json.loads(open('profiles/${tag}_bench_$c.json').read()); print('$c', round(b['value'],1), b['unit'], round(b['ms_per_step'],3), 'ms')"; done
