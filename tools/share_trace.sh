#!/bin/bash
# usage: tools/share_trace.sh <N> "<VAR=val ...>" ...   (GPU box) — kernel trace of tools/share_rate.py <N> per environment; prints the
# kernels of the last frame
root=$GRAFT_REPO_ROOT
n=$1; shift
cd /tmp && export TMPDIR=/tmp
for e in "$@"; do
  tag=$(echo "$e" | tr ' =/' '___')
  out=$root/gpurun_out/sharetrace/$tag
  mkdir -p $out
  export $e
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python3 $root/tools/share_rate.py $n > $out/out.txt 2>&1 || { echo "$e failed"; tail -3 $out/out.txt; exit 1; }
  unset ${e%%=*}
  echo "== $e"
  python3 - $out/t_kernel_trace.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = [r for r in rows[-60:] if "rocprim" not in r["Kernel_Name"] and "fillBuffer" not in r["Kernel_Name"]]
for r in last[-7:]:
    print(f"  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:9.1f} us  {r['Kernel_Name'][:60]}")
PY
done
