#!/bin/bash
# per-kernel times of the cfg4 key switch at batch 64 (tools/ckks_lab.py): usage  prof_ckks.sh <tag> [ckks_lab arguments]
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-ckks}; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/tools/ckks_lab.py --batch 64 --reps 20 "$@" > $R/gpurun_out/prof_$tag.log 2>&1
python3 - <<PY
import csv,glob
f=sorted(glob.glob("$R/gpurun_out/prof_$tag/**/*kernel_stats.csv",recursive=True))[-1]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if "rns_" in n or "ntt14w" in n: print(n[:60], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
tail -1 $R/gpurun_out/prof_$tag.log
