cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_rns -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/prof_rns.log 2>&1
python3 - <<PY
import csv,glob
f=sorted(glob.glob("$R/gpurun_out/prof_rns/**/*kernel_stats.csv",recursive=True))[-1]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if any(k in n for k in ("rns_","true, 3>","R0")) or "ntt14w" in n: print(n[:100], r["Calls"], r["AverageNs"], r["MaxNs"])
PY
grep -o '"ckks": {[^}]*}' $R/gpurun_out/prof_rns.log | cut -c1-300
