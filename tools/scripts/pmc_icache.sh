#!/bin/bash
# instruction-cache counters of the transform kernels (default bench command, short)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_icache
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_icache -- python3 $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 --preheat-ms 0 > $R/gpurun_out/pmc_icache.log 2>&1
python3 - <<PY
import csv,glob,collections
f=sorted(glob.glob("$R/gpurun_out/pmc_icache/**/*counter_collection.csv",recursive=True))[-1]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); best={}
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    if "ntt14w" not in n and "blind_rotate_kernel" not in n and "rns_" not in n: continue
    acc[(n[:95], r["Dispatch_Id"])][r["Counter_Name"]]+=float(r["Counter_Value"])
for (n,d),c in acc.items():
    if n not in best or c["SQ_WAVES"]>best[n]["SQ_WAVES"]: best[n]=c
for n,c in best.items():
    req=c["SQC_ICACHE_REQ"] or 1
    print(n); print("   waves %d  icache req/wave %.0f  hit %.3f  miss %.3f  dup-miss %.3f  ifetch/wave %.0f" % (c["SQ_WAVES"], req/(c["SQ_WAVES"] or 1), c["SQC_ICACHE_HITS"]/req, c["SQC_ICACHE_MISSES"]/req, c["SQC_ICACHE_MISSES_DUPLICATE"]/req, c["SQ_IFETCH"]/(c["SQ_WAVES"] or 1)))
PY
