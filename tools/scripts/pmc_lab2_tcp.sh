#!/bin/bash
# L1 (TCP) / L2 (TCC) request counters of the lab variants, one counter group per pass (never with other trace domains)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "TA_BUSY_avr TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc_tcp$i -- $R/tools/lab2 4096 2 > $R/gpurun_out/pmc_tcp$i.log 2>&1 || echo "group $i failed"
done
python3 - <<PY
import csv,glob,collections
for i in range(1,5):
    for f in glob.glob("$R/gpurun_out/pmc_tcp%d/**/*counter_collection.csv"%i, recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:110]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in acc.items():
            if "ntt14w" not in k: continue
            print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
