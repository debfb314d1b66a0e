#!/bin/bash
# SQ counters of the cfg4 kernels (bench.py secondary block), one counter pass; prints per-kernel sums for the batch-64 launches
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_rns
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_rns -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 --preheat-ms 0 > $R/gpurun_out/pmc_rns.log 2>&1
python3 - <<PY
import csv,glob,collections
f=sorted(glob.glob("$R/gpurun_out/pmc_rns/**/*counter_collection.csv",recursive=True))[-1]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); best={}
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    if not any(k in n for k in ("rns_","ntt14w")): continue
    key=(n[:90], r["Dispatch_Id"])
    acc[key][r["Counter_Name"]]+=float(r["Counter_Value"])
# keep, per kernel name, the dispatch with the most waves
for (n,d),c in acc.items():
    if n not in best or c["SQ_WAVES"]>best[n]["SQ_WAVES"]: best[n]=c
for n,c in best.items():
    w=c["SQ_WAVES"] or 1
    print(n); print("   waves %d  VALU/wave %.0f  wave_cycles/wave %.0f  wait_any %.2f  wait_inst %.2f  active_any %.2f  active_valu(/busy) %.2f" % (w, c["SQ_INSTS_VALU"]/w, c["SQ_WAVE_CYCLES"]/w, c["SQ_WAIT_ANY"]/c["SQ_WAVE_CYCLES"], c["SQ_WAIT_INST_ANY"]/c["SQ_WAVE_CYCLES"], c["SQ_ACTIVE_INST_ANY"]/c["SQ_WAVE_CYCLES"], c["SQ_ACTIVE_INST_VALU"]/(c["SQ_BUSY_CYCLES"] or 1)))
PY
