#!/bin/bash
# SQ counters of the lab variants, one counter group per pass (never with other trace domains)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc_sq$i -- $R/tools/lab2 4096 2 > $R/gpurun_out/pmc_sq$i.log 2>&1 || echo "group $i failed"
done
python3 - <<PY
import csv,glob,collections
for i in range(1,4):
    for f in glob.glob("$R/gpurun_out/pmc_sq%d/**/*counter_collection.csv"%i, recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:64]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in acc.items():
            if "ntt14w" not in k: continue
            print(k.replace("void fhe::ntt14w_",""), {c.replace("SQ_",""): round(sum(x)/len(x)) for c,x in v.items()})
PY
