#!/bin/bash
# kernel-trace stats of the secondary workloads (cfg3 / cfg4 / cfg5 blocks of bench.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats_secondary -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/prof_stats_secondary.log 2>&1
echo done
