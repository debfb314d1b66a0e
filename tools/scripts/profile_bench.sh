#!/bin/bash
# Collects, on the GPU box, what tools/summarize_profiles.py condenses into profiles/: kernel-trace stats of the default
# bench command, and FETCH_SIZE / WRITE_SIZE in separate counter passes (MI355X_MICROARCH.md: TCC slots), plus the
# calibration copies.  usage (from the repo root on the box): bash tools/scripts/profile_bench.sh
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-fhew --steps 10 --warmup 2"
# kernel-trace stats of THE default bench command (what the driver runs); the counter passes below use a short run
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python3 $R/bench.py > $R/gpurun_out/prof_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_pmc_fetch -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_pmc_write -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_pmc_write.log 2>&1
if [ -x $R/tools/lab ]; then
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_calib_fetch -- $R/tools/lab calib > $R/gpurun_out/prof_calib_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_calib_write -- $R/tools/lab calib > $R/gpurun_out/prof_calib_write.log 2>&1
fi
echo profile_bench done
