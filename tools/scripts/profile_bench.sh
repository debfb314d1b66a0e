#!/bin/bash
# Collects, on the GPU box, what tools/summarize_profiles.py condenses into profiles/: kernel-trace stats of the default
# bench command (headline and secondary workloads), and FETCH_SIZE / WRITE_SIZE / SQ counters in SEPARATE counter passes
# (MI355X_MICROARCH.md: TCC slots; --pmc never together with other trace domains), plus the calibration copies.
# usage (from the repo root on the box): bash tools/scripts/profile_bench.sh
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-fhew --steps 10 --warmup 2 --preheat-ms 0"
SEC="--no-cpu-baseline --no-verify-secondary --steps 2 --warmup 1 --preheat-ms 0"
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
# kernel-trace stats of THE default bench command (what the driver runs): headline + every secondary workload
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_stats.log 2>&1
echo stats done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_pmc_fetch -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_pmc_write -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $R/gpurun_out/prof_pmc_sq -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_pmc_sq.log 2>&1
echo headline counters done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_sec_fetch -- python3 $R/bench.py $SEC > $R/gpurun_out/prof_sec_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_sec_write -- python3 $R/bench.py $SEC > $R/gpurun_out/prof_sec_write.log 2>&1
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $R/gpurun_out/prof_sec_sq -- python3 $R/bench.py $SEC > $R/gpurun_out/prof_sec_sq.log 2>&1
echo secondary counters done
if [ -x $R/tools/lab2 ]; then
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_calib_fetch -- $R/tools/lab2 calib > $R/gpurun_out/prof_calib_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_calib_write -- $R/tools/lab2 calib > $R/gpurun_out/prof_calib_write.log 2>&1
fi
echo profile_bench done
