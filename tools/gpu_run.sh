#!/bin/bash
# Runs GPU steps one after another on the gpurun box, each under its own timeout; an ordinary failure
# (non-zero exit) is recorded and the next step still runs, a timeout/kill (124/137) stops everything.
# usage: tools/gpu_run.sh "<name>|<seconds>|<command>" ...
mkdir -p gpurun_out
status=0
for spec in "$@"; do
  name="${spec%%|*}"; rest="${spec#*|}"; secs="${rest%%|*}"; cmd="${rest#*|}"
  echo "=== [$name] $cmd" | tee -a gpurun_out/steps.log
  start=$(date +%s)
  timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "=== [$name] exit $rc after $(( $(date +%s) - start ))s" | tee -a gpurun_out/steps.log
  tail -n 15 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout/kill: stopping"; exit $rc; fi
  [ $rc -ne 0 ] && status=$rc
done
exit $status
