#!/bin/bash
# A/B of environment switches on the cfg4 block of bench.py: usage  ckks_ab.sh "VAR=val" "VAR=val" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
for spec in "" "$@"; do
  out=$(env $spec python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | grep '^{')
  echo "[$spec] $(echo "$out" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); c=d["ckks"]; print(c["key_switches_per_sec_batch8"], c["key_switches_per_sec_batch64"], c["muls_per_sec_batch16"])')"
done
