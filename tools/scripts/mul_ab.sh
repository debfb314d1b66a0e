#!/bin/bash
# A/B of environment switches on the ring-product block of bench.py: usage  mul_ab.sh "VAR=val" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
for spec in "" "$@"; do
  out=$(env $spec python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | grep '^{')
  echo "[$spec] $(echo "$out" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ntt_mul"]["products_per_sec"], d["value"])')"
done
