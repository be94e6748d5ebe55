#!/bin/bash
# Same-box A/B of the training-step bench: runs bench.py alternately with each environment setting given as an argument
# ("VAR=0" "VAR=1" ...), ROUNDS times, and prints ms/step of every run (boxes of the pool differ by ~3 %: compare within one call).
#   bash scripts/ab_bench.sh "FEA_AMD_P_EPI2_128=0" "FEA_AMD_P_EPI2_128=1"
ROUNDS=${ROUNDS:-2}
STEPS=${STEPS:-20}
for r in $(seq $ROUNDS); do
  for setting in "$@"; do
    out=$(env $setting timeout -k 10 200 python3 bench.py --steps $STEPS --warmup 5 --no-cpu-baseline --no-kernel-timing) || exit 1
    echo "$setting round $r: $(echo "$out" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.2f ms/step  %.1f img/s" % (d["ms_per_step"], d["value"]))')"
  done
done
