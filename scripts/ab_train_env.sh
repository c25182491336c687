#!/bin/bash
# Same-box A/B of one environment switch on the cfg3 training iteration (bench.py --train: utility terms) : bash scripts/ab_train_env.sh VAR "v0 v1" [rounds]
VAR=$1; VALS=${2:-"0 1"}; R=${3:-2}
for r in $(seq 1 $R); do
  for v in $VALS; do
    env $VAR=$v timeout -k 10 300 python bench.py --train --steps 8 --warmup 2 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v round $r: iteration ms', d['value'], flush=True)"
  done
done
