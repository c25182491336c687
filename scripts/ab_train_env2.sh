#!/bin/bash
# same-box A/B of one environment switch on the cfg3 training iteration with the privacy branch (scripts/bench_train.py --fb): bash scripts/ab_train_env2.sh VAR "v0 v1" [rounds]
VAR=$1; VALS=${2:-"0 1"}; R=${3:-2}
for r in $(seq 1 $R); do
  for v in $VALS; do
    echo -n "$VAR=$v round $r: "
    if [ "$v" = unset ]; then timeout -k 10 500 python scripts/bench_train.py --fb 2>&1 | tail -1 | cut -c1-120
    else env $VAR=$v timeout -k 10 500 python scripts/bench_train.py --fb 2>&1 | tail -1 | cut -c1-120; fi
  done
done
