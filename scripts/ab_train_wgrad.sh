#!/bin/bash
# same-box A/B of the cfg3 training iteration (UNet + I3Res50 + fb) on the patch weight-gradient kernel's loader waves: bash scripts/ab_train_wgrad.sh "0 4" [rounds]
R=${2:-2}
for r in $(seq 1 $R); do
  for v in ${1:-0 4}; do
    echo -n "TEDSPAD_WGRAD3P_LOADERS=$v round $r: "
    TEDSPAD_WGRAD3P_LOADERS=$v timeout -k 10 500 python scripts/bench_train.py --fb 2>&1 | tail -1 | cut -c1-300
  done
done
