#!/bin/bash
# same-box A/B of the cfg3 training iteration (UNet + I3Res50 + fb) with and without tile_cfg 40 among the tuner's candidates: bash scripts/ab_train_cfg40.sh [rounds]
R=${1:-1}
for r in $(seq 1 $R); do
  for skip in 40 ""; do
    echo -n "skip=[$skip] round $r: "
    TEDSPAD_SKIP_CFGS=$skip timeout -k 10 500 python scripts/bench_train.py --fb 2>&1 | tail -1 | cut -c1-400
  done
done
