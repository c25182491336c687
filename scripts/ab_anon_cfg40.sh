#!/bin/bash
# same-box A/B of anonymised extraction (unet++ fa + I3Res50) with and without tile_cfg 40 in the tuner's candidate list: bash scripts/ab_anon_cfg40.sh [rounds]
R=${1:-2}
for r in $(seq 1 $R); do
  for skip in 40 ""; do
    echo -n "skip=[$skip] round $r: "
    TEDSPAD_SKIP_CFGS=$skip timeout -k 10 400 python scripts/bench_anon_extract.py --arch-fa unet++ --clips 150 --batch 75 --steps 3 2>&1 | tail -1
  done
done
