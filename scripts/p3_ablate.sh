#!/bin/bash
# timing ablations of tile 40 (wrong results): bash scripts/p3_ablate.sh
for spec in "400,1,112,112 64 64" "400,1,112,112 320 64"; do
  set -- $spec
  for ab in 0 32 64 128 256 384; do
    echo -n "cin $2 ablate $ab: "
    TEDSPAD_P3_ABLATE=$ab timeout -k 10 120 python scripts/conv_probe.py --dims $1 --cin $2 --cout $3 --cfg 40 --k 1,3,3 --pads 0,1,1 --reps 10 2>&1 | tail -1
  done
done
