#!/bin/bash
# isolated timings of the persistent two-patch tile (40) against tiles 32 / 38 on the anonymizers' wide 3 x 3 layers: bash scripts/p3_probe.sh
for spec in "400,1,112,112 64 64" "400,1,112,112 128 64" "400,1,112,112 192 64" "400,1,112,112 320 64" "100,1,224,224 64 64"; do
  set -- $spec
  for cfg in 32 38 40; do
    timeout -k 10 120 python scripts/conv_probe.py --dims $1 --cin $2 --cout $3 --cfg $cfg --k 1,3,3 --pads 0,1,1 --reps 10 2>&1 | tail -1
  done
done
