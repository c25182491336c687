#!/bin/bash
# tile 40 as two 64-channel launches on the cout = 128 layers against the flat / patch tiles (31 / 33 / 34): bash scripts/p3_probe128.sh
for spec in "150,1,112,112 64 128" "150,1,112,112 128 128" "150,1,112,112 256 128" "150,1,112,112 384 128" "384,1,56,56 128 128" "384,1,56,56 256 128"; do
  set -- $spec
  for cfg in 31 33 34 40; do
    timeout -k 10 120 python scripts/conv_probe.py --dims $1 --cin $2 --cout $3 --cfg $cfg --k 1,3,3 --pads 0,1,1 --reps 10 2>&1 | grep -E "^cfg|rror" | tail -1
  done
done
