#!/bin/bash
trap 'cp ab/new.so ted_spad_amd/libtedspad_hip.so' EXIT
# same-box A/B of two library builds (ab/old.so, ab/new.so) on isolated conv probes: bash scripts/ab_probe.sh
for r in 1 2; do
for v in old new; do
  cp ab/$v.so ted_spad_amd/libtedspad_hip.so
  echo "== $v round $r"
  for spec in "400,1,112,112 64 64 32 1,3,3 0,1,1" "400,1,112,112 320 64 32 1,3,3 0,1,1" "384,1,56,56 128 128 33 1,3,3 0,1,1" "384,1,28,28 256 256 33 1,3,3 0,1,1" "225,8,56,56 64 192 33 3,3,3 1,1,1" "375,4,56,56 256 64 34 3,1,1 1,0,0"; do
    set -- $spec
    timeout -k 10 120 python scripts/conv_probe.py --dims $1 --cin $2 --cout $3 --cfg $4 --k $5 --pads $6 --reps 10 2>&1 | tail -1
  done
done
done
cp ab/new.so ted_spad_amd/libtedspad_hip.so
