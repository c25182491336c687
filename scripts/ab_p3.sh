#!/bin/bash
trap 'cp ab/new.so ted_spad_amd/libtedspad_hip.so' EXIT
# same-box A/B of two library builds (ab/old.so, ab/new.so) on tile 40's isolated probes: bash scripts/ab_p3.sh [rounds]
R=${1:-2}
for r in $(seq 1 $R); do
for v in old new; do
  cp ab/$v.so ted_spad_amd/libtedspad_hip.so
  echo -n "$v round $r:"
  for spec in "400,1,112,112 64" "400,1,112,112 128" "400,1,112,112 192" "400,1,112,112 320"; do
    set -- $spec
    echo -n " c$2 $(timeout -k 10 120 python scripts/conv_probe.py --dims $1 --cin $2 --cout 64 --cfg 40 --k 1,3,3 --pads 0,1,1 --reps 10 2>&1 | tail -1 | awk '{print $3}')"
  done
  echo
done
done
cp ab/new.so ted_spad_amd/libtedspad_hip.so
