#!/bin/bash
# same-box A/B of two library builds on isolated conv probes
for r in 1 2; do
for v in old new; do
  cp ab/$v.so ted_spad_amd/libtedspad_hip.so
  echo "== $v round $r"
  timeout -k 10 120 python scripts/conv_probe.py --dims 400,1,112,112 --cin 64 --cout 64 --cfg 32 --reps 10 2>&1 | tail -1
  timeout -k 10 120 python scripts/conv_probe.py --dims 400,1,112,112 --cin 320 --cout 64 --cfg 32 --reps 10 2>&1 | tail -1
  timeout -k 10 120 python scripts/conv_probe.py --dims 384,1,56,56 --cin 128 --cout 128 --cfg 33 --reps 10 2>&1 | tail -1
  timeout -k 10 200 python scripts/conv_probe.py --dims 225,8,56,56 --cin 64 --cout 192 --k 3,3,3 --pads 1,1,1 --cfg 33 --reps 5 2>&1 | tail -1
done
done
cp ab/new.so ted_spad_amd/libtedspad_hip.so
