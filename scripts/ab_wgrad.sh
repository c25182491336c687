#!/bin/bash
trap 'cp ab/new.so ted_spad_amd/libtedspad_hip.so' EXIT
# same-box A/B of two library builds (ab/old.so, ab/new.so) on the weight-gradient probe (scripts/wgrad_probe.py: the UNet's 3 x 3 layers at cfg3); the new build
# also with TEDSPAD_WGRAD3P_LOADERS=0 (the patch kernel without its loader waves): bash scripts/ab_wgrad.sh
for r in 1 2; do
  cp ab/old.so ted_spad_amd/libtedspad_hip.so
  echo "== old round $r"; timeout -k 10 200 python scripts/wgrad_probe.py 2>&1 | tail -9
  cp ab/new.so ted_spad_amd/libtedspad_hip.so
  echo "== new round $r"; timeout -k 10 200 python scripts/wgrad_probe.py 2>&1 | tail -9
  echo "== new, no loader waves, round $r"; TEDSPAD_WGRAD3P_LOADERS=0 timeout -k 10 200 python scripts/wgrad_probe.py 2>&1 | tail -9
done
cp ab/new.so ted_spad_amd/libtedspad_hip.so
