#!/bin/bash
# same-box A/B of two library builds (ab/old.so, ab/new.so) on the fused unet++ tail: bash scripts/ab_tail.sh [rounds]
trap 'cp ab/new.so ted_spad_amd/libtedspad_hip.so' EXIT
for r in $(seq 1 ${1:-3}); do for v in old new; do
  cp ab/$v.so ted_spad_amd/libtedspad_hip.so
  echo -n "$v round $r: "; timeout -k 10 200 python scripts/tail_probe.py 2>&1 | tail -1
done; done
