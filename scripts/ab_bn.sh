#!/bin/bash
trap 'cp ab/new.so ted_spad_amd/libtedspad_hip.so' EXIT
# same-box A/B of two library builds (ab/old.so, ab/new.so) on the BatchNorm pass probe (scripts/bn_probe.py): bash scripts/ab_bn.sh
for v in old new old new; do
  cp ab/$v.so ted_spad_amd/libtedspad_hip.so
  echo "== $v"; timeout -k 10 200 python scripts/bn_probe.py 2>&1 | grep bn_
done
