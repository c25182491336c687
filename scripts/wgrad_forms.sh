#!/bin/bash
# the patch weight-gradient kernel with and without its loader waves on one box (scripts/wgrad_probe.py): bash scripts/wgrad_forms.sh
# (round 6 also built 2 / 6 loaders, rings of 4 / 5 slots, a segment-staggered and an LDS-counter form behind this switch: profiles/r06_ab_experiments.md)
for r in 1 2; do
  for f in ${1:-0 4}; do
    echo "== loaders $f round $r"; TEDSPAD_WGRAD3P_LOADERS=$f timeout -k 10 200 python scripts/wgrad_probe.py 2>&1 | tail -9 | head -8
  done
done
