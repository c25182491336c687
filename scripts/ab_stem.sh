#!/bin/bash
# the fp32-clip stem without / with 4 / with 8 loader waves on one box (scripts/stem_clip_probe.py, 375 clips): bash scripts/ab_stem.sh ["0 4 8"]
for r in 1 2; do
  for v in ${1:-0 4 8}; do
    echo "== TEDSPAD_STEM_LOADERS=$v round $r"; TEDSPAD_STEM_LOADERS=$v timeout -k 10 300 python scripts/stem_clip_probe.py 375 2>&1 | tail -6
  done
done
