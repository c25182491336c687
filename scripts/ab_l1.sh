#!/bin/bash
# same-box A/B of the whole-block layer1 kernel inside the network: TEDSPAD_BNECK_L1 = 0 (two launches per block), 2 (the last, pooled block fused), 1 (both plain blocks)
for r in 1 2; do for v in 0 2 1; do for st in 2 1; do
TEDSPAD_BNECK_L1=$v timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-train --streams $st | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('L1=$v round $r streams $st: clips/s', round(j['value']), 'ms/fwd', j['roofline']['ms_per_forward'])"
done; done; done
