#!/bin/bash
trap 'cp ab/new.so ted_spad_amd/libtedspad_hip.so' EXIT
# same-box A/B of two library builds (ab/old.so, ab/new.so) on the cfg3 training iteration with the privacy branch (scripts/bench_train.py --fb): bash scripts/ab_train_lib.sh [rounds]
R=${1:-2}
for r in $(seq 1 $R); do
  for v in old new; do
    cp ab/$v.so ted_spad_amd/libtedspad_hip.so
    echo -n "$v round $r: "; timeout -k 10 500 python scripts/bench_train.py --fb 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k: d[k] for k in ('phase1_update_fa_ms', 'phase2_update_ft_ms')}, flush=True)"
  done
done
