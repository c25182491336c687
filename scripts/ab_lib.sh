#!/bin/bash
trap 'cp ab/new.so ted_spad_amd/libtedspad_hip.so' EXIT
# Same-box A/B of two builds of libtedspad_hip.so: ab/old.so and ab/new.so take turns as the library; bench.py (cfg2) with 2 streams and with 1.
# Usage (inside one gpurun call): bash scripts/ab_lib.sh [rounds]
set -e
R=${1:-2}
mkdir -p gpurun_out
for r in $(seq 1 $R); do
  for v in old new; do
    cp ab/$v.so ted_spad_amd/libtedspad_hip.so
    for st in 2 1; do
      timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-train --no-act-range --streams $st > gpurun_out/ab_${v}_${r}_s${st}.json 2> gpurun_out/ab_${v}_${r}_s${st}.err
      python - <<PY
import json
j=[json.loads(l) for l in open("gpurun_out/ab_${v}_${r}_s${st}.json") if l.startswith("{")][0]
print("$v round $r streams $st: clips/s", round(j["value"]), "ms/fwd", j["roofline"].get("ms_per_forward"), "relL2", j.get("feature_rel_l2_max"), flush=True)
PY
    done
  done
done
cp ab/new.so ted_spad_amd/libtedspad_hip.so
