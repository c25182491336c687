#!/bin/bash
# Same-box A/B of one environment switch: bench.py (cfg2) with VAR=0 and VAR=1 taking turns, 2 streams and 1.
# Usage (inside one gpurun call): bash scripts/ab_env.sh TEDSPAD_STEM_CLIP [rounds]
set -e
VAR=$1
R=${2:-2}
mkdir -p gpurun_out
for r in $(seq 1 $R); do
  for v in 0 1; do
    for st in 2 1; do
      env $VAR=$v timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-train --no-act-range --streams $st > gpurun_out/abenv_${VAR}_${v}_${r}_s${st}.json 2> gpurun_out/abenv_${VAR}_${v}_${r}_s${st}.err
      python - <<PY
import json
j=[json.loads(l) for l in open("gpurun_out/abenv_${VAR}_${v}_${r}_s${st}.json") if l.startswith("{")][0]
print("$VAR=$v round $r streams $st: clips/s", round(j["value"]), "ms/fwd", j["roofline"].get("ms_per_forward"), "relL2", j.get("feature_rel_l2_max"), flush=True)
PY
    done
  done
done
