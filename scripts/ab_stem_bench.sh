#!/bin/bash
trap 'cp ab/new.so ted_spad_amd/libtedspad_hip.so' EXIT
# headline bench (cfg2) on one box: ab/old.so, then ab/new.so with TEDSPAD_STEM_LOADERS=0 and with the loader waves: bash scripts/ab_stem_bench.sh [rounds]
R=${1:-2}
run() {
  timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-train --no-act-range > gpurun_out/ab_stem_bench.json 2> gpurun_out/ab_stem_bench.err
  python - <<PY
import json
j=[json.loads(l) for l in open("gpurun_out/ab_stem_bench.json") if l.startswith("{")][0]
print("$1: clips/s", round(j["value"]), "ms/fwd", j["roofline"].get("ms_per_forward"), "relL2", j.get("feature_rel_l2_max"), flush=True)
PY
}
for r in $(seq 1 $R); do
  cp ab/old.so ted_spad_amd/libtedspad_hip.so; run "old round $r"
  cp ab/new.so ted_spad_amd/libtedspad_hip.so
  TEDSPAD_STEM_LOADERS=0 run "new, 8-wave stem, round $r"
  run "new, loader waves, round $r"
done
