#!/bin/bash
trap 'cp ab/new.so ted_spad_amd/libtedspad_hip.so' EXIT
# as ab_stem_bench.sh with ONE stream (kernels run alone: the forward's time is the sum of its kernels)
run() {
  timeout -k 10 200 python bench.py --steps 6 --warmup 2 --streams 1 --no-cpu-baseline --no-train --no-act-range > gpurun_out/ab_stem_bench.json 2> gpurun_out/ab_stem_bench.err
  python - <<PY
import json
j=[json.loads(l) for l in open("gpurun_out/ab_stem_bench.json") if l.startswith("{")][0]
print("$1: clips/s", round(j["value"]), "ms/fwd", j["roofline"].get("ms_per_forward"), flush=True)
PY
}
for r in 1 2; do
  cp ab/old.so ted_spad_amd/libtedspad_hip.so; run "old round $r"
  cp ab/new.so ted_spad_amd/libtedspad_hip.so
  TEDSPAD_STEM_LOADERS=0 run "new, 8-wave stem, round $r"
  run "new, loader waves, round $r"
done
