#!/bin/bash
# Same-box A/B of two builds of libtedspad_hip.so (ab/old.so, ab/new.so take turns) on any bench.py command line.
# Usage (inside one gpurun call): bash scripts/ab_any.sh ROUNDS KEY[,KEY..] -- bench.py args ...     KEY = JSON paths to print, e.g. value or train_cfg3.iteration_ms
set -e
trap 'cp ab/new.so ted_spad_amd/libtedspad_hip.so' EXIT      # a failed / timed-out run must not leave the OLD library in the package
R=$1; KEYS=$2; shift 3
mkdir -p gpurun_out
for r in $(seq 1 $R); do
  for v in old new; do
    cp ab/$v.so ted_spad_amd/libtedspad_hip.so
    timeout -k 10 400 python bench.py "$@" > gpurun_out/abx_${v}_${r}.json 2> gpurun_out/abx_${v}_${r}.err
    python - "$KEYS" gpurun_out/abx_${v}_${r}.json "$v round $r" <<'PY'
import json, sys
j = [json.loads(l) for l in open(sys.argv[2]) if l.startswith("{")][0]
out = []
for k in sys.argv[1].split(","):
    v = j
    for part in k.split("."):
        v = v[part]
    out.append("%s %s" % (k, v))
print(sys.argv[3] + ": " + " | ".join(out), flush=True)
PY
  done
done
cp ab/new.so ted_spad_amd/libtedspad_hip.so
