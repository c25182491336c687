#!/bin/bash
# rocprofv3 kernel trace of `bench.py --arch i3d --streams 1` (InceptionI3d extraction, one stream: undisturbed per-kernel durations)
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/prof_i3d
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/one -o one -- python3 bench.py --arch i3d --batch 225 --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-train --no-act-range > $O/one.json 2> $O/one.err
python3 scripts/summarize_rocprof.py $(find $O/one -name '*kernel_trace.csv' | head -1) --arch i3d --batch 225 --forwards 20 --streams 1 --out $O/kernels_1stream.md \
  --title "${ROUND:-r04}: bench.py --arch i3d --streams 1 (InceptionI3d, 225 clips/forward, f16) - per-kernel durations" > $O/summary.json
find $O -name '*kernel_trace.csv' -size +20M -delete
cat $O/one.json | tail -1 | cut -c1-300; tail -4 $O/kernels_1stream.md
