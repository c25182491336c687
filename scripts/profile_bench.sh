#!/bin/bash
# Profiles of the default bench command on the GPU box (run through gpurun): kernel trace + stats (default streams and 1 stream)
# and the two HBM-traffic counter passes; summaries land in gpurun_out/ and are copied to profiles/ by hand.
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/prof
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/main -o main -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train --no-act-range > $O/main.json 2> $O/main.err
echo "main done"; cat $O/main.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/one -o one -- python3 bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-train --no-act-range > $O/one.json 2> $O/one.err
echo "one-stream done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o fetch -- python3 bench.py --steps 1 --warmup 0 --clip-times 75 --streams 1 --no-cpu-baseline --no-train --no-act-range > $O/fetch.json 2> $O/fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o write -- python3 bench.py --steps 1 --warmup 0 --clip-times 75 --streams 1 --no-cpu-baseline --no-train --no-act-range > $O/write.json 2> $O/write.err
echo "write done"
T=$(find $O/main -name '*kernel_trace.csv' | head -1); T1=$(find $O/one -name '*kernel_trace.csv' | head -1)
F=$(find $O/fetch -name '*counter_collection.csv' | head -1); W=$(find $O/write -name '*counter_collection.csv' | head -1)
python3 scripts/summarize_rocprof.py $T --fetch $F --write $W --batch 375 --forwards 12 --streams 2 --out $O/kernels.md \
  --title "${ROUND:-r03}: bench.py cfg2 (largei3d, 375 clips/forward, 2 streams, f16) - the default bench command: kernel time and HBM traffic" > $O/summary.json
python3 scripts/summarize_rocprof.py $T1 --batch 375 --forwards 12 --streams 1 --out $O/kernels_1stream.md \
  --title "${ROUND:-r03}: bench.py cfg2 --streams 1 (largei3d, 375 clips/forward, f16) - undisturbed per-kernel durations" > $O/summary_1stream.json
cp $(find $O/main -name '*kernel_stats.csv' | head -1) $O/main_kernel_stats.csv
cp $(find $O/one -name '*kernel_stats.csv' | head -1) $O/one_kernel_stats.csv
# keep the merged scratch small: the raw traces are not needed once summarised
find $O -name '*kernel_trace.csv' -size +20M -delete; find $O -name '*counter_collection.csv' -size +20M -delete
cat $O/summary.json
python3 - <<'PY'
import json, sys
sys.path.insert(0, '.')
import bench
s = json.load(open('gpurun_out/prof/summary.json'))
json.dump({'arch': 'largei3d', 'batch': 375, 'dtype': 'f16', 'conv_traffic_bytes_per_forward': s['conv_traffic_bytes_per_forward'],
           'all_traffic_bytes_per_forward': s['all_traffic_bytes_per_forward'], 'kernel_sources_sha': bench.kernel_sources_sha(),
           'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of `bench.py --steps 1 --warmup 0 --clip-times 75 --streams 1` on MI355X (scripts/profile_bench.sh), last forward (375 clips); FETCH_SIZE doubled (gfx950 correction)'},
          open('gpurun_out/prof/traffic_cfg2.json', 'w'), indent=1)
PY
# SQ counters of the last forward: MFMA utilisation and wave states per kernel (one more pass, kernels serialised)
S=gpurun_out/prof_sq; rm -rf $S; mkdir -p $S
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $S -o sq -- python3 bench.py --steps 1 --warmup 0 --clip-times 75 --streams 1 --no-cpu-baseline --no-train --no-act-range > $S/sq.json 2> $S/sq.err
python3 scripts/summarize_sq.py $(find $S -name '*counter_collection.csv' | head -1) --batch 375 --out $S/mfma_util.md
find $S -name '*counter_collection.csv' -size +20M -delete
tail -5 $S/mfma_util.md
