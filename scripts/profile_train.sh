#!/bin/bash
# rocprofv3 passes over the cfg3 training iteration (scripts/train_prof_run.py): kernel trace + stats, the two HBM-traffic counters, the SQ counters.
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/prof_train
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o trace -- python3 scripts/train_prof_run.py 3 fb > $O/trace.log 2> $O/trace.err
echo "trace done"; tail -1 $O/trace.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o fetch -- python3 scripts/train_prof_run.py 3 fb > $O/fetch.log 2> $O/fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o write -- python3 scripts/train_prof_run.py 3 fb > $O/write.log 2> $O/write.err
echo "write done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sq -o sq -- python3 scripts/train_prof_run.py 3 fb > $O/sq.log 2> $O/sq.err
echo "sq done"
python3 scripts/summarize_train.py --trace $(find $O/trace -name '*kernel_trace.csv' | head -1) --fetch $(find $O/fetch -name '*counter_collection.csv' | head -1) \
  --write $(find $O/write -name '*counter_collection.csv' | head -1) --sq $(find $O/sq -name '*counter_collection.csv' | head -1) --iters 3 \
  --out $O/train_kernels.md --title "${ROUND:-r03}: cfg3 training iteration (UNet fa + I3Res50 ft on 8 x 48 x 112^2, privacy branch fb + NT-Xent on 2 x 12 x 224^2, f16): kernel time, HBM traffic, MFMA utilisation" > $O/summary.txt
cp $(find $O/trace -name '*kernel_stats.csv' | head -1) $O/train_kernel_stats.csv
python3 scripts/train_timeline.py $(find $O/trace -name '*kernel_trace.csv' | head -1) 3 > $O/train_timeline.txt
find $O -name '*kernel_trace.csv' -size +8M -delete; find $O -name '*counter_collection.csv' -size +8M -delete
cat $O/summary.txt
python3 - <<'PY'
import json, sys
sys.path.insert(0, '.')
import bench
s = json.loads(open('gpurun_out/prof_train/summary.txt').read().strip().splitlines()[-1])
json.dump({'traffic_bytes_per_iteration': s['traffic_bytes_per_iteration'], 'ms_per_iteration_kernels': s['ms_per_iteration_kernels'],
           'launches_per_iteration': s['launches_per_iteration'], 'kernel_sources_sha': bench.train_sources_sha(), 'with_fb': True,
           'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes over scripts/train_prof_run.py (3 iterations of cfg3 with the privacy branch: UNet fa + I3Res50 ft on 8 x 48 x 112^2, fb + NT-Xent on 2 x 12 x 224^2) on MI355X, scripts/profile_train.sh; FETCH_SIZE doubled (gfx950 correction)'},
          open('gpurun_out/prof_train/traffic_train_cfg3.json', 'w'), indent=1)
PY

