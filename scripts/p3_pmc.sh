#!/bin/bash
# HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, gfx950 correction) of tiles 38 / 40 on two of the wide 3 x 3 layers: bash scripts/p3_pmc.sh
export TMPDIR=/tmp
O=gpurun_out/p3pmc; rm -rf $O; mkdir -p $O
for cin in 64 128; do for cfg in 38 40; do for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/${cin}_${cfg}_$c -o x -- python3 scripts/conv_probe.py --dims 400,1,112,112 --cin $cin --cout 64 --cfg $cfg --k 1,3,3 --pads 0,1,1 --reps 3 > $O/log_${cin}_${cfg}_$c.txt 2>&1
  F=$(find $O/${cin}_${cfg}_$c -name '*counter_collection.csv' | head -1)
  python3 - "$F" "$cin" "$cfg" "$c" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'conv_patch' in r['Kernel_Name']]
v = [float(r['Counter_Value']) for r in rows if r['Counter_Name'] == sys.argv[4]]
mul = 2048.0 if sys.argv[4] == 'FETCH_SIZE' else 1024.0
print('cin %s cfg %s %s: %d launches, %.3f GB per launch' % (sys.argv[2], sys.argv[3], sys.argv[4], len(v), sum(v) / max(1, len(v)) * mul / 1e9), flush=True)
PY
done; done; done
