#!/bin/bash
# HBM traffic of the whole-block layer1 kernel (and of the two launches it replaces) at bench size: separate --pmc FETCH_SIZE / WRITE_SIZE passes of scripts/bneck_l1_probe.py
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/l1prof; rm -rf $O; mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o fetch -- python3 scripts/bneck_l1_probe.py > $O/fetch.txt 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o write -- python3 scripts/bneck_l1_probe.py > $O/write.txt 2> $O/write.err
python3 - <<'PY'
import csv, glob, collections
for what in ("fetch", "write"):
    f = glob.glob("gpurun_out/l1prof/%s/**/*counter_collection.csv" % what, recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if any(s in k for s in ("bneck_l1", "tflat", "bneck_tail")):
            kib = sum(v[-3:]) / 3
            print("%s %-60s last launches: %.1f MB%s" % (what, k, kib * 1024 / 1e6 * (2 if what == "fetch" else 1), " (x2 corrected)" if what == "fetch" else ""))
PY
find $O -name '*counter_collection.csv' -size +5M -delete
