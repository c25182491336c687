#!/bin/bash
# rocprofv3 passes over the anonymised extraction (scripts/anon_prof_run.py: unet++ -> Q1 feed -> I3Res50 on 75 clips, 25 per forward): kernel trace, HBM traffic, SQ.
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/prof_anon
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o trace -- python3 scripts/anon_prof_run.py 2 150 > $O/trace.log 2> $O/trace.err
echo "trace done"; tail -1 $O/trace.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o fetch -- python3 scripts/anon_prof_run.py 2 150 > $O/fetch.log 2> $O/fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o write -- python3 scripts/anon_prof_run.py 2 150 > $O/write.log 2> $O/write.err
echo "write done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sq -o sq -- python3 scripts/anon_prof_run.py 2 150 > $O/sq.log 2> $O/sq.err
echo "sq done"
python3 scripts/summarize_train.py --trace $(find $O/trace -name '*kernel_trace.csv' | head -1) --fetch $(find $O/fetch -name '*counter_collection.csv' | head -1) \
  --write $(find $O/write -name '*counter_collection.csv' | head -1) --sq $(find $O/sq -name '*counter_collection.csv' | head -1) --iters 2 \
  --out $O/anon_extract_kernels_1stream.md --title "${ROUND:-r04}: anonymised extraction (unet++ fa on 16 x 224^2 per clip -> Q1 feed -> I3Res50), one pass = 150 clips, 75 per anonymizer and per encoder forward, f16, one stream: kernel time, HBM traffic, MFMA utilisation" > $O/summary.txt
find $O -name '*kernel_trace.csv' -size +8M -delete; find $O -name '*counter_collection.csv' -size +8M -delete
cat $O/summary.txt
