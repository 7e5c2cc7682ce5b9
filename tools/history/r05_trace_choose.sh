#!/bin/bash
# per-kernel times of tools/bench_choose.py (fused + choose with and without the table)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r05_trace_choose
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/bench_choose.py > $OUT/log.txt 2>&1
python3 tools/kstats.py $OUT "k_"
tail -2 $OUT/log.txt
find $OUT -name "*kernel_trace.csv" -delete
