#!/bin/bash
# tools/ktrace_c3.sh [regex] -- per-kernel times of tools/bench_c3.py (default: binned path) from a rocprofv3 kernel trace
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/ktrace_c3
rm -rf $OUT; mkdir -p $OUT
C3_PATHS=${C3_PATHS:-bin} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/bench_c3.py > $OUT/log.txt 2>&1
python3 tools/kstats.py $OUT "${1:-.}"
grep '^{' $OUT/log.txt | tail -1
