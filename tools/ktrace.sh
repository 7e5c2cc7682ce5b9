#!/bin/bash
# tools/ktrace.sh -- kernel-trace stats of a short bench run (timing per kernel)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/ktrace
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 10 --warmup 2 --no-cpu ${EXTRA} > $OUT/log.txt 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cut -d, -f1-4 $f | head -12
