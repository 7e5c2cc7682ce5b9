#!/bin/bash
# tools/r05_trace.sh WORKLOAD [steps] -- per-kernel times of `bench.py --workload W --no-also --no-cpu` from a rocprofv3 kernel trace
# (kept: gpurun_out/r05_trace_W/ + the stats table on stdout)
export TMPDIR=/tmp
W=$1; K=${2:-5}
OUT=$PWD/gpurun_out/r05_trace_$W
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --workload $W --no-also --no-cpu --steps $K --warmup 2 > $OUT/log.txt 2>&1
echo "== $W"
python3 tools/kstats.py $OUT .
grep '^{' $OUT/log.txt | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'parts', d['roofline']['pass_parts_ms'])"
find $OUT -name "*kernel_trace.csv" -delete      # (large; the stats table is what is kept)
