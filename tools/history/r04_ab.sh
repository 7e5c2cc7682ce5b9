#!/bin/bash
# tools/r04_ab.sh "ENVVAR=v1 ENVVAR=v2 ..." WORKLOAD... -- per-kernel times (rocprofv3 kernel trace of bench.py) for several settings of
# one environment variable and several workloads inside ONE run (boxes differ by several percent); BENCH_ARGS: more arguments for bench.py
export TMPDIR=/tmp
SETTINGS=$1; shift
for WL in "$@"; do
for S in $SETTINGS; do
  OUT=$PWD/gpurun_out/ab_${WL}_${S//[^A-Za-z0-9]/_}; rm -rf $OUT; mkdir -p $OUT
  export "$S"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps ${STEPS:-5} --warmup 2 --no-cpu --no-also --workload $WL $BENCH_ARGS > $OUT/log.txt 2>&1
  echo "== $WL $S rc=$?"
  grep -h '^{' $OUT/log.txt | tail -1 | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); r = d['roofline']; p = r['pass_parts_ms']
    print('   pass %.3f ms (frac %.3f)  scan %.3f (frac %.3f)  after %.3f  updates %d' % (p['pass'], r['pass_frac'], p['scan'], r['frac'], p['after_scan'], d['config']['table_updates_rank0']))
except Exception as e:
    print('   no bench line', e)"
  python3 tools/kstats.py $OUT "${KREGEX:-k_part|k_apply|k_sort|k_scan<|k_l1|k_l2}"
done
done
