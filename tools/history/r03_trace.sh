#!/bin/bash
# tools/r03_trace.sh WORKLOAD... -- rocprofv3 kernel-trace stats of bench.py per workload, printed as a table
export TMPDIR=/tmp
for WL in "$@"; do
  OUT=$PWD/gpurun_out/trace_$WL; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 5 --warmup 2 --no-cpu --no-also --workload $WL > $OUT/log.txt 2>&1
  echo "== $WL rc=$?"; grep -h '^{' $OUT/log.txt | tail -1 | cut -c1-400
  f=$(find $OUT -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-60s calls %5s avg %10.1f us total %10.1f us  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3, r["Percentage"]))
PY
done
