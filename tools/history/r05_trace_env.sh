#!/bin/bash
# tools/r05_trace_env.sh WORKLOAD -- like r05_trace.sh, with whatever LIME_* variables the caller exported (bench.py's refusals aside) and the raw trace kept
export TMPDIR=/tmp
W=$1
OUT=$PWD/gpurun_out/r05_trace_env_$W
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --workload $W --no-also --no-cpu --steps 5 --warmup 2 > $OUT/log.txt 2>&1
python3 tools/kstats.py $OUT "k_"
python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last complete pass: from the last k_zero2 on
idx = [i for i, r in enumerate(rows) if "k_zero2" in r["Kernel_Name"]]
a = idx[-2]; b = idx[-1]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    print(f"  +{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} us  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us  {r['Kernel_Name'][:60]}")
PY
