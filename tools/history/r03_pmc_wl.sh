#!/bin/bash
# tools/r03_pmc_wl.sh WORKLOAD -- instruction mix per 1024-position window of the scan on a bench.py workload (rocprofv3 --pmc, one pass)
export TMPDIR=/tmp
WL=${1:-text_tiled}
OUT=/tmp/pmcwl; rm -rf $OUT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT -- python3 bench.py --steps 3 --warmup 2 --no-cpu --no-also --workload $WL > $OUT.log 2>&1
grep -h '^{' $OUT.log | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$WL: scan %.3f ms frac %.3f pass %.3f ms symbols %d' % (r['kernel_ms_avg'], r['frac'], r['pass_ms_avg'], d['config']['symbols_total']))"
f=$(find $OUT -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append((r["Dispatch_Id"], float(r["Counter_Value"])))
for k, cs in acc.items():
    if "k_scan<" not in k: continue
    tot = {c: sum(v for _, v in vs) / len(set(d for d, _ in vs)) for c, vs in cs.items()}
    print(k, {c: round(v / 1e6, 2) for c, v in tot.items()}, "(millions per launch)")
PY
