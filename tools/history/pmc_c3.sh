#!/bin/bash
# tools/pmc_c3.sh "COUNTERS" [regex] -- rocprofv3 --pmc pass over tools/bench_c3.py (binned path), per-kernel counter averages
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_c3
rm -rf $OUT; mkdir -p $OUT
C3_PATHS=${C3_PATHS:-bin} rocprofv3 --pmc $1 --output-format csv -d $OUT -- python3 tools/bench_c3.py > $OUT/log.txt 2>&1
python3 - "$OUT" "${2:-.}" <<'PY'
import csv, glob, re, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[-1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
seen = set()
for row in csv.DictReader(open(f)):
    name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").replace("lime::", "")
    if not re.search(sys.argv[2], name): continue
    acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
    key = (name, row["Dispatch_Id"])
    if key not in seen: seen.add(key); calls[name] += 1
for name, cs in acc.items():
    print(name, "calls", calls[name], {k: round(v / calls[name], 1) for k, v in cs.items()})
PY
