export TMPDIR=/tmp
OUT=/tmp/atp; rm -rf $OUT
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $OUT -- python3 bench.py --steps 3 --warmup 2 --no-cpu --no-also --workload text_tiled > /tmp/atp.log 2>&1
f=$(find $OUT -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    if not any(x in k for x in ("k_apply_tiles", "k_part", "k_sort_tiles")): continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
for k, cs in acc.items():
    n = len(disp[k]); print(k, {c: round(v / n / 1e6, 3) for c, v in cs.items()}, "(millions per launch; 24.0 M records)")
PY
