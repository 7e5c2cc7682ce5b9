#!/bin/bash
# tools/r03_ab2.sh lib... -- min over REPS alternating runs of the scan / pass / after-scan times (HIP events): configs[2] (binned), N = 1e10
# (binned, 1 GB table) and configs[1] (cas), per library build
export TMPDIR=/tmp
REPS=${REPS:-3}
cp lime_amd/liblime_hip.so /tmp/lib_keep.so
rm -f /tmp/ab2_*.txt
for rep in $(seq $REPS); do
for lib in "$@"; do
  cp $lib lime_amd/liblime_hip.so
  t=$(basename $lib .so)
  C3_PATHS=bin python3 tools/bench_c3.py 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['bin']['parts_ms']; print('c3', b['scan'], b['pass'], b['after_scan'])" >> /tmp/ab2_$t.txt
  if [ "${N10:-1}" = "1" ]; then C3_N=10000000000 C3_NR=1000000 C3_NG=1000 C3_PATHS=bin python3 tools/bench_c3.py 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['bin']['parts_ms']; print('n1e10', b['scan'], b['pass'], b['after_scan'])" >> /tmp/ab2_$t.txt; fi
  C3_EBWT=1 C3_N=100000000 C3_NR=100000 C3_NG=500 C3_PATHS=cas python3 tools/bench_c3.py 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['cas']['parts_ms']; print('c2', b['scan'], b['pass'], b['after_scan'])" >> /tmp/ab2_$t.txt
done
done
cp /tmp/lib_keep.so lime_amd/liblime_hip.so
python3 - "$@" <<'PY'
import sys, os, collections
for lib in sys.argv[1:]:
    t = os.path.basename(lib)[:-3]
    rows = collections.defaultdict(list)
    for ln in open(f"/tmp/ab2_{t}.txt"):
        w, s, p, a = ln.split(); rows[w].append((float(s), float(p), float(a)))
    for w, v in rows.items():
        bps = {"c3": 8e9, "n1e10": 8e10, "c2": 9e8}[w]
        ms = min(x[0] for x in v)
        print(f"{t:14s} {w:6s} scan min {ms:8.3f} (all {' '.join('%.3f' % x[0] for x in v)}) frac {bps / ms / 8e9:.3f}  pass min {min(x[1] for x in v):8.3f}  after min {min(x[2] for x in v):7.3f}")
PY
