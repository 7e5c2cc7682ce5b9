#!/bin/bash
# tools/r04_part_phases_waves.sh -- k_part_lines' phases as seen by the workgroup's first, fourth and last wave (variants/lib_ppt{0,3,7}.so:
# -DLIME_PART_TIMING -DLIME_PT_WAVE=w): do the waves reach the barriers together?
# build them first:  for w in 0 3 7; do make -C lime_amd/csrc -s EXTRA="-DLIME_PART_TIMING -DLIME_PT_WAVE=$w" -B ../liblime_hip.so && cp lime_amd/liblime_hip.so variants/lib_ppt$w.so; done; make -C lime_amd/csrc -s -B ../liblime_hip.so
cp lime_amd/liblime_hip.so /tmp/lib_keep.so
for w in 0 3 7; do
cp variants/lib_ppt$w.so lime_amd/liblime_hip.so
LIME_PART_SPLIT=4 C3_PATHS=bin C3_N=${1:-4000000000} C3_NR=1000000 C3_NG=1000 python3 - $w <<'PY'
import os, sys, ctypes
sys.path.insert(0, os.getcwd())
import runpy, io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path("tools/bench_c3.py", run_name="__main__")
from lime_amd import _lib
out = (ctypes.c_ulonglong * 8)()
_lib.load().lime_debug_part_times(out)
v = list(out); tot = sum(v) or 1
names = ["(carry update) -> scan", "-", "scan", "place", "wait at barrier (stage)", "count next", "write-out lines + prefetch", "wait barrier + carry update"]
print("wave %s, N=%s: " % (sys.argv[1], os.environ["C3_N"]) + "; ".join("%s %.1f%% (%.0f M)" % (n, 100.0 * x / tot, x / 1e6) for n, x in zip(names, v) if x))
PY
done
cp /tmp/lib_keep.so lime_amd/liblime_hip.so
