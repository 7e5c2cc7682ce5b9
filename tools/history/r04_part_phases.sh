#!/bin/bash
# tools/r04_part_phases.sh -- where k_part's cycles go (variants/lib_ppt.so: -DLIME_PART_TIMING): per phase, summed over wave 0 of every workgroup
# build the instrumented library first:  make -C lime_amd/csrc -s EXTRA=-DLIME_PART_TIMING -B ../liblime_hip.so && mkdir -p variants && cp lime_amd/liblime_hip.so variants/lib_ppt.so && make -C lime_amd/csrc -s -B ../liblime_hip.so
cp lime_amd/liblime_hip.so /tmp/lib_keep.so; cp variants/lib_ppt.so lime_amd/liblime_hip.so
for shape in ${SHAPES:-"1000000000 1000000 5000" "10000000000 1000000 1000"}; do
set -- $shape
LIME_PART_LINES=${LINES:-0} LIME_PART_SPLIT=${SPLIT:-4} C3_PATHS=bin C3_N=$1 C3_NR=$2 C3_NG=$3 python3 - <<'PY'
import os, sys, ctypes, json, subprocess
sys.path.insert(0, os.getcwd())
import runpy, io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path("tools/bench_c3.py", run_name="__main__")
from lime_amd import _lib
out = (ctypes.c_ulonglong * 8)()
_lib.load().lime_debug_part_times(out)
v = list(out); tot = sum(v) or 1
names = ["->barrier", "wait at barrier 1", "scan", "place", "wait at barrier (stage)", "count next + prefetch", "write-out", "-"] if os.environ.get("LIME_PART_LINES") == "0" else ["(carry update) ->barrier", "wait at barrier 1", "scan", "place", "wait at barrier (stage)", "write-out lines", "count next + prefetch", "wait barrier + carry update"]
print("N=%s: k_part phases (share of wave 0's cycles): " % os.environ["C3_N"] + "; ".join("%s %.1f%%" % (n, 100.0 * x / tot) for n, x in zip(names, v) if x))
PY
done
cp /tmp/lib_keep.so lime_amd/liblime_hip.so
