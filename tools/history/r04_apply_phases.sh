#!/bin/bash
# tools/r04_apply_phases.sh -- where k_apply_tiles' cycles go (variants/lib_apt.so: -DLIME_APPLY_TIMING; the waits for the runs' loads are forced to
# vmcnt(0) there so that they can be told from the adds): per phase, summed over wave 0 of every workgroup
# build the instrumented library first:  make -C lime_amd/csrc -s EXTRA=-DLIME_APPLY_TIMING -B ../liblime_hip.so && mkdir -p variants && cp lime_amd/liblime_hip.so variants/lib_apt.so && make -C lime_amd/csrc -s -B ../liblime_hip.so
cp lime_amd/liblime_hip.so /tmp/lib_keep.so; cp variants/lib_apt.so lime_amd/liblime_hip.so
for shape in "1000000000 1000000 5000" "10000000000 1000000 1000" "100000000 452000 678"; do
set -- $shape
C3_PATHS=bin C3_N=$1 C3_NR=$2 C3_NG=$3 python3 - <<'PY'
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
names = ["clear LDS + barrier", "first loads issued", "wait for a step's loads", "next step's loads issued", "adds", "barrier after the adds", "index prefetch + write-out", "barrier after the write-out"]
print("N=%s table %sx%s: k_apply_tiles phases (share of wave 0's cycles): " % (os.environ["C3_N"], os.environ["C3_NR"], os.environ["C3_NG"]) + "; ".join("%s %.1f%%" % (n, 100.0 * x / tot) for n, x in zip(names, v) if x))
PY
done
cp /tmp/lib_keep.so lime_amd/liblime_hip.so
