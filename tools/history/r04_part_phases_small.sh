#!/bin/bash
# tools/r04_part_phases_small.sh -- k_part / k_part_lines phase shares on a text-like problem (1e8 symbols, clustered generator, 306 MB table)
# build the instrumented library first:  make -C lime_amd/csrc -s EXTRA=-DLIME_PART_TIMING -B ../liblime_hip.so && mkdir -p variants && cp lime_amd/liblime_hip.so variants/lib_ppt.so && make -C lime_amd/csrc -s -B ../liblime_hip.so
cp lime_amd/liblime_hip.so /tmp/lib_keep.so; cp variants/lib_ppt.so lime_amd/liblime_hip.so
for lines in 0 1; do
LIME_PART_LINES=$lines C3_PATHS=bin C3_MODE=1 C3_EBWT=1 C3_N=100000000 C3_NR=452000 C3_NG=678 python3 - <<'PY'
import os, sys, ctypes, io, contextlib, runpy
sys.path.insert(0, os.getcwd())
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path("tools/bench_c3.py", run_name="__main__")
from lime_amd import _lib
out = (ctypes.c_ulonglong * 8)()
_lib.load().lime_debug_part_times(out)
v = list(out); tot = sum(v) or 1
print("LINES=%s total wave-0 cycles %d (7 passes): " % (os.environ["LIME_PART_LINES"], tot) + "; ".join("p%d %.1f%%" % (i, 100.0 * x / tot) for i, x in enumerate(v) if x))
print(buf.getvalue()[-300:])
PY
done
cp /tmp/lib_keep.so lime_amd/liblime_hip.so
