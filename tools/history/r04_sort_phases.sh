#!/bin/bash
# tools/r04_sort_phases.sh -- where k_sort_tiles' cycles go (variants/lib_stt.so: -DLIME_SORT_TIMING; the wait for a tile's loads is forced to
# vmcnt(0) there): per phase, summed over wave 0 of every workgroup
# build the instrumented library first:  make -C lime_amd/csrc -s EXTRA=-DLIME_SORT_TIMING -B ../liblime_hip.so && mkdir -p variants && cp lime_amd/liblime_hip.so variants/lib_stt.so && make -C lime_amd/csrc -s -B ../liblime_hip.so
cp lime_amd/liblime_hip.so /tmp/lib_keep.so; cp variants/lib_stt.so lime_amd/liblime_hip.so
for shape in "1000000000 1000000 5000" "10000000000 1000000 1000" "100000000 452000 678"; do
set -- $shape
C3_PATHS=bin C3_N=$1 C3_NR=$2 C3_NG=$3 python3 - <<'PY'
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
names = ["wait for the tile's loads", "ranks (16 returning LDS adds)", "next tile's loads issued", "barrier 1", "scan of the regions' counts (2 barriers)", "records to their stage slots", "barrier 4", "row written out"]
print("N=%s table %sx%s: k_sort_tiles phases (share of wave 0's cycles): " % (os.environ["C3_N"], os.environ["C3_NR"], os.environ["C3_NG"]) + "; ".join("%s %.1f%%" % (n, 100.0 * x / tot) for n, x in zip(names, v) if x))
PY
done
cp /tmp/lib_keep.so lime_amd/liblime_hip.so
