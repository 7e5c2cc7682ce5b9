#!/bin/bash
# tools/r05_part_phases.sh -- k_part_lines: absolute cycles of wave 0 per phase and per tile, text-like (1e8 symbols, clustered, 306 MB table) against 2e9 iid symbols with a 1 GB table
# (variants/lib_ppt.so: -DLIME_PART_TIMING)
export LIME_LIB=$PWD/variants/lib_ppt.so LIME_TEST_HOOKS=1      # (round 6: the variant is LOADED, not copied over the installed library -- ADVICE r5)
for shape in "100000000 452000 678 1 1" "2000000000 1000000 1000 0 0"; do
set -- $shape
LIME_NO_PROBE=1 C3_PATHS=bin C3_N=$1 C3_NR=$2 C3_NG=$3 C3_EBWT=$4 C3_MODE=$5 python3 - <<'PY'
import os, sys, ctypes, io, contextlib, runpy, json
sys.path.insert(0, os.getcwd())
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path("tools/bench_c3.py", run_name="__main__")
from lime_amd import _lib
out = (ctypes.c_ulonglong * 10)()
_lib.load().lime_debug_part_times(out)
d = json.loads([l for l in buf.getvalue().splitlines() if l.startswith("{")][-1])
rec = d["table_updates_bin"]; passes = 7
v = list(out)[:8]; tot = sum(v) or 1
pro, epi = out[8], out[9]
tiles_per_wg = rec / 8192 / 512
print("N=%s: %d records = %.1f tiles per producer; wave 0 of 512 workgroups, %d passes: %.0f cycles per workgroup and pass = %.0f per tile" % (
    os.environ["C3_N"], rec, tiles_per_wg, passes, tot / 512 / passes, tot / 512 / passes / tiles_per_wg))
names = ["p0 scan top", "p1 barrier", "p2 scan", "p3 stage", "p4 barrier", "p5 count next", "p6 lines out + loads", "p7 carries"]
print("   before the tile loop %.0f cycles per workgroup and pass, behind it %.0f (wave 0)" % (pro / 512 / passes, epi / 512 / passes))
print("   " + "; ".join("%s %.1f%% (%.0f/tile)" % (n, 100.0 * x / tot, x / 512 / passes / tiles_per_wg) for n, x in zip(names, v) if x))
PY
done
# (nothing to restore)
