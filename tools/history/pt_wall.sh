#!/bin/bash
# tools/pt_wall.sh -- start/end wall clock of every wave of the scan (variants/lib_wall.so, -DLIME_WALL_TIMING): how uneven
# the waves' finishing times are
cp lime_amd/liblime_hip.so /tmp/lib_keep.so
cp variants/lib_wall.so lime_amd/liblime_hip.so
C3_WALL=1 C3_PATHS=${C3_PATHS:-bin} C3_EBWT=${C3_EBWT:-0} C3_NR=${C3_NR:-1000000} C3_N=${C3_N:-1000000000} C3_NG=${C3_NG:-5000} python3 tools/bench_c3.py 2>/dev/null | grep -v "^{"
cp /tmp/lib_keep.so lime_amd/liblime_hip.so
