#!/bin/bash
# tools/abl_c3.sh [lib ...] -- scan time on configs[2] with the scan cut after phase k (libraries built with -DLIME_ABLATE_BUILD;
# results invalid).  Several libraries are compared inside ONE run: boxes differ by ~10 %.
cp lime_amd/liblime_hip.so /tmp/lib_keep.so
for rep in 1 2; do
for lib in ${@:-variants/lib_abl.so}; do
  cp $lib lime_amd/liblime_hip.so
  for k in ${ABL_LIST:-1 3 4 10 11 0}; do
    echo -n "$lib ablate=$k  "
    LIME_ABLATE=$k C3_PATHS=${C3_PATHS:-bin} python3 tools/bench_c3.py 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); p=d[[k for k in d if k in ('bin','cas')][0]]; print('scan ms', round(p['parts_ms']['scan'],3), 'pass ms', round(p['parts_ms']['pass'],3))"
  done
done
done
cp /tmp/lib_keep.so lime_amd/liblime_hip.so
