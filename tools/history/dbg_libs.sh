#!/bin/bash
# tools/dbg_libs.sh "golden names" lib... -- tools/dbg_golden.py once per library build
G="$1"; shift
cp lime_amd/liblime_hip.so /tmp/lib_keep.so
for lib in "$@"; do
  cp $lib lime_amd/liblime_hip.so
  echo "== $lib"
  timeout -k 10 120 python3 tools/dbg_golden.py $G 2>&1 | grep -v amdgpu.ids
done
cp /tmp/lib_keep.so lime_amd/liblime_hip.so
