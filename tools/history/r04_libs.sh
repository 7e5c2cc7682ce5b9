#!/bin/bash
# tools/r04_libs.sh "lib1 lib2" WORKLOAD... -- tools/r04_ab.sh for several builds of the library inside ONE run
LIBS=$1; shift
cp lime_amd/liblime_hip.so /tmp/lib_keep.so
for lib in $LIBS; do
  cp $lib lime_amd/liblime_hip.so
  echo "#### $lib"
  bash tools/r04_ab.sh "LIME_X=0" "$@"
done
cp /tmp/lib_keep.so lime_amd/liblime_hip.so
