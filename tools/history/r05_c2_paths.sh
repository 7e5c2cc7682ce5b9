#!/bin/bash
# configs[1] and the same shape on the clustered generator: the steady pass by update path (after the binned pass's small launches were merged)
for p in cas bin cas bin; do
  echo "== LIME_UPDATE_PATH=$p"
  LIME_UPDATE_PATH=$p python3 tools/r05_probe.py 1e8,100000,500,1,0 1e8,100000,500,0,0 2e8,100000,500,1,0 2>&1 | grep -v amdgpu.ids
done
