#!/bin/bash
cp lime_amd/liblime_hip.so /tmp/lib_keep.so; cp variants/lib_apt.so lime_amd/liblime_hip.so
python3 tools/r05_apply_phases.py 2>&1 | grep -v amdgpu.ids
cp /tmp/lib_keep.so lime_amd/liblime_hip.so
