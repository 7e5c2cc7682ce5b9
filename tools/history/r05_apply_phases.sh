#!/bin/bash
export LIME_LIB=$PWD/variants/lib_apt.so LIME_TEST_HOOKS=1      # (round 6: the variant is LOADED, not copied over the installed library -- ADVICE r5)
python3 tools/r05_apply_phases.py 2>&1 | grep -v amdgpu.ids
# (nothing to restore)
