#!/bin/bash
export LIME_LIB=$PWD/variants/lib_abl.so LIME_TEST_HOOKS=1      # (round 6: the variant is LOADED, not copied over the installed library -- ADVICE r5)
for w in text_spread c2_clustered_n; do
for k in 1 3 4 10 11 8 0; do LIME_ABLATE=$k python3 tools/r05_text_abl.py text_spread 2>&1 | grep ablate; done
break
done
# (nothing to restore)
