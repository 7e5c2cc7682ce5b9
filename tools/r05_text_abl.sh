#!/bin/bash
cp lime_amd/liblime_hip.so /tmp/lib_keep.so; cp variants/lib_abl.so lime_amd/liblime_hip.so
for w in text_spread c2_clustered_n; do
for k in 1 3 4 10 11 8 0; do LIME_ABLATE=$k python3 tools/r05_text_abl.py text_spread 2>&1 | grep ablate; done
break
done
cp /tmp/lib_keep.so lime_amd/liblime_hip.so
