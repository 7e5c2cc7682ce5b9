#!/bin/bash
# tools/r03_valu_abl.sh "k..." -- instructions per window of the scan with parts cut (variants/lib_abl.so, LIME_ABLATE=k)
for k in $1; do echo -n "ablate=$k  "; LIME_ABLATE=$k bash tools/r03_valu.sh variants/lib_abl.so; done
