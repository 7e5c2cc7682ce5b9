#!/bin/bash
export TMPDIR=/tmp
for a in 6 7 0; do
  rm -rf gpurun_out/mt_$a; LIME_ABLATE=$a rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/mt_$a -- python3 bench.py --steps 5 --warmup 2 --no-cpu > /dev/null 2>&1
  echo "ablate=$a"; grep -E "k_score_med|k_scan" gpurun_out/mt_$a/*/*_kernel_stats.csv | cut -d, -f1,4 | cut -c1-120
done
