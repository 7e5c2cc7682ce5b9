#!/bin/bash
# tools/ablate.sh -- time k_tile cut after each phase (LIME_ABLATE; results are invalid, timing only)
run() { python3 bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('k_tile ms', round(d['roofline']['kernel_ms_avg'],4), 'step ms', round(d['ms_per_step'],4), 'GB/s', round(d['roofline']['achieved'],1))"; }
for a in 1 3 4 5 0; do echo -n "ablate=$a  "; LIME_ABLATE=$a run; done
for mb in 256 512 768 1024 2048 100000; do echo -n "max_blocks=$mb  "; LIME_MAX_BLOCKS=$mb run; done
echo -n "mode1 (correlated symbols) "; python3 bench.py --steps 10 --warmup 2 --no-cpu --mode 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('k_tile ms', round(d['roofline']['kernel_ms_avg'],4), 'step ms', round(d['ms_per_step'],4), 'updates', d['config']['table_updates'], 'clusters', d['config']['n_clusters'])"
