#!/bin/bash
# tools/quick.sh -- one bench line per setting (timing only)
run() { python3 bench.py --steps 10 --warmup 2 --no-cpu $EXTRA 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('k_scan ms', round(d['roofline']['kernel_ms_avg'],4), 'step ms', round(d['ms_per_step'],4), 'GB/s', round(d['roofline']['achieved'],1), 'upd', d['config']['table_updates'])"; }
for a in 1 3 4 10 11 5 0; do echo -n "ablate=$a  "; LIME_ABLATE=$a run; done
for mb in 1024; do echo -n "max_blocks=$mb  "; LIME_MAX_BLOCKS=$mb run; done
echo -n "mode1 "; EXTRA="--mode 1" run
