#!/bin/bash
# tools/r05_lds_abl.sh [lib] -- LDS bank conflicts of the scan by phase: the scan cut after phase k (LIME_ABLATE build; results invalid), configs[2]
# binned, counters per launch of k_scan.  Differences between consecutive cuts attribute the conflicts (and the LDS / vector instructions) to
# the phases: 1 = loads + staging, 3 = + chunk acceptance, 4 = + cluster list, 10 = + lengths / round bookkeeping, 11 = + 2-4-symbol scoring,
# 0 = everything (rows, repeats, 17-64, record drains).
export TMPDIR=/tmp
export LIME_NO_PROBE=1
export LIME_LIB=$PWD/${1:-variants/lib_abl.so} LIME_TEST_HOOKS=1      # (round 6: the variant is LOADED instead of copied over the installed library -- ADVICE r5)
for k in 1 3 4 10 11 0; do
  echo "== ablate=$k"
  LIME_ABLATE=$k C3_PATHS=bin bash tools/pmc_c3.sh "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" 'k_scan<'
done
# (nothing to restore)
