#!/bin/bash
# tools/profile.sh TAG -- rocprofv3 kernel-trace + separate PMC passes of `bench.py --no-cpu`
# on the GPU box; writes under gpurun_out/prof_TAG (copy the summaries into profiles/ by hand).
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 bench.py --steps 5 --warmup 2 --no-cpu"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
echo "trace rc=$?"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc1 -- $CMD > $OUT/pmc1.log 2>&1
echo "pmc1 rc=$?"
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc2 -- $CMD > $OUT/pmc2.log 2>&1
echo "pmc2 rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- $CMD > $OUT/pmc3.log 2>&1
echo "pmc3 rc=$?"
rocprofv3 --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum --output-format csv -d $OUT/pmc4 -- $CMD > $OUT/pmc4.log 2>&1
echo "pmc4 rc=$?"
find $OUT -name "*.csv" | head -40
