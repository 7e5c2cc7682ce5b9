#!/bin/bash
# tools/profile.sh TAG [WORKLOAD] -- rocprofv3 kernel-trace + separate PMC passes of `bench.py --no-cpu --no-also
# --workload WORKLOAD` (default c3 = the headline) on the GPU box; writes under gpurun_out/prof_TAG, then
# tools/summarize_profile.py TAG WORKLOAD copies the summaries into profiles/ (run it on the box or afterwards).
# The program itself follows `--` (python3 bench.py ...): no env / bash -c hop under the profiler.
TAG=${1:-r06}
WL=${2:-c3}
# (round 5: without the sampled density probe in front of the context's first pass -- it is a short launch of the same k_scan instance and
# would pull the kernel's average in the trace and the counters 5 % below that of a pass; the bench line of the driver has it)
# (round 6: bench.py refuses LIME_* variables; its --no-probe flag does this)
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
ARGS="bench.py --steps ${STEPS:-5} --warmup 2 --no-cpu --no-also --no-probe --workload $WL"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
echo "trace rc=$?"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS --output-format csv -d $OUT/pmc1 -- python3 $ARGS > $OUT/pmc1.log 2>&1
echo "pmc1 rc=$?"
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc2 -- python3 $ARGS > $OUT/pmc2.log 2>&1
echo "pmc2 rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- python3 $ARGS > $OUT/pmc3.log 2>&1
echo "pmc3 rc=$?"
rocprofv3 --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum --output-format csv -d $OUT/pmc4 -- python3 $ARGS > $OUT/pmc4.log 2>&1
echo "pmc4 rc=$?"
grep -h '^{' $OUT/trace.log | tail -1 > $OUT/bench_line.json
python3 tools/summarize_profile.py $TAG $WL
