#!/bin/bash
# Profile capture on the GPU box (run through gpurun): kernel trace + separate PMC passes, as MI355X_MICROARCH.md prescribes
# (no --pmc together with other trace domains). The profiled command is bench.py's timed loop alone (--no-secondary: the default
# run's extra figures launch the same kernel two at a time, which would enter the per-kernel average). Outputs under gpurun_out/$1; copy the summary into profiles/.
#   tools/capture_profiles.sh <tag> <config> <batch> [variant]
set -u
TAG=${1:-prof_final}
CFG=${2:-acrobot}
B=${3:-1024}
VAR=${4:-auto}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ARGS="bench.py --config $CFG --batch $B --variant $VAR --steps 4 --warmup 2 --no-cpu-baseline --no-pmc --no-secondary"
SHORT="bench.py --config $CFG --batch $B --variant $VAR --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-secondary"
rocprofv3 --kernel-trace --stats -d $OUT/kt -o r1 -- python3 $ARGS > $OUT/kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o r1 -- python3 $SHORT > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o r1 -- python3 $SHORT > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/sq -o r1 -- python3 $SHORT > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $OUT/mfma -o r1 -- python3 $SHORT > $OUT/mfma.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $OUT/tcc -o r1 -- python3 $SHORT > $OUT/tcc.log 2>&1
DB() { find $OUT/$1 -name "*results.db" | head -1; }
( echo "# $TAG: python3 $ARGS   (tree $(cat $GRAFT_REPO_ROOT/.tree_id 2>/dev/null || echo unstamped: run through tools/grun))"
  python3 tools/rocprof_summary.py $(DB kt) $(DB fetch) $(DB write) $(DB sq) $(DB mfma) $(DB tcc)
  echo; echo "# bench line under the kernel-trace pass"; grep "^{\"metric\"" $OUT/kt.log | tail -1 ) > $OUT/summary.txt 2>&1
rm -rf $OUT/kt $OUT/fetch $OUT/write $OUT/sq $OUT/mfma $OUT/tcc
cat $OUT/summary.txt | head -40
