#!/bin/bash
# Final-profile capture on the GPU box (run through gpurun): kernel trace + separate PMC passes, as
# MI355X_MICROARCH.md prescribes (no --pmc together with other trace domains). Outputs under gpurun_out/$1.
set -u
TAG=${1:-prof_final}
CFG=${2:-acrobot}
B=${3:-1024}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ARGS="bench.py --config $CFG --batch $B --steps 4 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/kt -o r1 -- python3 $ARGS > $OUT/kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o r1 -- python3 bench.py --config $CFG --batch $B --steps 2 --warmup 1 --no-cpu-baseline > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o r1 -- python3 bench.py --config $CFG --batch $B --steps 2 --warmup 1 --no-cpu-baseline > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY -d $OUT/sq -o r1 -- python3 bench.py --config $CFG --batch $B --steps 2 --warmup 1 --no-cpu-baseline > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/mfma -o r1 -- python3 bench.py --config $CFG --batch $B --steps 2 --warmup 1 --no-cpu-baseline > $OUT/mfma.log 2>&1
find $OUT -name "*.db" | head
DB() { find $OUT/$1 -name "*results.db" | head -1; }
python3 tools/rocprof_summary.py $(DB kt) $(DB fetch) $(DB write) $(DB sq) $(DB mfma) > $OUT/summary.txt 2>&1
tail -1 $OUT/kt.log > $OUT/bench_under_profiler.json
cat $OUT/summary.txt
