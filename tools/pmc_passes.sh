#!/bin/bash
# PMC passes over `bench.py --pmc-child` (2 solves of one config / batch / kernel variant), one rocprofv3 run per counter
# group (no --pmc together with other trace domains). Usage: tools/pmc_passes.sh <tag> <config> <batch> <variant>
set -u
TAG=${1:-pmc}; CFG=${2:-acrobot}; B=${3:-1024}; VAR=${4:-auto}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ARGS="bench.py --pmc-child --config $CFG --batch $B --steps 2 --variant $VAR"
i=0
for GROUP in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $GROUP -d $OUT/p$i -o r -- python3 $ARGS > $OUT/p$i.log 2>&1 || echo "pass $i ($GROUP) failed: $(tail -2 $OUT/p$i.log)"
done
python3 - <<PY
import glob, sqlite3
for db in sorted(glob.glob("$OUT/p*/**/*.db", recursive=True)):
    cur = sqlite3.connect(db).cursor()
    try:
        rows = list(cur.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"))
    except Exception as e:
        print(db, e); continue
    for name, cname, tot, n in rows:
        if "solve_kernel" in name:
            print("%-60s %-30s per_dispatch=%.6g (n=%d)" % (name[:60], cname, tot / n, n))
    try:
        for name, avg, n in cur.execute("select name, avg(duration), count(*) from kernels group by name"):
            if "solve_kernel" in name: print("   duration avg %.3f ms (n=%d) %s" % (avg / 1e6, n, name[:50]))
    except Exception as e:
        pass
PY
