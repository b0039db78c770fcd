#!/bin/bash
# Same-box A/B through bench.py (applies the config's solver options, e.g. synth32_tight11): A = lib_prof/ build, B = lib/ build.
#   tools/ab_bench.sh <config> <batch> [variant]
CFG=$1; B=$2; VAR=${3:-auto}
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for which in A B; do
    if [ $which = A ]; then export ILQR_LIB=$PWD/iterativelqr.jl_amd/lib_prof/libilqr_hip.so; else unset ILQR_LIB; fi
    python bench.py --config $CFG --batch $B --variant $VAR --steps 4 --warmup 1 --no-pmc --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$which $CFG B=$B kernel %.3f ms  frac %.3f  iterations mean %.1f max %d' % (r['kernel_ms_avg'], r['frac'], d['solve_stats']['inner_iterations_mean'], d['solve_stats']['iterations_max']))"
  done
done
