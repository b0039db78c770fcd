#!/bin/bash
# Same-box A/B of two builds of the library: iterativelqr.jl_amd/lib_prof/libilqr_hip.so (A, e.g. the previous commit) against
# iterativelqr.jl_amd/lib/libilqr_hip.so (B), alternating.  tools/ab.sh <config> <variant> <batch...>
CFG=$1; VAR=$2; shift 2
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for which in A B; do
    if [ $which = A ]; then export ILQR_LIB=$PWD/iterativelqr.jl_amd/lib_prof/libilqr_hip.so; else unset ILQR_LIB; fi
    ILQR_VARIANT=$VAR python tools/batch_sweep.py $CFG "$@" | grep "B=" | cut -c1-62 | sed "s/^/$which $CFG $VAR /"
  done
done
