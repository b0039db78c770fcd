#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
export ILQR_LIB=$PWD/iterativelqr.jl_amd/lib_sub/libilqr_hip.so
for B in 1 256 512; do
  echo "B=$B" >> gpurun_out/r03/sub_$1.txt
  python tools/subphase_cycles.py synth32_tight $B "A uh|T,B Qux Quu|T,C0 +g potrf,C1 potrs,C wait,D" >> gpurun_out/r03/sub_$1.txt 2>&1
done
