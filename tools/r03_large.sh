#!/bin/bash
# large-path profile on one box: sub-phase split of the Riccati step, phase split, bench lines; A = lib_prof (previous build)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
TAG=${1:-x}
export ILQR_LIB=$PWD/iterativelqr.jl_amd/lib_sub/libilqr_hip.so
python tools/subphase_cycles.py synth32_tight 512 "A uh|T,B Qux Quu|T,C0 +g potrf,C1 potrs,C wait,D" > gpurun_out/r03/sub_$TAG.txt 2>&1
export ILQR_LIB=$PWD/iterativelqr.jl_amd/lib_phase/libilqr_hip.so
python tools/phase_cycles.py synth32 512 > gpurun_out/r03/phase_$TAG.txt 2>&1
python tools/phase_cycles.py synth32_tight 512 >> gpurun_out/r03/phase_$TAG.txt 2>&1
unset ILQR_LIB
for cfg in "synth32 512" "synth32_tight 512" "synth32_tight11 512"; do
  set -- $cfg
  python bench.py --config $1 --batch $2 --steps 5 --warmup 1 --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/r03/bench_$TAG.jsonl
done
