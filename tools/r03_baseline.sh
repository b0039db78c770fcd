#!/bin/bash
# round-3 baseline on one box: sub-phase split of the large Riccati step, phase split, bench lines of the configs to improve
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
export ILQR_LIB=$PWD/iterativelqr.jl_amd/lib_sub/libilqr_hip.so
python tools/subphase_cycles.py synth32 512 > gpurun_out/r03/sub_synth32.txt 2>&1
python tools/subphase_cycles.py synth32_tight 512 >> gpurun_out/r03/sub_synth32.txt 2>&1
export ILQR_LIB=$PWD/iterativelqr.jl_amd/lib_phase/libilqr_hip.so
python tools/phase_cycles.py synth32 512 > gpurun_out/r03/phase.txt 2>&1
python tools/phase_cycles.py acrobot 1024 >> gpurun_out/r03/phase.txt 2>&1
unset ILQR_LIB
for cfg in "synth32 512" "synth32_tight 512" "acrobot 1024" ; do
  set -- $cfg
  python bench.py --config $1 --batch $2 --steps 5 --warmup 1 --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/r03/bench_base.jsonl
done
python bench.py --config acrobot --batch 8192 --variant packed --steps 3 --warmup 1 --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/r03/bench_base.jsonl
python bench.py --config car --batch 4096 --variant packed --steps 5 --warmup 1 --no-pmc --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/r03/bench_base.jsonl
