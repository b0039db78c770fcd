#!/bin/bash
# profile set of a round on one box (tools/capture_all.sh r05): rocprofv3 kernel-trace + PMC summaries per config (tools/capture_profiles.sh) and the bench lines
R=${1:-r05}
cd $GRAFT_REPO_ROOT
bash tools/capture_profiles.sh ${R}_acrobot_b1024 acrobot 1024 auto > /dev/null 2>&1
bash tools/capture_profiles.sh ${R}_acrobot_b8192_packed acrobot 8192 auto > /dev/null 2>&1
bash tools/capture_profiles.sh ${R}_car_b4096_packed car 4096 auto > /dev/null 2>&1
bash tools/capture_profiles.sh ${R}_car_b1024 car 1024 auto > /dev/null 2>&1
bash tools/capture_profiles.sh ${R}_synth32_b512 synth32 512 auto > /dev/null 2>&1
bash tools/capture_profiles.sh ${R}_synth32_tight_b512 synth32_tight 512 auto > /dev/null 2>&1
bash tools/capture_profiles.sh ${R}_synth32_tight11_b512 synth32_tight11 512 auto > /dev/null 2>&1
bash tools/capture_profiles.sh ${R}_synth12_b4096_mid synth12 4096 auto > /dev/null 2>&1
bash tools/capture_profiles.sh ${R}_synth12_b4096_fourwave synth12 4096 latency > /dev/null 2>&1
mkdir -p gpurun_out/${R}_bench
python bench.py > gpurun_out/${R}_bench/bench_default.json 2> gpurun_out/${R}_bench/bench_default.err
for cfg in "synth12 4096" "synth32 512" "synth32_tight 512" "synth32_tight11 512" "car 4096" "acrobot 8192" "car 1024"; do
  set -- $cfg
  python bench.py --config $1 --batch $2 --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${R}_bench/bench_$1_$2.json
done
