#!/bin/bash
# Everything a round's profiles/ holds, on one box:  tools/grun --timeout 3000 -- 'bash tools/capture_round.sh r05'
# (profile libraries first: make -C iterativelqr.jl_amd/csrc LIBDIR=../lib_prof1 EXTRA=-DILQR_PROFILE EXTRA_API=-DILQR_PROFILE and
#  LIBDIR=../lib_prof EXTRA="-DILQR_PROFILE -DILQR_PROFILE_SUB" EXTRA_API=-DILQR_PROFILE)
R=${1:-r06}
cd $GRAFT_REPO_ROOT
O=gpurun_out/${R}_final; mkdir -p $O
TREE="# tree $(cat .tree_id 2>/dev/null || echo unstamped)"
bash tools/capture_all.sh $R
( echo "# whole-solve parity against the CPU oracle (a restatement: the reference cannot run here) at BASELINE sizes; ${TREE#\# }"; python tools/parity_report.py 2>&1 | grep -v "^#" ) > $O/parity.txt
( echo "$TREE"; python tools/det_check_all.py 2>&1 ) > $O/determinism.txt
( echo "$TREE"; python tools/all_shards.py 264 2>&1 ) > $O/all_shards.txt
( echo "$TREE"; python tools/straggler_signature.py 2,6 2>&1 ) > $O/straggler_signature.txt
( echo "$TREE"; cd tools/probes && /opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 probe_evict.hip -o probe_evict 2>/dev/null; ./probe_evict 4 2>&1 ) > $O/probe_evict.txt
( echo "$TREE  (phases: library built with -DILQR_PROFILE; sub-phases of the large Riccati step: -DILQR_PROFILE -DILQR_PROFILE_SUB; ticks = shader clock)"
  export ILQR_LIB=$PWD/iterativelqr.jl_amd/lib_prof1/libilqr_hip.so
  python tools/phase_cycles.py acrobot 1024 splitmix64 slowest 2>&1      # the batch and, beside it, its last finisher alone on the chip
  python tools/phase_cycles.py acrobot 1024 pcg64 slowest 2>&1
  for a in "car 1024" "synth32_tight 512" "synth32_tight11 512"; do python tools/phase_cycles.py $a pcg64 2>&1; done
  export ILQR_LIB=$PWD/iterativelqr.jl_amd/lib_prof/libilqr_hip.so
  for a in "synth32_tight 512" "synth32_tight 256" "synth32_tight11 512"; do echo "sub-phases $a:"; python tools/subphase_cycles.py $a "A,B,C chain 1,C chain 2,C wait,D" 2>&1; done ) > $O/phase_cycles.txt
python bench.py --generator pcg64 2>/dev/null | tail -1 > $O/bench_default_pcg64.json
( echo "$TREE"; FINISH_REPS=6 python tools/finish_times.py acrobot 1024 splitmix64 2>&1 ) > $O/finish_times_splitmix64.txt
( echo "$TREE"; FINISH_REPS=6 python tools/finish_times.py acrobot 1024 pcg64 2>&1 ) > $O/finish_times_pcg64.txt
( echo "$TREE  (roles as launched, ILQR_ROLE_SLOTS=0: what the consensus of DESIGN 3.0 removes)"; ILQR_ROLE_SLOTS=0 python tools/finish_times.py acrobot 1024 splitmix64 2>&1 ) > $O/finish_times_roles_as_launched.txt
( echo "$TREE"; python tools/finish_times_large.py synth32_tight11 512 2>&1 ) > $O/finish_times_large.txt
ILQR_BENCH_SHARE_DEVICE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_2ranks_one_gpu.json
ILQR_BENCH_SHARE_DEVICE=1 python bench.py --sharded-handle --gpus 2 --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_sharded_handle_2x_one_gpu.json
ls -la $O
