"""Per-phase shader-clock ticks (wave 0) of the large path on synth12, four-wave kernel against the one-wave variant.
Needs a -DILQR_PROFILE build:  make -C iterativelqr.jl_amd/csrc LIBDIR=../lib_phase EXTRA="-DILQR_PROFILE -DILQR_BUILTIN_ONLY=Model_synth12"
    ILQR_LIB=iterativelqr.jl_amd/lib_phase/libilqr_hip.so python tools/mid_phases.py [B]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ilqr_amd_loader import load_package
pkg = load_package()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
model, T, x1, ub = pkg.workloads.make_inputs("synth12", B)
kw = pkg.workloads.CONFIG_OPTIONS["synth12"]
for variant in ("latency", "mid"):
    s = pkg.Solver(model=model, horizon=T, batch=B, options=pkg.Options(verbose=0, **kw))
    s.set_kernel_variant_(variant)
    for _ in range(2):
        s.reset_(); s.initialize_rollout_(x1, ub); s.solve_()
    sc = s.buffer("_scalars"); st = s.stats()
    prof = sc[:, 10:16]
    it = st["iterations"].astype(float) + st["outer_iterations"]; ro = st["rollouts"].astype(float)
    print("%s B=%d: backward passes %.1f rollouts %.1f per instance; total %.3e ticks" % (variant, B, it.mean(), ro.mean(), prof[:, 5].mean()))
    for i, nm in enumerate(["cost", "gradients", "backward", "delta", "rollout"]):
        per = prof[:, i].sum() / (ro.sum() if nm == "rollout" else it.sum())
        print("   %-10s %5.1f%%  %9.0f ticks per %s" % (nm, 100 * prof[:, i].mean() / prof[:, 5].mean(), per, "rollout" if nm == "rollout" else "pass"))
    s.close()
