"""Per-phase shader-clock ticks (wave 0) of the large path on synth12, four-wave kernel against the one-wave variant.
Needs a -DILQR_PROFILE build: ILQR_LIB=iterativelqr.jl_amd/lib_phase/libilqr_hip.so python tools/mid_phases.py [B]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ilqr_amd_loader import load_package
pkg = load_package()
T = 101
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
mdl = pkg.models.synth12()
kw = dict(max_iterations=15, max_dual_updates=3)
rng = np.random.default_rng(12)
x1 = 0.5 * rng.standard_normal((B, 12)); ub = 0.1 * rng.standard_normal((B, T - 1, 5))
# the generated model as a module compiled with -DILQR_PROFILE like the library (tools: hipcc ... -DILQR_PROFILE s12.hip -> lib_phase/)
import ctypes
ctypes.CDLL(os.path.join(os.path.dirname(os.environ["ILQR_LIB"]), "libmodel_synth12_prof.so"), mode=ctypes.RTLD_GLOBAL)
for variant in ("latency", "mid"):
    s = pkg.Solver(model="synth12_midtest", horizon=T, batch=B, options=pkg.Options(verbose=0, **kw))
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
