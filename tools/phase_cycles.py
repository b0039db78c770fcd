"""Per-phase shader-clock accounting inside the fused solve kernel.
Needs a library built with -DILQR_PROFILE:  make -C iterativelqr.jl_amd/csrc clean all EXTRA=-DILQR_PROFILE"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ilqr_amd_loader import load_package
pkg = load_package()
cfg = sys.argv[1] if len(sys.argv) > 1 else "acrobot"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
model, T, x1, ub = pkg.workloads.make_inputs(cfg, B)
sol = pkg.Solver(model=model, horizon=T, batch=B, options=pkg.Options(verbose=0, **pkg.workloads.CONFIG_OPTIONS.get(cfg, {})))
for _ in range(2):
    sol.reset_(); sol.initialize_rollout_(x1, ub); sol.solve_()
sc = sol.buffer("_scalars"); st = sol.stats()
names = ["cost", "gradients", "backward", "delta", "rollout", "total"]
prof = sc[:, 10:16]
it = st["iterations"].astype(float); ro = st["rollouts"].astype(float)
tot = prof[:, 5].mean()
print("config %s B=%d: iterations %.1f rollouts %.1f; total %.3e ticks/instance" % (cfg, B, it.mean(), ro.mean(), tot))
for i, nm in enumerate(names[:5]):
    per = prof[:, i].sum() / (ro.sum() if nm in ("rollout",) else it.sum())
    print("  %-10s %6.1f%%  %9.0f ticks per %s" % (nm, 100 * prof[:, i].mean() / tot, per, "rollout" if nm == "rollout" else "iteration"))
print("  %-10s %6.1f%%" % ("other", 100 * (1 - prof[:, :5].sum(1).mean() / tot)))
