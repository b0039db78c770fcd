"""Solve-kernel time vs batch size (occupancy / residency check)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ilqr_amd_loader import load_package
pkg = load_package()
cfg = sys.argv[1] if len(sys.argv) > 1 else "acrobot"
for B in [int(b) for b in (sys.argv[2:] or ["128", "256", "512", "768", "1024", "2048", "4096"])]:
    model, T, x1, ub = pkg.workloads.make_inputs(cfg, B)
    sol = pkg.Solver(model=model, horizon=T, batch=B, options=pkg.Options(verbose=0))
    sol.set_kernel_variant_(os.environ.get("ILQR_VARIANT", "auto"))
    for _ in range(2):
        sol.reset_(); sol.initialize_rollout_(x1, ub); sol.timing_reset(); sol.solve_()
    ms, n = sol.timing()
    sc = sol.buffer("_scalars")
    ticks = sc[:, 15]
    st = sol.stats()
    print("B=%5d kernel %.2f ms  -> %.0f traj/s; iterations mean %.1f max %d; ticks/instance mean %.3e max %.3e (-> %.2f GHz if resident)" % (
        B, ms, B / ms * 1e3, st["iterations"].mean(), st["iterations"].max(), ticks.mean(), ticks.max(), ticks.max() / ms / 1e6))
    sol.close()
