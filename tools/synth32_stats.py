import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from ilqr_amd_loader import load_package
pkg = load_package()
for cfg in ("synth32", "synth32_tight"):
    B = 512
    model, T, x1, ub = pkg.workloads.make_inputs(cfg, B)
    sol = pkg.Solver(model=model, horizon=T, batch=B, options=pkg.Options(verbose=0, **pkg.workloads.CONFIG_OPTIONS.get(cfg, {})))
    for _ in range(2):
        sol.reset_(); sol.initialize_rollout_(x1, ub); sol.timing_reset(); sol.solve_()
    ms, _ = sol.timing(); st = sol.stats()
    print(cfg, "kernel %.2f ms; iterations mean %.1f max %d; rollouts mean %.1f max %d; outer mean %.2f; conv %.3f" % (
        ms, st["iterations"].mean(), st["iterations"].max(), st["rollouts"].mean(), st["rollouts"].max(), st["outer_iterations"].mean(), (st["max_violation"] <= 5e-3).mean()))
    sol.close()
