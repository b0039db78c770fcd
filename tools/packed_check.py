import sys, time; import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
from ilqr_amd_loader import load_package
pkg = load_package()
for cfg, B in (("particle", 7), ("car", 9), ("acrobot51", 6), ("acrobot", 64), ("car_goal", 16)):
    model, T, x1, ub = pkg.workloads.make_inputs(cfg, B)
    out = {}
    for v in ("latency", "packed"):
        sol = pkg.Solver(model=model, horizon=T, batch=B, options=pkg.Options(verbose=0))
        sol.set_kernel_variant_(v); sol.initialize_rollout_(x1, ub)
        t0 = time.time(); sol.solve_(); dt = time.time() - t0
        out[v] = (sol.get_trajectory(), sol.get_policy(), sol.stats(), dt)
        sol.close()
    a, b = out["latency"], out["packed"]
    same = (a[2]["iterations"] == b[2]["iterations"]) & (a[2]["rollouts"] == b[2]["rollouts"]) & (a[2]["outer_iterations"] == b[2]["outer_iterations"])
    print(cfg, "same control flow", same.mean(), "dx", np.abs(a[0][0] - b[0][0]).max(), "du", np.abs(a[0][1] - b[0][1]).max(),
          "dK", np.abs(a[1][0] - b[1][0]).max() / np.abs(a[1][0]).max(), "it", a[2]["iterations"][:4], b[2]["iterations"][:4], "t %.3f %.3f" % (a[3], b[3]))
