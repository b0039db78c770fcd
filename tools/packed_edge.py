"""Edge shapes on the packed kernel against the latency kernel: minimal horizons (N = 1, 2), horizons around the 16-step
chunk boundary, batches of 1..5."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from ilqr_amd_loader import load_package
pkg = load_package()
rng = np.random.default_rng(3)
worst = 0.0
for model, n, m in (("particle", 2, 1), ("car", 3, 2)):
    for T in (2, 3, 16, 17, 18, 33, 34):
        for B in (1, 2, 3, 5):
            x1 = 0.1 * rng.standard_normal((B, n)); ub = 0.05 * rng.standard_normal((B, T - 1, m))
            res = []
            for v in ("latency", "packed"):
                sol = pkg.Solver(model=model, horizon=T, batch=B, options=pkg.Options(verbose=0))
                sol.set_kernel_variant_(v); sol.initialize_rollout_(x1, ub); sol.solve_()
                res.append((sol.get_trajectory(), sol.get_policy(), sol.stats())); sol.close()
            a, b = res
            ok = (a[2]["iterations"] == b[2]["iterations"]).all() and (a[2]["rollouts"] == b[2]["rollouts"]).all()
            dx = np.abs(a[0][0] - b[0][0]).max(); dK = np.abs(a[1][0] - b[1][0]).max() / max(1.0, np.abs(a[1][0]).max())
            worst = max(worst, dx, dK)
            if not ok or dx > 1e-9 or dK > 1e-8 or not np.isfinite(b[0][0]).all():
                print("MISMATCH", model, T, B, ok, dx, dK, a[2]["iterations"], b[2]["iterations"])
print("edge shapes done, worst deviation %.2e" % worst)
