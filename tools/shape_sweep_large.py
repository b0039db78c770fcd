"""Large-path twin of shape_sweep.py: (nx, nu) beyond 4 x even / odd / tiny horizons — the fused backward pass
(max_iterations = 0) must equal the staged one bitwise, and a short fused solve must be finite and reproducible."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from ilqr_amd_loader import load_package
pkg = load_package()
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(5, 1), (8, 3), (17, 2), (33, 9)]
bad = 0
for n, m in shapes:
    mdl = pkg.models.synth_nm(n, m)
    for T in (2, 3, 4, 5, 9, 10, 21):
        B = 4
        rng = np.random.default_rng(100 * n + m + T)
        x1 = 0.5 * rng.standard_normal((B, n)); ub = 0.4 * rng.standard_normal((B, T - 1, m))
        res = {}
        for mode, opt in (("solve", dict(max_iterations=5, max_dual_updates=2)), ("solve2", dict(max_iterations=5, max_dual_updates=2)),
                          ("bp_fused", dict(max_iterations=0, max_dual_updates=1)), ("bp_staged", dict(max_iterations=0, max_dual_updates=1))):
            sol = pkg.Solver([mdl["dynamics"]] * (T - 1), [mdl["cost_stage"]] * (T - 1) + [mdl["cost_term"]],
                             [mdl["con_stage"]] * (T - 1) + [mdl["con_term"]], batch=B, options=pkg.Options(verbose=0, **opt), name="lg%d_%d" % (n, m))
            sol.initialize_rollout_(x1, ub)
            if mode == "bp_staged":
                for st in ("al_begin", "cost_nominal", "gradients", "backward_pass"):
                    sol.run_stage_(st)
            else:
                sol.solve_()
            K, k = sol.get_policy(); x, u = sol.get_trajectory(); s = sol.stats()
            res[mode] = (K, k, x, u, s["iterations"], s["gradient_norm"])
            sol.close()
        msgs = []
        a, b = res["bp_fused"], res["bp_staged"]
        if not (np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[5], b[5], equal_nan=True)):
            msgs.append("fused!=staged backward pass (dK %.1e)" % np.abs(a[0] - b[0]).max())
        a, b = res["solve"], res["solve2"]
        if not (np.isfinite(a[2]).all() and np.array_equal(a[2], b[2]) and np.array_equal(a[0], b[0])):
            msgs.append("solve not finite / not reproducible")
        bad += len(msgs)
        print("nx=%d nu=%d T=%2d: %s" % (n, m, T, "ok" if not msgs else "; ".join(msgs)), flush=True)
print("MISMATCHES:", bad)
