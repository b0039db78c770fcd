"""Shape sweep: every (nx, nu) with nx, nu <= 4 x even / odd / tiny horizons x the three small-model kernels. For each case:
the fused solves of the kernel variants must agree (iteration counts exactly, trajectories to 1e-8), and the fused backward pass (max_iterations = 0)
with the staged one of the same variant. Catches shape-dependent code-generation hazards (the stale-P MFMA read of DESIGN §3.1
showed only for nu = 3 and even horizons).   usage: python tools/shape_sweep.py [nx,nu ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, sympy as sp
from ilqr_amd_loader import load_package
pkg = load_package()
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(n, m) for n in range(1, 5) for m in range(1, 5)]
bad = 0
for n, m in shapes:
    h = 0.1
    A = [[(-0.5 if i == j else 0.0) + 0.2 * np.cos(1.0 + i + 2 * j) for j in range(n)] for i in range(n)]
    Bm = [[np.sin(1.0 + 3 * i + j) for j in range(m)] for i in range(n)]
    f = lambda x, u: [x[i] + h * (sum(A[i][j] * x[j] for j in range(n)) + sum(Bm[i][j] * u[j] for j in range(m)) + 0.3 * sp.sin(x[i])) for i in range(n)]
    dyn = pkg.Dynamics(f, n, m)
    stage = pkg.Cost(lambda x, u: 0.5 * sum(xi * xi for xi in x) + 0.05 * sum((1 + j) * u[j] * u[j] for j in range(m)) + 0.01 * u[0] * u[m - 1], n, m)
    term = pkg.Cost(lambda x, u: 5.0 * sum(xi * xi for xi in x), n, 0)
    box = pkg.Constraint(lambda x, u: [u[0] - 0.8, -0.8 - u[0]], n, m, indices_inequality=[1, 2])
    goal = pkg.Constraint(lambda x, u: [x[0] - 0.3], n, 0)
    for T in (2, 3, 4, 5, 8, 9, 17, 18, 33):
        B = 6
        rng = np.random.default_rng(1000 * n + 100 * m + T)
        x1 = rng.standard_normal((B, n)); ub = 0.3 * rng.standard_normal((B, T - 1, m))
        res = {}
        for v in ("latency", "throughput", "packed"):
            for mode, opt in (("solve", dict()), ("bp_fused", dict(max_iterations=0, max_dual_updates=1)), ("bp_staged", dict(max_iterations=0, max_dual_updates=1))):
                sol = pkg.Solver([dyn] * (T - 1), [stage] * (T - 1) + [term], [box] * (T - 1) + [goal], batch=B,
                                 options=pkg.Options(verbose=0, **opt), name="sw%d%d" % (n, m))
                sol.set_kernel_variant_(v)
                sol.initialize_rollout_(x1, ub)
                if mode == "bp_staged":
                    for st in ("al_begin", "cost_nominal", "gradients", "backward_pass"):
                        sol.run_stage_(st)
                else:
                    sol.solve_()
                K, k = sol.get_policy(); x, u = sol.get_trajectory(); s = sol.stats()
                res[(v, mode)] = (K, k, x, u, s["iterations"], s["gradient_norm"])
                sol.close()
        msgs = []
        for v in ("latency", "throughput", "packed"):
            a, b = res[(v, "bp_fused")], res[(v, "bp_staged")]
            if not (np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[5], b[5], equal_nan=True)):
                msgs.append("%s fused!=staged backward pass (dK %.1e)" % (v, np.abs(a[0] - b[0]).max()))
        a = res[("latency", "solve")]
        for v in ("throughput", "packed"):
            b = res[(v, "solve")]
            # (the packed kernel forms the Gauss-Newton AL terms of STAGE constraints symbolically, the LDS kernels as dense products:
            # once a stage constraint is active the variants agree to rounding, not bitwise — same iteration counts required)
            if not (np.array_equal(a[4], b[4]) and np.nanmax(np.abs(a[2] - b[2])) <= 1e-8 * (1.0 + np.nanmax(np.abs(a[2])))):
                msgs.append("%s solve != latency (it %s vs %s, dx %.1e)" % (v, a[4], b[4], np.nanmax(np.abs(a[2] - b[2]))))
        bad += len(msgs)
        print("nx=%d nu=%d T=%2d: %s" % (n, m, T, "ok" if not msgs else "; ".join(msgs)), flush=True)
print("MISMATCHES:", bad)
