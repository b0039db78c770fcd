"""Large models whose matrices are single tiles (4 < nx <= 16): the four-wave kernel (two instances per CU) against the one-wave
variant (eight per CU) on synth12 (nx = 12, nu = 5, 10 stage inequalities, 3 terminal equalities; oracle twin "synth12").
    python tools/mid_bench.py [T] [B ...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ilqr_amd_loader import load_package
pkg = load_package()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 101
Bs = [int(a) for a in sys.argv[2:]] or [512, 2048, 4096]
mdl = pkg.models.synth12()
kw = dict(max_iterations=15, max_dual_updates=3)
for B in Bs:
    rng = np.random.default_rng(12)
    x1 = 0.5 * rng.standard_normal((B, 12)); ub = 0.1 * rng.standard_normal((B, T - 1, 5))
    res = {}
    for variant in ("latency", "mid", "auto"):
        s = pkg.Solver([mdl["dynamics"]] * (T - 1), [mdl["cost_stage"]] * (T - 1) + [mdl["cost_term"]],
                       [mdl["con_stage"]] * (T - 1) + [mdl["con_term"]], batch=B, options=pkg.Options(verbose=0, **kw), name="synth12")
        s.set_kernel_variant_(variant)
        ts = []
        for rep in range(4):
            s.reset_(); s.initialize_rollout_(x1, ub); s.timing_reset(); s.solve_(); ts.append(s.timing()[0])
        st = s.stats()
        res[variant] = (min(ts[1:]), s.get_trajectory()[0])
        print("synth12 T=%d B=%d %-8s kernel %s ms -> %.0f solves/s, %.2f M instance-iterations/s (iterations mean %.1f max %d)"
              % (T, B, variant, " ".join("%.2f" % t for t in ts[1:]), B / (1e-3 * min(ts[1:])), st["iterations"].sum() / (1e3 * min(ts[1:])),
                 st["iterations"].mean(), st["iterations"].max()))
        s.close()
    print("  one wave / four waves: %.2fx; results bitwise equal: %s" % (res["latency"][0] / res["mid"][0], np.array_equal(res["latency"][1], res["mid"][1], equal_nan=True)))
