"""Soak of the packed kernel's two-wave form: many batch sizes, horizons and models, each solve compared bitwise with the one-wave
form (a barrier mismatch between the two waves would hang the launch: run under `timeout`).   python tools/packed2_soak.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ilqr_amd_loader import load_package
pkg = load_package()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(7)
bad = 0
for r in range(rounds):
    cfg = ["car", "acrobot51", "particle", "car_goal", "pendulum"][r % 5] if r % 5 != 4 else "car"
    B = int(rng.integers(1, 300))
    model, T0, x1, ub = pkg.workloads.make_inputs(cfg, B)
    T = int(rng.integers(2, T0 + 1))                      # any horizon from the minimal one up (segments of 15, chunks of 16: all the edges)
    ub = ub[:, :T - 1]
    ho = [0, -1, 2][r % 3]
    max_it = int(rng.integers(1, 40)) if r % 4 == 0 else 100
    res = {}
    for v in ("packed1", "packed2"):
        s = pkg.Solver(model=model, horizon=T, batch=B, options=pkg.Options(verbose=0, max_iterations=max_it))
        s.set_kernel_variant_(v); s.set_handover_(ho)
        s.initialize_rollout_(x1, ub); s.solve_()
        st = s.stats()
        res[v] = (s.get_trajectory()[0], s.get_policy()[0], st["iterations"], st["rollouts"], st["max_violation"])
        s.close()
    same = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(res["packed1"], res["packed2"]))
    bad += not same
    print("round %2d %-9s B=%3d T=%2d hand-over %2d: %s (iterations max %d)" % (r, cfg, B, T, ho, "bitwise identical" if same else "DIFFERENT", res["packed1"][2].max()), flush=True)
print("soak ok" if bad == 0 else "%d round(s) differ" % bad)
