"""Two waves per pack (a linearisation server beside the solver wave) against one: bitwise results, then kernel times.
    python tools/packed2_check.py [quick]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ilqr_amd_loader import load_package
pkg = load_package()
quick = len(sys.argv) > 1
cases = [("particle", 9), ("car", 37), ("acrobot51", 45)] + ([] if quick else [("car", 4096), ("acrobot", 4096), ("car_obs", 2048)])
for cfg, B in cases:
    model, T, x1, ub = pkg.workloads.make_inputs(cfg, B)
    w = pkg.workloads.make_parameters(cfg, B) if cfg == "car_obs" else None
    out = {}
    for v in ("packed1", "packed2"):
        s = pkg.Solver(model=model, horizon=T, batch=B, options=pkg.Options(verbose=0))
        s.set_kernel_variant_(v); s.set_handover_(0)
        if w is not None: s.set_parameters_(w)
        ts = []
        for rep in range(3):
            s.reset_(); s.initialize_rollout_(x1, ub); s.timing_reset(); s.solve_(); ts.append(s.timing()[0])
        st = s.stats()
        out[v] = (s.get_trajectory()[0], s.get_trajectory()[1], s.get_policy()[0], st["iterations"], st["objective"], s.buffer("jacobian_state"), s.buffer("hessian_state_state"), min(ts[1:]))
        s.close()
    a, b = out["packed1"], out["packed2"]
    same = all(np.array_equal(a[i], b[i], equal_nan=True) for i in range(7))
    print("%-10s B=%5d: %s; one wave %.3f ms, two waves %.3f ms (iterations max %d)" % (cfg, B, "bitwise identical" if same else "DIFFERENT", a[7], b[7], a[3].max()), flush=True)
