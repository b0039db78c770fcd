import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ilqr_amd_loader import load_package
from oracle import oracle as O
pkg = load_package()
cfg = sys.argv[1] if len(sys.argv) > 1 else "particle"
B = 1
model, T, x1, ub = pkg.workloads.make_inputs(cfg, B)
sol = pkg.Solver(model=model, horizon=T, batch=B, options=pkg.Options(verbose=0))
sol.initialize_rollout_(x1, ub)
pr = O.Problem(model, T); xb = pr.rollout(x1[0], ub[0]); s = O.Solver(pr); s.initialize_controls(ub[0]); s.initialize_states(xb)
for st in ("reset_model_objective", "cost_nominal", "gradients", "backward_pass"): sol.run_stage_(st)
s.call("reset_model_objective"); s.call("cost_bang", 0); s.call("gradients"); s.call("backward_pass"); s.call("lagrangian_gradient")
np.set_printoptions(precision=6, linewidth=200, suppress=True)
n, m = sol.nx, sol.nu
for name, shape in (("K", (T - 1, n, m)), ("k", (T - 1, m)), ("P", (T, n, n)), ("p", (T, n))):
    g = sol.buffer(name)[0].reshape(shape); o = s.buffer(name).reshape(shape)
    err = np.abs(g - o).reshape(shape[0], -1).max(1)
    print(name, "max err", err.max(), "first bad t", np.argmax(err > 1e-9 * max(1, np.abs(o).max())), "of", shape[0])
    tt = shape[0] - 1 if name in ("P", "p") else shape[0] - 1
    for t in (tt, tt - 1):
        print("  t=%d gpu" % t, g[t].ravel(), "\n       orc", o[t].ravel())
