import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ilqr_amd_loader import load_package
from oracle import oracle as O
pkg = load_package()
cfg = sys.argv[1] if len(sys.argv) > 1 else "acrobot"
B = 2
model, T, x1, ub = pkg.workloads.make_inputs(cfg, B)
sol = pkg.Solver(model=model, horizon=T, batch=B, options=pkg.Options(verbose=0))
sol.initialize_rollout_(x1, ub)
pr = O.Problem(model, T); s = O.Solver(pr); s.initialize_controls(ub[0]); s.initialize_states(pr.rollout(x1[0], ub[0]))
for st in ("reset_model_objective", "cost_nominal", "gradients"): sol.run_stage_(st)
s.call("reset_model_objective"); s.call("cost_bang", 0); s.call("gradients")
np.set_printoptions(precision=5, linewidth=220, suppress=True)
for name in ("nominal_states", "jacobian_state", "jacobian_action", "gradient_state", "hessian_state_state"):
    g = sol.buffer(name)[0]; o = s.buffer(name)
    bad = np.nonzero(np.abs(g - o) > 1e-9 * max(1, np.abs(o).max()))[0]
    print(name, "len", o.size, "max err", np.abs(g - o).max(), "n bad", bad.size, "first bad idx", bad[:8])
    if bad.size: print("   gpu", g[bad[:6]], "\n   orc", o[bad[:6]])
