"""Determinism of the large path: the same solve repeated must give bitwise the same trajectories."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ilqr_amd_loader import load_package
pkg = load_package()
for T in (11, 19, 101):
    B = 3
    model, _, x1, ub = pkg.workloads.make_inputs("synth32", B)
    ub = ub[:, :T - 1] + 0.3
    sol = pkg.Solver(model="synth32", horizon=T, batch=B, options=pkg.Options(verbose=0))
    outs = []
    for rep in range(4):
        sol.reset_(); sol.initialize_rollout_(x1, ub); sol.solve_()
        outs.append((sol.get_trajectory()[0].copy(), sol.stats()["iterations"].copy(), sol.get_policy()[0].copy()))
    for r in range(1, 4):
        d = np.abs(outs[r][0] - outs[0][0]).max(); dk = np.abs(outs[r][2] - outs[0][2]).max()
        print("T=%d rep %d: max|dx| %.3e max|dK| %.3e iterations %s vs %s" % (T, r, d, dk, outs[r][1], outs[0][1]))
    sol.close()
T = 11; B = 3
model, _, x1, ub = pkg.workloads.make_inputs("synth32", B)
ub = ub[:, :T - 1] + 0.3
sol = pkg.Solver(model="synth32", horizon=T, batch=B, options=pkg.Options(verbose=0))
res = []
for rep in range(2):
    sol.reset_(); sol.initialize_rollout_(x1, ub); sol.solve_()
    res.append((sol.get_trajectory()[0].copy(), sol.get_trajectory()[1].copy(), sol.buffer("states").reshape(B, T, 32).copy()))
print("per-t max|dx| nominal:", np.abs(res[1][0] - res[0][0]).max(axis=(0, 2)))
print("per-t max|du|        :", np.abs(res[1][1] - res[0][1]).reshape(B, T - 1, -1).max(axis=(0, 2)))
