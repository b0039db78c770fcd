import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from ilqr_amd_loader import load_package
pkg = load_package()
cfg, B = sys.argv[1], int(sys.argv[2])
model, T, x1, ub = pkg.workloads.make_inputs(cfg, B)
sol = pkg.Solver(model=model, horizon=T, batch=B, options=pkg.Options(verbose=0))
sol.initialize_rollout_(x1, ub); sol.solve_()
st = sol.stats()
np.save(os.path.join(ROOT, "gpurun_out", "cycles_%s_%d.npy" % (cfg, B)), np.stack([st["rollouts"], st["outer_iterations"], st["iterations"]]))
