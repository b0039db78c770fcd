import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ilqr_amd_loader import load_package
from oracle import oracle as O
pkg = load_package()
B = 8192
lo, hi = pkg.distributed.shard_range(3, B)
model, T, x1, ub = pkg.workloads.make_inputs("acrobot", B, offset=lo)
sol = pkg.Solver(model=model, horizon=T, batch=B, options=pkg.Options(verbose=0))
sol.initialize_rollout_(x1, ub); sol.solve_()
x, u = sol.get_trajectory(); st = sol.stats()
bad = np.nonzero(~np.isfinite(x).all(axis=(1, 2)) | (st["potrf_info"] != 0))[0]
print("bad instances:", bad)
for b in bad[:6]:
    print("GPU b=%d" % b, {k: v[b] for k, v in st.items()}, "finite x:", np.isfinite(x[b]).all())
    ref = O.solve_batch(model, T, x1[b:b+1], ub[b:b+1], nthreads=1)
    print("ORC     ", {k: v[0] for k, v in ref["stats"].items()}, "finite x:", np.isfinite(ref["x"]).all())
    pr = O.Problem(model, T); s = O.Solver(pr); s.initialize_controls(ub[b]); s.initialize_states(pr.rollout(x1[b], ub[b])); s.enable_trace(); s.solve()
    tr = s.trace()
    for r in tr[:3] + tr[-6:]:
        print("   orc trace outer %d inner %d J %.6g gn %.4g viol %.4g step %.3g status %d" % (r.outer, r.inner, r.objective, r.gradient_norm, r.max_violation, r.step_size, r.status))
