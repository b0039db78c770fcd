"""Where a straggler spends its line-search trials: per-iteration trace of ONE instance of a shard (packed kernel, hand-over off)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ilqr_amd_loader import load_package
pkg = load_package()
shard, inst = int(sys.argv[1]), int(sys.argv[2])
model, T, x1, ub = pkg.workloads.make_inputs("acrobot", 4, offset=shard * 8192 + inst)
s = pkg.Solver(model=model, horizon=T, batch=4, options=pkg.Options(verbose=0))
s.set_kernel_variant_("packed"); s.set_handover_(0); s.enable_trace_(1000)
s.initialize_rollout_(x1, ub); s.solve_()
tr = s.trace()[0]; n = int(s.scalar("trace_len")[0])
print("iterations", n, "rollouts", tr[n - 1, 7])
rej = tr[:n, 7] - np.arange(1, n + 1)
for it in range(0, n, 25):
    print("iteration %4d: outer %d inner %3d rollouts so far %5d rejected so far %4d  step %.3g" % (it + 1, tr[it, 0], tr[it, 1], tr[it, 7], rej[it], tr[it, 5]))
first = int(np.argmax(rej >= 32)) if (rej >= 32).any() else -1
print("rejects reach 32 at iteration", first + 1, "= cycle", int(tr[first, 7]) if first >= 0 else -1, "of", int(tr[n - 1, 7]))
