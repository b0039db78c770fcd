"""Time each stage kernel (gradients / backward_pass / forward_pass) and the full solve on the GPU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ilqr_amd_loader import load_package
pkg = load_package()
cfg = sys.argv[1] if len(sys.argv) > 1 else "acrobot"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
model, T, x1, ub = pkg.workloads.make_inputs(cfg, B)
sol = pkg.Solver(model=model, horizon=T, batch=B, options=pkg.Options(verbose=0))
sol.initialize_rollout_(x1, ub)
sol.run_stage_("reset_model_objective"); sol.run_stage_("cost_nominal")
def t(stage, reps=20):
    sol.run_stage_(stage)
    t0 = time.perf_counter()
    for _ in range(reps): sol.run_stage_(stage)
    return (time.perf_counter() - t0) / reps * 1e6
print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "n/a")
for st in ("cost_nominal", "gradients", "backward_pass", "forward_pass", "reset_model_objective", "al_update"):
    print("%-24s %9.1f us per launch (B=%d)" % (st, t(st), B))
sol.reset_(); sol.initialize_rollout_(x1, ub)
t0 = time.perf_counter(); sol.solve_(); dt = time.perf_counter() - t0
st = sol.stats()
print("solve: %.2f ms; iterations mean %.1f rollouts mean %.1f -> %.1f us/iteration" % (dt * 1e3, st["iterations"].mean(), st["rollouts"].mean(), dt * 1e6 / st["iterations"].mean()))
