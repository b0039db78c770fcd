"""Where do the latency and the packed kernel first differ on car_obs (a model with parameters)? Trace rows, then the workspace after
a solve capped at the first differing iteration."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ilqr_amd_loader import load_package
pkg = load_package()
cfg, B = (sys.argv[1] if len(sys.argv) > 1 else "car_obs"), 21
model, T, x1, ub = pkg.workloads.make_inputs(cfg, B)
w = pkg.workloads.make_parameters(cfg, B) if cfg == "car_obs" else None
names = ("nominal_states", "nominal_actions", "K", "k", "constraint_dual", "constraint_penalty", "violations", "active_set", "jacobian_state", "jacobian_action",
         "gradient_state", "gradient_action", "hessian_state_state", "hessian_action_action", "hessian_action_state", "gradient_state_lagrangian", "gradient_action_lagrangian")
def run(v, **kw):
    s = pkg.Solver(model=model, horizon=T, batch=B, options=pkg.Options(verbose=0, **kw))
    s.set_kernel_variant_(v); s.set_handover_(0)
    if w is not None: s.set_parameters_(w)
    s.enable_trace_(1100); s.initialize_rollout_(x1, ub); s.solve_()
    out = dict(tr=s.trace(), tl=s.scalar("trace_len"), **{n: s.buffer(n) for n in names})
    s.close()
    return out
a, b = run("latency"), run("packed1")
cols = ("outer", "inner", "objective", "gradient_norm", "max_violation", "step_size", "status", "rollouts")
first = None
for inst in range(B):
    n = int(min(a["tl"][inst], b["tl"][inst]))
    d = np.nonzero((a["tr"][inst, :n] != b["tr"][inst, :n]).any(1))[0]
    if len(d):
        r = d[0]
        bad = [cols[c] for c in range(8) if a["tr"][inst, r, c] != b["tr"][inst, r, c]]
        print("instance %d: first differing trace row %d (outer %d inner %d): %s  rel %.2e" % (inst, r, a["tr"][inst, r, 0], a["tr"][inst, r, 1], bad,
              max(abs(a["tr"][inst, r, c] - b["tr"][inst, r, c]) / max(abs(a["tr"][inst, r, c]), 1e-300) for c in range(8))))
        if first is None or r < first[1]: first = (inst, r)
print("buffers after the whole solve that differ:", [n for n in names if not np.array_equal(a[n], b[n], equal_nan=True)])
if first is not None:
    inst, r = first
    it = int(a["tr"][inst, r, 1]); ou = int(a["tr"][inst, r, 0])
    print("re-running capped at outer %d, inner %d" % (ou, it))
    for cap in range(max(1, it - 1), it + 1):
        a2, b2 = run("latency", max_dual_updates=ou, max_iterations=cap if ou == 1 else 100), run("packed1", max_dual_updates=ou, max_iterations=cap if ou == 1 else 100)
        print(" cap", cap, "differ:", [(n, float(np.nanmax(np.abs(a2[n][inst] - b2[n][inst])))) for n in names if not np.array_equal(a2[n][inst], b2[n][inst], equal_nan=True)])
