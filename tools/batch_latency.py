"""Kernel time of the latency kernel against the number of resident instances (contention between the waves that share a SIMD):
    python tools/batch_latency.py [config] [B ...]   — prints ms per solve, the slowest instance's iterations and us per iteration of it"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ilqr_amd_loader import load_package
pkg = load_package()
cfg = sys.argv[1] if len(sys.argv) > 1 else "acrobot"
Bs = [int(b) for b in sys.argv[2:]] or [64, 256, 512, 768, 1024]
model, T, x1, ub = pkg.workloads.make_inputs(cfg, max(Bs))
# keep the slowest instance of the full batch in every subset (first slot), so that every launch lasts the same number of iterations
s = pkg.Solver(model=model, horizon=T, batch=max(Bs), options=pkg.Options(verbose=0, **pkg.workloads.CONFIG_OPTIONS.get(cfg, {})))
s.set_kernel_variant_("latency"); s.initialize_rollout_(x1, ub); s.solve_()
it = s.stats()["iterations"]; worst = int(np.argmax(it)); s.close()
order = np.array([worst] + [i for i in range(max(Bs)) if i != worst])
for B in Bs:
    idx = order[:B]
    s = pkg.Solver(model=model, horizon=T, batch=B, options=pkg.Options(verbose=0, **pkg.workloads.CONFIG_OPTIONS.get(cfg, {})))
    s.set_kernel_variant_("latency")
    ts = []
    for rep in range(4):
        s.reset_(); s.initialize_rollout_(x1[idx], ub[idx]); s.timing_reset(); s.solve_(); ts.append(s.timing()[0])
    st = s.stats()
    print("%s B=%d: %.2f ms, slowest instance %d iterations (%d rollouts) -> %.1f us per iteration; mean iterations %.0f"
          % (cfg, B, min(ts[1:]), st["iterations"].max(), st["rollouts"][np.argmax(st["iterations"])], 1e3 * min(ts[1:]) / st["iterations"].max(), st["iterations"].mean()))
    s.close()
