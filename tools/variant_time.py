import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
from ilqr_amd_loader import load_package
pkg = load_package()
cfg = sys.argv[1]; B = int(sys.argv[2])
model, T, x1, ub = pkg.workloads.make_inputs(cfg, B)
for v in sys.argv[3:]:
    s = pkg.Solver(model=model, horizon=T, batch=B, options=pkg.Options(verbose=0, **pkg.workloads.CONFIG_OPTIONS.get(cfg, {})))
    s.set_kernel_variant_(v)
    ts = []
    for rep in range(4):
        s.reset_(); s.initialize_rollout_(x1, ub); s.timing_reset(); s.solve_(); ts.append(s.timing()[0])
    st = s.stats()
    print("%s B=%d %-10s %s ms  (iterations max %d)" % (cfg, B, v, " ".join("%.2f" % t for t in ts[1:]), st["iterations"].max()))
    s.close()
