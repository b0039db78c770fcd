import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from ilqr_amd_loader import load_package
pkg = load_package()
B = int(sys.argv[1])
model, T, x1, ub = pkg.workloads.make_inputs("synth12", B)
kw = pkg.workloads.CONFIG_OPTIONS["synth12"]
s = pkg.Solver(model=model, horizon=T, batch=B, options=pkg.Options(verbose=0, **kw))
s.set_kernel_variant_("mid")
ts = []
for rep in range(4):
    s.reset_(); s.initialize_rollout_(x1, ub); s.timing_reset(); s.solve_(); ts.append(s.timing()[0])
print("B=%d pad=%s: %s" % (B, os.environ.get("ILQR_DBG_LDS_PAD", "0"), " ".join("%.2f" % t for t in ts[1:])))
