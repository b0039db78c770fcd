"""Kernel time of the packed kernel with / without the straggler hand-over: python tools/pk_time.py [config] [B ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ilqr_amd_loader import load_package
pkg = load_package()
cfg = sys.argv[1] if len(sys.argv) > 1 else "acrobot"
for B in [int(a) for a in sys.argv[2:]] or [4096, 8192]:
    model, T, x1, ub = pkg.workloads.make_inputs(cfg, B)
    for ho in (0, -1):
        s = pkg.Solver(model=model, horizon=T, batch=B, options=pkg.Options(verbose=0))
        s.set_kernel_variant_("packed"); s.set_handover_(ho)
        ts = []
        for rep in range(4):
            s.reset_(); s.initialize_rollout_(x1, ub); s.timing_reset(); s.solve_(); ts.append(s.timing()[0])
        print("%s B=%d handover=%s: %s ms" % (cfg, B, "off" if ho == 0 else "head count", " ".join("%.2f" % t for t in ts[1:])))
        s.close()
