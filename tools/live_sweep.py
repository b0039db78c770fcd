"""Hand-over threshold sweep: kernel time of an 8192-instance acrobot shard against ilqr_set_handover_live (how many instances
may still be running when the survivors leave the packed kernel for the latency kernel).  python tools/live_sweep.py [shard ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ilqr_amd_loader import load_package
pkg = load_package()
B = 8192
for r in [int(a) for a in sys.argv[1:]] or [0, 2]:
    model, T, x1, ub = pkg.workloads.make_inputs("acrobot", B, offset=r * B)
    for live in (512, 1024, 1536, 2048, 3072, 4096):
        s = pkg.Solver(model=model, horizon=T, batch=B, options=pkg.Options(verbose=0))
        s.set_kernel_variant_("packed"); s.set_handover_(-1); s.set_handover_live_(live)
        ts = []
        for rep in range(3):
            s.reset_(); s.initialize_rollout_(x1, ub); s.timing_reset(); s.solve_(); ts.append(s.timing()[0])
        print("shard %d live %4d: %s ms (iterations max %d)" % (r, live, " ".join("%.1f" % t for t in ts[1:]), s.stats()["iterations"].max()))
        s.close()
