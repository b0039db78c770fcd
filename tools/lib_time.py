"""Kernel time of one workload with an experiment library:  python tools/lib_time.py <libdir name under iterativelqr.jl_amd> <config> <B> [variant]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ILQR_LIB"] = os.path.join(ROOT, "iterativelqr.jl_amd", sys.argv[1], "libilqr_hip.so")
from ilqr_amd_loader import load_package
pkg = load_package()
cfg, B = sys.argv[2], int(sys.argv[3]); v = sys.argv[4] if len(sys.argv) > 4 else "auto"
model, T, x1, ub = pkg.workloads.make_inputs(cfg, B)
s = pkg.Solver(model=model, horizon=T, batch=B, options=pkg.Options(verbose=0, **pkg.workloads.CONFIG_OPTIONS.get(cfg, {})))
s.set_kernel_variant_(v)
ts = []
for rep in range(5):
    s.reset_(); s.initialize_rollout_(x1, ub); s.timing_reset(); s.solve_(); ts.append(s.timing()[0])
print("%s %s B=%d %s: %s ms" % (sys.argv[1], cfg, B, v, " ".join("%.3f" % t for t in ts[1:])))
