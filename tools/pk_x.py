"""Kernel experiments on the packed kernel: us per cycle in fixed-work mode (ILQR_PK_DEBUG bit 0) for each experiment library
lib_x_<NAME> (built with  make LIBDIR=../lib_x_<NAME> EXTRA_API=-DILQR_PK_DEBUG_HOOK EXTRA="-DILQR_BUILTIN_ONLY=Model_<cfg> -DPK_X_<NAME>").
    python tools/pk_x.py <cfg> <B> <NAME> [bits ...]      (one library per process: the loader binds once)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
cfg, B, name = sys.argv[1], int(sys.argv[2]), sys.argv[3]
bits = [int(b) for b in sys.argv[4:]] or [1, 9, 17, 33]
os.environ["ILQR_LIB"] = os.path.join(ROOT, "iterativelqr.jl_amd", "lib_x_" + name, "libilqr_hip.so")
from ilqr_amd_loader import load_package
pkg = load_package()
K = 100
model, T, x1, ub = pkg.workloads.make_inputs(cfg, B)
sol = pkg.Solver(model=model, horizon=T, batch=B, options=pkg.Options(verbose=0, max_iterations=K))
sol.set_kernel_variant_("packed")
out = []
for b in bits:
    os.environ["ILQR_PK_DEBUG"] = str(b)
    for _ in range(2):
        sol.reset_(); sol.initialize_rollout_(x1, ub); sol.timing_reset(); sol.solve_()
    out.append("bits %d: %.1f us" % (b, sol.timing()[0] / K * 1e3))
print("%s B=%d %-8s %s" % (cfg, B, name, "  ".join(out)))
