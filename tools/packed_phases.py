"""Needs a library built with the hook:  make -C iterativelqr.jl_amd/csrc LIBDIR=../lib_dbg EXTRA_API=-DILQR_PK_DEBUG_HOOK  and
ILQR_LIB pointing at it (the product library ignores ILQR_PK_DEBUG).
Per-phase time of the packed kernel in fixed-work mode (ILQR_PK_DEBUG bit 0: every instance takes every phase for
max_iterations cycles, nothing is accepted): kernel ms per cycle with one phase left out at a time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ilqr_amd_loader import load_package
pkg = load_package()
cfg = sys.argv[1] if len(sys.argv) > 1 else "acrobot"
K = 100
for B in [int(b) for b in (sys.argv[2:] or ["4096", "8192"])]:
    model, T, x1, ub = pkg.workloads.make_inputs(cfg, B)
    sol = pkg.Solver(model=model, horizon=T, batch=B, options=pkg.Options(verbose=0, max_iterations=K))
    sol.set_kernel_variant_("packed")
    res = {}
    for name, bits in (("all", 1), ("-delta", 3), ("-gradients", 5), ("-riccati", 9), ("-rollout", 17), ("-cost", 33), ("none", 63)):
        os.environ["ILQR_PK_DEBUG"] = str(bits)
        for _ in range(2):
            sol.reset_(); sol.initialize_rollout_(x1, ub); sol.timing_reset(); sol.solve_()
        ms, _ = sol.timing()
        res[name] = ms / K * 1e3
    del os.environ["ILQR_PK_DEBUG"]
    print("%s B=%d: us per cycle: " % (cfg, B) + "  ".join("%s %.0f" % (k, v) for k, v in res.items()))
    print("     phase cost (all minus without): " + "  ".join("%s %.0f" % (k[1:], res["all"] - v) for k, v in res.items() if k.startswith("-")))
    sol.close()
