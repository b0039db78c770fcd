"""Sub-phase cycle split of ONE phase (the ILQR_SUB_MARK sites, currently the large-model Riccati step).
Needs:  make -C iterativelqr.jl_amd/csrc clean all EXTRA="-DILQR_PROFILE -DILQR_PROFILE_SUB" """
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ilqr_amd_loader import load_package
pkg = load_package()
cfg = sys.argv[1] if len(sys.argv) > 1 else "synth32"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
names = sys.argv[3].split(",") if len(sys.argv) > 3 else ["stage+matvec", "gemm [T;Uh]", "gemm Qxx,Qux,Quu", "potrf", "potrs", "Uxt,P,p"]
model, T, x1, ub = pkg.workloads.make_inputs(cfg, B)
sol = pkg.Solver(model=model, horizon=T, batch=B, options=pkg.Options(verbose=0, **pkg.workloads.CONFIG_OPTIONS.get(cfg, {})))
for _ in range(2):
    sol.reset_(); sol.initialize_rollout_(x1, ub); sol.solve_()
sc = sol.buffer("_scalars"); st = sol.stats()
steps = (st["iterations"].astype(float) + st["outer_iterations"]) * (T - 1)   # one extra backward pass per ilqr_solve!
for i, nm in enumerate(names):
    print("  %-20s %9.0f ticks per timestep" % (nm, sc[:, 10 + i].sum() / steps.sum()))
