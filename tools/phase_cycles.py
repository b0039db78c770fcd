"""Per-phase shader-clock accounting inside the fused solve kernel.
    python tools/phase_cycles.py [config] [B] [generator] [alone]
alone = "slowest" or an instance index: after the batch, that ONE instance of it is solved alone on the chip and its phases are
printed beside its own phases inside the batch — what sharing the SIMDs with the rest of the batch costs it, phase by phase.
Needs a library built with -DILQR_PROFILE:  make -C iterativelqr.jl_amd/csrc LIBDIR=../lib_prof1 EXTRA=-DILQR_PROFILE EXTRA_API=-DILQR_PROFILE"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ilqr_amd_loader import load_package
pkg = load_package()
cfg = sys.argv[1] if len(sys.argv) > 1 else "acrobot"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
gen = sys.argv[3] if len(sys.argv) > 3 else "splitmix64"
alone = sys.argv[4] if len(sys.argv) > 4 else None
model, T, x1, ub = pkg.workloads.make_inputs(cfg, B, generator=gen)
sol = pkg.Solver(model=model, horizon=T, batch=B, options=pkg.Options(verbose=0, **pkg.workloads.CONFIG_OPTIONS.get(cfg, {})))
for _ in range(2):
    sol.reset_(); sol.initialize_rollout_(x1, ub); sol.solve_()
sc = sol.buffer("_scalars"); st = sol.stats()
names = ["cost", "gradients", "backward", "delta", "rollout", "total"]
prof = sc[:, 10:16]
it = st["iterations"].astype(float); ro = st["rollouts"].astype(float)
tot = prof[:, 5].mean()
print("config %s B=%d (%s): iterations %.1f rollouts %.1f; total %.3e ticks/instance" % (cfg, B, gen, it.mean(), ro.mean(), tot))
for i, nm in enumerate(names[:5]):
    per = prof[:, i].sum() / (ro.sum() if nm in ("rollout",) else it.sum())
    print("  %-10s %6.1f%%  %9.0f ticks per %s" % (nm, 100 * prof[:, i].mean() / tot, per, "rollout" if nm == "rollout" else "iteration"))
print("  %-10s %6.1f%%" % ("other", 100 * (1 - prof[:, :5].sum(1).mean() / tot)))
if alone is not None:
    b = int(np.argmax(prof[:, 5])) if alone == "slowest" else int(alone)
    s1 = pkg.Solver(model=model, horizon=T, batch=1, options=pkg.Options(verbose=0, **pkg.workloads.CONFIG_OPTIONS.get(cfg, {})))
    s1.set_kernel_variant_("latency" if model in ("acrobot", "car", "particle", "car_goal", "car_obs") else "auto")
    for _ in range(2):
        s1.reset_(); s1.initialize_rollout_(x1[b:b + 1], ub[b:b + 1]); s1.solve_()
    p1 = s1.buffer("_scalars")[0, 10:16]; st1 = s1.stats()
    assert st1["iterations"][0] == st["iterations"][b] and st1["rollouts"][0] == st["rollouts"][b]
    print("instance %d (the one with the longest lifetime in the batch): %d iterations, %d rollouts; ticks per iteration (rollout: per rollout)"
          % (b, it[b], ro[b]))
    print("  %-10s %12s %12s %8s" % ("phase", "in the batch", "alone", "ratio"))
    for i, nm in enumerate(names):
        d = ro[b] if nm == "rollout" else it[b]
        if nm == "delta": continue
        print("  %-10s %12.0f %12.0f %8.3f" % (nm, prof[b, i] / d, p1[i] / d, prof[b, i] / max(p1[i], 1.0)))
    s1.close()
