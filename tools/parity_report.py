"""Whole-solve parity statistics of the HIP path against the CPU oracle at BASELINE sizes (what the gpu tests assert
with thresholds, printed as numbers for profiles/)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ilqr_amd_loader import load_package
from oracle import oracle
pkg = load_package()
for cfg, B, variant in (("particle", 64, "auto"), ("acrobot", 1024, "auto"), ("acrobot", 1024, "packed"), ("acrobot", 4096, "auto"),
                        ("car", 4096, "auto"), ("car", 4096, "throughput"), ("car_goal", 1024, "auto"), ("car_goal", 2048, "auto"),
                        ("synth32", 512, "auto"), ("synth12", 1024, "latency"), ("synth12", 4096, "auto"), ("acrobot", 8192, "auto")):
    model, T, x1, ub = pkg.workloads.make_inputs(cfg, B)
    kw = pkg.workloads.CONFIG_OPTIONS.get(cfg, {})
    sol = pkg.Solver(model=model, horizon=T, batch=B, options=pkg.Options(verbose=0, **kw))
    sol.set_kernel_variant_(variant)
    sol.initialize_rollout_(x1, ub); sol.solve_()
    x, u = sol.get_trajectory(); K, k = sol.get_policy(); st = sol.stats()
    ref = oracle.solve_batch(model, T, x1, ub, options=oracle.default_options(**kw), nthreads=int(os.environ.get("ORACLE_THREADS", "16")))
    rs = ref["stats"]
    same = (st["iterations"] == rs["iterations"]) & (st["rollouts"] == rs["rollouts"]) & (st["outer_iterations"] == rs["outer_iterations"])
    fin = np.isfinite(ref["x"]).reshape(B, -1).all(1)
    s = same & fin
    dx = np.abs(x - ref["x"]).reshape(B, -1).max(1)[s].max(); du = np.abs(u - ref["u"]).reshape(B, -1).max(1)[s].max()
    dK = (np.abs(K - ref["K"]).reshape(B, -1).max(1) / np.maximum(np.abs(ref["K"]).reshape(B, -1).max(1), 1.0))[s].max()
    print("%-9s B=%5d T=%3d %-10s: control flow identical on %.2f%% of instances; on those max|dx| %.2e  max|du| %.2e  max|dK|/max|K| %.2e; "
          "iterations mean %.1f (oracle %.1f)" % (cfg, B, T, variant, 100 * same.mean(), dx, du, dK, st["iterations"].mean(), rs["iterations"].mean()))
    sol.close()
