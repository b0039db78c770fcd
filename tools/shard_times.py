"""Kernel time of every shard of BASELINE config 4 (acrobot T=101, 65536 = 8 x 8192) on the one GPU of the box, with the
straggler hand-over off / auto / at other outer iterations.   python tools/shard_times.py [config] [B] [handover,...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ilqr_amd_loader import load_package
pkg = load_package()
cfg = sys.argv[1] if len(sys.argv) > 1 else "acrobot"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
modes = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, -1]
res = {m: [] for m in modes}
for r in range(8):
    model, T, x1, ub = pkg.workloads.make_inputs(cfg, B, offset=r * B)
    ref = None
    for m in modes:
        sol = pkg.Solver(model=model, horizon=T, batch=B, options=pkg.Options(verbose=0, **pkg.workloads.CONFIG_OPTIONS.get(cfg, {})))
        sol.set_handover_(m)
        for _ in range(2):
            sol.reset_(); sol.initialize_rollout_(x1, ub); sol.timing_reset(); sol.solve_()
        ms, _ = sol.timing()
        st = sol.stats(); x = sol.get_trajectory()[0]
        if ref is None:
            ref = (st, x)
        same = (st["iterations"] == ref[0]["iterations"]).mean()
        dx = np.nanmax(np.abs(x - ref[1])[st["iterations"] == ref[0]["iterations"]])
        ho = pkg.Options().max_dual_updates // 2 + 1 if m < 0 else m
        moved = int((st["outer_iterations"] >= ho).sum()) if ho > 1 else 0
        res[m].append(ms)
        print("shard %d handover %2d: kernel %7.2f ms  iterations mean %.1f max %d  handed over %4d  vs first mode: same control flow %.4f max|dx| %.1e"
              % (r, m, ms, st["iterations"].mean(), st["iterations"].max(), moved, same, dx), flush=True)
        sol.close()
for m in modes:
    a = np.array(res[m])
    print("handover %2d: shards %.1f .. %.1f ms (spread %.0f %%), slowest => %.0f trajectories/s for 8 x %d" % (m, a.min(), a.max(), 100 * (a.max() - a.min()) / a.min(), 8 * B / a.max() * 1e3, B))
