"""Kernel time of every shard of BASELINE config 4 (acrobot T=101, 65536 = 8 x 8192) on the one GPU of the box, with the
straggler hand-over off (0) / auto (-1: by head count, automatic threshold) / at outer iteration k (k >= 2) / by head count with
head count L<live>.   python tools/shard_times.py [config] [B] [mode,...] [shards]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ilqr_amd_loader import load_package
pkg = load_package()
cfg = sys.argv[1] if len(sys.argv) > 1 else "acrobot"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
modes = sys.argv[3].split(",") if len(sys.argv) > 3 else ["0", "-1"]
shards = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else list(range(8))
res = {m: [] for m in modes}
for r in shards:
    model, T, x1, ub = pkg.workloads.make_inputs(cfg, B, offset=r * B)
    ref = None
    for m in modes:
        sol = pkg.Solver(model=model, horizon=T, batch=B, options=pkg.Options(verbose=0, **pkg.workloads.CONFIG_OPTIONS.get(cfg, {})))
        if m.startswith("L"):            # L<live>
            sol.set_handover_(-1); sol.set_handover_live_(int(m[1:]))
        elif m.startswith("M"):          # M<mark>: by head count, stragglers marked at <mark> rejected trials above the batch's mean (0 = never)
            sol.set_handover_(-1); sol.set_handover_mark_(int(m[1:]))
        else:
            sol.set_handover_(int(m))
        for _ in range(2):
            sol.reset_(); sol.initialize_rollout_(x1, ub); sol.timing_reset(); sol.solve_()
        ms, _ = sol.timing()
        st = sol.stats(); x = sol.get_trajectory()[0]
        if ref is None:
            ref = (st, x)
        same = (st["iterations"] == ref[0]["iterations"]).mean()
        dx = np.nanmax(np.abs(x - ref[1])[st["iterations"] == ref[0]["iterations"]])
        res[m].append(ms)
        print("shard %d handover %5s: kernel %7.2f ms  iterations mean %.1f max %d  vs first mode: same control flow %.4f max|dx| %.1e; through the queue %d, marked %d"
              % ((r, m, ms, st["iterations"].mean(), st["iterations"].max(), same, dx) + sol.handover_stats()), flush=True)
        sol.close()
for m in modes:
    a = np.array(res[m])
    print("handover %5s: shards %.1f .. %.1f ms (spread %.0f %%), slowest => %.0f trajectories/s per shard-set x %d" % (m, a.min(), a.max(), 100 * (a.max() - a.min()) / a.min(), len(shards) * B / a.max() * 1e3, B))
