"""BASELINE configs 4 and 5 as ALL EIGHT shards, one after the other on the one GPU of the box (the 8-GPU run is the
driver's): acrobot T=101 batch 65536 = 8 x 8192, synth32 batch 4096 = 8 x 512. Per shard: kernel time, iteration
statistics, the reference's end-to-end property (test/acrobot.jl:114), and whole-solve parity with the CPU oracle on a
sample of the shard's instances. The last line of each config is what an 8-GPU node would report if every rank behaved
like this GPU (slowest shard decides).   usage: python tools/all_shards.py [acrobot_sample_per_shard]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ilqr_amd_loader import load_package
from oracle import oracle
pkg = load_package()
nsample = int(sys.argv[1]) if len(sys.argv) > 1 else 512
threads = int(os.environ.get("ORACLE_THREADS", "16"))
for cfg, B, sample in (("acrobot", 8192, nsample), ("synth32", 512, 512)):
    worst_ms, tot_it = 0.0, 0.0
    print("== %s: 8 shards of %d instances (instances [r*%d, (r+1)*%d) of the global batch)" % (cfg, B, B, B))
    for r in range(8):
        model, T, x1, ub = pkg.workloads.make_inputs(cfg, B, offset=r * B)
        sol = pkg.Solver(model=model, horizon=T, batch=B, options=pkg.Options(verbose=0))
        for _ in range(2):
            sol.reset_(); sol.initialize_rollout_(x1, ub); sol.timing_reset(); sol.solve_()
        ms, _ = sol.timing()
        hq, hm = sol.handover_stats()
        x, u = sol.get_trajectory(); K, k = sol.get_policy(); st = sol.stats()
        idx = np.arange(B) if sample >= B else np.unique(np.r_[np.argsort(st["iterations"])[-8:], np.linspace(0, B - 1, sample).astype(int)])
        ref = oracle.solve_batch(model, T, x1[idx], ub[idx], nthreads=threads)
        rs = ref["stats"]
        same = (st["iterations"][idx] == rs["iterations"]) & (st["rollouts"][idx] == rs["rollouts"]) & \
               (st["outer_iterations"][idx] == rs["outer_iterations"])
        fin = np.isfinite(ref["x"]).reshape(len(idx), -1).all(1)
        s = same & fin
        ex = np.abs(x[idx] - ref["x"]).reshape(len(idx), -1).max(1); eu = np.abs(u[idx] - ref["u"]).reshape(len(idx), -1).max(1)
        eK = np.abs(K[idx] - ref["K"]).reshape(len(idx), -1).max(1) / np.maximum(np.abs(ref["K"]).reshape(len(idx), -1).max(1), 1.0)
        slow = np.isin(idx, np.argsort(st["iterations"])[-8:]) if sample < B else np.zeros(len(idx), bool)
        reg = s & ~slow
        dx, du, dK = ex[reg].max(), eu[reg].max(), eK[reg].max()
        slow_txt = ""
        if slow.any():
            # the 8 slowest instances of the shard (several hundred iterations on the iteration cap): how much of their difference is
            # conditioning? the ORACLE against itself with ū perturbed by one part in 1e15
            j = idx[slow]
            pert = oracle.solve_batch(model, T, x1[j], ub[j] * (1.0 + 1e-15), nthreads=threads)
            own = np.abs(pert["x"] - ref["x"][slow]).reshape(len(j), -1).max(1)
            okp = np.isfinite(own)
            slow_txt = "; the 8 slowest (%d..%d iterations): control flow identical %d/8, max|dx| %.1e — the oracle against itself with ū·(1+1e-15): max|dx| %.1e" % (
                st["iterations"][j].min(), st["iterations"][j].max(), int(same[slow].sum()), ex[slow & s].max() if (slow & s).any() else float("nan"),
                own[okp].max() if okp.any() else float("nan"))
        nanflow = same[~fin].all() if (~fin).any() else True
        prop = ""
        if cfg == "acrobot":
            ok = np.abs(x[:, -1, :] - [np.pi, 0, 0, 0]).max(1) < 5e-3
            prop = "; |x_T - goal| < 5e-3 on %.2f%%" % (100 * ok.mean())
        else:
            prop = "; max_violation <= 5e-3 on %.2f%%" % (100 * (st["max_violation"] <= 5e-3).mean())
        print("shard %d: kernel %7.2f ms (%s), iterations mean %.1f max %d%s; oracle sample %d: control flow identical "
              "%.2f%% (non-finite instances identical: %s), max|dx| %.1e max|du| %.1e max|dK|/max|K| %.1e on the regular sample%s"
              % (r, ms, "auto variant; %d instances marked as stragglers, %d through the workgroups' queue" % (hm, hq) if cfg == "acrobot" else "auto variant", st["iterations"].mean(), st["iterations"].max(), prop,
                 len(idx), 100 * same.mean(), nanflow, dx, du, dK, slow_txt))
        worst_ms = max(worst_ms, ms); tot_it += st["iterations"].sum()
        sol.close()
    print("-> %s: slowest shard %.2f ms => %.0f trajectories/s for the 8-shard job if each rank ran like this GPU (kernel time; %d instances)"
          % (cfg, worst_ms, 8 * B / worst_ms * 1e3, 8 * B))
