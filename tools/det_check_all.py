"""Determinism across the kernel families: the same solve repeated on one handle (first run with a cold instruction cache) and on a
fresh handle must give bitwise the same trajectories, policies and statistics — timing-dependent races show up here
(the large path's Qxx / T race of round 3 did)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ilqr_amd_loader import load_package
pkg = load_package()
bad = 0
for cfg, B, variants in (("acrobot", 1024, ("latency", "packed", "throughput")), ("car", 2048, ("latency", "packed1", "packed2")), ("synth32_tight", 256, ("auto",)),
                         ("synth12", 2048, ("latency", "mid")), ("particle", 64, ("latency", "packed"))):
    model, T, x1, ub = pkg.workloads.make_inputs(cfg, B)
    kw = pkg.workloads.CONFIG_OPTIONS.get(cfg, {})
    for v in variants:
        outs = []
        for handle in range(2):
            sol = pkg.Solver(model=model, horizon=T, batch=B, options=pkg.Options(verbose=0, **kw))
            sol.set_kernel_variant_(v)
            for rep in range(3):
                sol.reset_(); sol.initialize_rollout_(x1, ub); sol.solve_()
                st = sol.stats()
                outs.append((sol.get_trajectory()[0].copy(), sol.get_trajectory()[1].copy(), sol.get_policy()[0].copy(), st["iterations"].copy(), st["objective"].copy()))
            sol.close()
        names = ("x", "u", "K", "iterations", "objective")
        diff = sorted({names[i] for o in outs[1:] for i, (a, b) in enumerate(zip(o, outs[0])) if not np.array_equal(a, b, equal_nan=True)})
        # no exception for the objective: every kernel family forms J in one arithmetic (ilqr_device.hpp: objective_term), so an
        # instance that the packed kernel hands over to the latency kernel — which instances do depends on timing — reports the same bits
        bad += bool(diff)
        print("%-14s B=%5d %-10s 6 solves on 2 handles: %s" % (cfg, B, v, "bitwise identical" if not diff else "DIFFERENT in " + ", ".join(diff)), flush=True)
# across the kernel families of the small models: same instances, every family — objective, counts and trajectories must agree bitwise
for cfg, B in (("acrobot", 1024), ("car", 2048)):
    model, T, x1, ub = pkg.workloads.make_inputs(cfg, B)
    res = {}
    for v in ("latency", "throughput", "packed1", "packed2"):
        sol = pkg.Solver(model=model, horizon=T, batch=B, options=pkg.Options(verbose=0))
        sol.set_kernel_variant_(v); sol.initialize_rollout_(x1, ub); sol.solve_()
        st = sol.stats()
        res[v] = (sol.get_trajectory()[0].copy(), sol.get_policy()[0].copy(), st["iterations"].copy(), st["rollouts"].copy(), st["objective"].copy())
        sol.close()
    names = ("x", "K", "iterations", "rollouts", "objective")
    for v in ("throughput", "packed1", "packed2"):
        diff = [names[i] for i in range(5) if not np.array_equal(res[v][i], res["latency"][i], equal_nan=True)]
        bad += bool(diff)
        print("%-14s B=%5d %-10s against the latency kernel: %s" % (cfg, B, v, "bitwise identical" if not diff else "DIFFERENT in " + ", ".join(diff)), flush=True)
print("all deterministic" if bad == 0 else "%d case(s) differ" % bad)
