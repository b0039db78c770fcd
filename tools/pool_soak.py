"""Soak of the packed kernel's one-wave form as its own pool (two packs per workgroup, workgroups finishing hand-overs, stragglers
marked and leaving at once, the line search in rounds of four trials): random models, batch sizes, horizons, iteration caps, head counts and
marks; each solve compared bitwise with the latency kernel — trajectories, policies, duals, counters, objective, the cost gradients
and problem.states, the Jacobians (what the rounds of trials use as their extra buffers / must leave as the last trial evaluated) — and nothing may
be left for the launch behind. A lost wake-up would hang the launch: run under `timeout`.   python tools/pool_soak.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ilqr_amd_loader import load_package
pkg = load_package()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(11)
bad = 0
for r in range(rounds):
    cfg = ["car", "acrobot51", "particle", "car_goal", "acrobot", "car_obs"][r % 6]
    B = int(rng.integers(1, 700)) if r % 7 else int(rng.integers(2049, 2600))     # now and then more workgroups than CUs
    off = int(rng.integers(0, 60000))
    model, T0, x1, ub = pkg.workloads.make_inputs(cfg, B, offset=off)
    T = int(rng.integers(2, T0 + 1)) if r % 3 == 0 else T0
    ub = ub[:, :T - 1]
    w = pkg.workloads.make_parameters(cfg, B, offset=off)[:, :T] if cfg == "car_obs" else None
    live = int(rng.integers(1, B + 1)); mark = [1, 2, 6, 0][r % 4]
    max_it = int(rng.integers(1, 40)) if r % 5 == 0 else 100
    res = {}
    for v in ("latency", "packed1"):
        s = pkg.Solver(model=model, horizon=T, batch=B, options=pkg.Options(verbose=0, max_iterations=max_it))
        s.set_kernel_variant_(v)
        if v == "packed1":
            s.set_handover_(-1); s.set_handover_live_(live); s.set_handover_mark_(mark)
        if w is not None: s.set_parameters_(w)
        s.initialize_rollout_(x1, ub); s.solve_()
        st = s.stats()
        res[v] = (s.get_trajectory()[0], s.get_trajectory()[1], s.get_policy()[0], s.get_policy()[1], s.buffer("constraint_dual"), st["iterations"], st["rollouts"],
                  st["objective"], st["max_violation"], s.buffer("gradient_state"), s.buffer("gradient_action"), s.buffer("states"), s.buffer("actions"),
                  s.buffer("jacobian_state"), s.buffer("hessian_state_state"))
        ho = s.handover_stats() if v == "packed1" else None
        left = int((s.scalar("resume") != 0).sum())
        s.close()
    same = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(res["latency"], res["packed1"])) and left == 0
    bad += not same
    print("round %2d %-9s B=%4d T=%3d cap %3d live %4d mark %d: %s (iterations max %d; %d marked, %d through the queue)"
          % (r, cfg, B, T, max_it, live, mark, "bitwise identical" if same else "DIFFERENT", res["latency"][5].max(), ho[1], ho[0]), flush=True)
# ---- the two situations the advisor (round 5) found uncovered: a LONG horizon (the latency solver's LDS no longer fits beside two packs
# at four workgroups per CU: the launcher must drop marks and vacated CUs, nobody may wait) and TWO handles in flight on one device (a
# CU then hosts workgroups of both launches: no launch is a single round of its own)
for r, (cfg, B, T, mark) in enumerate([("acrobot", 2100, 301, 1), ("car", 2300, 401, 2)]):
    model, T0, x1, ub = pkg.workloads.make_inputs(cfg, B, offset=777)
    ub = np.concatenate([ub] * (T // T0 + 1), 1)[:, :T - 1]
    res = {}
    for v in ("packed1", "packed1_again"):
        s = pkg.Solver(model=model, horizon=T, batch=B, options=pkg.Options(verbose=0, max_iterations=12, max_dual_updates=3))
        s.set_kernel_variant_("packed1"); s.set_handover_(-1); s.set_handover_live_(B // 3); s.set_handover_mark_(mark)
        s.initialize_rollout_(x1, ub); s.solve_()
        res[v] = (s.get_trajectory()[0], s.get_policy()[0], s.stats()["iterations"], s.buffer("constraint_dual"))
        left = int((s.scalar("resume") != 0).sum())
        s.close()
    same = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(res["packed1"], res["packed1_again"])) and left == 0
    bad += not same
    print("long horizon %-8s B=%4d T=%3d mark %d: %s (packed kernel only: the LDS-resident kernels refuse this horizon)" % (cfg, B, T, mark, "bitwise repeatable, nothing left behind" if same else "DIFFERENT"), flush=True)
for r, (cfg, B) in enumerate([("acrobot51", 2200), ("car", 2500)]):
    model, T, x1, ub = pkg.workloads.make_inputs(cfg, B, offset=4242)
    ref = pkg.Solver(model=model, horizon=T, batch=B, options=pkg.Options(verbose=0)); ref.set_kernel_variant_("latency")
    ref.initialize_rollout_(x1, ub); ref.solve_()
    want = (ref.get_trajectory()[0], ref.get_policy()[0], ref.stats()["iterations"]); ref.close()
    pair = [pkg.Solver(model=model, horizon=T, batch=B, options=pkg.Options(verbose=0)) for _ in range(2)]
    for s in pair:
        s.set_kernel_variant_("packed1"); s.set_handover_(-1); s.set_handover_live_(B // 4); s.set_handover_mark_(2)
        s.initialize_rollout_(x1, ub)
    for s in pair: s.solve_(sync=False)                # both launches in flight on their own streams
    for s in pair: s.synchronize()
    same = True
    for s in pair:
        got = (s.get_trajectory()[0], s.get_policy()[0], s.stats()["iterations"])
        same = same and all(np.array_equal(a, b, equal_nan=True) for a, b in zip(want, got)) and int((s.scalar("resume") != 0).sum()) == 0
        s.close()
    bad += not same
    print("two handles in flight %-9s B=%4d: %s" % (cfg, B, "both bitwise the latency kernel's" if same else "DIFFERENT"), flush=True)
print("soak ok" if bad == 0 else "%d round(s) differ" % bad)
