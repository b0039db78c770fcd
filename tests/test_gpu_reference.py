"""GPU parity against the fixtures of the independent restatement (tests/golden/ref_*.npz: numpy + sympy + scipy's
real LAPACK dpotrf/dpotrs — see tests/golden/reference_restatement.py), through the C-ABI.

Bars (fp64; observed values in profiles/r02_parity.txt):
    stage level, identical inputs   linearisation 2e-11 relative; Riccati outputs (Qx..Qux, K, k, P, p, ∇L) 1e-8
                                    (the recursion amplifies rounding over 50-100 steps; observed ≤ 1e-10);
                                    Δ = ∇Lᵀ·Δz 1e-10 relative for EVERY device implementation (latency kernel: wave 1's
                                    delta_small; throughput kernel: MFMA ride-along in rollout_small; large path:
                                    delta_large_body); line-search decisions (α, status, number of trials) exact
    whole solve                     per-iteration trace: outer / inner / step size / status exact, objective 1e-8 relative;
                                    |Δx|, |Δu| ≤ 2e-8, |ΔK| ≤ 1e-7·max|K| (observed against the fixtures: 6.4e-10 / 1.1e-9 / 1.1e-9·max|K|)
"""
import numpy as np
import pytest

from ilqr_amd_loader import load_package
from refdata import CASES, colmajor, load, npoints, rel

pytestmark = pytest.mark.gpu

GRAD_TOL = 2e-11
RICCATI_TOL = 1e-8
SNAP_CASES = [c for c in CASES if npoints(load(c)) > 0]


@pytest.fixture(scope="module")
def pkg():
    p = load_package()
    if p._ffi.lib().ilqr_device_count() < 1:
        pytest.fail("no HIP device: the gpu tests must run on a GPU box")
    return p


def _variants(model, whole_solve=False):
    if model == "synth32":
        return ["latency"]
    return ["latency", "throughput", "packed"] if whole_solve else ["latency", "throughput"]


def _set(sol, name, v):
    sol.set_buffer(name, np.asarray(v, dtype=np.float64).reshape(1, -1))


def _load_pre_state(sol, d, j):
    p = "s%d_" % j
    _set(sol, "nominal_states", colmajor(d[p + "pre_nominal_states"]))
    _set(sol, "nominal_actions", colmajor(d[p + "pre_nominal_actions"]))
    _set(sol, "states", colmajor(d[p + "pre_states"]))
    _set(sol, "actions", colmajor(d[p + "pre_actions"]))
    _set(sol, "hessian_state_state", colmajor(d[p + "pre_gxx"]))
    _set(sol, "hessian_action_action", colmajor(d[p + "pre_guu"]))
    _set(sol, "hessian_action_state", colmajor(d[p + "pre_gux"]))
    if d[p + "pre_violations"].size:
        _set(sol, "violations", d[p + "pre_violations"])
        _set(sol, "constraint_dual", d[p + "pre_dual"])
        _set(sol, "constraint_penalty", d[p + "pre_penalty"])
        _set(sol, "active_set", d[p + "pre_active_set"])
    _set_scalars(sol, d[p + "pre_scalars"])


def _set_scalars(sol, sc4):
    L = sol._ffi_lib
    sc = sol.buffer("_scalars")
    sc[0, L.ilqr_scalar_slot(b"objective")] = sc4[0]
    sc[0, L.ilqr_scalar_slot(b"max_violation")] = sc4[1]
    sc[0, L.ilqr_scalar_slot(b"step_size")] = sc4[2]
    sc[0, L.ilqr_scalar_slot(b"status")] = sc4[3]
    sc[0, L.ilqr_scalar_slot(b"states_eq_nominal")] = 0.0      # always evaluate both trajectories (Q2)
    sol.set_buffer("_scalars", sc)


GRADS = (("fx", "jacobian_state"), ("fu", "jacobian_action"), ("gx", "gradient_state"), ("gu", "gradient_action"),
         ("gxx", "hessian_state_state"), ("guu", "hessian_action_action"), ("gux", "hessian_action_state"))


@pytest.mark.parametrize("case", SNAP_CASES)
def test_hip_stages_match_reference_fixture(pkg, case):
    d = load(case)
    T, model = d["T"], d["model"]
    worst = {}
    for variant in _variants(model):
        sol = pkg.Solver(model=model, horizon=T, batch=1, options=pkg.Options(verbose=0))
        sol._ffi_lib = pkg._ffi.lib()
        sol.set_kernel_variant_(variant)
        sol.enable_action_value_buffers_()
        n, m = sol.nx, sol.nu
        for j in range(npoints(d)):
            p = "s%d_" % j
            where = (case, variant, j)
            _load_pre_state(sol, d, j)
            sol.run_stage_("gradients")
            for key, name in GRADS:
                e = rel(sol.buffer(name)[0], colmajor(d[p + key]))
                worst["grad"] = max(worst.get("grad", 0.0), e)
                assert e < GRAD_TOL, (where, key, e)
            for key, name in GRADS:                       # backward pass from the fixture's exact linearisation
                _set(sol, name, colmajor(d[p + key]))
            sol.run_stage_("backward_pass")
            for key in ("Qx", "Qu", "Qxx", "Quu", "Qux", "K", "k", "P", "p"):
                e = rel(sol.buffer(key)[0], colmajor(d[p + key]))
                worst["riccati"] = max(worst.get("riccati", 0.0), e)
                assert e < RICCATI_TOL, (where, key, e)
            assert rel(sol.buffer("gradient_state_lagrangian")[0], d[p + "Lx"]) < RICCATI_TOL, where
            assert rel(sol.buffer("gradient_action_lagrangian")[0], d[p + "Lu"]) < RICCATI_TOL, where
            for key in ("K", "k"):                        # forward pass from the fixture's exact policy
                _set(sol, key, colmajor(d[p + key]))
            _set(sol, "gradient_state_lagrangian", colmajor(d[p + "Lx"]))
            _set(sol, "gradient_action_lagrangian", colmajor(d[p + "Lu"]))
            r0 = int(sol.stats()["rollouts"][0])
            sol.run_stage_("forward_pass")
            f = p + "fwd_"
            delta = float(sol.scalar("delta_grad_product")[0])
            ref_delta = float(d[f + "delta"][0])
            e = abs(delta - ref_delta) / max(1e-300, abs(ref_delta))
            worst["delta"] = max(worst.get("delta", 0.0), e)
            assert e < 1e-10, (where, delta, ref_delta)                                   # a8: ∇Lᵀ·Δz
            st = sol.stats()
            J, viol, alpha, status = d[f + "scalars"]
            assert st["step_size"][0] == alpha and st["status"][0] == int(status), where
            assert int(st["rollouts"][0]) - r0 == d[f + "trial_objectives"].size, where      # line-search trials
            assert st["objective"][0] == pytest.approx(J, rel=1e-11), where
            assert st["max_violation"][0] == pytest.approx(viol, rel=1e-9, abs=1e-13), where
            for key in ("nominal_states", "nominal_actions", "states", "actions"):
                e = rel(sol.buffer(key)[0], colmajor(d[f + key]))
                worst["forward"] = max(worst.get("forward", 0.0), e)
                assert e < 1e-10, (where, key, e)
            if d[f + "violations"].size:
                assert rel(sol.buffer("violations")[0], d[f + "violations"]) < 1e-10, where
                assert np.array_equal(sol.buffer("active_set")[0], d[f + "active_set"]), where
        sol.close()
    print("observed", case, {k: "%.1e" % v for k, v in worst.items()})


@pytest.mark.parametrize("case", CASES)
def test_hip_whole_solve_matches_reference_fixture(pkg, case):
    d = load(case)
    T, model = d["T"], d["model"]
    ref = d["trace"]
    for variant in _variants(model, whole_solve=True):
        sol = pkg.Solver(model=model, horizon=T, batch=1, options=pkg.Options(verbose=0))
        sol.set_kernel_variant_(variant)
        sol.enable_trace_(ref.shape[0] + 8)
        sol.initialize_rollout_(d["x1"][None], d["ubar"][None])
        assert rel(sol.buffer("nominal_states")[0], d["xbar"]) < 1e-12                    # rollout(), src/rollout.jl:33-42
        sol.solve_()
        nrow = int(sol.scalar("trace_len")[0])
        assert nrow == ref.shape[0], (case, variant, nrow, ref.shape[0])
        tr = sol.trace()[0][:nrow]
        assert np.array_equal(tr[:, [0, 1, 5, 6, 7]], ref[:, [0, 1, 5, 6, 7]]), (case, variant)   # outer, inner, α, status, rollouts
        assert np.allclose(tr[:, 2], ref[:, 2], rtol=1e-8, atol=1e-12)
        assert np.allclose(tr[:, 4], ref[:, 4], rtol=1e-5, atol=1e-11)
        assert np.abs(tr[:, 3] - ref[:, 3]).max() <= 1e-6 * max(1.0, np.abs(ref[:, 3]).max())
        st, rs = sol.stats(), d["stats"]
        got = (st["iterations"][0], st["outer_iterations"][0], st["status"][0], st["rollouts"][0], st["potrf_info"][0])
        assert tuple(int(v) for v in got) == tuple(int(v) for v in rs[4:9]), (case, variant)
        x, u = sol.get_trajectory()
        K, k = sol.get_policy()
        n, m = sol.nx, sol.nu
        Kd = K[0].transpose(0, 2, 1)                      # [t][nx][nu] column-major -> [t][nu][nx]
        if "K_steps" in d:
            Kd = Kd[d["K_steps"]]
        dx, du = np.abs(x[0] - d["x"]).max(), np.abs(u[0] - d["u"]).max()
        dK = np.abs(Kd - d["K"]).max() / max(1.0, np.abs(d["K"]).max())
        print("observed", case, variant, "dx %.1e du %.1e dK %.1e" % (dx, du, dK))
        assert dx <= 2e-8 and du <= 2e-8, (case, variant, dx, du)
        assert dK <= 1e-7, (case, variant, dK)
        assert np.abs(k[0] - d["k"]).max() <= 1e-6 * max(1.0, np.abs(d["k"]).max())
        sol.close()


@pytest.mark.parametrize("family,B", [("acrobot", 7), ("car", 5), ("particle", 6), ("car_goal", 5)])
def test_packed_kernel_with_several_fixture_instances_in_one_handle(pkg, family, B):
    """The fixtures' instances side by side in ONE handle on the packed kernel (four instances per wave: a full wave plus a
    ragged one, neighbours in different phases of their state machines): every instance must reproduce its own fixture —
    trace exact, trajectories and gains within the whole-solve tolerance — exactly as it does alone."""
    cases = [c for c in CASES if c.startswith(family + "_i") or (family == "particle" and c == "particle_sin")]
    assert len(cases) >= 2
    order = [cases[i % len(cases)] for i in [0, 1, 2, 1, 0, 2, 1][:B]]
    ds = [load(c) for c in order]
    T, model = ds[0]["T"], ds[0]["model"]
    assert all(d["T"] == T and d["model"] == model for d in ds)
    cap = max(d["trace"].shape[0] for d in ds) + 8
    sol = pkg.Solver(model=model, horizon=T, batch=B, options=pkg.Options(verbose=0))
    sol.set_kernel_variant_("packed")
    sol.enable_trace_(cap)
    sol.initialize_rollout_(np.stack([d["x1"] for d in ds]), np.stack([d["ubar"] for d in ds]))
    sol.solve_()
    x, u = sol.get_trajectory(); K, k = sol.get_policy(); st = sol.stats()
    tl = sol.scalar("trace_len").astype(int); tr = sol.trace()
    for b, d in enumerate(ds):
        ref = d["trace"]
        assert tl[b] == ref.shape[0], (order[b], b, tl[b], ref.shape[0])
        assert np.array_equal(tr[b, :tl[b]][:, [0, 1, 5, 6, 7]], ref[:, [0, 1, 5, 6, 7]]), (order[b], b)
        assert np.allclose(tr[b, :tl[b], 2], ref[:, 2], rtol=1e-8, atol=1e-12)
        got = (st["iterations"][b], st["outer_iterations"][b], st["status"][b], st["rollouts"][b], st["potrf_info"][b])
        assert tuple(int(v) for v in got) == tuple(int(v) for v in d["stats"][4:9]), (order[b], b)
        assert np.abs(x[b] - d["x"]).max() <= 2e-8 and np.abs(u[b] - d["u"]).max() <= 2e-8, (order[b], b)
        Kd = K[b].transpose(0, 2, 1)
        assert np.abs(Kd - d["K"]).max() <= 1e-7 * max(1.0, np.abs(d["K"]).max()), (order[b], b)
    # equal fixtures in different lanes of different waves give bitwise equal results
    for b in range(B):
        for c in range(b + 1, B):
            if order[b] == order[c]:
                assert np.array_equal(x[b], x[c]) and np.array_equal(K[b], K[c]), (b, c)
    sol.close()
