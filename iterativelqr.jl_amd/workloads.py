"""Deterministic synthetic inputs for the BASELINE.json configurations
(SURVEY.md §8(d)); one generator feeds the GPU path, the oracle and the bench."""
import numpy as np

SEED = 20240607

# name -> (model, T, constrained)
CONFIGS = {
    "particle": ("particle", 11, True),
    "acrobot": ("acrobot", 101, True),
    "acrobot51": ("acrobot", 51, True),
    "car": ("car", 51, True),
    "car_goal": ("car_goal", 51, True),
    "car_obs": ("car_obs", 51, True),
    "synth32": ("synth32", 101, True),
    # same model and inputs, convergence tolerances tightened to 1e-9 (CONFIG_OPTIONS): BASELINE's setting converges in 1.6
    # inner iterations on average, this one takes ~20 (4..50) — the workload on which the large-model kernels are profiled
    "synth32_tight": ("synth32", 101, True),
    "synth32_tight11": ("synth32", 101, True),     # tolerances 1e-11: ~76 iterations (17..104), few idle SIMDs at the end
    # a mid-size model (nx = 12, nu = 5, 10 stage inequalities, 3 terminal equalities; oracle twin "synth12"): the sizes for which
    # the large path has its one-wave-per-instance variant. Iteration caps as in the parity tests (the problem is hard for the
    # reference's AL loop: parity and throughput, not convergence)
    "synth12": ("synth12", 101, True),
}
# per-config solver options (src/options.jl:1-15 fields) that differ from the defaults
CONFIG_OPTIONS = {"synth32_tight": dict(objective_tolerance=1.0e-9, lagrangian_gradient_tolerance=1.0e-9),
                  "synth32_tight11": dict(objective_tolerance=1.0e-11, lagrangian_gradient_tolerance=1.0e-11),
                  "synth12": dict(max_iterations=15, max_dual_updates=3)}
DIMS = {"particle": (2, 1), "acrobot": (4, 1), "car": (3, 2), "car_goal": (3, 2), "car_obs": (3, 2),
        "pendulum_euler": (2, 1), "synth32": (32, 8), "synth12": (12, 5)}


def make_inputs(config, batch, seed=SEED, offset=0, generator="pcg64"):
    """Returns (model, T, x1[B,n], ubar[B,T-1,m]). `offset` shifts the instance
    index so that shards of a larger batch draw disjoint, reproducible instances.
    generator: "pcg64" — one numpy PCG64 stream per instance (the inputs of every profile of rounds 1-5) — or "splitmix64" — the
    library's own generator (ilqr_synthetic_inputs: SURVEY §8(d)'s splitmix64 / Box-Muller as a pure function of (seed, b, t, j)),
    which a Julia or C host calls for the same instances."""
    model, T, _ = CONFIGS[config]
    n, m = DIMS[model]
    x1 = np.zeros((batch, n))
    ub = np.zeros((batch, T - 1, m))
    if generator == "splitmix64":
        import ctypes as C
        from . import _ffi
        _ffi.check(_ffi.lib().ilqr_synthetic_inputs(model.encode(), T, seed, offset, batch, x1.ctypes.data_as(_ffi.c_double_p),
                                                    ub.ctypes.data_as(_ffi.c_double_p)))
        return model, T, x1, ub
    assert generator == "pcg64", generator
    for b in range(batch):
        rng = np.random.default_rng([seed, offset + b])
        if model == "particle":
            ub[b] = 0.1 * rng.standard_normal((T - 1, m))          # examples/particle.jl:30
        elif model == "acrobot":
            ub[b] = 1.0 * rng.standard_normal((T - 1, m))          # test/acrobot.jl:88
        elif model == "synth32":
            x1[b] = 0.5 * rng.standard_normal(n)                   # SURVEY.md §8(d) C5; ū = 0
        elif model == "synth12":
            x1[b] = 0.5 * rng.standard_normal(n)
            ub[b] = 0.1 * rng.standard_normal((T - 1, m))
        elif model in ("car", "car_goal", "car_obs"):
            ub[b] = 1.0e-2 * np.array([1.0, 0.1])                  # test/car.jl:28
            if offset + b > 0:
                ub[b] *= 1.0 + 0.5 * rng.uniform(-1.0, 1.0)
                x1[b, :2] = 0.05 * rng.standard_normal(2)
    return model, T, x1, ub


def make_parameters(config, batch, seed=SEED, offset=0):
    """Per-instance parameter trajectories θ[b, t, :] for parametrised models (car_obs: obstacle centre,
    drifting slowly along the horizon)."""
    model, T, _ = CONFIGS[config]
    assert model == "car_obs"
    w = np.zeros((batch, T, 2))
    for b in range(batch):
        rng = np.random.default_rng([seed, 7, offset + b])
        c0 = np.array([0.8, 0.5]) + 0.08 * rng.uniform(-1, 1, 2)       # on the car's way: the constraint binds
        drift = 0.05 * rng.uniform(-1, 1, 2)
        w[b] = c0 + np.linspace(0.0, 1.0, T)[:, None] * drift
    return w
