"""Inputs of the fixture cases for tests/golden/make_julia_fixtures.jl (the real IterativeLQR.jl package): per case
    julia_in/<case>.txt      model name, horizon, number of snapshot points, then one "outer inner" line per point
    julia_in/<case>.x1.f64   x1[nx]            julia_in/<case>.u.f64   ubar[T-1][nu]   (little-endian float64)
taken from the committed fixtures, i.e. exactly the instances the oracle and the HIP path are tested on."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_reference_fixtures import CASES  # noqa: E402


# SURVEY §8 f3 (round 6): whole solves of the two problems with per-step objects / dimensions, no stage snapshots. Inputs are
# generated here (no restatement fixture exists for them: they are checked live), padded to the largest dimensions.
F3_CASES = {"ragged_T9": ("ragged", 9), "ragged_T41": ("ragged", 41), "car_tv_T21": ("car_tv", 21)}
RAGGED_N, RAGGED_M = [3, 3, 4, 4, 2, 2, 3, 3], [2, 1, 2, 1, 1, 2, 2, 1]


def f3_inputs(case):
    """(model, T, x1[nx_max], ubar[T-1][nu_max], state_dims, action_dims) of an f3 case: ragged from the distribution tests/test_oracle_ragged.py draws
    from (0.5 N(0,1) on the first state, 0.2 N(0,1) on every action, seed 5), car_tv from the reference's own initialisation (test/car.jl:24-29)."""
    model, T = F3_CASES[case]
    if model == "ragged":
        n_t = [RAGGED_N[t % 8] for t in range(T)]
        m_t = [RAGGED_M[t % 8] for t in range(T - 1)]
        rng = np.random.default_rng(5)
        x1 = np.zeros(max(n_t)); x1[:n_t[0]] = 0.5 * rng.standard_normal((1, n_t[0]))[0]
        ub = np.zeros((T - 1, max(m_t)))
        for t in range(T - 1):
            ub[t, :m_t[t]] = 0.2 * rng.standard_normal((1, m_t[t]))[0]
        return model, T, x1, ub, n_t, m_t
    x1 = np.zeros(3)
    ub = np.tile(1.0e-2 * np.array([1.0, 0.1]), (T - 1, 1))
    return model, T, x1, ub, [3] * T, [2] * (T - 1)


def main(out_dir=None):
    out_dir = out_dir or os.path.join(HERE, "julia_in")
    os.makedirs(out_dir, exist_ok=True)
    for case in F3_CASES:
        model, T, x1, ub, _, _ = f3_inputs(case)
        with open(os.path.join(out_dir, case + ".txt"), "w") as f:
            f.write("%s\n%d\n0\n" % (model, T))
        x1.astype("<f8").tofile(os.path.join(out_dir, case + ".x1.f64"))
        ub.astype("<f8").tofile(os.path.join(out_dir, case + ".u.f64"))
    for case, (model, T, _, _, points) in CASES.items():
        d = np.load(os.path.join(HERE, "ref_%s.npz" % case))
        with open(os.path.join(out_dir, case + ".txt"), "w") as f:
            f.write("%s\n%d\n%d\n" % (model, T, len(points)))
            for o, i in points:
                f.write("%d %d\n" % (o, i))
        d["x1"].astype("<f8").tofile(os.path.join(out_dir, case + ".x1.f64"))
        d["ubar"].astype("<f8").tofile(os.path.join(out_dir, case + ".u.f64"))
    return out_dir


if __name__ == "__main__":
    print("wrote", main(sys.argv[1] if len(sys.argv) > 1 else None))
