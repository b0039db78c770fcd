"""Loading of the reference-restatement fixtures (tests/golden/ref_*.npz, made by tests/golden/make_reference_fixtures.py
from the independent numpy + scipy-LAPACK restatement) and their conversion to the flat, Julia-ordered (column-major
per timestep) buffers that both the oracle and the C-ABI use."""
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# ref_<case>_julia.npz — the same schema produced by the REAL package (tests/golden/make_julia_fixtures.jl + julia_to_npz.py) —
# takes precedence over the restatement's ref_<case>.npz wherever it exists (none can be produced in this image: no Julia)
CASES = sorted(os.path.basename(p)[4:-4] for p in glob.glob(os.path.join(GOLDEN, "ref_*.npz")) if not p.endswith("_julia.npz"))


def source_of(case):
    """"julia" if the reference itself produced this case's numbers, "restatement" otherwise."""
    return "julia" if os.path.exists(os.path.join(GOLDEN, "ref_%s_julia.npz" % case)) else "restatement"

# fixture key -> reference field name used by orc_buffer / ilqr_get_buffer
FIELD = {"fx": "jacobian_state", "fu": "jacobian_action", "gx": "gradient_state", "gu": "gradient_action",
         "gxx": "hessian_state_state", "guu": "hessian_action_action", "gux": "hessian_action_state",
         "K": "K", "k": "k", "P": "P", "p": "p", "Qx": "Qx", "Qu": "Qu", "Qxx": "Qxx", "Quu": "Quu", "Qux": "Qux"}

MODEL_OF = {"particle": "particle", "car": "car", "car_goal": "car_goal", "acrobot": "acrobot", "acrobot51": "acrobot",
            "synth32": "synth32"}


def load(case):
    path = os.path.join(GOLDEN, "ref_%s_julia.npz" % case)
    if not os.path.exists(path):
        path = os.path.join(GOLDEN, "ref_%s.npz" % case)
    d = dict(np.load(path))
    d["model"] = "car_goal" if case.startswith("car_goal") else MODEL_OF[case.split("_")[0]]
    d["T"] = int(d["horizon"][0])
    return d


def colmajor(a):
    """[t][row][col] stack (or [t][i] vectors) -> flat buffer, column-major per timestep like Julia."""
    a = np.asarray(a)
    if a.ndim == 3:
        return np.ascontiguousarray(a.transpose(0, 2, 1)).ravel()
    return np.ascontiguousarray(a).ravel()


def npoints(d):
    return int(d["points"].shape[0])


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64).ravel(), np.asarray(b, dtype=np.float64).ravel()
    if a.size == 0:
        return 0.0
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))
