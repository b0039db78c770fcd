"""Generates tests/golden/oracle_regression.json from THIS repository's CPU oracle (the reference is pure
Julia and ships no golden vectors; these pin the oracle — and through it the HIP path — against
regressions). Inputs come from iterativelqr.jl_amd/workloads.py (seeded).   python tests/golden/make_golden.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ilqr_amd_loader import load_package  # noqa: E402
from oracle import oracle as O  # noqa: E402

pkg = load_package()
out = {}
for config, B in (("particle", 3), ("acrobot51", 3), ("acrobot", 2), ("car", 3), ("car_goal", 2), ("car_obs", 2)):
    model, T, x1, ub = pkg.workloads.make_inputs(config, B)
    w = pkg.workloads.make_parameters(config, B) if config == "car_obs" else None
    r = O.solve_batch(model, T, x1, ub, w=w, nthreads=1)
    out[config] = dict(
        model=model, T=T, B=B,
        iterations=r["stats"]["iterations"].tolist(), outer_iterations=r["stats"]["outer_iterations"].tolist(),
        rollouts=r["stats"]["rollouts"].tolist(), objective=r["stats"]["objective"].tolist(),
        max_violation=r["stats"]["max_violation"].tolist(),
        x_final=r["x"][:, -1, :].tolist(), u_first=r["u"][:, 0, :].tolist(), u_last=r["u"][:, -1, :].tolist(),
        K_first=r["K"][:, 0].reshape(B, -1).tolist(), k_first=r["k"][:, 0].tolist(),
        x_checksum=[float(np.sum(r["x"][b] * np.cos(np.arange(r["x"][b].size)).reshape(r["x"][b].shape))) for b in range(B)],
    )
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "oracle_regression.json"), "w"), indent=1)
print("wrote", {k: v["iterations"] for k, v in out.items()})
