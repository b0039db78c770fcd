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


def main(out_dir=None):
    out_dir = out_dir or os.path.join(HERE, "julia_in")
    os.makedirs(out_dir, exist_ok=True)
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
