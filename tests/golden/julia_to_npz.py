"""tests/golden/julia_out/<case>/ (written by make_julia_fixtures.jl from the real package) -> tests/golden/ref_<case>_julia.npz
in the schema of make_reference_fixtures.py. Julia arrays are column-major: a stack of matrices arrives as (rows, cols, T)
and becomes [t][row][col]; a stack of vectors (n, T) becomes [t][i].

    python tests/golden/julia_to_npz.py [julia_out dir] [destination dir]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
INT_KEYS = ("points", "horizon", "K_steps")


def convert_case(src, dst_path):
    out = {}
    for line in open(os.path.join(src, "manifest.txt")):
        parts = line.split()
        if not parts:
            continue
        key, dims = parts[0], [int(v) for v in parts[1:]]
        a = np.fromfile(os.path.join(src, key + ".f64"), dtype="<f8")
        if len(dims) >= 2:
            a = a.reshape(dims[::-1])                  # Julia (d1, d2, d3) column-major == numpy (d3, d2, d1) row-major
            if len(dims) == 3:
                a = a.transpose(0, 2, 1)               # (T, cols, rows) -> [t][row][col]
        if key in INT_KEYS:
            a = np.rint(a).astype(np.int64)
        out[key] = a
    if out["points"].size == 0:
        out["points"] = np.zeros((0, 2), dtype=np.int64)
    out["trace"] = out["trace"].reshape(-1, 8)
    np.savez_compressed(dst_path, **out)
    return out


def main(src_root=None, dst=None):
    src_root = src_root or os.path.join(HERE, "julia_out")
    dst = dst or HERE
    done = []
    for case in sorted(os.listdir(src_root)):
        if os.path.exists(os.path.join(src_root, case, "manifest.txt")):
            convert_case(os.path.join(src_root, case), os.path.join(dst, "ref_%s_julia.npz" % case))
            done.append(case)
    return done


if __name__ == "__main__":
    print("converted", main(*(sys.argv[1:3])))
