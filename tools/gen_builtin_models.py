"""Regenerate csrc/models/model_<name>.h for the built-in model zoo (needs sympy).
The generated headers are committed so that build() only needs hipcc."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ilqr_amd_loader import load_package  # noqa: E402

pkg = load_package()
out_dir = os.path.join(ROOT, "iterativelqr.jl_amd", "csrc", "models")
names = sys.argv[1:] or list(pkg.models.BUILTIN)
for name in names:
    sname, src = pkg.models.builtin_source(name)
    path = os.path.join(out_dir, "model_%s.h" % name)
    with open(path, "w") as f:
        f.write("#pragma once\n" + src)
    print("wrote", path, len(src.splitlines()), "lines")
