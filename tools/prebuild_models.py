"""Compile the C-source model modules the examples and the GPU tests use (ilqr_compile_model: hipcc as a child process, no GPU
needed) into iterativelqr.jl_amd/lib/models/, the library's module cache — so that a GPU box that receives the tree finds them
built (the dense 64 x 16 module alone takes a minute). Called by __graft_entry__.build()."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Src(C.Structure):
    _fields_ = [("name", C.c_char_p), ("nx", C.c_int32), ("nu", C.c_int32), ("nw", C.c_int32), ("nc_stage", C.c_int32),
                ("nc_term", C.c_int32), ("ineq_stage", C.c_uint64), ("ineq_term", C.c_uint64), ("source", C.c_char_p), ("flags", C.c_int32)]


def main(verbose=True):
    from ilqr_amd_loader import load_package
    pkg = load_package()
    L = pkg._ffi.lib()
    ex = lambda f: open(os.path.join(ROOT, "examples", f), "rb").read()
    jobs = [  # (name, dims, source, probe) exactly as tests/test_gpu_parity.py and tools/c_model_bench.py pass them
        (b"synth32_c", (32, 8, 0, 16, 0, (1 << 16) - 1, 0), ex("synth32_model.c"), True),
        (b"synth32_c", (32, 8, 0, 16, 0, (1 << 16) - 1, 0), ex("synth32_model.c"), False),
        (b"synth64_c", (64, 16, 0, 32, 0, (1 << 32) - 1, 0), pkg.models.synth_c_source(64, 16).encode(), True),
        (b"synth64_c", (64, 16, 0, 32, 0, (1 << 32) - 1, 0), pkg.models.synth_c_source(64, 16).encode(), False),
        (b"synth12_t", (12, 5, 0, 10, 3, (1 << 10) - 1, 0), ex("synth12_model.c"), True),
    ]
    old = os.environ.get("ILQR_NO_STRUCTURE_PROBE")
    try:
        for name, dims, text, probe in jobs:
            if probe:
                os.environ.pop("ILQR_NO_STRUCTURE_PROBE", None)
            else:
                os.environ["ILQR_NO_STRUCTURE_PROBE"] = "1"
            ms = Src(name, *dims, text)
            reg = C.create_string_buffer(128); path = C.create_string_buffer(1024)
            rc = L.ilqr_compile_model(C.byref(ms), reg, 128, path, 1024)
            if rc != 0:
                raise RuntimeError("ilqr_compile_model(%s, probe=%s): %s" % (name.decode(), probe, L.ilqr_last_error().decode()[-600:]))
            if verbose:
                print("model module %s (%s tables): %s" % (reg.value.decode(), "probed" if probe else "dense", os.path.basename(path.value.decode())))
        # per-step objects through ilqr_compile_model_stages (the GPU tests' ragged_c and car_tv_c; horizon 41 / 51 as there)
        F = pkg._ffi
        for name, (kinds, src) in ((b"ragged_c", pkg.models.ragged_c_stages(41)), (b"car_tv_c", pkg.lowering.c_stage_sources(*pkg.models.car_tv(51)))):
            cap = kinds.horizon * (kinds.n_dynamics + kinds.n_costs + kinds.n_constraints)
            plan, sel = F.StagePlan(), (C.c_double * max(cap, 1))()
            reg = C.create_string_buffer(160); path = C.create_string_buffer(1024)
            rc = L.ilqr_compile_model_stages(name, C.byref(kinds), src.encode(), C.byref(plan), sel, cap, None, None, reg, 160, path, 1024)
            if rc != 0:
                raise RuntimeError("ilqr_compile_model_stages(%s): %s" % (name.decode(), L.ilqr_last_error().decode()[-600:]))
            if verbose:
                print("model module %s (per-step kinds): %s" % (reg.value.decode(), os.path.basename(path.value.decode())))
    finally:
        if old is None:
            os.environ.pop("ILQR_NO_STRUCTURE_PROBE", None)
        else:
            os.environ["ILQR_NO_STRUCTURE_PROBE"] = old


if __name__ == "__main__":
    main()
