"""synth32 defined by C callables (examples/synth32_model.c through ilqr_compile_model, structure found by probing on the host)
against the model the symbolic generator makes: kernel time of BASELINE config 5's shard and agreement of the results.
    python tools/c_model_bench.py [B] [config]"""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ilqr_amd_loader import load_package
pkg = load_package()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
cfg = sys.argv[2] if len(sys.argv) > 2 else "synth32"


class Src(C.Structure):
    _fields_ = [("name", C.c_char_p), ("nx", C.c_int32), ("nu", C.c_int32), ("nw", C.c_int32), ("nc_stage", C.c_int32),
                ("nc_term", C.c_int32), ("ineq_stage", C.c_uint64), ("ineq_term", C.c_uint64), ("source", C.c_char_p), ("flags", C.c_int32)]


def compile_c(dense):
    if dense:
        os.environ["ILQR_NO_STRUCTURE_PROBE"] = "1"
    else:
        os.environ.pop("ILQR_NO_STRUCTURE_PROBE", None)
    L = pkg._ffi.lib()
    text = open(os.path.join(ROOT, "examples", "synth32_model.c"), "rb").read()
    ms = Src(b"synth32_c", 32, 8, 0, 16, 0, (1 << 16) - 1, 0, text)
    name = C.create_string_buffer(128); path = C.create_string_buffer(1024)
    t0 = time.time()
    if L.ilqr_compile_model(C.byref(ms), name, 128, path, 1024) != 0:
        print("ilqr_compile_model (%s) FAILED: %s" % ("dense" if dense else "probed", L.ilqr_last_error().decode()[-400:]))
        return None
    jv, hs = C.c_int32(), C.c_int32()
    L.ilqr_model_compact_sizes(name.value, C.byref(jv), C.byref(hs))
    print("ilqr_compile_model (%s): %.1f s, %d state-dependent Jacobian entries, %d non-zero Hessian entries" % ("dense" if dense else "probed", time.time() - t0, jv.value, hs.value))
    return name.value.decode()


model, T, x1, ub = pkg.workloads.make_inputs(cfg, B, offset=5 * B)
opts = pkg.workloads.CONFIG_OPTIONS.get(cfg, {})
res = {}
for label, mdl in (("generated", model), ("C callables, probed", compile_c(False)), ("C callables, dense", compile_c(True))):
    if mdl is None:
        continue
    s = pkg.Solver(model=mdl, horizon=T, batch=B, options=pkg.Options(verbose=0, **opts))
    ts = []
    for rep in range(4):
        s.reset_(); s.initialize_rollout_(x1, ub); s.timing_reset(); s.solve_(); ts.append(s.timing()[0])
    res[label] = (s.get_trajectory()[0], s.stats())
    print("%-24s kernel %s ms; iterations mean %.2f max %d" % (label, " ".join("%.2f" % t for t in ts[1:]), res[label][1]["iterations"].mean(), res[label][1]["iterations"].max()))
    s.close()
g = res["generated"]
for label in ("C callables, probed", "C callables, dense"):
    if label not in res:
        continue
    r = res[label]
    same = (r[1]["iterations"] == g[1]["iterations"]) & (r[1]["rollouts"] == g[1]["rollouts"])
    print("%s vs generated: control flow identical on %.1f %%, max |dx| %.3e" % (label, 100 * same.mean(), np.abs(r[0] - g[0])[same].max()))

# where the C-callable model's time goes: wall time of single stage launches (synchronous calls), generated model beside it
print("stage times (ms per launch, %d instances):" % B)
for label, mdl in (("generated", model), ("C callables, probed", compile_c(False))):
    s = pkg.Solver(model=mdl, horizon=T, batch=B, options=pkg.Options(verbose=0, **opts))
    s.initialize_rollout_(x1, ub)
    out = []
    for stage in ("cost_nominal", "gradients", "backward_pass", "forward_pass"):
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter(); s.run_stage_(stage); s.synchronize(); best = min(best, time.perf_counter() - t0)
        out.append("%s %.2f" % (stage, 1e3 * best))
    t0 = time.perf_counter(); s.reset_(); s.initialize_rollout_(x1, ub); s.synchronize()
    out.append("reset + initial rollout %.2f" % (1e3 * (time.perf_counter() - t0)))
    print("   %-22s %s" % (label, "  ".join(out)))
    s.close()
