"""Finishing times and wave placement of the four-wave large-model kernel (two instances per CU).
    python tools/finish_times_large.py [config] [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from collections import defaultdict
from ilqr_amd_loader import load_package
pkg = load_package()
cfg = sys.argv[1] if len(sys.argv) > 1 else "synth32_tight11"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
model, T, x1, ub = pkg.workloads.make_inputs(cfg, B)
opts = pkg.Options(verbose=0, **pkg.workloads.CONFIG_OPTIONS.get(cfg, {}))
s = pkg.Solver(model=model, horizon=T, batch=B, options=opts)
L = pkg._ffi.lib()
slot0, slot1 = L.ilqr_scalar_slot(b"t_start"), L.ilqr_scalar_slot(b"t_end")
hs = [L.ilqr_scalar_slot(b"hw_id_wave%d" % w) for w in range(4)]
for rep in range(3):
    s.reset_(); s.initialize_rollout_(x1, ub); s.timing_reset(); s.solve_(); ms = s.timing()[0]
sc = s.buffer("_scalars"); st = s.stats()
t0 = sc[:, slot0]; t1 = sc[:, slot1]
end = (t1 - t0.min()) / 1e5; life = (t1 - t0) / 1e5
it = np.maximum(st["iterations"].astype(float), 1)
print("# %s B=%d: kernel %.2f ms; iterations mean %.1f max %d" % (cfg, B, ms, it.mean(), it.max()))
print("finish percentiles (ms): " + "  ".join("p%g=%.2f" % (q, np.percentile(end, q)) for q in [0, 25, 50, 75, 90, 99, 100]))
rate = 1e3 * life / it
print("us per iteration: mean %.1f p5 %.1f p50 %.1f p95 %.1f max %.1f" % (rate.mean(), np.percentile(rate, 5), np.percentile(rate, 50), np.percentile(rate, 95), rate.max()))
def where(v):
    v = int(v); xcc = v >> 32; id_ = v & 0xffffffff
    return (xcc, (id_ >> 13) & 7, (id_ >> 12) & 1, (id_ >> 8) & 15, (id_ >> 4) & 3, id_ & 15)
W = [[where(sc[b, h]) for h in hs] for b in range(B)]
cu = defaultdict(list)
for b in range(B): cu[W[b][0][:4]].append(b)
print("CUs in use %d; workgroups per CU: %s" % (len(cu), dict(zip(*np.unique([len(v) for v in cu.values()], return_counts=True)))))
same0 = 0; groups = defaultdict(list)
for k, bs in cu.items():
    if len(bs) != 2: continue
    a, b = sorted(bs)
    share = W[a][0][4] == W[b][0][4]
    same0 += share
    groups[("first", share)].append(rate[a]); groups[("second", share)].append(rate[b])
print("CUs whose two hardware wave-0s share a SIMD: %d of %d" % (same0, sum(1 for v in cu.values() if len(v) == 2)))
for k, v in sorted(groups.items()): print("  %s workgroup of its CU, wave-0s share a SIMD = %s: %d instances, %.1f us per iteration" % (k[0], k[1], len(v), np.mean(v)))
print("examples (instance: simd.slot of hardware waves 0..3, us/iteration):")
for k, bs in list(cu.items())[:6]:
    print("  %s: %s" % (k, "  |  ".join("%d: %s  %.1f" % (b, " ".join("%d.%d" % (W[b][w][4], W[b][w][5]) for w in range(4)), rate[b]) for b in sorted(bs))))
order = np.argsort(-end)
print("last finishers: " + "  ".join("%d (%.2f ms, %d it, %.1f us/it)" % (b, end[b], it[b], rate[b]) for b in order[:6]))
s.close()
