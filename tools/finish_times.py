"""When the instances of a latency-kernel batch finish (S_T_START / S_T_END: s_memrealtime, 100 MHz, one counter per device).
    python tools/finish_times.py [config] [B] [generator]
Prints the finishing-time profile of the batch, the last finishers with their iteration / rollout counts, and how an instance's
lifetime relates to its own work (us per iteration in the batch against the same instance alone on the chip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ilqr_amd_loader import load_package
pkg = load_package()
cfg = sys.argv[1] if len(sys.argv) > 1 else "acrobot"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
gen = sys.argv[3] if len(sys.argv) > 3 else "splitmix64"
alone = [int(v) for v in sys.argv[4:]]
model, T, x1, ub = pkg.workloads.make_inputs(cfg, B, generator=gen)
opts = pkg.Options(verbose=0, **pkg.workloads.CONFIG_OPTIONS.get(cfg, {}))
s = pkg.Solver(model=model, horizon=T, batch=B, options=opts)
s.set_kernel_variant_("latency")
slot0, slot1 = pkg._ffi.lib().ilqr_scalar_slot(b"t_start"), pkg._ffi.lib().ilqr_scalar_slot(b"t_end")
reps = []
for rep in range(int(os.environ.get("FINISH_REPS", "3"))):
    s.reset_(); s.initialize_rollout_(x1, ub); s.timing_reset(); s.solve_(); ms = s.timing()[0]
    sc = s.buffer("_scalars"); st_ = s.stats()
    reps.append((ms, 1e-2 * ((sc[:, slot1] - sc[:, slot0]) / st_["iterations"]).max()))
print("# per launch: kernel ms (slowest instance's us per iteration): " + "  ".join("%.2f (%.1f)" % r for r in reps))
sc = s.buffer("_scalars"); st = s.stats()
t0 = sc[:, slot0]; t1 = sc[:, slot1]
start = (t0 - t0.min()) / 1e5; end = (t1 - t0.min()) / 1e5          # ms
it = st["iterations"].astype(float); ro = st["rollouts"].astype(float)
print("# %s B=%d generator=%s: kernel %.2f ms; starts within %.3f ms; iterations mean %.1f max %d; rollouts mean %.1f max %d"
      % (cfg, B, gen, ms, start.max(), it.mean(), it.max(), ro.mean(), ro.max()))
qs = [0, 5, 25, 50, 75, 90, 95, 99, 99.9, 100]
print("finish time percentiles (ms): " + "  ".join("p%g=%.2f" % (q, np.percentile(end, q)) for q in qs))
life = end - start
print("us per iteration in the batch: mean %.2f  p5 %.2f  p50 %.2f  p95 %.2f  max %.2f" % tuple(
    1e3 * v for v in ((life / it).mean(), np.percentile(life / it, 5), np.percentile(life / it, 50), np.percentile(life / it, 95), (life / it).max())))
# least squares: lifetime = a * iterations + b * (rollouts - iterations)
A = np.stack([it, ro - it], 1); coef, *_ = np.linalg.lstsq(A, life, rcond=None)
print("lifetime ~ %.2f us x iterations + %.2f us x rejected trials (rms residual %.3f ms)" % (1e3 * coef[0], 1e3 * coef[1], np.sqrt(np.mean((A @ coef - life) ** 2))))
order = np.argsort(-end)
print("last finishers: instance end_ms iterations rollouts us_per_iteration")
for b in order[:12]:
    print("  %5d %7.2f %5d %5d %7.2f" % (b, end[b], it[b], ro[b], 1e3 * life[b] / it[b]))
# placement: which SIMD hosted which waves
hs = [pkg._ffi.lib().ilqr_scalar_slot(b"hw_id_wave0"), pkg._ffi.lib().ilqr_scalar_slot(b"hw_id_wave1")]
def where(v):
    v = int(v); xcc = v >> 32; id_ = v & 0xffffffff
    return (xcc, (id_ >> 13) & 7, (id_ >> 12) & 1, (id_ >> 8) & 15, (id_ >> 4) & 3)      # xcc, se, sh, cu, simd
W = [[where(sc[b, h]) for h in hs] for b in range(B)]
from collections import defaultdict
simd = defaultdict(list); cu = defaultdict(list)
for b in range(B):
    for w in range(2):
        simd[W[b][w]].append((b, w)); cu[W[b][w][:4]].append((b, w))
hist = defaultdict(int)
for k, v in simd.items(): hist[(sum(1 for _, w in v if w == 0), sum(1 for _, w in v if w == 1))] += 1
print("SIMDs in use %d, CUs in use %d; SIMDs by (wave-0s, wave-1s) hosted: %s" % (len(simd), len(cu), dict(hist)))
print("workgroups per CU: %s" % dict(zip(*np.unique([len(set(b for b, _ in v)) for v in cu.values()], return_counts=True))))
same = sum(1 for b in range(B) if W[b][0] == W[b][1])
print("instances whose two waves share a SIMD: %d" % same)
# lifetime per iteration against what the instance's SIMDs host
load0 = np.array([len(simd[W[b][0]]) for b in range(B)]); load1 = np.array([len(simd[W[b][1]]) for b in range(B)])
n0_on0 = np.array([sum(1 for _, w in simd[W[b][0]] if w == 0) for b in range(B)])
for key in sorted(set(zip(load0, n0_on0, load1))):
    m = (load0 == key[0]) & (n0_on0 == key[1]) & (load1 == key[2])
    print("  wave 0's SIMD hosts %d waves (%d of them wave-0s), wave 1's SIMD hosts %d: %4d instances, %.2f us per iteration, finish %.2f ms (mean)"
          % (key[0], key[1], key[2], m.sum(), 1e3 * (life[m] / it[m]).mean(), end[m].mean()))
print("last finishers' placement (xcc, se, sh, cu, simd of wave 0 | wave 1):")
for b in order[:8]: print("  %5d %s | %s" % (b, W[b][0], W[b][1]))
rate = 1e3 * life / it
print("us per iteration by instance index, rows of 32 (mean | min max):")
for r0 in range(0, B, 32):
    v = rate[r0:r0 + 32]
    print("  %4d: %.1f | %.1f %.1f" % (r0, v.mean(), v.min(), v.max()))
print("CUs sorted by the mean rate of their four instances (xcc, se, sh, cu): instances -> rates")
rows = []
for k, v in cu.items():
    bs = sorted(set(b for b, _ in v))
    rows.append((np.mean([rate[b] for b in bs]), k, bs))
rows.sort(reverse=True)
for m_, k, bs in rows[:12] + rows[-4:]:
    print("  %s: %s -> %s" % (k, bs, " ".join("%.1f" % rate[b] for b in bs)))
slot = lambda v: int(v) & 15
print("CUs with a SIMD hosting two wave-0s: per workgroup (instance: wave-0 simd.slot, wave-1 simd.slot, us/iteration)")
shown = 0
for k, v in cu.items():
    bs = sorted(set(b for b, _ in v))
    s0 = [W[b][0][4] for b in bs]
    ab = len(set(s0)) < len(s0)
    if ab or shown < 6:
        if not ab: shown += 1
        print("  %s %s: %s" % ("ABNORMAL" if ab else "normal  ", k, "  ".join("%d: %d.%d %d.%d %.1f" % (b, W[b][0][4], slot(sc[b, hs[0]]), W[b][1][4], slot(sc[b, hs[1]]), rate[b]) for b in bs)))
live = [(end > t).sum() for t in np.arange(0, end.max() + 1, 1.0)]
print("instances still running at t = 0, 1, 2, ... ms: " + " ".join(str(v) for v in live))
s.close()
for b in (alone or [int(order[0]), int(order[1]), int(np.argsort(it)[len(it) // 2])]):
    s1 = pkg.Solver(model=model, horizon=T, batch=1, options=opts); s1.set_kernel_variant_("latency")
    for rep in range(3):
        s1.reset_(); s1.initialize_rollout_(x1[b:b + 1], ub[b:b + 1]); s1.timing_reset(); s1.solve_(); ms1 = s1.timing()[0]
    print("instance %d alone: %.2f ms (%.2f us per iteration); in the batch %.2f ms -> x %.3f" % (b, ms1, 1e3 * ms1 / it[b], life[b], life[b] / ms1))
    s1.close()
