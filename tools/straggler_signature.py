"""When does a straggler of BASELINE config 4 (acrobot, 8 shards of 8192) show? From the per-iteration trace of every instance,
the measure of ilqr_set_handover_mark — rejected line-search trials of forward passes that ended in an acceptance — against the
batch's mean at the same cycle (a cycle of the packed kernel = one trial per instance: the instance's rollout count), in a
lock-step model of the batch: which instances a mark of K rejected trials above the mean hits, at which cycle, and what they still
have to do then.      python tools/straggler_signature.py [shards, e.g. 2,6] [SLOW iterations]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ilqr_amd_loader import load_package
pkg = load_package()
shards = [int(s) for s in sys.argv[1].split(",")] if len(sys.argv) > 1 else list(range(8))
SLOW = int(sys.argv[2]) if len(sys.argv) > 2 else 520
B, cfg, W, C = 8192, "acrobot", 1000, 2200
model, T, _ = pkg.workloads.CONFIGS[cfg]
sol = pkg.Solver(model=model, horizon=T, batch=B, options=pkg.Options(verbose=0))
sol.enable_trace_(W)
for r in shards:
    _, _, x1, ub = pkg.workloads.make_inputs(cfg, B, offset=r * B)
    sol.reset_(); sol.initialize_rollout_(x1, ub); sol.timing_reset(); sol.solve_(); ms = sol.timing()[0]
    st = sol.stats(); it = st["iterations"].astype(int); ro = st["rollouts"].astype(int)
    tr = sol.trace(); ln = sol.scalar("trace_len").astype(int)
    rows = np.arange(W)[None, :]
    with np.errstate(divide="ignore"):
        rejected = np.where(tr[:, :, 5] > 0, np.round(-np.log2(np.maximum(tr[:, :, 5], 1e-300))), 0).astype(int)    # step 2^-k: k rejected trials
    acc = np.cumsum(np.where((rows < ln[:, None]) & (tr[:, :, 6] > 0), rejected, 0), 1)                            # ... of accepted passes only
    at = np.zeros((B, C), np.int32)                                     # the measure as a function of the cycle (= rollouts so far)
    for b in range(B):
        idx = np.searchsorted(tr[b, :ln[b], 7], np.arange(C), side="right") - 1
        at[b] = np.where(idx >= 0, acc[b, np.clip(idx, 0, W - 1)], 0)
    mean = at.mean(0)
    q, m = sol.handover_stats()
    print("== shard %d: kernel %.1f ms (through the queue %d, marked %d); iterations mean %.1f max %d; rejected trials per instance mean %.1f, failed searches excluded %.1f"
          % (r, ms, q, m, it.mean(), it.max(), (ro - it).mean(), acc[np.arange(B), np.maximum(ln - 1, 0)].mean()))
    for K in (6, 8):
        hit = at - mean[None, :] >= K
        first = np.where(hit.any(1), hit.argmax(1), -1)
        print("  mark at %d above the mean: %d instances" % (K, (first >= 0).sum()))
        for b in np.nonzero((first >= 0) | (it > SLOW))[0]:
            if first[b] >= 0:
                k = int(np.searchsorted(tr[b, :ln[b], 7], first[b], side="right"))
                print("      instance %5d (%d iterations, %d rollouts): marked in cycle %d = iteration %d; %d iterations and %d rollouts to go"
                      % (b, it[b], ro[b], first[b], k, it[b] - k, ro[b] - first[b]))
            else:
                print("      instance %5d (%d iterations, %d rollouts): never marked" % (b, it[b], ro[b]))
