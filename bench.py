#!/usr/bin/env python3
"""Headline benchmark: trajectories/sec, acrobot T=101, batch=1024 per GPU (BASELINE.json).

One step = one pass of the hot path over one batch: fresh-solver reset, open-loop
rollout initialisation from device-resident (x1, ū), and the whole AL/iLQR solve
of every instance (`solve!`), all on the GPU. Inputs are resident in HBM before
the timed region starts.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

The batch shards embarrassingly: every rank solves its own 1024 instances with no
data-path collective (weak scaling); the only collectives are the timing barrier
and a MAX over ranks of the elapsed time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes(n, m, T, C, iterations, rollouts):
    """SURVEY.md Appendix D stage-materialised traffic model (bytes).
    One inner iteration with one line-search trial + extra trials."""
    N = T - 1
    g_w = N * (n * n + n * m + m + m * m + m * n) + T * (n + n * n)
    g_r = T * n + N * m + 3.5 * C + (T * n * n + N * m * m + N * m * n)
    b_r = g_w
    b_w = N * (m * n + m) + N * (n + m)
    f_s = N * (n * n + n * m) + N * (m * n + m) + N * (n + m)
    tr = (T * n + N * m + N * (m * n + m)) + (T * n + N * m + C)
    per_iter = g_w + g_r + b_r + b_w + f_s + tr
    extra = np.maximum(rollouts - iterations, 0)
    return 8.0 * (iterations * per_iter + extra * tr)


def host_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1024, help="instances per GPU")
    ap.add_argument("--config", default="acrobot")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--distinct-shards", action="store_true",
                    help="rank r solves instances [r*B, (r+1)*B) of one big synthetic batch instead of the same B instances "
                         "on every rank (weak scaling then also measures how unlucky the worst shard's slowest instance is)")
    ap.add_argument("--inflight", type=int, default=1,
                    help="solver handles used round-robin on separate HIP streams (1 = strictly sequential steps, "
                         "the headline setting; 2 lets the next batch fill SIMDs freed by early finishers)")
    args = ap.parse_args()

    import torch
    from ilqr_amd_loader import load_package
    pkg = load_package()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # test hook: ILQR_BENCH_SHARE_DEVICE=1 lets several ranks share GPU 0 over gloo, to exercise the
    # multi-rank code path on a one-GPU box; the real runs use one GPU per rank over RCCL
    share = os.environ.get("ILQR_BENCH_SHARE_DEVICE") == "1"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    gpu = 0 if share else local_rank
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(gpu)
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", gpu))
    torch.cuda.set_device(gpu)
    dev = torch.device("cuda", gpu)
    local_rank = gpu

    B = args.batch
    # weak scaling with IDENTICAL per-GPU work by default: a step lasts as long as the slowest instance of the batch
    # (iteration counts are data dependent: 500 for the slowest of these 1024 instances, 381..649 for the slowest of
    # other shards), so distinct shards would fold that data lottery into the scaling figure
    model, T, x1, ub = pkg.workloads.make_inputs(args.config, B, offset=(rank * B if args.distinct_shards else 0))
    d_x1 = torch.from_numpy(x1).to(dev)
    d_u = torch.from_numpy(ub).to(dev)
    sols = [pkg.Solver(model=model, horizon=T, batch=B, device=local_rank, options=pkg.Options(verbose=0))
            for _ in range(max(1, args.inflight))]
    sol = sols[0]
    if len(sols) > 1:      # several batches in flight: which kernel fills the SIMDs freed by early finishers best
        for s_ in sols:
            s_.set_kernel_variant_(os.environ.get("ILQR_INFLIGHT_VARIANT", "throughput"))
    torch.cuda.synchronize()
    counter = [0]

    def step():
        s = sols[counter[0] % len(sols)]
        counter[0] += 1
        s.reset_()
        s.initialize_rollout_device_(d_x1.data_ptr(), d_u.data_ptr())
        s.solve_(sync=False)

    def barrier():
        for s in sols:
            s.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    for s_ in sols:
        s_.timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    for s_ in sols:
        s_.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kernel_ms, launches = sol.timing()
    st = sol.stats()

    # the same step with the boundary fed HOST buffers: x1, ū over PCIe in, x, u, K, k over PCIe out
    # (reported beside `value`, never as `value`)
    h0 = time.perf_counter()
    host_steps = 3
    for _ in range(host_steps):
        sol.reset_()
        sol.initialize_rollout_(x1, ub)
        sol.solve_(sync=True)
        sol.get_trajectory()
        sol.get_policy()
    host_elapsed = (time.perf_counter() - h0) / host_steps
    n_, m_ = sol.nx, sol.nu
    io_bytes = 8.0 * B * (n_ + (T - 1) * m_ + T * n_ + (T - 1) * m_ + (T - 1) * (m_ * n_ + m_))
    C = (T - 1) * sol.nc_stage + sol.nc_term
    abytes = float(algorithmic_bytes(sol.nx, sol.nu, T, C, st["iterations"].astype(np.float64),
                                     st["rollouts"].astype(np.float64)).sum())
    achieved = abytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    value = world * B * args.steps / elapsed
    traffic = None
    try:   # HBM bytes per launch from the committed rocprofv3 PMC passes (separate runs, see profiles/)
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        traffic = pmc.get("%s:%d" % (args.config, B), {}).get("traffic_bytes_per_launch")
    except (OSError, ValueError):
        pass

    out = {
        "metric": "trajectories/sec (whole node), acrobot T=101 batch=1024/GPU",
        "value": value, "unit": "trajectories/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s (nx=%d, nu=%d, T=%d) AL-iLQR solve!, batch=%d per GPU, fp64, faithful reference semantics"
                               % (args.config, sol.nx, sol.nu, T, B),
                   "global_batch": world * B, "horizon": T, "parallelism": "batch-shard x%d (no collective), %s" % (world, "distinct shards" if args.distinct_shards else "same 1024 instances per GPU"),
                   "batches_in_flight": len(sols)},
        "solve_stats": {"inner_iterations_mean": float(st["iterations"].mean()),
                        "rollouts_mean": float(st["rollouts"].mean()),
                        "outer_iterations_mean": float(st["outer_iterations"].mean()),
                        "converged_frac": float((st["max_violation"] <= 5e-3).mean()),
                        "trajectory_iterations_per_s": float(world * st["iterations"].sum() * args.steps / elapsed)},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": "solve_kernel<Model_%s>" % model, "kernel_ms_avg": kernel_ms, "launches": launches,
                     "algorithmic_bytes_per_launch": abytes,
                     # pure I/O of a fully fused solve (x1, ū in; x, u, K, k out): how far below the
                     # stage-materialised model a resident solve sits
                     "io_lower_bound_bytes_per_launch": io_bytes,
                     "io_lower_bound_GBs": io_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0},
        "host_boundary": {"value": B / host_elapsed, "unit": "trajectories/s",
                          "note": "one rank, inputs from host memory and x, u, K, k copied back (PCIe-inclusive)"},
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        threads = host_cores()
        sample = min(B, 64 * threads)
        c0 = time.perf_counter()
        O.solve_batch(model, T, x1[:sample], ub[:sample], nthreads=threads, want_policy=False)
        c1 = time.perf_counter() - c0
        one = min(sample, 48)
        c0 = time.perf_counter()
        O.solve_batch(model, T, x1[:one], ub[:one], nthreads=1, want_policy=False)
        c_one = time.perf_counter() - c0
        out["cpu_baseline"] = {"value": sample / c1, "unit": "trajectories/s", "cores": threads, "kind": "port",
                               "single_thread_value": one / c_one,
                               "sample": "first %d of the %d instances of this workload, C++ oracle "
                                         "(literal restatement of the Julia reference, which cannot run here), "
                                         "OpenMP over instances, %.1f s wall" % (sample, B, c1)}
    if rank == 0:
        print(json.dumps(out))
    for s_ in sols:
        s_.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
