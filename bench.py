#!/usr/bin/env python3
"""Headline benchmark: trajectories/sec, acrobot T=101, batch=1024 per GPU (BASELINE.json).

One step = one pass of the hot path over one batch: fresh-solver reset, open-loop rollout initialisation from
device-resident (x1, ū), and the whole AL/iLQR solve of every instance (`solve!`), all on the GPU. Inputs are
resident in HBM before the timed region starts.

    python bench.py --gpus N --steps K --warmup W          # N > 1: this process starts N ranks itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

The batch shards embarrassingly: every rank solves its own 1024 instances with no data-path collective (weak
scaling); the only collectives are the timing barrier, a MAX over ranks of the elapsed time and an all-gather of
the per-rank figures. `--gpus N` without a torchrun environment launches the N ranks from here, BEFORE this
process touches the GPU, and fails if fewer than N devices are visible. Under torchrun, `--gpus` must equal
WORLD_SIZE.
"""
import argparse
import json
import os
import sqlite3
import subprocess
import sys
import tempfile
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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1024, help="instances per GPU")
    ap.add_argument("--config", default="acrobot")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true",
                    help="skip the two rocprofv3 --pmc child passes that measure HBM traffic of the solve kernel live")
    ap.add_argument("--distinct-shards", action="store_true",
                    help="rank r solves instances [r*B, (r+1)*B) of one big synthetic batch (BASELINE config 4's meaning of sharding; "
                         "the DEFAULT for --gpus N > 1; at N = 1 the shard is [0, B) either way)")
    ap.add_argument("--same-instances", action="store_true",
                    help="every rank solves the SAME B instances [0, B) (fixed per-GPU work: weak scaling is then perfect by construction; "
                         "reported as secondary.same_instances in the default run)")
    ap.add_argument("--inflight", type=int, default=1,
                    help="solver handles used round-robin on separate HIP streams (1 = strictly sequential steps, "
                         "the headline setting; 2 lets the next batch fill SIMDs freed by early finishers)")
    ap.add_argument("--generator", default="splitmix64", choices=["pcg64", "splitmix64"],
                    help="synthetic inputs: the library's own ilqr_synthetic_inputs (SURVEY 8(d) / BASELINE.md: splitmix64 / Box-Muller, "
                         "seed 20240607 — what the cpu_baseline leg, a C host and bench/julia_ref.jl regenerate; the default since round 6) "
                         "or numpy PCG64 streams (the inputs of the figures of rounds 1-5; reported as secondary.pcg64 by the default run)")
    ap.add_argument("--variant", default="auto", choices=["auto", "latency", "throughput", "packed", "mid", "packed1", "packed2"])
    ap.add_argument("--shared-step", action="store_true",
                    help="optional mode, NOT the reference's behaviour and not the headline: one Armijo step size per inner iteration "
                         "for the whole (multi-GPU) batch, decided on the summed merit — one all-reduce (RCCL over xGMI) of three "
                         "doubles per line-search trial, line search stepped from the host (Solver.solve_shared_step_)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the two short secondary passes (distinct shards per rank; two batches in flight) reported as extra keys")
    ap.add_argument("--sharded-handle", action="store_true",
                    help="ONE process, ONE solver handle over the first N devices (ilqr_create_sharded: what a Julia host holding one "
                         "Solver would use) instead of one process per GPU; same JSON line, ranks.launcher says so")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ PMC traffic
def pmc_child(args):
    """Body of the rocprofv3 child passes: the same solves as the timed steps, through the C-ABI with host inputs
    (no torch: the counters are per kernel dispatch, the init path does not matter)."""
    from ilqr_amd_loader import load_package
    pkg = load_package()
    model, T, x1, ub = pkg.workloads.make_inputs(args.config, args.batch, generator=args.generator)
    sol = pkg.Solver(model=model, horizon=T, batch=args.batch,
                     options=pkg.Options(verbose=0, **pkg.workloads.CONFIG_OPTIONS.get(args.config, {})))
    sol.set_kernel_variant_(args.variant)
    for _ in range(max(1, args.steps)):
        sol.reset_()
        sol.initialize_rollout_(x1, ub)
        sol.solve_()
    sol.close()


def measure_traffic(args, kernel_prefix="void ilqr::solve_kernel"):
    """HBM bytes per launch of the solve kernel from rocprofv3 PMC counters, collected the way
    MI355X_MICROARCH.md §HBM prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slots), each with
    --kernel-trace only; both counters are in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced
    reads, so it is doubled. Returns (dict, None) or (None, reason)."""
    res = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        with tempfile.TemporaryDirectory(prefix="ilqr_pmc_", dir="/tmp") as d:
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "-d", d, "-o", "pmc", "--",
                   sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child", "--config", args.config,
                   "--batch", str(args.batch), "--steps", "2", "--variant", args.variant, "--generator", args.generator]
            env = dict(os.environ, TMPDIR="/tmp")
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
                env.pop(k, None)
            try:
                p = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240)
            except (OSError, subprocess.TimeoutExpired) as e:
                return None, "rocprofv3 %s pass failed to run: %s" % (counter, e)
            if p.returncode != 0:
                return None, "rocprofv3 %s pass exited %d: %s" % (counter, p.returncode, p.stdout.decode()[-300:])
            dbs = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith(".db")]
            if not dbs:
                return None, "rocprofv3 %s pass wrote no database" % counter
            try:
                cur = sqlite3.connect(dbs[0]).cursor()
                rows = list(cur.execute("select kernel_name, sum(value), count(*) from counters_collection "
                                        "where counter_name = ? group by kernel_name", (counter,)))
            except sqlite3.Error as e:
                return None, "cannot read %s from the rocprofv3 database: %s" % (counter, e)
            rows = [r for r in rows if "solve_kernel" in r[0]]
            if not rows:
                return None, "no solve_kernel dispatch in the %s pass" % counter
            name, tot, n = max(rows, key=lambda r: r[1])
            res[counter] = tot / n
            res["kernel"] = name
            res["dispatches"] = n
    # correction factors calibrated on THIS code's access patterns (tools/probes/probe_traffic.hip under the same two --pmc
    # passes, tools/calibrate_counters.py -> profiles/r03_counter_calibration.json): FETCH_SIZE reports half the bytes for 8-byte
    # per-lane loads (coalesced, 128-byte tile walks, strided read-modify-write) exactly as for the guide's 16-byte streaming
    # reads; WRITE_SIZE is exact for 8-byte coalesced, row-wise and strided stores
    f_fetch, f_write, cal_src = 2.0, 1.0, "MI355X_MICROARCH.md defaults (calibration file missing)"
    try:
        cal = json.load(open(os.path.join(ROOT, "profiles", "r03_counter_calibration.json")))
        rf = [v["fetch_over_true"] for k, v in cal.items() if k in ("read8_coalesced", "read8_tile128", "rmw8_strided") and v["fetch_over_true"]]
        rw = [v["write_over_true"] for k, v in cal.items() if k in ("write8_coalesced", "write8_rows16", "rmw8_strided") and v["write_over_true"]]
        f_fetch, f_write = len(rf) / sum(rf), len(rw) / sum(rw)
        cal_src = "profiles/r03_counter_calibration.json (tools/probes/probe_traffic.hip: measured/true %.3f for 8-byte loads, %.3f for 8-byte stores)" % (sum(rf) / len(rf), sum(rw) / len(rw))
    except (OSError, ValueError, KeyError, ZeroDivisionError):
        pass
    fetch_b, write_b = f_fetch * res["FETCH_SIZE"] * 1024.0, f_write * res["WRITE_SIZE"] * 1024.0
    return {"fetch_bytes": fetch_b, "write_bytes": write_b, "traffic_bytes_per_launch": fetch_b + write_b,
            "fetch_kib_raw": res["FETCH_SIZE"], "write_kib_raw": res["WRITE_SIZE"], "kernel": res["kernel"],
            "fetch_factor": f_fetch, "write_factor": f_write,
            "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE, separate child passes of this bench run "
                      "(2 launches each); KiB -> bytes; factors from " + cal_src}, None


# ------------------------------------------------------------------------------------------------ stub (CPU tests)
class _StubSolver:
    """Stands in for the GPU solver in the CPU test of the multi-rank path (ILQR_BENCH_STUB=1): same call sequence,
    the 'solve' is a sleep that depends on the rank so that MAX-over-ranks is observable."""

    def __init__(self, rank, batch):
        self.rank, self.B, self.nx, self.nu, self.nc_stage, self.nc_term = rank, batch, 4, 1, 0, 4

    def reset_(self): pass
    def initialize_rollout_device_(self, a, b): pass
    def initialize_rollout_(self, a, b): pass
    def set_kernel_variant_(self, v): pass
    def solve_(self, sync=True): time.sleep(0.002 * (self.rank + 1))
    def synchronize(self): pass
    def timing(self): return 2.0 * (self.rank + 1), 1
    def timing_reset(self): pass
    def get_trajectory(self): return None
    def get_policy(self): return None
    def close(self): pass

    def stats(self):
        it = np.full(self.B, 10 + self.rank)
        return {"iterations": it, "rollouts": it + 1, "outer_iterations": np.full(self.B, 2),
                "max_violation": np.zeros(self.B)}


# ------------------------------------------------------------------------------------------------ one rank
def worker(args, solver_factory=None):
    """One rank. solver_factory(rank, model, T, B, x1, ub) -> solver-like object replaces the GPU solver (tests only: the CPU
    tests of the N > 1 path run this very function over gloo with a sleeping stub or with a CPU solve of the rank's shard)."""
    stub = os.environ.get("ILQR_BENCH_STUB") == "1" or solver_factory is not None
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    sharded = bool(getattr(args, "sharded_handle", False))
    ndev = args.gpus if sharded else 1           # devices under this process's handle
    if sharded and world != 1:
        raise SystemExit("bench.py: --sharded-handle is one process over N devices, not a rank of %d" % world)
    if not sharded and world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d — launch with `python bench.py --gpus N` (it starts the ranks "
                         "itself) or `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`" % (args.gpus, world))
    import torch
    from ilqr_amd_loader import load_package
    pkg = load_package()
    dist = None
    # test hook: ILQR_BENCH_SHARE_DEVICE=1 lets several ranks share GPU 0 over gloo, to exercise the
    # multi-rank code path on a one-GPU box; the real runs use one GPU per rank over RCCL
    share = os.environ.get("ILQR_BENCH_SHARE_DEVICE") == "1"
    cpu_group = share or stub
    if not stub:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (no CPU fallback)")
        if not share and torch.cuda.device_count() <= local_rank:
            raise SystemExit("bench.py: rank %d has no GPU (%d visible)" % (rank, torch.cuda.device_count()))
        if sharded and not share and torch.cuda.device_count() < ndev:
            raise SystemExit("bench.py: --sharded-handle --gpus %d but only %d device(s) visible" % (ndev, torch.cuda.device_count()))
    gpu = 0 if share else local_rank
    backend = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if cpu_group:
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(gpu)
            dist.init_process_group("nccl", device_id=torch.device("cuda", gpu))
        backend = dist.get_backend()
    if not stub:
        torch.cuda.set_device(gpu)
    dev = None if stub else torch.device("cuda", gpu)
    cdev = "cpu" if cpu_group else dev          # where the collectives' tensors live

    B = args.batch
    # N > 1: rank r solves ITS OWN shard [rB, (r+1)B) of one synthetic batch of N * B instances — the batch split BASELINE.json's
    # configs[3] describes. A step lasts as long as the slowest instance of a shard (iteration counts are data dependent: 500 for the
    # slowest of the first 1024 acrobot instances, 381..649 for the slowest of other shards), and that is part of the whole-node
    # figure. --same-instances puts the same B instances on every rank (fixed per-GPU work; secondary.same_instances by default).
    distinct = bool(args.distinct_shards) or not bool(getattr(args, "same_instances", False))
    args.distinct_shards = distinct
    lo, _ = pkg.distributed.shard_range(rank, B)
    model, T, x1, ub = pkg.workloads.make_inputs(args.config, B, offset=(lo if distinct else 0), generator=args.generator)
    if sharded:
        # one handle, ndev contiguous ranges of B instances: the same B instances on every device (fixed per-GPU work, like the
        # process-per-GPU default) or ndev * B distinct ones
        if args.distinct_shards:
            model, T, x1, ub = pkg.workloads.make_inputs(args.config, ndev * B, generator=args.generator)
        else:
            x1, ub = np.tile(x1, (ndev, 1)), np.tile(ub, (ndev, 1, 1))
    if stub:
        sols = [solver_factory(rank, model, T, B, x1, ub) if solver_factory is not None else _StubSolver(rank, B)]
        d_x1 = d_u = None
    else:
        if sharded:
            d_x1 = d_u = None
            devs = [0] * ndev if share else list(range(ndev))
            sols = [pkg.Solver(model=model, horizon=T, batch=ndev * B, devices=devs,
                               options=pkg.Options(verbose=0, **pkg.workloads.CONFIG_OPTIONS.get(args.config, {})))]
            sols[0].initialize_rollout_(x1, ub)        # inputs to every device's HBM, once, outside the timed region
        else:
            d_x1 = torch.from_numpy(x1).to(dev)
            d_u = torch.from_numpy(ub).to(dev)
            sols = [pkg.Solver(model=model, horizon=T, batch=B, device=gpu,
                               options=pkg.Options(verbose=0, **pkg.workloads.CONFIG_OPTIONS.get(args.config, {})))
                    for _ in range(max(1, args.inflight))]
    sol = sols[0]
    for s_ in sols:
        s_.set_kernel_variant_(args.variant)
    if len(sols) > 1 and args.variant == "auto":   # several batches in flight: the latency kernel fills the SIMDs freed by early finishers best
        for s_ in sols:                            # (the secondary inflight_2 figure below pins the same kernel: set_kernel_variant_(args.variant) resolves to it at 1024 instances)
            s_.set_kernel_variant_("latency")
    counter = [0]

    def step():
        s = sols[counter[0] % len(sols)]
        counter[0] += 1
        s.reset_()
        if sharded and not stub:
            s.initialize_rollout_resident_()
        else:
            s.initialize_rollout_device_(d_x1.data_ptr() if d_x1 is not None else 0, d_u.data_ptr() if d_u is not None else 0)
        if args.shared_step and not stub:
            s.solve_shared_step_(pkg.distributed.torch_allreduce_sum(dist, cdev))
        else:
            s.solve_(sync=False)

    def barrier():
        for s in sols:
            s.synchronize()
        if not stub:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            if not stub:
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
    if not stub:
        torch.cuda.synchronize()
    my_elapsed = time.perf_counter() - t0
    barrier()
    elapsed = pkg.distributed.max_over_ranks(my_elapsed, dist, cdev)

    kernel_ms, launches = sol.timing()
    st = sol.stats()
    ho_stats = None
    if not stub and not sharded:
        try:       # one-wave form of the packed kernel: what its workgroups did with the stragglers of this rank's last timed solve
            ho_stats = sol.handover_stats()
        except Exception:
            pass
    it_sum, it_max = float(st["iterations"].sum()), float(st["iterations"].max())
    # when the instances of this rank's last timed launch finished (latency kernel: S_T_START / S_T_END, 100 MHz device counter)
    finish = None
    if not stub and not sharded:
        try:
            sc_ = sol.buffer("_scalars")
            i0, i1 = pkg._ffi.lib().ilqr_scalar_slot(b"t_start"), pkg._ffi.lib().ilqr_scalar_slot(b"t_end")
            if i0 >= 0 and i1 >= 0 and (sc_[:, i1] > 0).all():
                t_end = (sc_[:, i1] - sc_[:, i0].min()) / 1e5
                life = (sc_[:, i1] - sc_[:, i0]) / 1e5
                finish = {"life_ms": life, "end_ms": t_end}
        except Exception:
            finish = None

    # ---- secondary figures of the same run (extra keys of the JSON line, never `value`): rank r on ITS OWN shard
    # [rB, (r+1)B) of one big synthetic batch (BASELINE config 4's situation: the step then lasts as long as the unluckiest
    # shard's slowest instance), and two batches in flight per GPU (the next batch fills SIMDs freed by early finishers)
    secondary = {}
    if not args.shared_step and not args.no_secondary and solver_factory is None and not sharded:
        k2 = max(1, min(3, args.steps))

        def timed(step_fn, handles):
            for s_ in handles:
                s_.synchronize()
            barrier()
            t_ = time.perf_counter()
            for _ in range(k2):
                step_fn()
            for s_ in handles:
                s_.synchronize()
            if not stub:
                torch.cuda.synchronize()
            mine = time.perf_counter() - t_
            barrier()
            return pkg.distributed.max_over_ranks(mine, dist, cdev)

        if world > 1 or not args.distinct_shards:
            # the other instance assignment than the timed one: distinct shards when the run was --same-instances, the same [0, B) on
            # every rank when it was distinct (only differs from the timed run for N > 1)
            other_lo = lo if not args.distinct_shards else 0
            m2, T2, x1b, ubb = pkg.workloads.make_inputs(args.config, B, offset=other_lo, generator=args.generator)
            if stub:
                d2 = (None, None)
            else:
                d2 = (torch.from_numpy(x1b).to(dev), torch.from_numpy(ubb).to(dev))

            def step_distinct():
                sol.reset_()
                sol.initialize_rollout_device_(d2[0].data_ptr() if d2[0] is not None else 0, d2[1].data_ptr() if d2[1] is not None else 0)
                sol.solve_(sync=False)
            el = timed(step_distinct, [sol])
            itmax = pkg.distributed.gather_over_ranks([float(sol.stats()["iterations"].max())], dist, cdev)
            key = "same_instances" if args.distinct_shards else "distinct_shards"
            secondary[key] = {"value": world * B * k2 / el, "unit": "trajectories/s", "ms_per_step": 1e3 * el / k2, "steps": k2,
                              "iterations_max_per_rank": [r[0] for r in itmax],
                              "note": ("every rank solves the same instances [0, B): fixed per-GPU work" if args.distinct_shards else
                                       "rank r solves instances [r*B, (r+1)*B): the data-dependent iteration counts of the shards enter the figure")}
        if len(sols) == 1 and not stub:
            extra = pkg.Solver(model=model, horizon=T, batch=B, device=gpu,
                               options=pkg.Options(verbose=0, **pkg.workloads.CONFIG_OPTIONS.get(args.config, {})))
            pair = [sol, extra]
            for s_ in pair:
                s_.set_kernel_variant_(args.variant)       # (same kernel as the headline: the second batch's workgroups take the LDS of early finishers)
            cnt2 = [0]

            def step_pair():
                s_ = pair[cnt2[0] % 2]
                cnt2[0] += 1
                s_.reset_()
                s_.initialize_rollout_device_(d_x1.data_ptr(), d_u.data_ptr())
                s_.solve_(sync=False)
            for _ in range(2):
                step_pair()
            el = timed(lambda: (step_pair(), step_pair()), pair)
            secondary["inflight_2"] = {"value": world * B * 2 * k2 / el, "unit": "trajectories/s", "ms_per_batch": 1e3 * el / (2 * k2),
                                       "note": "two solver handles per GPU on separate streams, alternating batches"}
            extra.close()
        if rank == 0 and not stub and world == 1 and args.generator != "pcg64":
            # continuity with rounds 1-5, whose figures were measured on numpy PCG64 inputs (same distribution, other draws: one
            # instance of that batch needs 500 inner iterations where this batch's slowest needs 411)
            _, _, x1p, ubp = pkg.workloads.make_inputs(args.config, B, generator="pcg64")
            dp = (torch.from_numpy(x1p).to(dev), torch.from_numpy(ubp).to(dev))

            def step_pcg():
                sol.reset_()
                sol.initialize_rollout_device_(dp[0].data_ptr(), dp[1].data_ptr())
                sol.solve_(sync=False)
            step_pcg()
            sol.timing_reset()
            el = timed(step_pcg, [sol])
            stp = sol.stats()
            secondary["pcg64"] = {"value": B * k2 / el, "unit": "trajectories/s", "ms_per_step": 1e3 * el / k2, "steps": k2,
                                  "solve_kernel_ms": sol.timing()[0], "iterations_max": float(stp["iterations"].max()),
                                  "inner_iterations_mean": float(stp["iterations"].mean()),
                                  "note": "the same step on the numpy-PCG64 inputs every figure of rounds 1-5 was measured on"}
            # back to the headline inputs: the last solve of this handle is what the statistics below describe
            sol.reset_(); sol.initialize_rollout_device_(d_x1.data_ptr(), d_u.data_ptr()); sol.solve_(sync=True)
        if rank == 0 and not stub and world == 1:
            # the batch's slowest instance ALONE on the chip, same kernel: what of the step is one instance's serial latency and
            # what is the sharing of SIMDs with the other instances of the batch (tools/batch_latency.py has the whole curve)
            st_ = sol.stats()
            # the instance that finished LAST in the batch (its stamps), else the one with the most iterations
            worst = int(np.argmax(finish["end_ms"])) if finish is not None else int(np.argmax(st_["iterations"]))
            lone = pkg.Solver(model=model, horizon=T, batch=1, device=gpu,
                              options=pkg.Options(verbose=0, **pkg.workloads.CONFIG_OPTIONS.get(args.config, {})))
            batch_kernel = sol.resolved_kernel_variant()         # "auto" on a batch of ONE would pick the latency kernel whatever the batch ran on
            lone.set_kernel_variant_(batch_kernel)
            lone.set_handover_(0)
            lone_ms = []
            for _ in range(3):
                lone.reset_(); lone.initialize_rollout_(x1[worst:worst + 1], ub[worst:worst + 1]); lone.timing_reset(); lone.solve_(sync=True)
                lone_ms.append(lone.timing()[0])
            secondary["slowest_instance_alone"] = {"kernel_ms": min(lone_ms[1:]), "instance": worst, "iterations": int(lone.stats()["iterations"][0]),
                                                   "batch_iterations_max": int(st_["iterations"].max()),
                                                   "chosen_by": "last to finish in the batch (device time stamps)" if finish is not None else "most iterations",
                                                   "lifetime_in_the_batch_ms": float(finish["life_ms"][worst]) if finish is not None else None,
                                                   "kernel_variant": batch_kernel,
                                                   "note": "one launch of the same kernel with only the batch's slowest instance on the GPU: "
                                                           "the serial latency no batch size can go below; the step's kernel time minus this is "
                                                           "what sharing the SIMDs with the rest of the batch costs that instance"}
            lone.close()
    per_rank = pkg.distributed.gather_over_ranks([1e3 * my_elapsed / args.steps, kernel_ms, it_sum, it_max], dist, cdev)
    if rank != 0:
        for s_ in sols:
            s_.close()
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    rank_ms = [r[0] for r in per_rank]
    n_gpus = ndev if sharded else world
    value = n_gpus * B * args.steps / elapsed
    # BASELINE.json's metric is quoted on acrobot T=101 at 1024 instances per GPU; any other --config / --batch names itself
    headline = args.config == "acrobot" and B == 1024
    out = {
        "metric": "trajectories/sec (whole node), acrobot T=101 batch=1024/GPU" if headline
                  else "trajectories/sec (whole node), %s T=%d batch=%d/GPU (NOT the BASELINE metric's configuration)" % (args.config, T, B),
        "value": value, "unit": "trajectories/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic (%s)" % args.generator,
        "config": {"workload": "%s (nx=%d, nu=%d, T=%d) AL-iLQR solve!, batch=%d per GPU, fp64, faithful reference semantics"
                               % (args.config, sol.nx, sol.nu, T, B),
                   "global_batch": n_gpus * B, "horizon": T,
                   "parallelism": "batch-shard x%d (no collective), %s" % (n_gpus, "distinct shards: rank r solves instances [r*%d, (r+1)*%d)" % (B, B)
                                                                           if args.distinct_shards else "same %d instances per GPU" % B),
                   "batches_in_flight": len(sols), "kernel_variant": args.variant,
                   "mode": ("shared_step: one step size per iteration for the global batch, all-reduce(sum) of 3 doubles per trial over %s, "
                            "host-stepped line search — changes the iterates, not comparable with the reference" % (backend or "one rank"))
                           if args.shared_step else "independent line search per instance (the reference's semantics)"},
        "ranks": {"world_size": world, "collective_backend": backend, "group_world_size": (dist.get_world_size() if dist is not None else 1),
                  "ms_per_step_per_rank": rank_ms, "slowest_rank": int(np.argmax(rank_ms)),
                  "solve_kernel_ms_per_rank": [r[1] for r in per_rank],
                  "launcher": ("one process, ilqr_create_sharded over %d device(s)" % ndev) if sharded
                              else os.environ.get("ILQR_BENCH_LAUNCHER", "external (torchrun)" if world > 1 else "single process"),
                  "shared_device_test_hook": share, "stub": stub},
        "solve_stats": {"inner_iterations_mean": float(st["iterations"].mean()),
                        "iterations_max": it_max,       # the kernel lasts as long as its slowest instance
                        "iterations_max_per_rank": [r[3] for r in per_rank],
                        "rollouts_mean": float(st["rollouts"].mean()),
                        "outer_iterations_mean": float(st["outer_iterations"].mean()),
                        "converged_frac": float((st["max_violation"] <= 5e-3).mean()),
                        "trajectory_iterations_per_s": float(sum(r[2] for r in per_rank) * args.steps / elapsed)},
    }
    if secondary:
        out["secondary"] = secondary
    if finish is not None:
        e_, l_ = finish["end_ms"], finish["life_ms"]
        per_it = 1e3 * l_ / np.maximum(st["iterations"], 1)
        out["solve_stats"]["finish_profile"] = {
            "finish_ms_percentiles": {"p50": float(np.percentile(e_, 50)), "p90": float(np.percentile(e_, 90)), "p99": float(np.percentile(e_, 99)), "max": float(e_.max())},
            "us_per_iteration_in_the_batch": {"p5": float(np.percentile(per_it, 5)), "p50": float(np.percentile(per_it, 50)), "p95": float(np.percentile(per_it, 95)), "max": float(per_it.max())},
            "last_to_finish": {"instance": int(np.argmax(e_)), "iterations": int(st["iterations"][int(np.argmax(e_))])},
            "note": "device time stamps of the last timed launch (tools/finish_times.py prints the whole profile and the wave placement)"}
    if ho_stats is not None:
        out["solve_stats"]["handover"] = {"marked_as_stragglers": ho_stats[1], "through_the_workgroups_queue": ho_stats[0]}
    if not stub:
        n_, m_ = sol.nx, sol.nu
        io_bytes = 8.0 * B * (n_ + (T - 1) * m_ + T * n_ + (T - 1) * m_ + (T - 1) * (m_ * n_ + m_))     # per device
        C = (T - 1) * sol.nc_stage + sol.nc_term
        abytes = float(algorithmic_bytes(sol.nx, sol.nu, T, C, st["iterations"].astype(np.float64),
                                         st["rollouts"].astype(np.float64)).sum()) / ndev     # (per device: one handle over ndev of them)
        achieved = abytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "basis": "ALGORITHMIC bytes of SURVEY §8(d) (stage-materialised model) / measured kernel time: a modelled figure, "
                         "the fused kernel keeps most of these bytes in LDS (see counter_frac and actual_bound)",
                "kernel": "solve_kernel<Model_%s>" % model, "kernel_ms_avg": kernel_ms, "launches": launches,
                "algorithmic_bytes_per_launch": abytes,
                # pure I/O of a fully fused solve (x1, ū in; x, u, K, k out): how far below the
                # stage-materialised model a resident solve sits
                "io_lower_bound_bytes_per_launch": io_bytes,
                "io_lower_bound_GBs": io_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0}
        if world == 1 and not args.no_pmc and not args.shared_step and not sharded:
            for s_ in sols:
                s_.synchronize()
            tr, why = measure_traffic(args)
            if tr is not None:
                roof["traffic"] = tr["traffic_bytes_per_launch"]
                roof["traffic_detail"] = tr
                roof["counter_GBs"] = tr["traffic_bytes_per_launch"] / (kernel_ms * 1e-3) / 1e9
                roof["counter_frac"] = roof["counter_GBs"] / HBM_PEAK_GBS
                roof["traffic_over_algorithmic"] = tr["traffic_bytes_per_launch"] / abytes
            else:
                roof["traffic_unmeasured_reason"] = why
        if sol.nx > 4 or sol.nu > 4:
            # large path: the Riccati step's tile products on v_mfma_f64_16x16x4_f64 (zero-padded 16x16 tiles), one backward pass per
            # inner iteration plus one per ilqr_solve! call; peak = MI355X fp64 matrix rate (78.6 TFLOP/s spec; the instruction issues
            # once per 71.5 clk per SIMD, tools/probes/probe_mfma16.hip: 70 TFLOP/s at 2.4 GHz)
            NPt, m4 = (sol.nx + 15) // 16, (sol.nu + 3) // 4
            n4 = (sol.nx + 3) // 4
            mfma_per_step = NPt * n4 + NPt * NPt * n4 + NPt * n4 + n4 + NPt * NPt * n4 + NPt * NPt * 4 * m4
            passes = float((st["iterations"] + st["outer_iterations"]).sum())
            flops = passes * (T - 1) * mfma_per_step * 2048.0
            # the flops the recursion needs (4n^3 + 12mn^2 + 6m^2n + m^3/3 + 2n^2 + 8mn per timestep), i.e. without
            # the zero padding of the 16x16 tiles
            n_l, m_l = float(sol.nx), float(sol.nu)
            useful = passes * (T - 1) * (4 * n_l ** 3 + 12 * m_l * n_l ** 2 + 6 * m_l ** 2 * n_l + m_l ** 3 / 3 + 2 * n_l ** 2 + 8 * m_l * n_l) / ndev
            flops /= ndev
            roof["mfma"] = {"achieved": flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0, "peak": 78.6, "unit": "TFLOP/s",
                            "frac": flops / (kernel_ms * 1e-3) / 1e12 / 78.6 if kernel_ms > 0 else 0.0,
                            "useful_achieved": useful / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0,
                            "useful_frac": useful / (kernel_ms * 1e-3) / 1e12 / 78.6 if kernel_ms > 0 else 0.0,
                            "mfma_per_riccati_step": mfma_per_step, "backward_passes": passes,
                            "note": "achieved / frac: v_mfma_f64_16x16x4_f64 flops INCLUDING the zero padding of the tiles (what the matrix pipe "
                                    "executes); useful_*: the recursion's own flops (SURVEY Appendix D); backward passes counted as inner "
                                    "iterations + one per ilqr_solve! call; the matrix pipe is shared by the two instances of a CU"}
            roof["actual_bound"] = ("per-timestep latency of the Riccati step: MFMA-tile windows (three for 17 <= nx <= 32, four otherwise) around the "
                                    "serial potrf / potrs chain — 6.7 k ticks per timestep with two instances per CU (tools/subphase_cycles.py), of which "
                                    "4.9 k are the matrix pipe's floor (136 MFMAs x 71.5 clk x 2 instances over 4 SIMDs); the launch lasts as long as "
                                    "its slowest instance")
        # instruction-issue model of the critical path: the per-step instruction lists of the serial loops, read off the assembly
        # the library's own compilation kept (csrc/Makefile -> lib/issue_model.json, stamped with the device-source hash)
        im, why_not = pkg._ffi.issue_model(model)
        if im is None:
            roof["issue_model_unavailable"] = why_not
        elif kernel_ms > 0 and args.variant in ("auto", "latency") and B <= 1024 and sol.nx <= 4 and sol.nu <= 4:
            rollouts_per_iter = float(st["rollouts"].sum()) / max(1.0, it_sum)
            # the two serial loops alone (the slower wave of the Riccati recursion sets its pace): nothing measured enters the floor —
            # the cost pass, the linearisation, copies and barriers of an iteration are what `other_ms` is left with
            ric_clk = max(im["riccati_step_occupancy_clk"], im.get("wave1_riccati_step_occupancy_clk", 0.0))
            per_iter_clk = (T - 1) * (im["rollout_step_occupancy_clk"] * rollouts_per_iter + ric_clk)
            slots_iter = (T - 1) * (im["rollout_step_instr"] * rollouts_per_iter + im["riccati_step_instr"])
            floor_ms = it_max * per_iter_clk / (im["clock_ghz"] * 1e6)
            chain_ms = it_max * (T - 1) * (im["rollout_step_chain"]["clk"] * rollouts_per_iter + im["riccati_step_chain"]["clk"]) / (im["clock_ghz"] * 1e6) \
                if "rollout_step_chain" in im else None
            roof["bound"] = "issue"
            roof["roofline_kind"] = "hbm (SURVEY §8(d): achieved / peak / frac are the modelled HBM figure the survey asks for)"
            roof["actual_bound"] = ("instruction issue of ONE wave: the slowest instance's critical wave issues one instruction per 5-6 clk "
                                    "whatever its class (tools/probes/probe_issue.hip), so its serial loops last as long as their "
                                    "instruction lists; predicted_floor_ms prices this build's lists at those rates")
            roof["issue_model"] = dict(im, iterations_max=it_max, rollouts_per_iteration=rollouts_per_iter,
                                       issue_slots_per_iteration=slots_iter,
                                       measured_clk_per_issue_slot=kernel_ms * im["clock_ghz"] * 1e6 / (it_max * slots_iter),
                                       predicted_floor_ms=floor_ms, achieved_over_floor=kernel_ms / floor_ms,
                                       other_ms=kernel_ms - floor_ms, dependent_chain_ms=chain_ms,
                                       floor_basis="issue time of the instruction lists of the two serial loops (rollout, Riccati) of the slowest "
                                                   "instance, single-wave rates; other_ms = everything else of its iterations (cost pass, linearisation, "
                                                   "copies, barriers, SIMD sharing); dependent_chain_ms = the same loops if only their register "
                                                   "dependences counted (rollout_step_chain, riccati_step_chain): what more issue bandwidth could reach")
        out["roofline"] = roof

        if world == 1:
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
            out["host_boundary"] = {"value": ndev * B / host_elapsed, "unit": "trajectories/s",
                                    "note": "one rank, inputs from host memory and x, u, K, k copied back (PCIe-inclusive)"}

        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle as O
            threads = host_cores()
            sample = min(B, 64 * threads)
            oopt = O.default_options(**pkg.workloads.CONFIG_OPTIONS.get(args.config, {}))   # the same solver options as the GPU leg
            c0 = time.perf_counter()
            O.solve_batch(model, T, x1[:sample], ub[:sample], options=oopt, nthreads=threads, want_policy=False)
            c1 = time.perf_counter() - c0
            one = min(sample, 48)
            c0 = time.perf_counter()
            O.solve_batch(model, T, x1[:one], ub[:one], options=oopt, nthreads=1, want_policy=False)
            c_one = time.perf_counter() - c0
            out["cpu_baseline"] = {"value": sample / c1, "unit": "trajectories/s", "cores": threads, "kind": "port",
                                   "single_thread_value": one / c_one, "sample_instances": sample, "single_thread_sample_instances": one,
                                   "wall_s": c1, "single_thread_wall_s": c_one,
                                   "sample": "first %d of the %d instances of this workload, C++ oracle "
                                             "(literal restatement of the Julia reference, which cannot run here), "
                                             "OpenMP over instances, %.1f s wall" % (sample, B, c1)}
    print(json.dumps(out), flush=True)
    for s_ in sols:
        s_.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main(argv=None):
    args = parse_args(argv)
    if args.pmc_child:
        return pmc_child(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1 and not args.sharded_handle:
        # launcher: N fresh ranks, started before this process has made any HIP call
        from ilqr_amd_loader import load_package
        pkg = load_package()
        os.environ["ILQR_BENCH_LAUNCHER"] = "bench.py --gpus %d (subprocess per rank)" % args.gpus
        try:
            rc, out0 = pkg.distributed.launch_ranks(os.path.abspath(__file__), sys.argv[1:] if argv is None else list(argv), args.gpus,
                                                    share_device=os.environ.get("ILQR_BENCH_SHARE_DEVICE") == "1",
                                                    stub=os.environ.get("ILQR_BENCH_STUB") == "1", timeout=3600)
        except RuntimeError as e:
            raise SystemExit("bench.py: %s" % e)
        sys.stdout.write(out0)
        sys.stdout.flush()
        if rc != 0:
            raise SystemExit("bench.py: a rank exited with code %d" % rc)
        return
    worker(args)


if __name__ == "__main__":
    main()
