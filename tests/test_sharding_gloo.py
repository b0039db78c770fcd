"""N > 1 path on CPU, world size 2 over gloo: bench.py's REAL per-rank `worker` — shard_range, make_inputs(offset), barriers,
MAX over ranks, all-gather of the per-rank figures, the JSON line — with the GPU solver replaced by a CPU solve of the rank's
own shard through the oracle (test infrastructure). The union of what the two ranks solved must equal a single-process solve
of the whole batch, instance for instance."""
import json
import os
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIG, B_PER_RANK, WORLD = "acrobot51", 3, 2


class OracleSolver:
    """Solver-shaped stand-in for bench.worker: solves (x1, ū) of ITS rank with the CPU oracle and leaves the result on disk."""

    def __init__(self, rank, model, T, B, x1, ub):
        from oracle import oracle as O
        self.O, self.rank, self.model, self.T, self.B, self.x1, self.ub = O, rank, model, T, B, x1, ub
        self.nx, self.nu, self.nc_stage, self.nc_term = x1.shape[1], ub.shape[2], 0, 4
        self.res, self.ms, self.launches = None, 0.0, 0

    def reset_(self): pass
    def initialize_rollout_device_(self, a, b): pass
    def set_kernel_variant_(self, v): pass
    def synchronize(self): pass
    def timing_reset(self): self.ms, self.launches = 0.0, 0

    def solve_(self, sync=True):
        import time
        t0 = time.perf_counter()
        self.res = self.O.solve_batch(self.model, self.T, self.x1, self.ub, nthreads=1)
        self.ms += 1e3 * (time.perf_counter() - t0)
        self.launches += 1

    def timing(self): return self.ms / max(1, self.launches), self.launches

    def stats(self):
        s = self.res["stats"]
        return {k: np.asarray(s[k]) for k in ("iterations", "rollouts", "outer_iterations", "max_violation")}

    def close(self):
        np.savez(os.path.join(os.environ["ILQR_TEST_DUMP_DIR"], "rank%d.npz" % self.rank), x=self.res["x"], u=self.res["u"],
                 iterations=self.res["stats"]["iterations"], x1=self.x1, ub=self.ub)


def _worker(rank, world, port, dump, q, extra=()):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      ILQR_TEST_DUMP_DIR=dump)
    import contextlib
    import io
    import bench
    # no flag: for N > 1 distinct shards ARE the default (BASELINE config 4's meaning of sharding)
    args = bench.parse_args(["--gpus", str(world), "--steps", "1", "--warmup", "0", "--batch", str(B_PER_RANK), "--config", CONFIG] + list(extra))
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.worker(args, solver_factory=OracleSolver)
    q.put((rank, buf.getvalue()))


def test_world_size_2_worker_solves_its_shard_and_the_union_is_the_whole_batch(tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, WORLD, port, str(tmp_path), q)) for r in range(WORLD)]
    for p in procs: p.start()
    outs = dict(q.get(timeout=300) for _ in range(WORLD))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    # rank 0 printed the ONE json line, with both ranks' figures gathered
    lines = [ln for ln in outs[0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and not [ln for ln in outs[1].splitlines() if ln.startswith("{")]
    out = json.loads(lines[0])
    assert out["n_gpus"] == WORLD and out["config"]["global_batch"] == WORLD * B_PER_RANK and "distinct shards" in out["config"]["parallelism"]
    r = out["ranks"]
    assert r["world_size"] == WORLD and r["group_world_size"] == WORLD and r["collective_backend"] == "gloo"
    assert len(r["ms_per_step_per_rank"]) == WORLD and out["ms_per_step"] >= max(r["ms_per_step_per_rank"]) - 1e-9       # MAX over ranks
    assert abs(out["value"] - WORLD * B_PER_RANK / (out["ms_per_step"] * 1e-3)) < 1e-6 * out["value"]                    # whole-job units / max time
    # what the ranks solved: contiguous, disjoint shards whose union is the global batch, each instance solved as in one process
    sys.path.insert(0, ROOT)
    from ilqr_amd_loader import load_package
    from oracle import oracle as O
    pkg = load_package()
    model, T, x1, ub = pkg.workloads.make_inputs(CONFIG, WORLD * B_PER_RANK, generator="splitmix64")      # bench.py's default inputs
    parts = [np.load(tmp_path / ("rank%d.npz" % rk)) for rk in range(WORLD)]
    assert np.array_equal(np.concatenate([p["x1"] for p in parts]), x1) and np.array_equal(np.concatenate([p["ub"] for p in parts]), ub)
    whole = O.solve_batch(model, T, x1, ub, nthreads=2)
    assert np.array_equal(np.concatenate([p["iterations"] for p in parts]), whole["stats"]["iterations"])
    assert np.array_equal(np.concatenate([p["x"] for p in parts]), whole["x"])
    assert np.array_equal(np.concatenate([p["u"] for p in parts]), whole["u"])
    assert out["solve_stats"]["iterations_max_per_rank"] == [float(p["iterations"].max()) for p in parts]


def test_same_instances_flag_puts_the_first_shard_on_every_rank(tmp_path):
    """--same-instances: the round-4 default, now opt-in — every rank solves instances [0, B) (fixed per-GPU work)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, WORLD, port, str(tmp_path), q, ("--same-instances",))) for r in range(WORLD)]
    for p in procs: p.start()
    outs = dict(q.get(timeout=300) for _ in range(WORLD))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    out = json.loads([ln for ln in outs[0].splitlines() if ln.startswith("{")][0])
    assert "same %d instances per GPU" % B_PER_RANK in out["config"]["parallelism"]
    sys.path.insert(0, ROOT)
    from ilqr_amd_loader import load_package
    pkg = load_package()
    model, T, x1, ub = pkg.workloads.make_inputs(CONFIG, B_PER_RANK, generator="splitmix64")
    for rk in range(WORLD):
        part = np.load(tmp_path / ("rank%d.npz" % rk))
        assert np.array_equal(part["x1"], x1) and np.array_equal(part["ub"], ub)
