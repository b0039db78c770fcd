"""`bench.py --gpus N` launches N ranks itself (one process per GPU) — exercised on CPU with gloo and a stub
solve (ILQR_BENCH_STUB=1): same launcher, rendezvous, barrier, MAX-over-ranks and per-rank gather code as on GPUs."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env)
    return subprocess.run([sys.executable, BENCH] + args, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)


def test_gpus_2_launches_two_ranks_over_gloo():
    p = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "8"], ILQR_BENCH_STUB="1")
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # rank 0 prints ONE json line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 16
    r = out["ranks"]
    assert r["world_size"] == 2 and r["group_world_size"] == 2 and r["collective_backend"] == "gloo"
    assert "bench.py --gpus 2" in r["launcher"]
    assert len(r["ms_per_step_per_rank"]) == 2 and r["slowest_rank"] == 1          # the stub's rank 1 sleeps twice as long
    assert out["ms_per_step"] >= max(r["ms_per_step_per_rank"]) - 1e-9             # MAX over ranks
    assert r["solve_kernel_ms_per_rank"] == [2.0, 4.0]
    assert abs(out["value"] - 16 * 3 / (out["ms_per_step"] * 3e-3)) < 1e-6 * out["value"]   # whole-job units / max time
    assert out["solve_stats"]["iterations_max_per_rank"] == [10.0, 11.0]


def test_gpus_n_refuses_when_fewer_devices_are_visible():
    """No GPU in this container: asking for 2 must fail loudly, not report a 1-GPU number as n_gpus = 2."""
    import torch
    if torch.cuda.device_count() >= 2:
        return
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert p.returncode != 0
    assert "GPU(s) are visible" in p.stderr.decode()


def test_world_size_mismatch_is_an_error():
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], ILQR_BENCH_STUB="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr.decode()


def test_driver_style_torchrun_launch():
    """The driver's own launch line for N > 1: python -m torch.distributed.run ... bench.py --gpus N (stub solve, gloo)."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e["ILQR_BENCH_STUB"] = "1"
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8"],
                       env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks"]["group_world_size"] == 2 and "torchrun" in out["ranks"]["launcher"]


def test_a_rank_that_dies_at_start_up_ends_the_job_at_once(tmp_path):
    """launch_ranks polls every child: rank 1 exiting non-zero must not leave rank 0 waiting in a barrier until a timeout."""
    import time
    from ilqr_amd_loader import load_package
    pkg = load_package()
    script = tmp_path / "ranks.py"
    script.write_text("import os, sys, time\nif os.environ['RANK'] == '1':\n    sys.exit(3)\nprint('rank0 up', flush=True)\ntime.sleep(120)\n")
    t0 = time.time()
    rc, out = pkg.distributed.launch_ranks(str(script), [], 2, stub=True, timeout=100)
    assert rc == 3 and time.time() - t0 < 20
    assert "rank0 up" in out or out == ""


def test_sharded_handle_mode_is_one_process_over_n_devices():
    """`bench.py --gpus N --sharded-handle`: ONE process whose solver handle spans N devices (ilqr_create_sharded — the path a Julia
    host holding one Solver takes, /root/reference/src/solver.jl:28-46), same JSON line. Stub solve on CPU: the launcher must not
    start ranks, n_gpus and the whole-job value must count all N ranges."""
    p = _run(["--gpus", "2", "--sharded-handle", "--steps", "3", "--warmup", "1", "--batch", "8"], ILQR_BENCH_STUB="1")
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 16
    assert out["ranks"]["world_size"] == 1 and "ilqr_create_sharded over 2 device(s)" in out["ranks"]["launcher"]
    assert abs(out["value"] - 16 * 3 / (out["ms_per_step"] * 3e-3)) < 1e-6 * out["value"]
    assert "NOT the BASELINE metric" in out["metric"]              # batch 8 is not the configuration the metric is quoted on


def test_sharded_handle_mode_refuses_without_enough_devices():
    import torch
    if torch.cuda.device_count() >= 2:
        return
    p = _run(["--gpus", "2", "--sharded-handle", "--steps", "1", "--warmup", "0"])
    assert p.returncode != 0
    err = p.stderr.decode()
    assert "needs a GPU" in err or "device(s) visible" in err
