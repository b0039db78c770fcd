"""N>1 path on CPU: world-size-2 gloo run of the batch sharding + timing aggregation used by bench.py."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from ilqr_amd_loader import load_package
    pkg = load_package()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    r, lr, w = pkg.distributed.rank_info()
    B = 6
    lo, hi = pkg.distributed.shard_range(r, B)
    model, T, x1, ub = pkg.workloads.make_inputs("acrobot51", B, offset=lo)
    dist.barrier()
    tmax = pkg.distributed.max_over_ranks(0.5 + r, dist)
    tot = pkg.distributed.sum_over_ranks(float(B), dist)
    q.put((r, lo, hi, ub.copy(), tmax, tot))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    from ilqr_amd_loader import load_package
    pkg = load_package()
    _, _, _, full = pkg.workloads.make_inputs("acrobot51", 12)
    assert (res[0][1], res[0][2], res[1][1], res[1][2]) == (0, 6, 6, 12)          # contiguous, disjoint
    assert np.array_equal(np.concatenate([res[0][3], res[1][3]]), full)             # union == global batch
    assert res[0][4] == res[1][4] == 1.5                                            # MAX over ranks
    assert res[0][5] == res[1][5] == 12.0
