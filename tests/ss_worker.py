"""One rank of the shared-step test (started by tests/test_gpu_parity.py through distributed.launch_ranks): solves its
shard of a car batch with Solver.solve_shared_step_ over a gloo group, both ranks on device 0, and writes the accepted
step sizes and its trajectories to <out>.<rank>.npz."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ilqr_amd_loader import load_package  # noqa: E402

out, B_total = sys.argv[1], int(sys.argv[2])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
import torch.distributed as dist  # noqa: E402
dist.init_process_group("gloo")
pkg = load_package()
B = B_total // world
lo, hi = pkg.distributed.shard_range(rank, B)
model, T, x1, ub = pkg.workloads.make_inputs("car", B_total)
sol = pkg.Solver(model=model, horizon=T, batch=B, device=0, options=pkg.Options(verbose=0))
sol.initialize_rollout_(x1[lo:hi], ub[lo:hi])
steps = sol.solve_shared_step_(pkg.distributed.torch_allreduce_sum(dist))
x, u = sol.get_trajectory()
st = sol.stats()
np.savez(out + ".%d.npz" % rank, steps=np.array(steps), x=x, u=u, iterations=st["iterations"], max_violation=st["max_violation"])
sol.close()
dist.barrier()
dist.destroy_process_group()
