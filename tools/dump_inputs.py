"""Write the synthetic inputs of a bench workload as raw little-endian float64 files, so that another
implementation (bench/julia_ref.jl with the real IterativeLQR.jl package) can be timed on exactly the
instances bench.py solves.

    python tools/dump_inputs.py acrobot 1024 /tmp/acrobot_inputs [splitmix64 | pcg64]      (default: bench.py's default, splitmix64)
      -> /tmp/acrobot_inputs.x1.f64  [B][nx]      /tmp/acrobot_inputs.u.f64  [B][T-1][nu]
         /tmp/acrobot_inputs.json    {"model", "T", "B", "nx", "nu"}
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ilqr_amd_loader import load_package  # noqa: E402


def main():
    config, batch, prefix = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    generator = sys.argv[4] if len(sys.argv) > 4 else "splitmix64"
    model, T, x1, ub = load_package().workloads.make_inputs(config, batch, generator=generator)
    x1.astype("<f8").tofile(prefix + ".x1.f64")
    ub.astype("<f8").tofile(prefix + ".u.f64")
    json.dump({"model": model, "T": T, "B": batch, "nx": x1.shape[1], "nu": ub.shape[2], "generator": generator}, open(prefix + ".json", "w"))


if __name__ == "__main__":
    main()
