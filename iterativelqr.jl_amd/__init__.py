"""MI355X-native batched iLQR: host-side mirror of IterativeLQR.jl's API over a
C-ABI (include/ilqr_hip.h) into hand-written HIP kernels (csrc/)."""
from . import _ffi, codegen, distributed, lowering, models, workloads  # noqa: F401
from .api import (Options, Solver, compile_model, get_trajectory, initialize_controls_,  # noqa: F401
                  initialize_states_, solve_)
from .codegen import Constraint, Cost, Dynamics  # noqa: F401
