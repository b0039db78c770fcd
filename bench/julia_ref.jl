# Times the REAL reference (thowell/IterativeLQR.jl, installed in the active Julia environment) on the
# instances bench.py solves, so that the `cpu_baseline` of bench.py (a C++ restatement, kind "port") can be
# replaced by a measured reference number by anyone who has Julia. NOT run in this build's containers
# (no Julia there); nothing in tests/, smoke() or bench.py depends on it.
#
#   python tools/dump_inputs.py acrobot 1024 /tmp/acrobot_inputs        # bench.py's default inputs: ilqr_synthetic_inputs (SURVEY 8(d))
#   (or without Python: ccall((:ilqr_synthetic_inputs, "libilqr_hip.so"), Cint, (Cstring, Int32, UInt64, Int64, Int32, Ptr{Float64}, Ptr{Float64}),
#    "acrobot", 101, 20240607, 0, 1024, x1, ubar) fills the same two arrays — no device needed)
#   julia -t auto bench/julia_ref.jl /tmp/acrobot_inputs [n_instances]
#
# Prints one JSON line: trajectories/s over the sampled instances (one Solver per thread, instances split
# over threads — the reference has no batching of its own) plus per-instance iterations for comparison
# with `solve_stats` of bench.py.
using IterativeLQR
using LinearAlgebra
using Printf

prefix = ARGS[1]
meta = read(prefix * ".json", String)
getint(key) = parse(Int, match(Regex("\"$key\": *([0-9]+)"), meta).captures[1])
T, B, nx, nu = getint("T"), getint("B"), getint("nx"), getint("nu")
nsample = length(ARGS) > 1 ? min(parse(Int, ARGS[2]), B) : B
x1_all = reshape(reinterpret(Float64, read(prefix * ".x1.f64")), nx, B)
u_all = reshape(reinterpret(Float64, read(prefix * ".u.f64")), nu, T - 1, B)

# acrobot of SURVEY Appendix B (constants as data; midpoint rule, h = 0.1)
function qdd(q, v, τ)
    c2, s1, s2, s12 = cos(q[2]), sin(q[1]), sin(q[2]), sin(q[1] + q[2])
    a = 0.33 + 0.33 + 1.0 + 2.0 * 0.5 * c2
    b = 0.33 + 0.5 * c2
    c = 0.33
    g1 = -9.81 * 0.5 * s1 - 9.81 * (s1 + 0.5 * s12)
    g2 = -9.81 * 0.5 * s12
    r1 = -(-2.0 * 0.5 * s2 * v[2] * v[1] - 0.5 * s2 * v[2] * v[2]) + g1 - 0.1 * v[1]
    r2 = -(0.5 * s2 * v[1] * v[1]) + g2 + τ - 0.1 * v[2]
    d = a * c - b * b
    return [(c * r1 - b * r2) / d, (-b * r1 + a * r2) / d]
end
fc(x, u) = vcat(x[3:4], qdd(x[1:2], x[3:4], u[1]))
step(x, u) = x + 0.1 * fc(x + 0.05 * fc(x, u), u)

function make_solver()
    dyn = IterativeLQR.Dynamics(step, nx, nu)
    stage = IterativeLQR.Cost((x, u) -> 0.1 * dot(x[3:4], x[3:4]) + 0.1 * dot(u, u), nx, nu)
    term = IterativeLQR.Cost((x, u) -> 0.1 * dot(x[3:4], x[3:4]), nx, 0)
    free = IterativeLQR.Constraint()
    goal = IterativeLQR.Constraint((x, u) -> x - [π, 0.0, 0.0, 0.0], nx, 0)
    return IterativeLQR.Solver([dyn for t = 1:T-1], [[stage for t = 1:T-1]..., term],
                               [[free for t = 1:T-1]..., goal];
                               options = IterativeLQR.Options(verbose = false))
end

function solve_one!(s, b)
    ū = [u_all[:, t, b] for t = 1:T-1]
    x̄ = IterativeLQR.rollout(s.problem.model.dynamics, x1_all[:, b], ū)
    IterativeLQR.initialize_controls!(s, ū)
    IterativeLQR.initialize_states!(s, x̄)
    IterativeLQR.solve!(s)
    return s.data.iterations[1]
end

nt = Threads.nthreads()
solvers = [make_solver() for _ = 1:nt]           # Symbolics code generation happens here, untimed
# a fresh Solver per instance would be the literal bench.py semantics (all buffers zero); building one costs
# seconds of Symbolics work, so the per-thread solver is reused — iteration counts can differ slightly from
# bench.py's because `problem.states` (where the first violations are evaluated, SURVEY A.Q2) then holds the
# previous instance's last trial instead of zeros.
for s in solvers
    solve_one!(s, 1)                               # warm-up / JIT
end
iters = zeros(Int, nsample)
elapsed = @elapsed Threads.@threads for b = 1:nsample
    iters[b] = solve_one!(solvers[Threads.threadid()], b)
end
@printf("{\"metric\": \"trajectories/s\", \"value\": %.3f, \"threads\": %d, \"instances\": %d, \"kind\": \"reference\", \"inner_iterations_mean\": %.2f}\n",
        nsample / elapsed, nt, nsample, sum(iters) / nsample)
