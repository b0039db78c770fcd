# make_julia_fixtures.jl — the route to a REFERENCE-PRODUCED pin of the oracle (SURVEY.md §8(c)).
#
# NOT RUN in this repository's image (no Julia). For anyone who has Julia >= 1.6 and the reference package:
#
#     python tests/golden/dump_fixture_inputs.py                 # inputs of the 15 cases -> tests/golden/julia_in/
#     julia --project=/path/to/IterativeLQR.jl tests/golden/make_julia_fixtures.jl [case ...]
#     python tests/golden/julia_to_npz.py                        # tests/golden/julia_out/ -> tests/golden/ref_<case>_julia.npz
#
# tests/refdata.py prefers ref_<case>_julia.npz over ref_<case>.npz, so every fixture test (oracle and HIP path) then
# compares against numbers the reference itself computed.
#
# How the numbers are taken: the REAL `solve!` of the package runs; nothing of src/solve.jl is restated here. The script only
# wraps five package functions — ilqr_solve!, gradients!(problem), backward_pass!, forward_pass!, lagrangian_gradient!
# (plus rollout! and cost! for counting) — with methods that record and then call the ORIGINAL method through
# `Base.invoke_in_world(W0, ...)` (the world age captured before the wrappers were defined). The hook points are the ones of
# tests/golden/reference_restatement.py (before gradients!, after backward_pass!, after forward_pass!, per inner iteration),
# the output schema is the one of tests/golden/make_reference_fixtures.py.
#
# Models: the reference's own example problems (examples/particle.jl, test/car.jl, test/acrobot.jl; constants in
# SURVEY.md Appendix B), the synthetic nx = 32 model of SURVEY.md §8(d) C5, and (round 6) the two problems of SURVEY §8 f3 —
# car_tv (distinct objects per step) and ragged (distinct DIMENSIONS per step) — written against the package's constructors.
# The f3 cases (ragged_T9, ragged_T41, car_tv_T21: whole solves, no stage snapshots) are read by
# tests/test_oracle_ragged.py::test_oracle_against_reference_produced_f3_fixtures, which is skipped while no *_julia.npz exists.
using IterativeLQR
using LinearAlgebra

const ILQR = IterativeLQR
const HERE = @__DIR__
const IN_DIR = joinpath(HERE, "julia_in")
const OUT_DIR = joinpath(HERE, "julia_out")

# ------------------------------------------------------------------------------------------------ models
# (user functions take (x, u) when num_parameter == 0, as the reference's constructors call them: src/dynamics.jl:22)
function midpoint(fc, h)
    return (x, u) -> x + h * fc(x + 0.5 * h * fc(x, u), u)
end

function particle_problem(T)
    f = (x, u) -> [x[1] + x[2]; x[2] + u[1]]
    dyn = [Dynamics(f, 2, 1) for t = 1:T-1]
    ot = (x, u) -> 0.1 * dot(x, x) + 0.1 * dot(u, u)
    oT = (x, u) -> 0.1 * dot(x, x)
    costs = [[Cost(ot, 2, 1) for t = 1:T-1]..., Cost(oT, 2, 0)]
    xT = [1.0; 0.0]
    cons = [[Constraint() for t = 1:T-1]..., Constraint((x, u) -> x - xT, 2, 0)]
    return dyn, costs, cons
end

function car_problem(T; goal_only = false)
    fc = (x, u) -> [u[1] * cos(x[3]); u[1] * sin(x[3]); u[2]]
    dyn = [Dynamics(midpoint(fc, 0.1), 3, 2) for t = 1:T-1]
    xT = [1.0; 1.0; 0.0]
    ot = (x, u) -> 1.0 * dot(x - xT, x - xT) + 1.0e-2 * dot(u, u)
    oT = (x, u) -> 1000.0 * dot(x - xT, x - xT)
    costs = [[Cost(ot, 3, 2) for t = 1:T-1]..., Cost(oT, 3, 0)]
    ul = -5.0 * ones(2); uu = 5.0 * ones(2)
    p_obs = [0.5; 0.5]; r_obs = 0.1
    obs = x -> r_obs^2.0 - dot(x[1:2] - p_obs, x[1:2] - p_obs)
    if goal_only
        cons = [[Constraint() for t = 1:T-1]..., Constraint((x, u) -> x - xT, 3, 0)]
    else
        stage = (x, u) -> [ul - u; u - uu; obs(x)]
        term = (x, u) -> [x - xT; obs(x)]
        cons = [[Constraint(stage, 3, 2, indices_inequality = collect(1:5)) for t = 1:T-1]...,
                Constraint(term, 3, 0, indices_inequality = collect(4:4))]
    end
    return dyn, costs, cons
end

function acrobot_continuous(x, u)
    mass1 = 1.0; inertia1 = 0.33; length1 = 1.0; lengthcom1 = 0.5
    mass2 = 1.0; inertia2 = 0.33; length2 = 1.0; lengthcom2 = 0.5
    gravity = 9.81; friction1 = 0.1; friction2 = 0.1
    q = x[1:2]; v = x[3:4]
    a = inertia1 + inertia2 + mass2 * length1 * length1 + 2.0 * mass2 * length1 * lengthcom2 * cos(q[2])
    b = inertia2 + mass2 * length1 * lengthcom2 * cos(q[2])
    c = inertia2
    det = a * c - b * b
    Minv = [c / det  -b / det; -b / det  a / det]
    tau = [-1.0 * mass1 * gravity * lengthcom1 * sin(q[1]) - mass2 * gravity * (length1 * sin(q[1]) + lengthcom2 * sin(q[1] + q[2]));
           -1.0 * mass2 * gravity * lengthcom2 * sin(q[1] + q[2])]
    Cm = [-2.0 * mass2 * length1 * lengthcom2 * sin(q[2]) * v[2]  -1.0 * mass2 * length1 * lengthcom2 * sin(q[2]) * v[2];
          mass2 * length1 * lengthcom2 * sin(q[2]) * v[1]          0.0]
    Bm = [0.0; 1.0]
    qdd = Minv * (-1.0 * Cm * v + tau + Bm * u[1] - [friction1; friction2] .* v)
    return [x[3]; x[4]; qdd[1]; qdd[2]]
end

function acrobot_problem(T)
    dyn = [Dynamics(midpoint(acrobot_continuous, 0.1), 4, 1) for t = 1:T-1]
    ot = (x, u) -> 0.1 * dot(x[3:4], x[3:4]) + 0.1 * dot(u, u)
    oT = (x, u) -> 0.1 * dot(x[3:4], x[3:4])
    costs = [[Cost(ot, 4, 1) for t = 1:T-1]..., Cost(oT, 4, 0)]
    xT = [pi; 0.0; 0.0; 0.0]
    cons = [[Constraint() for t = 1:T-1]..., Constraint((x, u) -> x - xT, 4, 0)]
    return dyn, costs, cons
end

function synth32_problem(T)
    n = 32; m = 8; h = 0.05
    A = [(i == j ? -1.0 : 0.0) + 0.3 * cos(Float64(i + 2 * j)) / 32.0 for i = 1:n, j = 1:n]
    B = [sin(Float64(3 * i + j)) / sqrt(32.0) for i = 1:n, j = 1:m]
    f = (x, u) -> x + h * (A * x + B * u + 0.1 * sin.(x))
    dyn = [Dynamics(f, n, m) for t = 1:T-1]
    xg = 0.5 * ones(n)
    ot = (x, u) -> 0.1 * dot(x - xg, x - xg) + 0.01 * dot(u, u)
    oT = (x, u) -> 10.0 * dot(x - xg, x - xg)
    costs = [[Cost(ot, n, m) for t = 1:T-1]..., Cost(oT, n, 0)]
    stage = (x, u) -> [-1.0 .- u; u .- 1.0]
    cons = [[Constraint(stage, n, m, indices_inequality = collect(1:2m)) for t = 1:T-1]..., Constraint()]
    return dyn, costs, cons
end

# ---- SURVEY §8 f3: objects and DIMENSIONS that differ from step to step (README.md:26, src/dynamics.jl:5-7). Twins of
# iterativelqr.jl_amd/models.py: car_tv / ragged, oracle/models.cpp "car_tv" / "ragged", reference_restatement.py: ragged_problem.
function car_tv_problem(T)
    fc = (x, u) -> [u[1] * cos(x[3]); u[1] * sin(x[3]); u[2]]
    base_dyn, base_costs, base_cons = car_problem(T)
    dyn_a = base_dyn[1]
    dyn_b = Dynamics((x, u) -> x + 0.05 * fc(x, u), 3, 2)                   # explicit Euler, h = 0.05
    cost_a = base_costs[1]
    q = [5.0; 2.0; 0.5]; xg = [0.9; 1.1; 0.2]; r = [0.05; 0.02]
    cost_b = Cost((x, u) -> sum(q[i] * (x[i] - xg[i])^2 for i = 1:3) + sum(r[j] * u[j]^2 for j = 1:2), 3, 2)
    con_a = base_cons[1]
    con_eq = Constraint((x, u) -> [u[2] - 0.3 * x[3] - 0.05], 3, 2)
    dyn = [((t - 1) % 3 == 2 ? dyn_b : dyn_a) for t = 1:T-1]
    costs = [[(2 * (t - 1) >= T - 1 ? cost_b : cost_a) for t = 1:T-1]..., base_costs[end]]
    cons = [[((t - 1) % 4 == 0 ? con_a : ((t - 1) % 4 == 2 ? con_eq : Constraint())) for t = 1:T-1]..., base_cons[end]]
    return dyn, costs, cons
end

const RAGGED_N = [3, 3, 4, 4, 2, 2, 3, 3]
const RAGGED_M = [2, 1, 2, 1, 1, 2, 2, 1]
function ragged_problem(T)
    n_t = [RAGGED_N[(t - 1) % 8 + 1] for t = 1:T]
    m_t = [RAGGED_M[(t - 1) % 8 + 1] for t = 1:T-1]
    dcache = Dict{Tuple{Int,Int,Int},Any}(); ccache = Dict{Tuple{Int,Int},Any}()
    function dynk(n0, m0, n1)
        get!(dcache, (n0, m0, n1)) do
            A = [(i == j ? 0.9 : 0.0) + 0.1 * cos(1.0 + (i - 1) + 2 * (j - 1) + n0) for i = 1:n1, j = 1:n0]
            Bm = [0.3 * sin(2.0 + 3 * (i - 1) + (j - 1) + m0) for i = 1:n1, j = 1:m0]
            Dynamics((x, u) -> [sum(A[i, j] * x[j] for j = 1:n0) + sum(Bm[i, j] * u[j] for j = 1:m0) + (i == 1 ? 0.1 * sin(x[1]) : 0.0)
                                for i = 1:n1], n0, m0)
        end
    end
    function costk(n0, m0)
        get!(ccache, (n0, m0)) do
            Cost((x, u) -> 0.5 * sum((1.0 + 0.1 * (i - 1)) * x[i] * x[i] for i = 1:n0) + 0.05 * sum(Float64(j) * u[j] * u[j] for j = 1:m0), n0, m0)
        end
    end
    dyn = [dynk(n_t[t], m_t[t], n_t[t+1]) for t = 1:T-1]
    nT = n_t[end]
    costs = [[costk(n_t[t], m_t[t]) for t = 1:T-1]..., Cost((x, u) -> 5.0 * sum(x[i] * x[i] for i = 1:nT), nT, 0)]
    cons = [[Constraint() for t = 1:T-1]..., Constraint((x, u) -> [x[1] - 0.2; x[2] + 0.1], nT, 0)]
    return dyn, costs, cons
end

const PROBLEMS = Dict("particle" => particle_problem, "car" => car_problem, "car_goal" => T -> car_problem(T; goal_only = true),
                      "acrobot" => acrobot_problem, "synth32" => synth32_problem, "car_tv" => car_tv_problem, "ragged" => ragged_problem)

# ------------------------------------------------------------------------------------------------ recording
mutable struct Recorder
    solver::Any
    points::Vector{Tuple{Int,Int}}
    outer::Int
    inner::Int
    rollouts::Int
    in_forward::Bool
    trial_objectives::Vector{Float64}
    pending::Dict{Tuple{Int,Int},Int}
    trace::Vector{Vector{Float64}}
    out::Dict{String,Array{Float64}}
end
const REC = Ref{Union{Nothing,Recorder}}(nothing)

# per-step dimensions: vectors / matrices of a problem whose num_state, num_action vary are zero-padded to the largest (the layout
# of the device template and of the oracle's getters); for uniform dimensions these are stackv / stackm
padv(vs, n) = isempty(vs) ? zeros(0) : hcat([vcat(Float64.(v), zeros(n - length(v))) for v in vs]...)
padm(ms, r, c) = isempty(ms) ? zeros(0) : cat([[Float64.(m) zeros(size(m, 1), c - size(m, 2)); zeros(r - size(m, 1), c)] for m in ms]...; dims = 3)
stackm(ms) = isempty(ms) ? zeros(0) : cat([Float64.(m) for m in ms]...; dims = 3)      # (rows, cols, T)
stackv(vs) = isempty(vs) ? zeros(0) : (all(isempty, vs) ? zeros(0) : hcat([Float64.(v) for v in vs]...))   # (n, T)
catv(vs) = isempty(vs) ? zeros(0) : vcat([Float64.(v) for v in vs]...)

al_costs(s) = s.problem.objective.costs                      # AugmentedLagrangianCosts (every fixture case is constrained)

function snapshot_pre!(r::Recorder, j::Int)
    s = r.solver; p = "s$(j)_pre_"; pr = s.problem; al = al_costs(s)
    r.out[p * "nominal_states"] = stackv(pr.nominal_states)
    r.out[p * "nominal_actions"] = stackv(pr.nominal_actions[1:end-1])
    r.out[p * "states"] = stackv(pr.states)
    r.out[p * "actions"] = stackv(pr.actions[1:end-1])
    r.out[p * "gxx"] = stackm(pr.objective.hessian_state_state)
    r.out[p * "guu"] = stackm(pr.objective.hessian_action_action)
    r.out[p * "gux"] = stackm(pr.objective.hessian_action_state)
    r.out[p * "violations"] = catv(al.constraint_data.violations)
    r.out[p * "dual"] = catv(al.constraint_dual)
    r.out[p * "penalty"] = catv(al.constraint_penalty)
    r.out[p * "active_set"] = catv(al.active_set)
    d = s.data
    r.out[p * "scalars"] = [d.objective[1], d.max_violation[1], d.step_size[1], d.status[1] ? 1.0 : 0.0]
end

function snapshot_backward!(r::Recorder, j::Int)
    s = r.solver; p = "s$(j)_"; pr = s.problem; po = s.policy; av = po.action_value
    r.out[p * "fx"] = stackm(pr.model.jacobian_state); r.out[p * "fu"] = stackm(pr.model.jacobian_action)
    r.out[p * "gx"] = stackv(pr.objective.gradient_state); r.out[p * "gu"] = stackv(pr.objective.gradient_action)
    r.out[p * "gxx"] = stackm(pr.objective.hessian_state_state); r.out[p * "guu"] = stackm(pr.objective.hessian_action_action)
    r.out[p * "gux"] = stackm(pr.objective.hessian_action_state)
    r.out[p * "Qx"] = stackv(av.gradient_state); r.out[p * "Qu"] = stackv(av.gradient_action)
    r.out[p * "Qxx"] = stackm(av.hessian_state_state); r.out[p * "Quu"] = stackm(av.hessian_action_action)
    r.out[p * "Qux"] = stackm(av.hessian_action_state)
    r.out[p * "K"] = stackm(po.K); r.out[p * "k"] = stackv(po.k)
    r.out[p * "P"] = stackm(po.value.hessian); r.out[p * "p"] = stackv(po.value.gradient)
    # Lagrangian gradient of THIS linearisation: Lx = Qx - p, Lu = Qu (what lagrangian_gradient! writes, src/solve.jl:67-83),
    # formed here without touching data.gradient
    H = length(pr.states)
    r.out[p * "Lx"] = stackv([av.gradient_state[t] - po.value.gradient[t] for t = 1:H-1])
    r.out[p * "Lu"] = stackv([copy(av.gradient_action[t]) for t = 1:H-1])
end

function snapshot_forward!(r::Recorder, j::Int)
    s = r.solver; p = "s$(j)_fwd_"; pr = s.problem; d = s.data; al = al_costs(s)
    r.out[p * "delta"] = [dot(d.gradient, pr.trajectory)]                  # src/forward_pass.jl:20 (both operands untouched since)
    r.out[p * "trial_objectives"] = copy(r.trial_objectives)
    r.out[p * "scalars"] = [d.objective[1], d.max_violation[1], d.step_size[1], d.status[1] ? 1.0 : 0.0]
    r.out[p * "nominal_states"] = stackv(pr.nominal_states)
    r.out[p * "nominal_actions"] = stackv(pr.nominal_actions[1:end-1])
    r.out[p * "states"] = stackv(pr.states)
    r.out[p * "actions"] = stackv(pr.actions[1:end-1])
    r.out[p * "violations"] = catv(al.constraint_data.violations)
    r.out[p * "active_set"] = catv(al.active_set)
    r.out[p * "trajectory"] = copy(pr.trajectory)
end

point_index(r::Recorder, o, i) = findfirst(==((o, i)), r.points)

# ------------------------------------------------------------------------------------------------ the wrappers
# World age BEFORE the wrappers exist: invoke_in_world(W0, f, ...) runs the package's own method.
const W0 = Base.get_world_counter()

@eval IterativeLQR begin
    function ilqr_solve!(solver::Solver; iteration = true)
        r = Main.REC[]
        if r !== nothing
            r.outer += 1; r.inner = 0
        end
        Base.invoke_in_world(Main.W0, ilqr_solve!, solver; iteration = iteration)
    end
    function gradients!(problem::ProblemData; mode = :nominal)
        r = Main.REC[]
        if r !== nothing
            j = Main.point_index(r, r.outer, r.inner)
            j === nothing || Main.snapshot_pre!(r, j - 1)
        end
        Base.invoke_in_world(Main.W0, gradients!, problem; mode = mode)
    end
    function backward_pass!(policy::PolicyData, problem::ProblemData; mode = :nominal)
        Base.invoke_in_world(Main.W0, backward_pass!, policy, problem; mode = mode)
        r = Main.REC[]
        if r !== nothing
            j = Main.point_index(r, r.outer, r.inner)
            if j !== nothing
                Main.snapshot_backward!(r, j - 1)
                r.pending[(r.outer, r.inner + 1)] = j - 1
            end
        end
        return nothing
    end
    function rollout!(policy::PolicyData, problem::ProblemData; step_size = 1.0)
        r = Main.REC[]
        r === nothing || (r.rollouts += 1)
        Base.invoke_in_world(Main.W0, rollout!, policy, problem; step_size = step_size)
    end
    function cost!(data::SolverData, problem::ProblemData; mode = :nominal)
        res = Base.invoke_in_world(Main.W0, cost!, data, problem; mode = mode)
        r = Main.REC[]
        (r !== nothing && r.in_forward && mode == :current) && push!(r.trial_objectives, data.objective[1])
        return res
    end
    function forward_pass!(policy::PolicyData, problem::ProblemData, data::SolverData; kwargs...)
        r = Main.REC[]
        if r !== nothing
            r.in_forward = true; empty!(r.trial_objectives)
        end
        Base.invoke_in_world(Main.W0, forward_pass!, policy, problem, data; kwargs...)
        if r !== nothing
            r.in_forward = false
            r.inner += 1
            j = pop!(r.pending, (r.outer, r.inner), nothing)
            j === nothing || Main.snapshot_forward!(r, j)
        end
        return nothing
    end
    function lagrangian_gradient!(data::SolverData, policy::PolicyData, problem::ProblemData)
        Base.invoke_in_world(Main.W0, lagrangian_gradient!, data, policy, problem)
        r = Main.REC[]
        if r !== nothing && !r.in_forward
            # the call after backward_pass! inside the iteration loop (src/solve.jl:32): one trace row per inner iteration —
            # outer, inner, objective, gradient_norm, max_violation, step_size, status, rollouts so far (src/solve.jl:36-45)
            push!(r.trace, [Float64(r.outer), Float64(r.inner), data.objective[1], norm(data.gradient, Inf), data.max_violation[1],
                            data.step_size[1], data.status[1] ? 1.0 : 0.0, Float64(r.rollouts)])
        end
        return nothing
    end
end

# ------------------------------------------------------------------------------------------------ driver
function read_f64(path, dims...)
    a = Array{Float64}(undef, dims...)
    read!(path, a)
    return a
end

function write_outputs(case, out)
    dir = joinpath(OUT_DIR, case)
    mkpath(dir)
    open(joinpath(dir, "manifest.txt"), "w") do man
        for (key, arr) in out
            a = Float64.(arr)
            open(joinpath(dir, key * ".f64"), "w") do io
                write(io, vec(a))
            end
            println(man, key, " ", join(size(a), " "))          # Julia (column-major) dimensions
        end
    end
end

function generate(case)
    lines = readlines(joinpath(IN_DIR, case * ".txt"))
    model = strip(lines[1]); T = parse(Int, lines[2])
    npts = parse(Int, lines[3])
    points = Tuple{Int,Int}[]
    for l in lines[4:3+npts]
        a, b = split(l)
        push!(points, (parse(Int, a), parse(Int, b)))
    end
    dyn, costs, cons = PROBLEMS[model](T)
    # (dimensions may vary per step: the inputs on disk are padded to the largest, src/dynamics.jl:5-7)
    ns = [[d.num_state for d in dyn]..., dyn[end].num_next_state]; ms_ = [d.num_action for d in dyn]
    n = maximum(ns); m = maximum(ms_)
    x1 = vec(read_f64(joinpath(IN_DIR, case * ".x1.f64"), n))[1:ns[1]]
    ub = read_f64(joinpath(IN_DIR, case * ".u.f64"), m, T - 1)              # row-major [T-1][m] on disk = (m, T-1) here
    ubar = [ub[1:ms_[t], t] for t = 1:T-1]
    xbar = rollout(dyn, x1, ubar)
    solver = Solver(dyn, costs, cons, options = Options{Float64}(verbose = false))
    initialize_controls!(solver, ubar)
    initialize_states!(solver, xbar)
    r = Recorder(solver, points, 0, 0, 0, false, Float64[], Dict{Tuple{Int,Int},Int}(), Vector{Float64}[], Dict{String,Array{Float64}}())
    REC[] = r
    t0 = time()
    solve!(solver)
    dt = time() - t0
    REC[] = nothing
    out = r.out
    # points the solve never reached are dropped and the rest renumbered (as make_reference_fixtures.py does)
    taken = sort([j for j = 0:length(points)-1 if haskey(out, "s$(j)_fx") && haskey(out, "s$(j)_fwd_delta")])
    renum = Dict(j => i - 1 for (i, j) in enumerate(taken))
    final = Dict{String,Array{Float64}}()
    for (key, val) in out
        j = parse(Int, key[2:findfirst('_', key)-1])
        haskey(renum, j) && (final["s$(renum[j])" * key[findfirst('_', key):end]] = val)
    end
    final["points"] = isempty(taken) ? zeros(0) : Float64.(hcat([[points[j+1][1], points[j+1][2]] for j in taken]...))   # (2, npoints)
    final["x1"] = vcat(x1, zeros(n - length(x1))); final["ubar"] = padv(ubar, m); final["xbar"] = padv(xbar, n); final["horizon"] = [Float64(T)]
    final["state_dims"] = Float64.(ns); final["action_dims"] = Float64.(ms_)
    final["trace"] = isempty(r.trace) ? zeros(0) : hcat(r.trace...)                                       # (8, n_iter)
    xs, us = get_trajectory(solver)
    final["x"] = padv(xs, n); final["u"] = padv(us, m)
    if model == "synth32" && T > 11
        sel = [0, 1, div(T, 2), T - 2]
        final["K_steps"] = Float64.(sel)
        final["K"] = stackm([solver.policy.K[t+1] for t in sel])
    else
        final["K"] = padm(solver.policy.K, m, n)
    end
    final["k"] = padv(solver.policy.k, m)
    d = solver.data
    # potrf_info: the reference ignores LAPACK's info (src/backward_pass.jl:69); a factorisation that failed would have thrown
    # from LAPACK.potrs! in Julia, so a solve that ends here had info == 0 throughout
    final["stats"] = [d.objective[1], norm(d.gradient, Inf), d.max_violation[1], d.step_size[1], Float64(d.iterations[1]),
                      Float64(r.outer), d.status[1] ? 1.0 : 0.0, Float64(r.rollouts), 0.0]
    write_outputs(case, final)
    println(rpad(case, 16), " iterations ", d.iterations[1], "  outer ", r.outer, "  rollouts ", r.rollouts, "  J ", d.objective[1],
            "  max_violation ", d.max_violation[1], "  (", round(dt, digits = 1), " s)")
end

cases = isempty(ARGS) ? sort([splitext(f)[1] for f in readdir(IN_DIR) if endswith(f, ".txt")]) : ARGS
for c in cases
    generate(c)
end
