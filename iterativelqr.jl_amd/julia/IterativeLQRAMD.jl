# IterativeLQRAMD.jl — Julia host side of the MI355X-native batched iLQR solver.
#
# NOT EXECUTED in this repository's CI: the build image has no Julia (SURVEY.md §0).
# It is the binding a maintainer of IterativeLQR.jl would add: the same exported
# names (src/IterativeLQR.jl:30-45) over `ccall`s into libilqr_hip.so
# (include/ilqr_hip.h). The tested twin of this file is iterativelqr.jl_amd/api.py.
#
# One `Solver` here owns a BATCH of B independent problem instances of one model;
# states are B×T×nx, actions B×(T-1)×nu (row-major on the C side, so Julia arrays
# are passed as (nx, T, B) / (nu, T-1, B) column-major views of the same memory).
module IterativeLQRAMD

export Options, Solver, Dynamics, Cost, Constraint, initialize_controls!, initialize_states!, initialize_rollout!,
       set_parameters!, solve!, get_trajectory, get_policy, stats, set_kernel_variant!, set_handover!, set_handover_live!, enable_trace!, trace

const LIB = Ref{String}(joinpath(@__DIR__, "..", "lib", "libilqr_hip.so"))

# Options{T} — src/options.jl:1-15, field for field (C layout of ilqr_options)
Base.@kwdef mutable struct Options
    line_search::Int32 = 1                      # 1 = :armijo, 0 = :none
    max_iterations::Int32 = 100
    max_dual_updates::Int32 = 10
    min_step_size::Float64 = 1.0e-5
    objective_tolerance::Float64 = 1.0e-3
    lagrangian_gradient_tolerance::Float64 = 1.0e-3
    constraint_tolerance::Float64 = 5.0e-3
    constraint_norm::Float64 = Inf
    initial_constraint_penalty::Float64 = 1.0
    scaling_penalty::Float64 = 10.0
    max_penalty::Float64 = 1.0e8
    reset_cache::Int32 = 0
    verbose::Int32 = 0
end

struct ProblemDesc
    model::Cstring
    model_library::Cstring
    horizon::Int32
    batch::Int32
    device::Int32
    constrained::Int32
end

struct Stats
    objective::Float64
    gradient_norm::Float64
    max_violation::Float64
    step_size::Float64
    iterations::Int32
    outer_iterations::Int32
    status::Int32
    potrf_info::Int32
    rollouts::Int32
    reserved::Int32
end

function check(rc::Cint)
    rc == 0 && return
    msg = unsafe_string(ccall((:ilqr_last_error, LIB[]), Cstring, ()))
    error("ilqr error $rc: $msg")
end

mutable struct Solver
    handle::Ptr{Cvoid}
    nx::Int; nu::Int; T::Int; B::Int
    options::Options
end

"""
    Solver(model; horizon, batch, constrained=true, options=Options(), device=0, devices=Int[], model_library="")

Batched counterpart of `Solver(dynamics, costs, constraints)` (src/solver.jl:28-46).
`model` names a built-in ("acrobot", "car", "particle", ...) or a model compiled by
the code generator (iterativelqr.jl_amd/codegen.py) whose module is `model_library`.
`devices = [0, 1, ..., 7]` spreads the batch over the GPUs of the node (contiguous ranges of
ceil(batch / length(devices)) instances, `ilqr_create_sharded`); every other call is unchanged.
"""
function Solver(model::AbstractString; horizon::Integer, batch::Integer, constrained::Bool=true,
                options::Options=Options(), device::Integer=0, devices::AbstractVector{<:Integer}=Int[],
                model_library::AbstractString="")
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve model model_library begin
        desc = ProblemDesc(Base.unsafe_convert(Cstring, model),
                           isempty(model_library) ? Cstring(C_NULL) : Base.unsafe_convert(Cstring, model_library),
                           Int32(horizon), Int32(batch), Int32(device), Int32(constrained))
        if isempty(devices)
            check(ccall((:ilqr_create, LIB[]), Cint, (Ref{ProblemDesc}, Ref{Ptr{Cvoid}}), desc, h))
        else
            devs = Int32.(collect(devices))
            check(ccall((:ilqr_create_sharded, LIB[]), Cint, (Ref{ProblemDesc}, Ptr{Int32}, Int32, Ref{Ptr{Cvoid}}),
                        desc, devs, Int32(length(devs)), h))
        end
    end
    d = [Ref{Int32}(0) for _ in 1:7]
    check(ccall((:ilqr_get_dims, LIB[]), Cint,
                (Ptr{Cvoid}, Ref{Int32}, Ref{Int32}, Ref{Int32}, Ref{Int32}, Ref{Int32}, Ref{Int32}, Ref{Int32}),
                h[], d...))
    s = Solver(h[], d[1][], d[2][], d[6][], d[7][], options)
    finalizer(x -> ccall((:ilqr_destroy, LIB[]), Cint, (Ptr{Cvoid},), x.handle), s)
    return s
end

# initialize_controls!(solver, ū) — src/solver.jl:56-60; ū :: Array{Float64,3} of size (nu, T-1, B)
function initialize_controls!(s::Solver, u::Array{Float64,3})
    @assert size(u) == (s.nu, s.T - 1, s.B)
    check(ccall((:ilqr_initialize_controls, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Float64}), s.handle, u))
end

# initialize_states!(solver, x̄) — src/solver.jl:62-66; x̄ :: (nx, T, B)
function initialize_states!(s::Solver, x::Array{Float64,3})
    @assert size(x) == (s.nx, s.T, s.B)
    check(ccall((:ilqr_initialize_states, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Float64}), s.handle, x))
end

# x̄ = rollout(dynamics, x1, ū) on the device (src/rollout.jl:33-42) + both initialisers
function initialize_rollout!(s::Solver, x1::Matrix{Float64}, u::Array{Float64,3})
    @assert size(x1) == (s.nx, s.B) && size(u) == (s.nu, s.T - 1, s.B)
    check(ccall((:ilqr_initialize_rollout, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), s.handle, x1, u))
end

# Solver(...; parameters = θ) — src/solver.jl:12,29; θ :: (nw, T, B)
function set_parameters!(s::Solver, w::Array{Float64,3})
    check(ccall((:ilqr_set_parameters, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Float64}), s.handle, w))
end

# solve!(solver[, states, actions]) — src/solve.jl:137-143, 131-135
function solve!(s::Solver, args...)
    if length(args) == 2
        initialize_controls!(s, args[2]); initialize_states!(s, args[1])
    end
    check(ccall((:ilqr_set_options, LIB[]), Cint, (Ptr{Cvoid}, Ref{Options}), s.handle, s.options))
    check(ccall((:ilqr_solve, LIB[]), Cint, (Ptr{Cvoid},), s.handle))
    check(ccall((:ilqr_synchronize, LIB[]), Cint, (Ptr{Cvoid},), s.handle))
    return nothing
end

# get_trajectory(solver) — src/solver.jl:48-50
function get_trajectory(s::Solver)
    x = Array{Float64,3}(undef, s.nx, s.T, s.B); u = Array{Float64,3}(undef, s.nu, s.T - 1, s.B)
    check(ccall((:ilqr_get_trajectory, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), s.handle, x, u))
    return x, u
end

# solver.policy.K / .k — K[:, :, t, b] is the nu×nx gain (column-major, as in the reference)
function get_policy(s::Solver)
    K = Array{Float64,4}(undef, s.nu, s.nx, s.T - 1, s.B); k = Array{Float64,3}(undef, s.nu, s.T - 1, s.B)
    check(ccall((:ilqr_get_policy, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), s.handle, K, k))
    return K, k
end

# solver.data.* per instance
function stats(s::Solver)
    st = Vector{Stats}(undef, s.B)
    check(ccall((:ilqr_get_stats, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Stats}), s.handle, st))
    return st
end

# ------------------------------------------------------------------------------------------------------------
# Dynamics / Cost / Constraint with the reference's constructors (src/dynamics.jl:16-34, src/costs.jl:17-44,
# src/constraints.jl:17-43): the user function is traced on Symbolics variables exactly as the reference does, but
# instead of `eval(build_function(...)[2])` (a Julia closure) the expressions are emitted as C
# (`Symbolics.build_function(expr, x, u, w; target = Symbolics.CTarget())` — from memory of the Symbolics docs; the
# image has no Julia, so this half has never run) and handed to `ilqr_compile_model`, which wraps them for the kernels
# (csrc/ilqr_model_adapter.hpp) and compiles them with hipcc. Requires `using Symbolics` in the caller's environment.
struct Dynamics;  body::String; num_state::Int; num_action::Int; num_parameter::Int; end
struct Cost;      body::String; num_state::Int; num_action::Int; num_parameter::Int; end
struct Constraint; body::String; num_constraint::Int; num_state::Int; num_action::Int; num_parameter::Int
                   indices_inequality::Vector{Int}; end
Constraint() = Constraint("", 0, 0, 0, 0, Int[])

# one `ILQR_MODEL_FN void name(double* out, const double* x, const double* u, const double* w)` per expression array
function c_function(Symbolics, name::String, exprs, x, u, w)
    src = Symbolics.build_function(exprs, x, u, w; target = Symbolics.CTarget(), fname = name * "_raw",
                                   lhsname = :out, rhsnames = [:x, :u, :w])
    return string(src, "\nILQR_MODEL_FN void ", name,
                  "(double* out, const double* x, const double* u, const double* w) { ", name, "_raw(out, x, u, w); }\n")
end

function Dynamics(Symbolics, f::Function, num_state::Int, num_action::Int; num_parameter::Int = 0)
    x = Symbolics.variables(:x, 1:num_state); u = Symbolics.variables(:u, 1:num_action); w = Symbolics.variables(:w, 1:num_parameter)
    y = num_parameter > 0 ? f(x, u, w) : f(x, u)
    body = c_function(Symbolics, "dynamics", y, x, u, w) *
           c_function(Symbolics, "dynamics_jacobian_state", vec(Symbolics.jacobian(y, x)), x, u, w) *
           c_function(Symbolics, "dynamics_jacobian_action", vec(Symbolics.jacobian(y, u)), x, u, w)
    Dynamics(body, num_state, num_action, num_parameter)
end

function Cost(Symbolics, f::Function, num_state::Int, num_action::Int; num_parameter::Int = 0, terminal::Bool = num_action == 0)
    x = Symbolics.variables(:x, 1:num_state); u = Symbolics.variables(:u, 1:num_action); w = Symbolics.variables(:w, 1:num_parameter)
    l = num_parameter > 0 ? f(x, u, w) : f(x, u)
    gx = Symbolics.gradient(l, x); gu = Symbolics.gradient(l, u)
    p = terminal ? "cost_terminal" : "cost_stage"
    body = c_function(Symbolics, p, [l], x, u, w) * c_function(Symbolics, p * "_gradient_state", gx, x, u, w) *
           c_function(Symbolics, p * "_hessian_state_state", vec(Symbolics.jacobian(gx, x)), x, u, w)
    if !terminal
        body *= c_function(Symbolics, p * "_gradient_action", gu, x, u, w) *
                c_function(Symbolics, p * "_hessian_action_action", vec(Symbolics.jacobian(gu, u)), x, u, w) *
                c_function(Symbolics, p * "_hessian_action_state", vec(Symbolics.jacobian(gu, x)), x, u, w)
    end
    Cost(body, num_state, num_action, num_parameter)
end

function Constraint(Symbolics, f::Function, num_state::Int, num_action::Int; indices_inequality::Vector{Int} = Int[],
                    num_parameter::Int = 0, terminal::Bool = num_action == 0)
    x = Symbolics.variables(:x, 1:num_state); u = Symbolics.variables(:u, 1:num_action); w = Symbolics.variables(:w, 1:num_parameter)
    c = num_parameter > 0 ? f(x, u, w) : f(x, u)
    p = terminal ? "constraint_terminal" : "constraint_stage"
    body = c_function(Symbolics, p, c, x, u, w) * c_function(Symbolics, p * "_jacobian_state", vec(Symbolics.jacobian(c, x)), x, u, w)
    terminal || (body *= c_function(Symbolics, p * "_jacobian_action", vec(Symbolics.jacobian(c, u)), x, u, w))
    Constraint(body, length(c), num_state, num_action, num_parameter, indices_inequality)
end

# The reference's own signatures — Dynamics(f, nx, nu; num_parameter), Cost(f, nx, nu; num_parameter),
# Constraint(f, nx, nu; indices_inequality, num_parameter) (src/dynamics.jl:16, src/costs.jl:17, src/constraints.jl:17) —
# with Symbolics taken from the session (`using Symbolics` before the first call, as a user of the reference has anyway)
function symbolics_module()
    isdefined(Main, :Symbolics) || error("IterativeLQRAMD: `using Symbolics` first (the constructors trace the user function symbolically, as the reference does)")
    return getfield(Main, :Symbolics)
end
Dynamics(f::Function, num_state::Int, num_action::Int; kwargs...) = Dynamics(symbolics_module(), f, num_state, num_action; kwargs...)
Cost(f::Function, num_state::Int, num_action::Int; kwargs...) = Cost(symbolics_module(), f, num_state, num_action; kwargs...)
Constraint(f::Function, num_state::Int, num_action::Int; kwargs...) = Constraint(symbolics_module(), f, num_state, num_action; kwargs...)

struct ModelSource
    name::Cstring; nx::Int32; nu::Int32; nw::Int32; nc_stage::Int32; nc_term::Int32
    ineq_stage::UInt64; ineq_term::UInt64; source::Cstring
end
ineq_mask(idx) = reduce(|, (UInt64(1) << (i - 1) for i in idx if i <= 64); init = UInt64(0))
# rows beyond 64 (ilqr_compile_model_rows): row i = bit (i - 1) % 64 of word (i - 1) ÷ 64, at least one word
function ineq_words(idx, num_constraint)
    w = zeros(UInt64, max(1, cld(num_constraint, 64)))
    for i in idx
        w[(i - 1) ÷ 64 + 1] |= UInt64(1) << ((i - 1) % 64)
    end
    return w
end

"""
    Solver(dynamics, costs, constraints; batch, options, name)

`Solver(dynamics, costs, constraints)` of the reference (src/solver.jl:28-46) for a batch: `dynamics[1]`, `costs[1]`,
`constraints[1]` are the stage objects (uniform over the horizon in this wrapper; distinct per-step objects are lowered by
the Python host, lowering.py), `costs[end]` / `constraints[end]` the terminal ones.
"""
function Solver(dynamics::Vector{Dynamics}, costs::Vector{Cost}, constraints::Vector{Constraint};
                batch::Integer, options::Options = Options(), name::AbstractString = "user", device::Integer = 0,
                devices::AbstractVector{<:Integer} = Int[], constrained::Bool = true)
    # The device kernels are compiled for ONE stage template. Distinct per-step objects (README.md:26 of the reference) are
    # lowered onto it by the Python host only (lowering.py); here they are refused instead of silently solving with [1].
    length(costs) == length(dynamics) + 1 && length(constraints) == length(costs) ||
        error("Solver: expected T-1 dynamics, T costs and T constraints (src/solver.jl:28-46)")
    all(x -> x.body == dynamics[1].body, dynamics) ||
        error("Solver: per-step Dynamics objects differ; this wrapper takes one stage template (use the Python host's lowering)")
    all(x -> x.body == costs[1].body, costs[1:end-1]) ||
        error("Solver: per-step stage Cost objects differ; this wrapper takes one stage template (use the Python host's lowering)")
    all(x -> x.body == constraints[1].body && x.indices_inequality == constraints[1].indices_inequality, constraints[1:end-1]) ||
        error("Solver: per-step stage Constraint objects differ; this wrapper takes one stage template (use the Python host's lowering)")
    d, cs, ct = dynamics[1], constraints[1], constraints[end]
    source = d.body * costs[1].body * costs[end].body * cs.body * ct.body
    regname = Vector{UInt8}(undef, 128); path = Vector{UInt8}(undef, 1024)
    GC.@preserve name source begin
        ms = ModelSource(Base.unsafe_convert(Cstring, name), d.num_state, d.num_action, d.num_parameter,
                         cs.num_constraint, ct.num_constraint, ineq_mask(cs.indices_inequality), ineq_mask(ct.indices_inequality),
                         Base.unsafe_convert(Cstring, source))
        ws, wt = ineq_words(cs.indices_inequality, cs.num_constraint), ineq_words(ct.indices_inequality, ct.num_constraint)
        check(ccall((:ilqr_compile_model_rows, LIB[]), Cint,
                    (Ref{ModelSource}, Ptr{UInt64}, Ptr{UInt64}, Ptr{UInt8}, Csize_t, Ptr{UInt8}, Csize_t),
                    ms, ws, wt, regname, length(regname), path, length(path)))
    end
    Solver(unsafe_string(pointer(regname)); horizon = length(costs), batch = batch, constrained = constrained, options = options,
           device = device, devices = devices, model_library = unsafe_string(pointer(path)))
end

# Solver(dynamics, costs) — src/solver.jl:11-26: no constraints, plain iLQR
function Solver(dynamics::Vector{Dynamics}, costs::Vector{Cost}; kwargs...)
    T = length(costs)
    return Solver(dynamics, costs, [Constraint() for _ in 1:T]; constrained = false, kwargs...)
end

# solve!(solver; augmented_lagrangian_callback! = cb) — src/solve.jl:88,125: the outer AL loop stepped from the host, one launch
# per outer iteration (ILQR_STAGE_AL_BEGIN = 7, ILQR_STAGE_AL_OUTER = 8), cb(solver) after every dual update
function solve!(s::Solver, augmented_lagrangian_callback!::Function)
    check(ccall((:ilqr_set_options, LIB[]), Cint, (Ptr{Cvoid}, Ref{Options}), s.handle, s.options))
    check(ccall((:ilqr_run_stage, LIB[]), Cint, (Ptr{Cvoid}, Int32), s.handle, 7))
    done = ccall((:ilqr_scalar_slot, LIB[]), Cint, (Cstring,), "done")
    nsc = ccall((:ilqr_scalar_slot, LIB[]), Cint, (Cstring,), "count")
    sc = Matrix{Float64}(undef, nsc, s.B)
    for _ in 1:s.options.max_dual_updates
        check(ccall((:ilqr_run_stage, LIB[]), Cint, (Ptr{Cvoid}, Int32), s.handle, 8))
        check(ccall((:ilqr_get_buffer, LIB[]), Cint, (Ptr{Cvoid}, Cstring, Ptr{Float64}), s.handle, "_scalars", sc))
        all(sc[done + 1, :] .!= 0.0) && break
        augmented_lagrangian_callback!(s)
    end
    return nothing
end

# 0 = auto, 1 = latency, 2 = throughput, 3 = packed (four instances per wave, no horizon limit)
set_kernel_variant!(s::Solver, v::Integer) = check(ccall((:ilqr_set_kernel_variant, LIB[]), Cint, (Ptr{Cvoid}, Int32), s.handle, v))
# straggler hand-over of the packed kernel (include/ilqr_hip.h): outer = -1 by head count (default), 0 off, k >= 2 by outer iteration;
# live = survivors of the batch at which they all leave (-1 auto)
set_handover!(s::Solver, outer::Integer) = check(ccall((:ilqr_set_handover, LIB[]), Cint, (Ptr{Cvoid}, Int32), s.handle, outer))
set_handover_live!(s::Solver, live::Integer) = check(ccall((:ilqr_set_handover_live, LIB[]), Cint, (Ptr{Cvoid}, Int32), s.handle, live))

# what `verbose` prints per inner iteration (src/solve.jl:40-45), recorded on the device: rows of
# (outer, inner, objective, gradient_norm, max_violation, step_size, status, rollouts) per instance
enable_trace!(s::Solver, capacity::Integer) = check(ccall((:ilqr_enable_trace, LIB[]), Cint, (Ptr{Cvoid}, Int32), s.handle, capacity))
function trace(s::Solver, capacity::Integer)
    out = Array{Float64,3}(undef, 8, capacity, s.B)
    check(ccall((:ilqr_get_trace, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Float64}), s.handle, out))
    return out
end

end # module
