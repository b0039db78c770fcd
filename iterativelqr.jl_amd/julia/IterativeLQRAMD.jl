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

export Options, Solver, initialize_controls!, initialize_states!, initialize_rollout!, set_parameters!,
       solve!, get_trajectory, get_policy, stats

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
    Solver(model; horizon, batch, constrained=true, options=Options(), device=0, model_library="")

Batched counterpart of `Solver(dynamics, costs, constraints)` (src/solver.jl:28-46).
`model` names a built-in ("acrobot", "car", "particle", ...) or a model compiled by
the code generator (iterativelqr.jl_amd/codegen.py) whose module is `model_library`.
"""
function Solver(model::AbstractString; horizon::Integer, batch::Integer, constrained::Bool=true,
                options::Options=Options(), device::Integer=0, model_library::AbstractString="")
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve model model_library begin
        desc = ProblemDesc(Base.unsafe_convert(Cstring, model),
                           isempty(model_library) ? Cstring(C_NULL) : Base.unsafe_convert(Cstring, model_library),
                           Int32(horizon), Int32(batch), Int32(device), Int32(constrained))
        check(ccall((:ilqr_create, LIB[]), Cint, (Ref{ProblemDesc}, Ref{Ptr{Cvoid}}), desc, h))
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

end # module
