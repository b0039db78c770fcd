# IterativeLQRAMD.jl — Julia host side of the MI355X-native batched iLQR solver.
#
# THIS FILE HAS NEVER RUN. The build image has no Julia (SURVEY.md §0), so nothing below has been parsed, let alone
# executed: the `ccall` signatures are written against include/ilqr_hip.h by hand, and the Symbolics calls
# (`build_function(...; target = CTarget(), fname, lhsname, rhsnames)`) from memory of its documentation. Treat it as the
# sketch of the binding a maintainer of IterativeLQR.jl would add — the same exported names (src/IterativeLQR.jl:30-45) over
# `ccall`s into libilqr_hip.so — not as a tested component. What IS tested is the C-ABI it binds (tests/test_abi.py, the C
# programs under examples/) and the Python twin of this file, iterativelqr.jl_amd/api.py, which drives every entry point
# used here (including ilqr_compile_model_stages / ilqr_set_stage_selectors, through Solver(stage_sources = ...)).
#
# One `Solver` here owns a BATCH of B independent problem instances of one model;
# states are B×T×nx, actions B×(T-1)×nu (row-major on the C side, so Julia arrays
# are passed as (nx, T, B) / (nu, T-1, B) column-major views of the same memory).
module IterativeLQRAMD

export Options, Solver, Dynamics, Cost, Constraint, initialize_controls!, initialize_states!, initialize_rollout!,
       set_parameters!, solve!, solve_shared_step!, get_trajectory, get_policy, stats, set_kernel_variant!, set_handover!, set_handover_live!, set_handover_mark!, enable_trace!, trace

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
    state_dims::Vector{Int}; action_dims::Vector{Int}     # real entries per step (all nx / nu unless the problem's dimensions vary)
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
    s = Solver(h[], d[1][], d[2][], d[6][], d[7][], options, fill(Int(d[1][]), d[6][]), fill(Int(d[2][]), d[6][] - 1))
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
# `body` is C source whose function names start with the placeholder @P@ (e.g. @P@_jacobian_state): Solver(...) gives every
# distinct object of a problem its own prefix — dynamics_<k>, cost_stage_<k>, constraint_stage_<k>, cost_terminal,
# constraint_terminal — as ilqr_compile_model_stages expects them (include/ilqr_hip.h).
struct Dynamics;  body::String; num_next_state::Int; num_state::Int; num_action::Int; num_parameter::Int; end
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
    # (name carries the @P@ placeholder; it is substituted in the whole body, raw function included, before compilation)
end

function Dynamics(Symbolics, f::Function, num_state::Int, num_action::Int; num_parameter::Int = 0)
    x = Symbolics.variables(:x, 1:num_state); u = Symbolics.variables(:u, 1:num_action); w = Symbolics.variables(:w, 1:num_parameter)
    y = num_parameter > 0 ? f(x, u, w) : f(x, u)
    body = c_function(Symbolics, "@P@", y, x, u, w) *
           c_function(Symbolics, "@P@_jacobian_state", vec(Symbolics.jacobian(y, x)), x, u, w) *
           c_function(Symbolics, "@P@_jacobian_action", vec(Symbolics.jacobian(y, u)), x, u, w)
    Dynamics(body, length(y), num_state, num_action, num_parameter)      # num_next_state = length(f(x, u)), src/dynamics.jl:22-29
end

function Cost(Symbolics, f::Function, num_state::Int, num_action::Int; num_parameter::Int = 0, terminal::Bool = num_action == 0)
    x = Symbolics.variables(:x, 1:num_state); u = Symbolics.variables(:u, 1:num_action); w = Symbolics.variables(:w, 1:num_parameter)
    l = num_parameter > 0 ? f(x, u, w) : f(x, u)
    gx = Symbolics.gradient(l, x); gu = Symbolics.gradient(l, u)
    p = "@P@"
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
    p = "@P@"
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

# ilqr_stage_kinds / ilqr_stage_plan (include/ilqr_hip.h), field for field
struct StageKinds
    horizon::Int32; num_parameter::Int32
    n_dynamics::Int32; dynamics_nx::Ptr{Int32}; dynamics_nu::Ptr{Int32}; dynamics_nx_next::Ptr{Int32}; dynamics_of_step::Ptr{Int32}
    n_costs::Int32; cost_nx::Ptr{Int32}; cost_nu::Ptr{Int32}; cost_of_step::Ptr{Int32}
    n_constraints::Int32; constraint_nc::Ptr{Int32}; constraint_nx::Ptr{Int32}; constraint_nu::Ptr{Int32}
    constraint_ineq::Ptr{UInt64}; constraint_of_step::Ptr{Int32}
    nx_term::Int32; nc_term::Int32; ineq_term::NTuple{4,UInt64}
end
struct StagePlan
    nx::Int32; nu::Int32; nw::Int32; nc_stage::Int32; nc_term::Int32; n_selectors::Int32
    sel_dynamics::Int32; sel_cost::Int32; sel_constraint::Int32
    constraint_row0::NTuple{16,Int32}; ineq_stage_words::NTuple{4,UInt64}
end
# inequality rows as four 64-bit words: row i (1-based, as indices_inequality is) = bit (i - 1) % 64 of word (i - 1) ÷ 64
function ineq_words(idx)
    w = zeros(UInt64, 4)
    for i in idx
        w[(i - 1) ÷ 64 + 1] |= UInt64(1) << ((i - 1) % 64)
    end
    return w
end

# distinct objects of a vector in order of first appearance, and the (0-based) kind of every element
function kinds_of(objs)
    kinds = eltype(objs)[]; index = Int32[]
    for o in objs
        k = findfirst(q -> q == o, kinds)
        if k === nothing
            push!(kinds, o); k = length(kinds)
        end
        push!(index, Int32(k - 1))
    end
    return kinds, index
end
Base.:(==)(a::Dynamics, b::Dynamics) = a.body == b.body && (a.num_next_state, a.num_state, a.num_action) == (b.num_next_state, b.num_state, b.num_action)
Base.:(==)(a::Cost, b::Cost) = a.body == b.body && (a.num_state, a.num_action) == (b.num_state, b.num_action)
Base.:(==)(a::Constraint, b::Constraint) = a.body == b.body && a.indices_inequality == b.indices_inequality &&
                                           (a.num_constraint, a.num_state, a.num_action) == (b.num_constraint, b.num_state, b.num_action)

"""
    Solver(dynamics, costs, constraints; batch, options, name)

`Solver(dynamics, costs, constraints)` of the reference (src/solver.jl:28-46) for a batch: T-1 Dynamics, T Costs, T Constraints,
the last Cost / Constraint being the terminal objects (they are evaluated with `u = zeros(0)`, src/costs.jl:52, src/constraints.jl:69).
The objects MAY DIFFER from step to step and so may their dimensions (README.md:26, src/dynamics.jl:5-7): the distinct ones are
handed to the library kind by kind (`ilqr_compile_model_stages`), which lowers them onto its one-stage template — selectors in θ_t,
stacked constraint rows, zero padding to the largest dimensions — and the handle is given the selector table
(`ilqr_set_stage_selectors`). Arrays passed to / returned by the solver are the PADDED ones: (solver.nx, T, B) with
`solver.state_dims[t]` real entries at step t, likewise `solver.action_dims`.
"""
function Solver(dynamics::Vector{Dynamics}, costs::Vector{Cost}, constraints::Vector{Constraint};
                batch::Integer, options::Options = Options(), name::AbstractString = "user", device::Integer = 0,
                devices::AbstractVector{<:Integer} = Int[], constrained::Bool = true)
    T = length(costs)
    length(dynamics) == T - 1 && length(constraints) == T ||
        error("Solver: expected T-1 dynamics, T costs and T constraints (src/solver.jl:28-46)")
    dk, di = kinds_of(dynamics)
    ck, ci = kinds_of(costs[1:end-1])
    kk, ki = kinds_of(constraints[1:end-1])
    ct, kt = costs[end], constraints[end]
    nwu = maximum(o.num_parameter for o in vcat(dk, ck, kk, [ct], [kt]))
    prefix(body, p) = replace(body, "@P@" => p)
    source = join([prefix(d.body, "dynamics_$(k - 1)") for (k, d) in enumerate(dk)]) *
             join([prefix(c.body, "cost_stage_$(k - 1)") for (k, c) in enumerate(ck)]) *
             join([prefix(c.body, "constraint_stage_$(k - 1)") for (k, c) in enumerate(kk) if c.num_constraint > 0]) *
             prefix(ct.body, "cost_terminal") * (kt.num_constraint > 0 ? prefix(kt.body, "constraint_terminal") : "")
    i32(v) = Int32.(collect(v))
    dnx, dnu, dnn = i32(d.num_state for d in dk), i32(d.num_action for d in dk), i32(d.num_next_state for d in dk)
    cnx, cnu = i32(c.num_state for c in ck), i32(c.num_action for c in ck)
    knc, knx, knu = i32(c.num_constraint for c in kk), i32(c.num_state for c in kk), i32(c.num_action for c in kk)
    kin = reduce(vcat, [ineq_words(c.indices_inequality) for c in kk]; init = UInt64[])
    cap = T * (length(dk) + length(ck) + length(kk))
    selectors = zeros(Float64, max(cap, 1)); state_dims = zeros(Int32, T); action_dims = zeros(Int32, T - 1)
    plan = Ref(StagePlan(0, 0, 0, 0, 0, 0, 0, 0, 0, ntuple(_ -> Int32(0), 16), ntuple(_ -> UInt64(0), 4)))
    regname = Vector{UInt8}(undef, 160); path = Vector{UInt8}(undef, 1024)
    GC.@preserve name source dnx dnu dnn di cnx cnu ci knc knx knu kin ki begin
        kinds = StageKinds(Int32(T), Int32(nwu),
                           Int32(length(dk)), pointer(dnx), pointer(dnu), pointer(dnn), pointer(di),
                           Int32(length(ck)), pointer(cnx), pointer(cnu), pointer(ci),
                           Int32(length(kk)), pointer(knc), pointer(knx), pointer(knu), pointer(kin), pointer(ki),
                           Int32(ct.num_state), Int32(kt.num_constraint), Tuple(ineq_words(kt.indices_inequality)))
        check(ccall((:ilqr_compile_model_stages, LIB[]), Cint,
                    (Cstring, Ref{StageKinds}, Cstring, Ref{StagePlan}, Ptr{Float64}, Csize_t, Ptr{Int32}, Ptr{Int32},
                     Ptr{UInt8}, Csize_t, Ptr{UInt8}, Csize_t),
                    name, kinds, source, plan, selectors, cap, state_dims, action_dims, regname, length(regname), path, length(path)))
    end
    s = Solver(unsafe_string(pointer(regname)); horizon = T, batch = batch, constrained = constrained, options = options,
               device = device, devices = devices, model_library = unsafe_string(pointer(path)))
    S = plan[].n_selectors
    S > 0 && check(ccall((:ilqr_set_stage_selectors, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int32), s.handle, selectors, S))
    s.state_dims = Int.(state_dims); s.action_dims = Int.(action_dims)
    return s
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

# solve! with ONE step size per inner iteration for the whole batch, over every process of a multi-GPU job (not a reference behaviour:
# the optional mode of include/ilqr_hip.h, ilqr_solve_shared_step). `reduce!(v::Vector{Float64})` sums v in place over all processes
# (MPI.Allreduce!(v, +, comm), an RCCL binding, ...); without it the batch of this handle shares its step size alone.
function solve_shared_step!(s::Solver, reduce!::Union{Function,Nothing} = nothing)
    check(ccall((:ilqr_set_options, LIB[]), Cint, (Ptr{Cvoid}, Ref{Options}), s.handle, s.options))
    cap = Int(s.options.max_dual_updates) * Int(s.options.max_iterations)
    steps = zeros(Float64, max(cap, 1)); n = Ref{Int32}(0)
    if reduce! === nothing
        check(ccall((:ilqr_solve_shared_step, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Int32, Ref{Int32}),
                    s.handle, C_NULL, C_NULL, steps, cap, n))
    else
        cb = function (values::Ptr{Float64}, len::Int32, ::Ptr{Cvoid})::Cint
            v = unsafe_wrap(Array, values, Int(len)); reduce!(v); return Cint(0)
        end
        cfn = @cfunction($cb, Cint, (Ptr{Float64}, Int32, Ptr{Cvoid}))
        GC.@preserve cfn check(ccall((:ilqr_solve_shared_step, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Int32, Ref{Int32}),
                                     s.handle, cfn, C_NULL, steps, cap, n))
    end
    return steps[1:min(Int(n[]), cap)]      # n = inner iterations taken; at most `cap` entries were written
end

# 0 = auto, 1 = latency, 2 = throughput, 3 = packed (four instances per wave, no horizon limit)
set_kernel_variant!(s::Solver, v::Integer) = check(ccall((:ilqr_set_kernel_variant, LIB[]), Cint, (Ptr{Cvoid}, Int32), s.handle, v))
# straggler hand-over of the packed kernel (include/ilqr_hip.h): outer = -1 by head count (default), 0 off, k >= 2 by outer iteration;
# live = survivors of the batch at which they all leave (-1 auto)
set_handover!(s::Solver, outer::Integer) = check(ccall((:ilqr_set_handover, LIB[]), Cint, (Ptr{Cvoid}, Int32), s.handle, outer))
set_handover_live!(s::Solver, live::Integer) = check(ccall((:ilqr_set_handover_live, LIB[]), Cint, (Ptr{Cvoid}, Int32), s.handle, live))
set_handover_mark!(s::Solver, rejected::Integer) = check(ccall((:ilqr_set_handover_mark, LIB[]), Cint, (Ptr{Cvoid}, Int32), s.handle, rejected))
function handover_stats(s::Solver)      # (instances through the workgroups' queue, instances marked as stragglers) of the last solve!
    q = Ref{Int32}(0); m = Ref{Int32}(0)
    check(ccall((:ilqr_get_handover_stats, LIB[]), Cint, (Ptr{Cvoid}, Ref{Int32}, Ref{Int32}), s.handle, q, m))
    return (queued = Int(q[]), marked = Int(m[]))
end

# what `verbose` prints per inner iteration (src/solve.jl:40-45), recorded on the device: rows of
# (outer, inner, objective, gradient_norm, max_violation, step_size, status, rollouts) per instance
enable_trace!(s::Solver, capacity::Integer) = check(ccall((:ilqr_enable_trace, LIB[]), Cint, (Ptr{Cvoid}, Int32), s.handle, capacity))
function trace(s::Solver, capacity::Integer)
    out = Array{Float64,3}(undef, 8, capacity, s.B)
    check(ccall((:ilqr_get_trace, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Float64}), s.handle, out))
    return out
end

end # module
