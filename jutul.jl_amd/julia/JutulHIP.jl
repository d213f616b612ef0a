# JutulHIP.jl -- Julia-side binding of libjutul_hip.so (C ABI: include/jutul_hip.h).
#
# STATUS: written against Jutul.jl v0.4.25's dispatch seams; NOT executed in the build environment (no Julia there).
# tests/test_gpu_julia_sequence.py parses the `@jh :name` calls of every function below and drives the SAME entry points in
# the SAME order through ctypes on the MI355X for one perform_step! (set-up -> update_equation! -> apply_forces! ->
# update_linearized_system_equation! -> check_convergence -> linear_solve! -> update_primary_variables! -> update_after_step!),
# comparing the result with the oracle's Newton step.  Keep the two in sync: the test fails when they diverge.
#
# Host code stays Julia: `simulate!` drives time stepping / Newton control as before; the per-Newton hot calls go to the GPU:
#   setup_equation_storage      conservation.jl:137       -> discretisation, law, Jacobian, vectors, face / cell data
#   update_equation!            conservation.jl:572       -> state / state0 upload ONLY when the host changed them, sources reset
#   get_diagonal_entries        conservation.jl:667-674   -> sparse host accumulator the forces add into (apply_forces!, models.jl:889-901)
#   update_linearized_system_equation!  conservation.jl:298 -> sources upload only when they changed + ONE fused flux / accumulation / fill kernel
#   convergence_criterion       equations.jl:619-629      -> max |r_e| on the device (check_convergence, models.jl:830-883)
#   linear_solve!               linsolve/krylov.jl:71-182 -> ILU(0) refactor + device BiCGStab / GMRES, dx = -x stays in HBM
#   update_primary_variables!   models.jl:928-965         -> increment norms + clamp chain on the device
#   update_after_step!          models.jl:983-1011        -> variable_change_report + state0 <- state on the device
#   get_output_state            models.jl:1048-1058       -> the one place the state comes back to the host (report steps)
#
# State residency (the reference keeps dx and the primary variables as views into the LinearizedSystem buffers and never
# copies them, models.jl:928-953,1105-1172; conservation.jl:549-555): after the first upload the DEVICE copy of the primary
# variables is the state.  `device_state_valid` / `device_state0_valid` say whether the device copies equal what the host last
# wrote (cleared by every host-side writer: reset_variables!, reset_previous_state!, invalidate_device_state!);
# `host_state_stale` says the device has moved on (update_primary_variables!) and storage.state must be refreshed
# (sync_host_state!) before host code reads it; get_output_state refreshes storage.state0 (what it copies), once per report step.  Per Newton iteration nothing
# crosses PCIe except a handful of scalars.
module JutulHIP

using Jutul, LinearAlgebra
import Jutul: JutulContext, GPUJutulContext, matrix_layout, float_type, index_type, transfer, synchronize,
              setup_equation_storage, update_equation!, update_linearized_system_equation!, align_to_jacobian!,
              declare_pattern, get_diagonal_entries, convergence_criterion, update_primary_variables!, update_after_step!,
              reset_state_to_previous_state!, post_update_linearized_system!, reset_variables!, reset_previous_state!,
              get_output_state, setup_linearized_system!,
              linear_solve!, update_preconditioner!, apply!, operator_nrows, ConservationLaw,
              TwoPointPotentialFlowHardCoded, GenericKrylov, ILUZeroPreconditioner, linear_solve_return,
              BlockMajorLayout, EquationMajorLayout, SimulationModel, Cells

const libjutul_hip = get(ENV, "JUTUL_HIP_LIBRARY", "libjutul_hip.so")

# ---- error handling: non-zero status -> Julia exception (keeps failure_cuts_timestep semantics, simulator.jl:509-518)
function check(rc::Int32)
    rc == 0 && return nothing
    buf = Vector{UInt8}(undef, 4096)
    ccall((:jh_last_error, libjutul_hip), Int32, (Ptr{UInt8}, Int64), buf, length(buf))
    error("libjutul_hip: " * unsafe_string(pointer(buf)))
end
macro jh(name, argtypes, args...)
    return esc(:(check(ccall(($name, libjutul_hip), Int32, $argtypes, $(args...)))))
end
const Handle = Ptr{Cvoid}

# ---- context (seam: JutulContext, core_types.jl:86-88; precedent SingleCUDAContext, contexts/cuda.jl) ------------
mutable struct HIPContext <: GPUJutulContext
    handle::Handle
    device::Int
    matrix_layout
    block_rows::Int
    n_owned::Int            # > 0: rank-local model whose cells are [owned..., ghosts...] (ext/.../utils.jl:178-184)
    # options: the named integer options of the context (jh_context_set_option, include/jutul_hip.h) as keyword arguments, e.g.
    # HIPContext(0; consumer_reduce = 0, comm_timeout_ms = 60_000) -- the role of ParallelCSRContext's keyword arguments
    # (contexts/csr.jl:3-23)
    function HIPContext(device = 0; matrix_layout = BlockMajorLayout(), block_rows = 0, n_owned = 0, options...)
        h = Ref{Handle}(C_NULL)
        if device < 0   # planning context: set-up tables only (jh_context_create_host), every compute call throws
            @jh :jh_context_create_host (Ref{Handle},) h
        else
            @jh :jh_context_create (Int32, Ref{Handle}) Int32(device) h
        end
        ctx = new(h[], device, matrix_layout, block_rows, n_owned)
        finalizer(c -> ccall((:jh_context_destroy, libjutul_hip), Int32, (Handle,), c.handle), ctx)
        for (k, v) in options
            set_option!(ctx, k, v)
        end
        return ctx
    end
end
set_option!(c::HIPContext, key, value) = @jh :jh_context_set_option (Handle, Cstring, Int64) c.handle String(key) Int64(value)
function get_option(c::HIPContext, key)
    v = Ref{Int64}(0)
    @jh :jh_context_get_option (Handle, Cstring, Ref{Int64}) c.handle String(key) v
    return v[]
end
function plan_checksum(c::HIPContext)   # planning contexts (device < 0) with plan_checksum = 1: checksum of the tables a device would get
    v = Ref{Int64}(0)
    @jh :jh_context_plan_checksum (Handle, Ref{Int64}) c.handle v
    return v[]
end
matrix_layout(c::HIPContext) = c.matrix_layout
float_type(::HIPContext) = Float64          # context.jl:76
index_type(::HIPContext) = Int64            # context.jl:77 (host side; the device uses 0-based Int32)
synchronize(c::HIPContext) = @jh :jh_synchronize (Handle,) c.handle
# host arrays stay on the host (device copies live behind handles).  Only the array method is specialised: a method
# `transfer(::HIPContext, v)` would be ambiguous with the reference's `transfer(context, v::Real / ::NamedTuple / ::AbstractDict /
# ::AbstractFloat / ::Integer)` (context.jl:15-60), which keep doing their conversions with float_type / index_type above
transfer(::HIPContext, v::AbstractArray) = v

const HIPModel = SimulationModel{<:Any, <:Any, <:Any, HIPContext}

# ---- physics hooks -------------------------------------------------------------------------------------------------------------
# Jutul's ConservationLaw carries no concrete flux (SURVEY a-5); a model package states which built-in law of the library its
# equation is (or hands over device source for jh_law_create_custom) and where its face / cell data live.  Defaults: the
# VariablePoisson test system (variable_poisson.jl:90-133): K on faces, unit accumulation coefficient, no gravity.
hip_law_kind(model, eq) = Int32(0)                       # JH_LAW_POISSON | 1 JH_LAW_COMPRESSIBLE | 2 JH_LAW_TWOPHASE
hip_law_params(model, eq) = nothing                      # rho0[2], compressibility[2], viscosity[2], p_ref
hip_law_source(model, eq) = nothing                      # String: HIP device code for jh_flux / jh_mass (generic-AD path)
hip_face_trans(model, storage) = haskey(storage.parameters, :Transmissibilities) ? storage.parameters[:Transmissibilities] : storage.parameters[:K]
hip_face_gdz(model, storage) = haskey(storage.parameters, :TwoPointGravityDifference) ? storage.parameters[:TwoPointGravityDifference] : nothing
hip_cell_volumes(model, storage) = haskey(storage.parameters, :FluidVolume) ? storage.parameters[:FluidVolume] : nothing
hip_update_limits(model) = nothing                       # 5N doubles: scale, abs_max, rel_max, minimum, maximum per variable (NaN = unset)
hip_download_increment(model) = false                    # true: linear_solve! also copies dx into sys.dx_buffer (adjoints, custom reports)

# primary variables <-> the library's [N, nc] block-major layout (one scalar per cell and variable on this path)
function pack_primary!(X::Matrix{Float64}, model, state)
    for (i, k) in enumerate(keys(Jutul.get_primary_variables(model)))
        X[i, :] .= Jutul.value.(state[k])
    end
    return X
end
function unpack_primary!(state, model, X::Matrix{Float64})
    for (i, k) in enumerate(keys(Jutul.get_primary_variables(model)))
        v = state[k]
        # v may hold ForwardDiff Duals with seeded partials that Jutul's own secondary-variable evaluation expects: shift the
        # VALUE and keep the partials, exactly like update_value does (v + dv, variables/utils.jl:169-174)
        @inbounds for c in eachindex(v)
            v[c] += X[i, c] - Jutul.value(v[c])
        end
    end
    return state
end

# Sparse accumulator handed to apply_forces! as `get_diagonal_entries` (models.jl:889-901): forces do `d[c] += f.value` (N = 1)
# or `d[e, c] += ...` on a handful of cells; remembering the touched cells makes the flush O(#sources) instead of an O(nc)
# scan per Newton iteration.
mutable struct SourceAccumulator <: AbstractMatrix{Float64}
    N::Int
    nc::Int
    slot::Dict{Int, Int}          # cell -> column of `vals`
    cells::Vector{Int64}
    vals::Vector{Float64}         # N values per touched cell
end
SourceAccumulator(N, nc) = SourceAccumulator(N, nc, Dict{Int, Int}(), Int64[], Float64[])
Base.size(a::SourceAccumulator) = (a.N, a.nc)
Base.IndexStyle(::Type{SourceAccumulator}) = IndexCartesian()
function Base.getindex(a::SourceAccumulator, e::Int, c::Int)
    j = get(a.slot, c, 0)
    return j == 0 ? 0.0 : a.vals[(j - 1) * a.N + e]
end
function Base.setindex!(a::SourceAccumulator, v, e::Int, c::Int)
    j = get!(a.slot, c) do
        push!(a.cells, c); append!(a.vals, zeros(a.N)); length(a.cells)
    end
    a.vals[(j - 1) * a.N + e] = v
    return a
end
Base.getindex(a::SourceAccumulator, c::Int) = a[1, c]                 # vector use when N == 1 (variable_poisson.jl:78-84)
Base.setindex!(a::SourceAccumulator, v, c::Int) = setindex!(a, v, 1, c)
function reset!(a::SourceAccumulator)                                 # reset_sources! (conservation.jl:660-664)
    empty!(a.slot); empty!(a.cells); empty!(a.vals)
    return a
end

# ---- equation storage (seam: setup_equation_storage, conservation.jl:137) -------------------------------------------
mutable struct HIPConservationLawStorage
    disc::Handle     # jh_tpfa
    law::Handle      # jh_law
    jac::Handle      # jh_csr
    r::Handle        # jh_vec
    dx::Handle       # jh_vec
    nc::Int
    N::Int
    n_owned::Int
    dt::Float64
    X::Matrix{Float64}             # [N, nc] staging of the primary variables
    sources::SourceAccumulator     # what apply_forces! adds to the diagonal entries (values only)
    src_cells::Vector{Int64}       # the source list the device holds
    src_vals::Vector{Float64}
    err::Vector{Float64}
    device_state_valid::Bool       # device X  == what the host last wrote into storage.state
    device_state0_valid::Bool      # device X0 == what the host last wrote into storage.state0
    host_state_stale::Bool         # the device has updated X since storage.state was last refreshed
    host_state0_stale::Bool        # the device has updated X0 (update_after_step!) since storage.state0 was last refreshed
    registered::Dict{Symbol, Ptr{Cvoid}}  # storage.state0[k] arrays that are page-locked (jh_host_register), by address
    assemblies::Int                # update_linearized_system_equation! calls so far
    err_of_assembly::Int           # the assembly `err` was reduced for (convergence_criterion of several equations: one reduction)
end
# Several conservation laws on Cells in one model (setup_storage_equations! loops over all of model.equations, models.jl:549-572;
# Jutul gives every equation its rows of the block through equation offsets, conservation.jl:143 `equation_offset`): on the device
# they are ONE law with N = sum of their components -- the storage of the FIRST law (the "group") owns the discretisation, the law,
# the Jacobian and the vectors and does all the work; every later law gets a member that names its rows of the block.  The physics
# hooks (hip_law_kind / hip_law_source / hip_law_params) of the first law describe the whole block, components in equation order.
mutable struct HIPEquationMember
    group::Union{Nothing, HIPConservationLawStorage}   # linked by setup_linearized_system! (all equation storages exist by then)
    offset::Int                                         # rows offset+1 .. offset+ne of the block
    ne::Int
end
# get_diagonal_entries of a member: its rows of the group's source accumulator (apply_forces! does d[c] += v or d[e, c] += v)
struct SourceRows <: AbstractMatrix{Float64}
    acc::SourceAccumulator
    offset::Int
    ne::Int
end
Base.size(a::SourceRows) = (a.ne, a.acc.nc)
Base.IndexStyle(::Type{SourceRows}) = IndexCartesian()
Base.getindex(a::SourceRows, e::Int, c::Int) = a.acc[a.offset + e, c]
Base.setindex!(a::SourceRows, v, e::Int, c::Int) = setindex!(a.acc, v, a.offset + e, c)
Base.getindex(a::SourceRows, c::Int) = a[1, c]
Base.setindex!(a::SourceRows, v, c::Int) = setindex!(a, v, 1, c)
# the model's equations as the device sees them: (key, equation, components) in equation order -- TPFA conservation laws only
function hip_device_laws(model)
    laws = Tuple{Symbol, Any, Int}[]
    for (k, e) in pairs(model.equations)
        e isa ConservationLaw{<:Any, <:TwoPointPotentialFlowHardCoded, <:Any, <:Any} ||
            error("JutulHIP: equation :$k is a $(typeof(e)); the device path assembles TPFA conservation laws on Cells only, " *
                  "models mixing them with host equations are not supported")
        push!(laws, (k, e, Jutul.number_of_equations_per_entity(model, e)))
    end
    return laws
end
# the page-locked ranges are handed back when the storage goes away (a later simulator may get the same addresses: registering a
# range twice is an error of the runtime) -- also called by hand when a simulator is torn down
function unregister_host_arrays!(s::HIPConservationLawStorage)
    for p in values(s.registered)
        ccall((:jh_host_unregister, libjutul_hip), Int32, (Ptr{Cvoid},), p)   # (no error out of a finalizer)
    end
    empty!(s.registered)
    return s
end

function setup_equation_storage(model::HIPModel,
        eq::ConservationLaw{<:Any, <:TwoPointPotentialFlowHardCoded, <:Any, <:Any}, storage; kwarg...)
    ctx = model.context
    laws = hip_device_laws(model)
    if laws[1][2] !== eq     # a later law of the model: rows of the first law's block
        i = findfirst(l -> l[2] === eq, laws)
        isnothing(i) && error("JutulHIP: setup_equation_storage for an equation that is not in model.equations")
        return HIPEquationMember(nothing, sum(l[3] for l in laws[1:i - 1]), laws[i][3])
    end
    N = Jutul.get_neighborship(model.domain.representation)      # 2 x nf, Int64, 1-based (as stored)
    nc = Jutul.number_of_cells(model.domain)
    ne = sum(l[3] for l in laws)                                  # block size: the components of all conservation laws
    ne == length(keys(Jutul.get_primary_variables(model))) ||
        error("JutulHIP: $ne conservation equations per cell but $(length(keys(Jutul.get_primary_variables(model)))) primary variables")
    disc = Ref{Handle}(C_NULL)
    # the face transmissibilities double as the weights of the device-block partition: weak couplings are cut first, like the Metis
    # partition of the |A|-weighted graph behind ILUZeroPreconditioner (precond/ilu.jl:37-60, partitioning.jl:64-78)
    @jh :jh_tpfa_create_weighted (Handle, Int64, Int64, Ptr{Int64}, Ptr{Float64}, Int32, Int32, Ptr{Int64}, Int64, Int64, Ref{Handle}) ctx.handle nc size(N, 2) N Float64.(hip_face_trans(model, storage)) Int32(ne) Int32(1) C_NULL ctx.block_rows ctx.n_owned disc
    law = Ref{Handle}(C_NULL)
    src = hip_law_source(model, eq)
    par = hip_law_params(model, eq)
    if isnothing(src)
        @jh :jh_law_create (Handle, Int32, Ptr{Float64}, Ref{Handle}) disc[] hip_law_kind(model, eq) (isnothing(par) ? C_NULL : par) law
    else
        @jh :jh_law_create_custom (Handle, Cstring, Ptr{Float64}, Int32, Ref{Handle}) disc[] src (isnothing(par) ? C_NULL : par) Int32(isnothing(par) ? 0 : length(par)) law
    end
    # static data of the discretisation: face transmissibilities, gravity differences, accumulation coefficients
    # (compute_face_trans / compute_face_gdz, finite-volume.jl:224-233,304-313; pore volumes) -- uploaded once, re-upload
    # with the same call if parameters change
    @jh :jh_law_set_data (Handle, Int32, Ptr{Float64}) law[] Int32(0) Float64.(hip_face_trans(model, storage))
    gdz = hip_face_gdz(model, storage)
    if !isnothing(gdz)
        @jh :jh_law_set_data (Handle, Int32, Ptr{Float64}) law[] Int32(1) Float64.(gdz)
    end
    vol = hip_cell_volumes(model, storage)
    if !isnothing(vol)
        @jh :jh_law_set_data (Handle, Int32, Ptr{Float64}) law[] Int32(2) Float64.(vol)
    end
    jac = Ref{Handle}(C_NULL); r = Ref{Handle}(C_NULL); dx = Ref{Handle}(C_NULL)
    @jh :jh_csr_create (Handle, Ref{Handle}) disc[] jac
    @jh :jh_vec_create (Handle, Ref{Handle}) disc[] r
    @jh :jh_vec_create (Handle, Ref{Handle}) disc[] dx
    n_owned = ctx.n_owned > 0 ? ctx.n_owned : nc
    s = HIPConservationLawStorage(disc[], law[], jac[], r[], dx[], nc, ne, n_owned, NaN, zeros(ne, nc), SourceAccumulator(ne, nc), Int64[], Float64[],
                                  zeros(ne), false, false, false, false, Dict{Symbol, Ptr{Cvoid}}(), 0, -1)
    finalizer(unregister_host_arrays!, s)
    return s
end

# ---- linearized system (seam: setup_linearized_system!, models.jl:654-668; LinearizedSystem, linsolve/default.jl:34-42) -----
# The Jacobian / residual / increment live on the device behind the equation storage's handles; this wrapper is what
# `storage[:LinearizedSystem]` holds so that `linear_solve!(sys, ...)` dispatches here.  dx_buffer is the host mirror
# linear_solve! fills (its default `dx = sys.dx_buffer`, krylov.jl:79) and storage.views.primary_variables points into.
struct HIPLinearizedSystem <: Jutul.JutulLinearSystem
    eq_s::HIPConservationLawStorage
    jac::Handle                    # the device Jacobian (LinearizedSystem.jac, default.jl:34-42); host code gets values via jh_csr_get_values
    jac_buffer::Nothing            # no host nzval: update_linearized_system! passes it on as `nzval = nothing`
    r_buffer::Vector{Float64}
    dx_buffer::Vector{Float64}     # host mirror, filled only on request (hip_download_increment)
    matrix_layout
end

# the group among the model's equation storages (every other storage must be one of its members: hip_device_laws has already
# rejected models with host equations); members are linked to it here, once
function hip_equation_storage(storage)
    all = collect(values(storage[:equations]))
    found = [v for v in all if v isa HIPConservationLawStorage]
    length(found) == 1 || error("JutulHIP: expected exactly one device conservation-law group, found $(length(found)) " *
                                "among $(length(all)) equations")
    for v in all
        v === found[1] && continue
        v isa HIPEquationMember || error("JutulHIP: models mixing the device conservation laws with host equations are not supported")
        v.group = found[1]
    end
    return found[1]
end

function setup_linearized_system!(storage, model::HIPModel)
    eq_s = hip_equation_storage(storage)
    n = eq_s.nc * eq_s.N
    lsys = HIPLinearizedSystem(eq_s, eq_s.jac, nothing, zeros(n), zeros(n), matrix_layout(model.context))
    storage[:LinearizedSystem] = lsys
    return lsys
end
Jutul.align_equations_to_linearized_system!(storage, model::HIPModel; kwarg...) = nothing
# update_linearized_system! hands `nzval = lsys.jac_buffer` and a residual view to every equation (models.jl:774-783): there
# are no host buffers to fill on this path
function Jutul.update_linearized_system!(lsys::HIPLinearizedSystem, equations, eqs_storage, eqs_views, model::HIPModel; kwarg...)
    for key in keys(equations)
        update_linearized_system_equation!(nothing, nothing, model, equations[key], eqs_storage[key])
    end
end

# members of a group: the group's kernel assembles their rows, uploads their sources and reduces their errors
update_equation!(m::HIPEquationMember, law::ConservationLaw, storage, model, dt) = nothing
update_linearized_system_equation!(nz, r, model, law::ConservationLaw, m::HIPEquationMember) = nothing
get_diagonal_entries(eq::ConservationLaw, m::HIPEquationMember) = SourceRows(m.group.sources, m.offset, m.ne)
declare_pattern(model, eq::ConservationLaw, m::HIPEquationMember, ::Cells) = (Int64[], Int64[])   # (in the group's pattern)
align_to_jacobian!(m::HIPEquationMember, eq::ConservationLaw, jac, model, u::Cells; kwarg...) = nothing

# pattern / alignment: the library owns the device pattern; the host tables are available bit-exact if Jutul needs
# them (conservation.jl:486-505, :143-216)
function declare_pattern(model, eq::ConservationLaw, s::HIPConservationLawStorage, ::Cells)
    nnzb = Ref{Int64}(0)
    @jh :jh_tpfa_sizes (Handle, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ref{Int64}, Ptr{Int32}) s.disc C_NULL C_NULL C_NULL nnzb C_NULL
    rowptr = Vector{Int64}(undef, s.nc + 1); colidx = Vector{Int64}(undef, nnzb[])
    @jh :jh_tpfa_get_pattern (Handle, Ptr{Int64}, Ptr{Int64}) s.disc rowptr colidx
    I = similar(colidx)
    for row in 1:s.nc, k in rowptr[row]:(rowptr[row + 1] - 1)
        I[k] = row
    end
    return (I, colidx)
end
align_to_jacobian!(s::HIPConservationLawStorage, eq::ConservationLaw, jac, model, u::Cells; kwarg...) = nothing

# ---- assembly (seams: update_equation!, conservation.jl:572; update_linearized_system_equation!, :298) ---------------
function update_equation!(s::HIPConservationLawStorage, law::ConservationLaw, storage, model, dt)
    # state_pair (conservation.jl:549-555): the accumulation needs state AND state0.  Both live on the device; they are
    # uploaded only when the host has written them since (first iteration, restarts, user edits)
    if !s.device_state_valid
        @jh :jh_law_set_state (Handle, Ptr{Float64}) s.law pack_primary!(s.X, model, storage.state)
        s.device_state_valid = true
        s.host_state_stale = false
    end
    if !s.device_state0_valid
        @jh :jh_law_set_state0 (Handle, Ptr{Float64}) s.law pack_primary!(s.X, model, storage.state0)
        s.device_state0_valid = true
        s.host_state0_stale = false
    end
    reset!(s.sources)                      # reset_sources! (conservation.jl:660-664)
    s.dt = dt
    return nothing  # the flux + fill are fused into one kernel launched by update_linearized_system_equation!
end

# apply_forces! (models.jl:889-901) adds every force to `get_diagonal_entries(eq, eq_s)`, e.g. `d[c] += f.value` for a
# PoissonSource (variable_poisson.jl:78-84).  Here that array is a sparse host accumulator; it reaches the device with the
# assembly, and only when it differs from the list the device already holds.
get_diagonal_entries(eq::ConservationLaw, s::HIPConservationLawStorage) = s.sources

function update_linearized_system_equation!(nz, r, model, law::ConservationLaw, s::HIPConservationLawStorage)
    a = s.sources
    if a.cells != s.src_cells || a.vals != s.src_vals
        @jh :jh_law_set_sources (Handle, Int64, Ptr{Int64}, Ptr{Float64}) s.law length(a.cells) a.cells a.vals
        s.src_cells = copy(a.cells); s.src_vals = copy(a.vals)
    end
    @jh :jh_assemble (Handle, Float64, Handle, Handle) s.law s.dt s.jac s.r
    s.assemblies += 1
    # host copies only if the caller insists on host buffers (parity / debugging); the solve reads device memory
    if !isnothing(nz)
        @jh :jh_csr_get_values (Handle, Ptr{Float64}) s.jac nz
    end
    if !isnothing(r)
        @jh :jh_vec_download (Handle, Ptr{Float64}) s.r r
    end
end

# ---- host <-> device state hand-over ----------------------------------------------------------------------------------------
# every host-side writer of the primary variables invalidates the device copy; the next update_equation! uploads it again
function invalidate_device_state!(storage; state = true, state0 = true)
    haskey(storage, :LinearizedSystem) || return storage      # (initial reset_variables! during set-up: nothing uploaded yet)
    s = storage.LinearizedSystem.eq_s
    state && (s.device_state_valid = false)
    state0 && (s.device_state0_valid = false)
    return storage
end
function reset_variables!(storage, model::HIPModel, new_vars; type = :state)      # models.jl:1079-1081 (simulator.jl:664-669)
    invoke(reset_variables!, Tuple{Any, Jutul.JutulModel, Any}, storage, model, new_vars; type = type)
    invalidate_device_state!(storage, state = type == :state, state0 = type == :state0)
end
function reset_previous_state!(storage, model::HIPModel, state0)                  # models.jl:1075-1077
    invoke(reset_previous_state!, Tuple{Any, Jutul.JutulModel, Any}, storage, model, state0)
    invalidate_device_state!(storage, state = false, state0 = true)
end
# device -> storage.state / storage.state0 (values only, partials kept): before any host code reads the primary variables
function sync_host_state!(storage, model::HIPModel)
    s = storage.LinearizedSystem.eq_s
    if s.host_state_stale
        @jh :jh_law_get_state (Handle, Ptr{Float64}) s.law s.X
        unpack_primary!(storage.state, model, s.X)
        s.host_state_stale = false
    end
    return storage
end
# get_output_state (models.jl:1048-1058) copies storage.state0[k] after a converged step (state0 == state there): the ONE regular
# device -> host transfer of the state, once per report step.  state0 holds plain Float64 arrays ("state without AD"), so each
# variable is DMA-ed straight into Jutul's own array (page-locked on first use); the Dual-valued storage.state stays stale
# until host code asks for it (sync_host_state!).
function get_output_state(storage, model::HIPModel)
    s = storage.LinearizedSystem.eq_s
    if s.host_state0_stale
        for (i, k) in enumerate(keys(Jutul.get_primary_variables(model)))
            v0 = storage.state0[k]::Array{Float64}
            p0 = Ptr{Cvoid}(pointer(v0))
            if get(s.registered, k, C_NULL) != p0      # first use, or Jutul has replaced the array since
                haskey(s.registered, k) && ccall((:jh_host_unregister, libjutul_hip), Int32, (Ptr{Cvoid},), s.registered[k])
                @jh :jh_host_register (Ptr{Cvoid}, Int64) v0 Int64(sizeof(v0))     # (a range that is already page-locked is accepted)
                s.registered[k] = p0
            end
            GC.@preserve v0 begin
                @jh :jh_law_get_variable (Handle, Int32, Int32, Ptr{Float64}) s.law Int32(1) Int32(i - 1) v0
            end
        end
        s.host_state0_stale = false
    end
    return invoke(get_output_state, Tuple{Any, Jutul.JutulModel}, storage, model)
end

# ---- convergence (seam: convergence_criterion, equations.jl:619-629, called by check_convergence, models.jl:830-883) ------
function reduce_errors!(s::HIPConservationLawStorage)     # max |r_e| of every component of the block, once per assembly
    if s.err_of_assembly != s.assemblies
        @jh :jh_convergence (Handle, Handle, Int64, Ptr{Float64}) s.law s.r Int64(s.n_owned) s.err
        s.err_of_assembly = s.assemblies
    end
    return s.err
end
function convergence_criterion(model::HIPModel, storage, eq::ConservationLaw, s::HIPConservationLawStorage, r; dt = 1.0, update_report = missing)
    ne = Jutul.number_of_equations_per_entity(model, eq)     # the group reports the rows of ITS law; members report theirs
    err = reduce_errors!(s)[1:ne]
    names = ne == 1 ? "R" : map(i -> "R_$i", 1:ne)
    return (AbsMax = (errors = err, names = names), )
end
function convergence_criterion(model::HIPModel, storage, eq::ConservationLaw, m::HIPEquationMember, r; dt = 1.0, update_report = missing)
    err = reduce_errors!(m.group)[m.offset + 1:m.offset + m.ne]
    names = m.ne == 1 ? "R" : map(i -> "R_$i", 1:m.ne)
    return (AbsMax = (errors = err, names = names), )
end

# ---- preconditioner (seams: update_preconditioner!, precond/ilu.jl:37; apply!, :62) -----------------------------------
abstract type HIPPreconditioner <: Jutul.JutulPreconditioner end
mutable struct HIPILUZero <: HIPPreconditioner       # ILUZeroPreconditioner (precond/ilu.jl:4-14): block-Jacobi ILU(0) over the device blocks
    handle::Handle
    dim
    HIPILUZero() = new(C_NULL, nothing)
end
mutable struct HIPJacobi <: HIPPreconditioner        # JacobiPreconditioner(w) (precond/jacobi.jl:5-18)
    handle::Handle
    dim
    w::Float64
    HIPJacobi(; w = 1.0) = new(C_NULL, nothing, w)
end
mutable struct HIPSPAI0 <: HIPPreconditioner         # SPAI0Preconditioner (precond/spai.jl:40-60)
    handle::Handle
    dim
    HIPSPAI0() = new(C_NULL, nothing)
end
function hip_preconditioner_create!(p::HIPILUZero, s::HIPConservationLawStorage)
    h = Ref{Handle}(C_NULL)
    @jh :jh_ilu0_create (Handle, Ptr{Int64}, Int64, Ref{Handle}) s.jac C_NULL -1 h   # device blocks
    return h[]
end
function hip_preconditioner_create!(p::Union{HIPJacobi, HIPSPAI0}, s::HIPConservationLawStorage)
    h = Ref{Handle}(C_NULL)
    @jh :jh_diag_precond_create (Handle, Int32, Float64, Ref{Handle}) s.jac Int32(p isa HIPJacobi ? 1 : 2) Float64(p isa HIPJacobi ? p.w : 1.0) h
    return h[]
end
function update_preconditioner!(p::HIPPreconditioner, s::HIPConservationLawStorage, b, context::HIPContext, executor)
    if p.handle == C_NULL
        p.handle = hip_preconditioner_create!(p, s)
        p.dim = (s.nc * s.N, s.nc * s.N)
    end
    @jh :jh_ilu0_factor (Handle,) p.handle        # update_preconditioner! of every kind (jutul_hip.h: jh_diag_precond_create)
end
operator_nrows(p::HIPPreconditioner) = p.dim[1]

# ---- linear solve (seam: linear_solve!, linsolve/krylov.jl:71-85) -------------------------------------------------------
mutable struct HIPKrylov
    handle::Handle
    r_norm::Float64
    HIPKrylov() = new(C_NULL, NaN)
end
function linear_solve!(sys::HIPLinearizedSystem, krylov::GenericKrylov, context::HIPContext, model, storage = nothing,
        dt = nothing, recorder = nothing, executor = nothing; dx = sys.dx_buffer, r = nothing,
        atol = Jutul.linear_solver_tolerance(krylov, :absolute), rtol = Jutul.linear_solver_tolerance(krylov, :relative),
        rtol_nl = Jutul.linear_solver_tolerance(krylov, :nonlinear_relative),
        rtol_relaxed = Jutul.linear_solver_tolerance(krylov, :relaxed_relative), kwarg...)
    s = sys.eq_s
    cfg = krylov.config
    if krylov.scaling != :none      # krylov_scale_system! (krylov.jl:194; default.jl:325-385)
        @jh :jh_scale_system (Handle, Handle, Int32, Float64) s.jac s.r Int32(krylov.scaling == :diagonal ? 1 : 2) Float64(isnothing(dt) ? 1.0 : dt)
    end
    prec = krylov.preconditioner::HIPPreconditioner   # HIPILUZero / HIPJacobi / HIPSPAI0
    t_prec = @elapsed update_preconditioner!(prec, s, nothing, context, executor)
    ws = krylov.storage
    if !(ws isa HIPKrylov)
        ws = HIPKrylov(); h = Ref{Handle}(C_NULL)
        @jh :jh_krylov_create (Handle, Ref{Handle}) s.jac h
        ws.handle = h[]; krylov.storage = ws
    end
    @jh :jh_krylov_set_min_iterations (Handle, Int64) ws.handle Int64(cfg.min_iterations)
    # true_residual / relaxed tolerance from the Newton history (krylov.jl:96-118)
    it = isnothing(recorder) ? 0 : Jutul.subiteration(recorder)
    use_relaxed = !isnothing(rtol_nl) && it > 0
    if use_relaxed || cfg.true_residual
        rr = Ref{Float64}(0.0)
        @jh :jh_vec_dot (Handle, Handle, Ref{Float64}) s.r s.r rr
        r_k = sqrt(rr[])
        if cfg.true_residual
            atol, rtol = atol + rtol * r_k, 0.0
        end
        if use_relaxed
            if it == 1
                ws.r_norm = r_k
            elseif !isnan(ws.r_norm)
                rtol = max(min(ws.r_norm * rtol_nl / r_k, rtol_relaxed), rtol)
            end
        end
    end
    side = cfg.precond_side == :right ? Int32(2) : Int32(1)
    iters = Ref{Int64}(0); status = Ref{Int32}(0)
    hist = zeros(cfg.max_iterations + 2)
    x = s.dx  # solution lands in dx, then negated in place (update_dx_from_vector!, default.jl:444-446)
    if krylov.solver == :gmres   # the reference's second Krylov method (linsolve/krylov.jl:214-218)
        @jh :jh_gmres (Handle, Handle, Int32, Handle, Handle, Float64, Float64, Int64, Ref{Int64}, Ref{Int32}, Ptr{Float64}, Int64) ws.handle prec.handle side s.r x rtol atol cfg.max_iterations iters status hist length(hist)
    else
        @jh :jh_bicgstab (Handle, Handle, Int32, Handle, Handle, Float64, Float64, Int64, Ref{Int64}, Ref{Int32}, Ptr{Float64}, Int64) ws.handle prec.handle side s.r x rtol atol cfg.max_iterations iters status hist length(hist)
    end
    @jh :jh_vec_negate_into (Handle, Handle) s.dx x
    n = iters[]
    solved = status[] == 0
    manual = cfg.min_iterations > 1
    bad = (manual && n == cfg.max_iterations) || (!manual && !solved)
    if bad && n > 0 && hist[n + 1] / hist[1] > 1.0
        error("Bad linear solve: final residual $(hist[n + 1]), rel. value $(hist[n + 1] / hist[1])")   # krylov.jl:161-166
    end
    # The increment stays in HBM: update_primary_variables! below applies it there and computes its norms there.  A host copy
    # (the reference's `dx = sys.dx_buffer`, krylov.jl:79) is made only for callers that read it (hip_download_increment).
    if !isnothing(dx) && hip_download_increment(model)
        @jh :jh_vec_download (Handle, Ptr{Float64}) s.dx dx
    end
    return linear_solve_return(solved, n, (residuals = hist[1:n + 1], solved = solved); prepare = t_prec)
end

# ---- primary update (seam: update_primary_variables!, models.jl:928-953; choose_increment chain, variables/utils.jl:110-174)
function update_primary_variables!(storage, model::HIPModel; relaxation = 1.0, check = false, kwarg...)
    s = storage.LinearizedSystem.eq_s
    norms = zeros(2 * s.N)         # increment_norm (models.jl:955-965) of every primary variable, reduced on the device
    @jh :jh_increment_norm (Handle, Handle, Int64, Ptr{Float64}) s.law s.dx Int64(s.n_owned) norms
    if check && !all(isfinite, norms)                      # check_increment: NaN / Inf propagate into the sums
        error("Primary variables recieved invalid updates.")
    end
    lim = hip_update_limits(model)
    @jh :jh_update_primary (Handle, Handle, Float64, Ptr{Float64}) s.law s.dx Float64(relaxation) (isnothing(lim) ? C_NULL : lim)
    s.host_state_stale = true      # storage.state is refreshed on demand (sync_host_state!, get_output_state)
    report = Dict{Symbol, Any}()
    for (i, (k, p)) in enumerate(pairs(Jutul.get_primary_variables(model)))
        scale = something(Jutul.variable_scale(p), 1.0)
        report[k] = (sum = scale * norms[2i - 1], max = scale * norms[2i])
    end
    return report
end

# ---- end of a step (seams: update_after_step!, models.jl:983-1011; reset_state_to_previous_state!, :1068-1073) ----------
function update_after_step!(storage, model::HIPModel, dt, forces; kwarg...)
    s = storage.LinearizedSystem.eq_s
    defs = storage.variable_definitions
    pvar, svar = defs.primary_variables, defs.secondary_variables
    nloc = s.nc                    # the reference reports over ALL local cells (length(X), models.jl:1023-1038): ghosts included
    rep4 = zeros(4 * s.N)          # variable_change_report of the primary variables on the device, before state0 <- state
    @jh :jh_law_change_report (Handle, Int64, Ptr{Float64}) s.law Int64(nloc) rep4
    report = Jutul.OrderedDict{Symbol, Any}()
    for (i, k) in enumerate(keys(pvar))
        o = 4 * (i - 1)
        report[k] = (dx = (sum = rep4[o + 1], max = rep4[o + 2]), x = (sum = rep4[o + 3], max = rep4[o + 4]), n = nloc)
    end
    # Secondary variables and extra fields live on the host (Jutul evaluates them): a model that has any gets the host's primary
    # variables refreshed, its secondary variables re-evaluated from them, their change reports and their state0 <- state sync
    # exactly as in the generic method (models.jl:983-1011); a model without them (the bench laws) pays nothing here
    host_fields = !isempty(keys(svar)) || !isempty(defs.extra_variable_fields)
    if host_fields
        sync_host_state!(storage, model)
        Jutul.update_secondary_variables!(storage, model)
        for k in keys(svar)
            report[k] = Jutul.variable_change_report(storage.state[k], storage.state0[k], svar[k])
        end
    end
    # the hooks of the generic method
    update_after_step!(storage, model.domain, model, dt, forces; kwarg...)
    update_after_step!(storage, model.system, model, dt, forces; kwarg...)
    update_after_step!(storage, model.formulation, model, dt, forces; kwarg...)
    # state0 <- state: primary variables on the device (the host copy follows at the next get_output_state), the rest on the host
    @jh :jh_law_update_state0 (Handle,) s.law
    s.host_state0_stale = true
    if host_fields
        for key in keys(svar)
            Jutul.update_values!(storage.state0[key], storage.state[key])
        end
        for key in defs.extra_variable_fields
            Jutul.update_values!(storage.state0[key], storage.state[key])
        end
    end
    return report
end
function reset_state_to_previous_state!(storage, model::HIPModel)
    s = storage.LinearizedSystem.eq_s
    if s.device_state0_valid
        @jh :jh_law_reset_state (Handle,) s.law        # state <- state0 on the device (time-step cut)
        s.device_state_valid = true
        s.host_state_stale = true
    else                                               # state0 was edited on the host and not uploaded yet: host path, upload later
        invoke(reset_state_to_previous_state!, Tuple{Any, Jutul.JutulModel}, storage, model)
        s.device_state_valid = false
    end
end

# ---- distributed (seams: the PArraySimulator hooks, src/ext/partitionedarrays_ext.jl:3-33; ext/JutulPartitionedArraysExt) -------
# One process per GPU.  `bcast(bytes, root)` and `allgather(x)` are the host's collectives (MPI.bcast / MPI.Allgather from the
# MPI extension); all data-path communication then runs inside the library.
struct HIPDistributedExecutor <: Jutul.JutulExecutor
    rank::Int
    nranks::Int
end

# exclusive_devices: does every rank drive a GPU (or a CU-masked share of one) of its own?  The Krylov loop then finishes its dot
# products over the ranks inside the consuming kernels (jh_comm_set_exclusive), whose wavefronts all wait for the peers -- on a
# device shared without CU masks that wait can keep the peer off the chip.  `nothing` (default): what the library observed when the
# mailboxes were attached (PCI bus ids of the ranks' devices, jh_comm_devices_distinct); `true` on a shared device is an error of
# jh_comm_set_exclusive, not a hang.
function setup_distributed!(ctx::HIPContext, nranks::Integer, rank::Integer, bcast, allgather; exclusive_devices::Union{Nothing, Bool} = nothing)
    id = zeros(UInt8, 128)
    if rank == 0
        @jh :jh_comm_unique_id (Ptr{UInt8},) id
    end
    id = bcast(id, 0)
    @jh :jh_comm_init (Handle, Int32, Int32, Ptr{UInt8}) ctx.handle Int32(nranks) Int32(rank) id
    # scalar all-reduces of the Krylov loop through peer-mapped mailboxes when every rank of the node passes the self-test
    h = zeros(UInt8, 64)
    @jh :jh_comm_ipc_export (Handle, Ptr{UInt8}) ctx.handle h
    all = reduce(vcat, allgather(h))         # nranks*64 bytes in rank order
    ok = Ref{Int32}(0)
    @jh :jh_comm_ipc_attach (Handle, Ptr{UInt8}, Ref{Int32}) ctx.handle all ok
    everyone = minimum(allgather(ok[])) == 1
    @jh :jh_comm_ipc_enable (Handle, Int32) ctx.handle Int32(everyone)
    if everyone && isnothing(exclusive_devices)
        distinct = Ref{Int32}(-1)
        @jh :jh_comm_devices_distinct (Handle, Ref{Int32}) ctx.handle distinct
        exclusive_devices = minimum(allgather(distinct[])) == 1
    end
    exclusive = everyone && exclusive_devices === true
    xok = Ref{Int32}(0)
    if exclusive   # collective self-test of the consumer-side all-reduce: all ranks or none
        @jh :jh_comm_xrank_selftest (Handle, Ref{Int32}) ctx.handle xok
    end
    @jh :jh_comm_set_exclusive (Handle, Int32) ctx.handle Int32(exclusive && minimum(allgather(xok[])) == 1)
    info = zeros(Int64, 8)
    @jh :jh_comm_info (Handle, Ptr{Int64}) ctx.handle info
    info[1] == nranks && info[3] == nranks || error("communicator has $(info[1]) ranks ($(info[3]) in RCCL), expected $nranks")
    return HIPDistributedExecutor(rank, nranks)
end

# Halo plan of the rank-local model built by distribute_case (ext/JutulPartitionedArraysExt/utils.jl:91-148): cells are
# [owned..., ghosts...]; per neighbour rank the local owned cells to send and the local ghost cells to receive (1-based),
# then (ranks of one node, mailboxes attached) the push halo for the exchanges inside the Krylov loop.  `global_ids` = global
# cell id of every local cell (the self-test exchanges them).  The ctypes twin is jutul.jl_amd/dd.py:setup_push_halo.
function set_halo!(s::HIPConservationLawStorage, rank::Integer, neighbors::Vector{Int32}, send::Vector{Vector{Int64}},
        recv::Vector{Vector{Int64}}, global_ids::Vector{Int64}, allgather)
    sp = Int64.(cumsum([0; length.(send)])); rp = Int64.(cumsum([0; length.(recv)]))
    @jh :jh_halo_create (Handle, Int64, Int32, Ptr{Int32}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}) s.disc Int64(s.n_owned) Int32(length(neighbors)) neighbors sp reduce(vcat, send; init = Int64[]) rp reduce(vcat, recv; init = Int64[])
    h = zeros(UInt8, 64)
    @jh :jh_halo_ipc_export (Handle, Ptr{UInt8}) s.disc h
    info = allgather((h, neighbors, length.(recv)))        # every rank's (handle, neighbours, receive counts)
    hs = UInt8[]; off = Int64[]; stride = Int64[]
    for q in neighbors
        hq, nbr_q, nrecv_q = info[q + 1]
        j = findfirst(==(Int32(rank)), nbr_q)
        append!(hs, hq); push!(off, sum(nrecv_q[1:j - 1])); push!(stride, sum(nrecv_q))
    end
    ok = Ref{Int32}(0)
    @jh :jh_halo_ipc_attach (Handle, Ptr{UInt8}, Ptr{Int64}, Ptr{Int64}, Ref{Int32}) s.disc hs off stride ok
    if ok[] == 1    # ghosts must receive their owners' global cell ids
        val = [Float64(global_ids[c]) + 0.25 * (e - 1) for e in 1:s.N, c in 1:s.nc]
        val[:, s.n_owned + 1:end] .= -1.0
        v = Ref{Handle}(C_NULL)
        @jh :jh_vec_create (Handle, Ref{Handle}) s.disc v
        @jh :jh_vec_upload (Handle, Ptr{Float64}) v[] val
        expect = [Float64(global_ids[c]) + 0.25 * (e - 1) for e in 1:s.N, c in reduce(vcat, recv; init = Int64[])]
        @jh :jh_halo_ipc_selftest (Handle, Handle, Ptr{Float64}, Ref{Int32}) s.disc v[] expect ok
        @jh :jh_vec_destroy (Handle,) v[]
    end
    everyone = minimum(allgather(ok[])) == 1
    @jh :jh_halo_ipc_enable (Handle, Int32) s.disc Int32(everyone)
    return everyone
end

# parray_synchronize_primary_variables (ext/.../interface.jl:189-220): owner values -> ghost copies, on the device
synchronize_primary_variables!(s::HIPConservationLawStorage) = @jh :jh_halo_exchange_state (Handle,) s.law

# post_update_linearized_system! (models.jl:770; ext/.../overloads.jl:270-275 -> unit_diagonalize!, linalg.jl:18-35)
function post_update_linearized_system!(lsys::HIPLinearizedSystem, executor::HIPDistributedExecutor, storage, model)
    s = lsys.eq_s
    @jh :jh_unit_diagonalize (Handle, Handle, Int64) s.jac s.r Int64(s.n_owned)
end

# mpi_scalar_allreduce (ext/.../utils.jl:232-234), e.g. the convergence consensus of perform_step!(::PArraySimulator)
function scalar_allreduce!(ctx::HIPContext, values::Vector{Float64}, op::Symbol = :sum)
    @jh :jh_allreduce (Handle, Ptr{Float64}, Int32, Int32) ctx.handle values Int32(length(values)) Int32(op == :max ? 1 : 0)
    return values
end

end # module
