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
#   update_equation!            conservation.jl:572       -> state + state0 upload (state_pair, :549-555), sources reset
#   get_diagonal_entries        conservation.jl:667-674   -> host accumulator the forces add into (apply_forces!, models.jl:889-901)
#   update_linearized_system_equation!  conservation.jl:298 -> sources upload + ONE fused flux / accumulation / fill kernel
#   convergence_criterion       equations.jl:619-629      -> max |r_e| on the device (check_convergence, models.jl:830-883)
#   linear_solve!               linsolve/krylov.jl:71-182 -> ILU(0) refactor + device BiCGStab / GMRES, dx = -x
#   update_primary_variables!   models.jl:928-953         -> clamp chain on the device, state downloaded into storage.state
#   update_after_step!          models.jl:983-1011        -> state0 <- state on the device
module JutulHIP

using Jutul, LinearAlgebra
import Jutul: JutulContext, GPUJutulContext, matrix_layout, float_type, index_type, transfer, synchronize,
              setup_equation_storage, update_equation!, update_linearized_system_equation!, align_to_jacobian!,
              declare_pattern, get_diagonal_entries, convergence_criterion, update_primary_variables!, update_after_step!,
              reset_state_to_previous_state!, post_update_linearized_system!,
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
    function HIPContext(device = 0; matrix_layout = BlockMajorLayout(), block_rows = 0, n_owned = 0)
        h = Ref{Handle}(C_NULL)
        @jh :jh_context_create (Int32, Ref{Handle}) Int32(device) h
        ctx = new(h[], device, matrix_layout, block_rows, n_owned)
        finalizer(c -> ccall((:jh_context_destroy, libjutul_hip), Int32, (Handle,), c.handle), ctx)
        return ctx
    end
end
matrix_layout(c::HIPContext) = c.matrix_layout
float_type(::HIPContext) = Float64          # context.jl:76
index_type(::HIPContext) = Int64            # context.jl:77 (host side; the device uses 0-based Int32)
synchronize(c::HIPContext) = @jh :jh_synchronize (Handle,) c.handle
transfer(::HIPContext, v) = v               # host arrays stay on the host; device copies live behind handles

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
        @. v = X[i, :]        # values only: the device owns the derivatives
    end
    return state
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
    X::Matrix{Float64}         # [N, nc] staging of the primary variables
    sources::Matrix{Float64}   # [N, nc] what apply_forces! adds to the diagonal entries (values only)
    err::Vector{Float64}
end

function setup_equation_storage(model::HIPModel,
        eq::ConservationLaw{<:Any, <:TwoPointPotentialFlowHardCoded, <:Any, <:Any}, storage; kwarg...)
    ctx = model.context
    N = Jutul.get_neighborship(model.domain.representation)      # 2 x nf, Int64, 1-based (as stored)
    nc = Jutul.number_of_cells(model.domain)
    ne = Jutul.number_of_equations_per_entity(model, eq)
    disc = Ref{Handle}(C_NULL)
    @jh :jh_tpfa_create (Handle, Int64, Int64, Ptr{Int64}, Int32, Int32, Ptr{Int64}, Int64, Int64, Ref{Handle}) ctx.handle nc size(N, 2) N Int32(ne) Int32(1) C_NULL ctx.block_rows ctx.n_owned disc
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
    return HIPConservationLawStorage(disc[], law[], jac[], r[], dx[], nc, ne, n_owned, NaN, zeros(ne, nc), zeros(ne, nc), zeros(ne))
end

# ---- linearized system (seam: setup_linearized_system!, models.jl:654-668; LinearizedSystem, linsolve/default.jl:34-42) -----
# The Jacobian / residual / increment live on the device behind the equation storage's handles; this wrapper is what
# `storage[:LinearizedSystem]` holds so that `linear_solve!(sys, ...)` dispatches here.  dx_buffer is the host mirror
# linear_solve! fills (its default `dx = sys.dx_buffer`, krylov.jl:79) and storage.views.primary_variables points into.
struct HIPLinearizedSystem <: Jutul.JutulLinearSystem
    eq_s::HIPConservationLawStorage
    r_buffer::Vector{Float64}
    dx_buffer::Vector{Float64}
end

function Jutul.setup_linearized_system!(storage, model::HIPModel)
    eq_s = first(values(storage[:equations]))::HIPConservationLawStorage   # single conservation law per model on this path
    n = eq_s.nc * eq_s.N
    lsys = HIPLinearizedSystem(eq_s, zeros(n), zeros(n))
    storage[:LinearizedSystem] = lsys
    return lsys
end
Jutul.align_equations_to_linearized_system!(storage, model::HIPModel; kwarg...) = nothing
# update_linearized_system! hands `nzval = lsys.jac_buffer` and a residual view to every equation (models.jl:774-783): there
# are no host buffers to fill on this path
Jutul.update_linearized_system!(lsys::HIPLinearizedSystem, equations, eqs_storage, eqs_views, model::HIPModel; kwarg...) =
    for key in keys(equations)
        update_linearized_system_equation!(nothing, nothing, model, equations[key], eqs_storage[key])
    end

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
    # state_pair (conservation.jl:549-555): the accumulation needs state AND state0
    @jh :jh_law_set_state (Handle, Ptr{Float64}) s.law pack_primary!(s.X, model, storage.state)
    @jh :jh_law_set_state0 (Handle, Ptr{Float64}) s.law pack_primary!(s.X, model, storage.state0)
    fill!(s.sources, 0.0)                  # reset_sources! (conservation.jl:660-664)
    s.dt = dt
    return nothing  # the flux + fill are fused into one kernel launched by update_linearized_system_equation!
end

# apply_forces! (models.jl:889-901) adds every force to `get_diagonal_entries(eq, eq_s)`, e.g. `d[c] += f.value` for a
# PoissonSource (variable_poisson.jl:78-84).  Here that array is a host accumulator; it reaches the device with the assembly.
get_diagonal_entries(eq::ConservationLaw, s::HIPConservationLawStorage) = s.N == 1 ? vec(s.sources) : s.sources

function update_linearized_system_equation!(nz, r, model, law::ConservationLaw, s::HIPConservationLawStorage)
    cells = Int64[c for c in 1:s.nc if any(!iszero, view(s.sources, :, c))]
    @jh :jh_law_set_sources (Handle, Int64, Ptr{Int64}, Ptr{Float64}) s.law length(cells) cells s.sources[:, cells]
    @jh :jh_assemble (Handle, Float64, Handle, Handle) s.law s.dt s.jac s.r
    # host copies only if the caller insists on host buffers (parity / debugging); the solve reads device memory
    if !isnothing(nz)
        @jh :jh_csr_get_values (Handle, Ptr{Float64}) s.jac nz
    end
    if !isnothing(r)
        @jh :jh_vec_download (Handle, Ptr{Float64}) s.r r
    end
end

# ---- convergence (seam: convergence_criterion, equations.jl:619-629, called by check_convergence, models.jl:830-883) ------
function convergence_criterion(model::HIPModel, storage, eq::ConservationLaw, s::HIPConservationLawStorage, r; dt = 1.0, update_report = missing)
    @jh :jh_convergence (Handle, Handle, Int64, Ptr{Float64}) s.law s.r Int64(s.n_owned) s.err
    names = s.N == 1 ? "R" : map(i -> "R_$i", 1:s.N)
    return (AbsMax = (errors = copy(s.err), names = names), )
end

# ---- preconditioner (seams: update_preconditioner!, precond/ilu.jl:37; apply!, :62) -----------------------------------
mutable struct HIPILUZero <: Jutul.JutulPreconditioner
    handle::Handle
    dim
    HIPILUZero() = new(C_NULL, nothing)
end
function update_preconditioner!(p::HIPILUZero, s::HIPConservationLawStorage, b, context::HIPContext, executor)
    if p.handle == C_NULL
        h = Ref{Handle}(C_NULL)
        @jh :jh_ilu0_create (Handle, Ptr{Int64}, Int64, Ref{Handle}) s.jac C_NULL -1 h   # device blocks
        p.handle = h[]
        p.dim = (s.nc * s.N, s.nc * s.N)
    end
    @jh :jh_ilu0_factor (Handle,) p.handle
end
operator_nrows(p::HIPILUZero) = p.dim[1]

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
    prec = krylov.preconditioner::HIPILUZero
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
    if !isnothing(dx)          # dx .= -x for everything downstream that reads the host increment (increment norms, reports)
        @jh :jh_vec_download (Handle, Ptr{Float64}) s.dx dx
    end
    return linear_solve_return(solved, n, (residuals = hist[1:n + 1], solved = solved); prepare = t_prec)
end

# ---- primary update (seam: update_primary_variables!, models.jl:928-953; choose_increment chain, variables/utils.jl:110-174)
function update_primary_variables!(storage, model::HIPModel; relaxation = 1.0, check = false, kwarg...)
    s = storage.LinearizedSystem.eq_s
    dxh = reshape(storage.LinearizedSystem.dx_buffer, s.N, s.nc)
    if check && !all(isfinite, dxh)
        error("Primary variables recieved invalid updates.")
    end
    lim = hip_update_limits(model)
    @jh :jh_update_primary (Handle, Handle, Float64, Ptr{Float64}) s.law s.dx Float64(relaxation) (isnothing(lim) ? C_NULL : lim)
    @jh :jh_law_get_state (Handle, Ptr{Float64}) s.law s.X
    unpack_primary!(storage.state, model, s.X)
    report = Dict{Symbol, Any}()   # increment_norm (models.jl:955-965)
    for (i, k) in enumerate(keys(Jutul.get_primary_variables(model)))
        report[k] = (sum = sum(abs, view(dxh, i, :)), max = maximum(abs, view(dxh, i, :)))
    end
    return report
end

# ---- end of a step (seams: update_after_step!, models.jl:983-1011; reset_state_to_previous_state!, :1068-1073) ----------
function update_after_step!(storage, model::HIPModel, dt, forces; kwarg...)
    s = storage.LinearizedSystem.eq_s
    @jh :jh_law_update_state0 (Handle,) s.law
    rep = invoke(update_after_step!, Tuple{Any, Jutul.JutulModel, Any, Any}, storage, model, dt, forces; kwarg...)  # host state0 <- state
    return rep
end
function reset_state_to_previous_state!(storage, model::HIPModel)
    s = storage.LinearizedSystem.eq_s
    @jh :jh_law_reset_state (Handle,) s.law
    invoke(reset_state_to_previous_state!, Tuple{Any, Jutul.JutulModel}, storage, model)
end

# ---- distributed (seams: the PArraySimulator hooks, src/ext/partitionedarrays_ext.jl:3-33; ext/JutulPartitionedArraysExt) -------
# One process per GPU.  `bcast(bytes, root)` and `allgather(x)` are the host's collectives (MPI.bcast / MPI.Allgather from the
# MPI extension); all data-path communication then runs inside the library.
struct HIPDistributedExecutor <: Jutul.JutulExecutor
    rank::Int
    nranks::Int
end

function setup_distributed!(ctx::HIPContext, nranks::Integer, rank::Integer, bcast, allgather)
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
