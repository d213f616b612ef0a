# JutulHIP.jl -- Julia-side binding of libjutul_hip.so (C ABI: include/jutul_hip.h).
#
# STATUS: written against Jutul.jl v0.4.25's dispatch seams; NOT executed in the build environment (no Julia
# there).  Every ccall below has a line-for-line twin in jutul.jl_amd/_lib.py + __init__.py, which IS executed by
# the test-suite on MI355X.  Keep the two in sync.
#
# Host code stays Julia: `simulate!` drives time stepping / Newton control as before, the three hot calls
# (`update_equation!`, `update_linearized_system_equation!`, `linear_solve!`) and the preconditioner go to the GPU.
module JutulHIP

using Jutul, LinearAlgebra
import Jutul: JutulContext, GPUJutulContext, matrix_layout, float_type, index_type, transfer, synchronize,
              setup_equation_storage, update_equation!, update_linearized_system_equation!, align_to_jacobian!,
              declare_pattern, linear_solve!, update_preconditioner!, apply!, operator_nrows, ConservationLaw,
              TwoPointPotentialFlowHardCoded, GenericKrylov, ILUZeroPreconditioner, linear_solve_return,
              BlockMajorLayout, EquationMajorLayout

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

# ---- context (seam: JutulContext, core_types.jl:86-88; precedent SingleCUDAContext, contexts/cuda.jl) ------------
mutable struct HIPContext <: GPUJutulContext
    handle::Ptr{Cvoid}
    device::Int
    matrix_layout
    block_rows::Int
    function HIPContext(device = 0; matrix_layout = EquationMajorLayout(), block_rows = 512)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        @jh :jh_context_create (Int32, Ref{Ptr{Cvoid}}) Int32(device) h
        ctx = new(h[], device, matrix_layout, block_rows)
        finalizer(c -> ccall((:jh_context_destroy, libjutul_hip), Int32, (Ptr{Cvoid},), c.handle), ctx)
        return ctx
    end
end
matrix_layout(c::HIPContext) = c.matrix_layout
float_type(::HIPContext) = Float64          # context.jl:76
index_type(::HIPContext) = Int64            # context.jl:77 (host side; the device uses 0-based Int32)
synchronize(c::HIPContext) = @jh :jh_synchronize (Ptr{Cvoid},) c.handle
transfer(::HIPContext, v) = v               # host arrays stay on the host; device copies live behind handles

# ---- equation storage (seam: setup_equation_storage, conservation.jl:137) -------------------------------------------
mutable struct HIPConservationLawStorage
    disc::Ptr{Cvoid}     # jh_tpfa
    law::Ptr{Cvoid}      # jh_law
    jac::Ptr{Cvoid}      # jh_csr
    r::Ptr{Cvoid}        # jh_vec
    dx::Ptr{Cvoid}       # jh_vec
    nc::Int
    N::Int
    dt::Float64
end

function setup_equation_storage(model::SimulationModel{<:Any, <:Any, <:Any, HIPContext},
        eq::ConservationLaw{<:Any, <:TwoPointPotentialFlowHardCoded, <:Any, <:Any}, storage; kind = 0, params = nothing, kwarg...)
    ctx = model.context
    N = Jutul.get_neighborship(model.domain.representation)      # 2 x nf, Int64, 1-based (as stored)
    nc = Jutul.number_of_cells(model.domain)
    ne = Jutul.number_of_equations_per_entity(model, eq)
    disc = Ref{Ptr{Cvoid}}(C_NULL)
    @jh :jh_tpfa_create (Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Int32, Int32, Ptr{Int64}, Int64, Int64, Ref{Ptr{Cvoid}}) ctx.handle nc size(N, 2) N Int32(ne) Int32(1) C_NULL ctx.block_rows 0 disc
    law = Ref{Ptr{Cvoid}}(C_NULL)
    @jh :jh_law_create (Ptr{Cvoid}, Int32, Ptr{Float64}, Ref{Ptr{Cvoid}}) disc[] Int32(kind) (isnothing(params) ? C_NULL : params) law
    jac = Ref{Ptr{Cvoid}}(C_NULL); r = Ref{Ptr{Cvoid}}(C_NULL); dx = Ref{Ptr{Cvoid}}(C_NULL)
    @jh :jh_csr_create (Ptr{Cvoid}, Ref{Ptr{Cvoid}}) disc[] jac
    @jh :jh_vec_create (Ptr{Cvoid}, Ref{Ptr{Cvoid}}) disc[] r
    @jh :jh_vec_create (Ptr{Cvoid}, Ref{Ptr{Cvoid}}) disc[] dx
    return HIPConservationLawStorage(disc[], law[], jac[], r[], dx[], nc, ne, NaN)
end

# ---- linearized system (seam: setup_linearized_system!, models.jl:654-668; LinearizedSystem, linsolve/default.jl:34-42) -----
# The Jacobian / residual / increment live on the device behind the equation storage's handles; this wrapper is what
# `storage[:LinearizedSystem]` holds so that `linear_solve!(sys, ...)` dispatches here.
struct HIPLinearizedSystem <: Jutul.JutulLinearSystem
    eq_s::HIPConservationLawStorage
    r_buffer::Vector{Float64}    # host mirrors, filled only on request (check_convergence on the host, debugging)
    dx_buffer::Vector{Float64}
end

function Jutul.setup_linearized_system!(storage, model::SimulationModel{<:Any, <:Any, <:Any, HIPContext})
    eq_s = first(values(storage[:equations]))::HIPConservationLawStorage   # single conservation law per model on this path
    n = eq_s.nc * eq_s.N
    lsys = HIPLinearizedSystem(eq_s, zeros(n), zeros(n))
    storage[:LinearizedSystem] = lsys
    return lsys
end
Jutul.align_equations_to_linearized_system!(storage, model::SimulationModel{<:Any, <:Any, <:Any, HIPContext}; kwarg...) = nothing

# pattern / alignment: the library owns the device pattern; the host tables are available bit-exact if Jutul needs
# them (conservation.jl:486-505, :143-216)
function declare_pattern(model, eq::ConservationLaw, s::HIPConservationLawStorage, ::Cells)
    nnzb = Ref{Int64}(0)
    @jh :jh_tpfa_sizes (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ref{Int64}, Ptr{Int32}) s.disc C_NULL C_NULL C_NULL nnzb C_NULL
    rowptr = Vector{Int64}(undef, s.nc + 1); colidx = Vector{Int64}(undef, nnzb[])
    @jh :jh_tpfa_get_pattern (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}) s.disc rowptr colidx
    I = similar(colidx)
    for row in 1:s.nc, k in rowptr[row]:(rowptr[row + 1] - 1)
        I[k] = row
    end
    return (I, colidx)
end
align_to_jacobian!(s::HIPConservationLawStorage, eq::ConservationLaw, jac, model, u::Cells; kwarg...) = nothing

# ---- assembly (seams: update_equation!, conservation.jl:572; update_linearized_system_equation!, :298) ---------------
function update_equation!(s::HIPConservationLawStorage, law::ConservationLaw, storage, model, dt)
    X = Jutul.vectorize_variables(model, storage.state, :primary)        # [N, nc] block-major
    @jh :jh_law_set_state (Ptr{Cvoid}, Ptr{Float64}) s.law X
    s.dt = dt
    return nothing  # the flux + fill are fused into one kernel launched by update_linearized_system_equation!
end

function update_linearized_system_equation!(nz, r, model, law::ConservationLaw, s::HIPConservationLawStorage)
    @jh :jh_assemble (Ptr{Cvoid}, Float64, Ptr{Cvoid}, Ptr{Cvoid}) s.law s.dt s.jac s.r
    # host copies only if the caller insists on host buffers (parity / debugging); the solve reads device memory
    if !isnothing(nz)
        @jh :jh_csr_get_values (Ptr{Cvoid}, Ptr{Float64}) s.jac nz
    end
    if !isnothing(r)
        @jh :jh_vec_download (Ptr{Cvoid}, Ptr{Float64}) s.r r
    end
end

# ---- preconditioner (seams: update_preconditioner!, precond/ilu.jl:37; apply!, :62) -----------------------------------
mutable struct HIPILUZero <: Jutul.JutulPreconditioner
    handle::Ptr{Cvoid}
    dim
    HIPILUZero() = new(C_NULL, nothing)
end
function update_preconditioner!(p::HIPILUZero, s::HIPConservationLawStorage, b, context::HIPContext, executor)
    if p.handle == C_NULL
        h = Ref{Ptr{Cvoid}}(C_NULL)
        @jh :jh_ilu0_create (Ptr{Cvoid}, Ptr{Int64}, Int64, Ref{Ptr{Cvoid}}) s.jac C_NULL -1 h   # device blocks
        p.handle = h[]
        p.dim = (s.nc * s.N, s.nc * s.N)
    end
    @jh :jh_ilu0_factor (Ptr{Cvoid},) p.handle
end
operator_nrows(p::HIPILUZero) = p.dim[1]

# ---- linear solve (seam: linear_solve!, linsolve/krylov.jl:71-85) -------------------------------------------------------
mutable struct HIPKrylov
    handle::Ptr{Cvoid}
    HIPKrylov() = new(C_NULL)
end
function linear_solve!(sys::HIPLinearizedSystem, krylov::GenericKrylov, context::HIPContext, model, storage = nothing,
        dt = nothing, recorder = nothing, executor = nothing; dx = nothing, r = nothing,
        atol = Jutul.linear_solver_tolerance(krylov, :absolute), rtol = Jutul.linear_solver_tolerance(krylov, :relative), kwarg...)
    s = sys.eq_s
    prec = krylov.preconditioner::HIPILUZero
    t_prec = @elapsed update_preconditioner!(prec, s, nothing, context, executor)
    ws = krylov.storage
    if !(ws isa HIPKrylov)
        ws = HIPKrylov(); h = Ref{Ptr{Cvoid}}(C_NULL)
        @jh :jh_krylov_create (Ptr{Cvoid}, Ref{Ptr{Cvoid}}) s.jac h
        ws.handle = h[]; krylov.storage = ws
    end
    cfg = krylov.config
    side = cfg.precond_side == :right ? Int32(2) : Int32(1)
    iters = Ref{Int64}(0); status = Ref{Int32}(0)
    hist = zeros(cfg.max_iterations + 2)
    x = s.dx  # solution lands in dx, then negated in place (update_dx_from_vector!, default.jl:444-446)
    if krylov.solver == :gmres   # the reference's second Krylov method (linsolve/krylov.jl:214-218)
        @jh :jh_gmres (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Float64, Int64, Ref{Int64}, Ref{Int32}, Ptr{Float64}, Int64) ws.handle prec.handle side s.r x rtol atol cfg.max_iterations iters status hist length(hist)
    else
        @jh :jh_bicgstab (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Float64, Int64, Ref{Int64}, Ref{Int32}, Ptr{Float64}, Int64) ws.handle prec.handle side s.r x rtol atol cfg.max_iterations iters status hist length(hist)
    end
    @jh :jh_vec_negate_into (Ptr{Cvoid}, Ptr{Cvoid}) s.dx x
    n = iters[]
    solved = status[] == 0
    if !solved && n > 0 && hist[n + 1] / hist[1] > 1.0
        error("Bad linear solve: final residual $(hist[n + 1]), rel. value $(hist[n + 1] / hist[1])")   # krylov.jl:161-166
    end
    if !isnothing(dx)
        @jh :jh_vec_download (Ptr{Cvoid}, Ptr{Float64}) s.dx dx
    end
    return linear_solve_return(solved, n, (residuals = hist[1:n + 1], solved = solved); prepare = t_prec)
end

# ---- distributed set-up (seam: the PArraySimulator hooks, src/ext/partitionedarrays_ext.jl:3-33) ---------------------------
# One process per GPU.  `bcast(bytes, root)` and `allgather(bytes)` are the host's collectives (MPI.Bcast! / MPI.Allgather from
# the MPI extension); all data-path communication then runs inside the library.
function setup_distributed!(ctx::HIPContext, nranks::Integer, rank::Integer, bcast, allgather)
    id = zeros(UInt8, 128)
    rank == 0 && @jh :jh_comm_unique_id (Ptr{UInt8},) id
    id = bcast(id, 0)
    @jh :jh_comm_init (Ptr{Cvoid}, Int32, Int32, Ptr{UInt8}) ctx.handle Int32(nranks) Int32(rank) id
    # scalar all-reduces of the Krylov loop through peer-mapped mailboxes when every rank of the node passes the self-test
    h = zeros(UInt8, 64)
    @jh :jh_comm_ipc_export (Ptr{Cvoid}, Ptr{UInt8}) ctx.handle h
    all = allgather(h)                       # nranks*64 bytes in rank order
    ok = Ref{Int32}(0)
    @jh :jh_comm_ipc_attach (Ptr{Cvoid}, Ptr{UInt8}, Ref{Int32}) ctx.handle all ok
    everyone = minimum(reinterpret(Int32, allgather(collect(reinterpret(UInt8, [ok[]]))))) == 1
    @jh :jh_comm_ipc_enable (Ptr{Cvoid}, Int32) ctx.handle Int32(everyone)
    return ctx
end

# Halo plan of the rank-local model built by distribute_case (ext/JutulPartitionedArraysExt/utils.jl:91-148): cells are
# [owned..., ghosts...]; per neighbour rank the local owned cells to send and the local ghost cells to receive (1-based).
function set_halo!(disc::Ptr{Cvoid}, n_owned::Integer, neighbors::Vector{Int32}, send::Vector{Vector{Int64}}, recv::Vector{Vector{Int64}})
    sp = cumsum([0; length.(send)]); rp = cumsum([0; length.(recv)])
    @jh :jh_halo_create (Ptr{Cvoid}, Int64, Int32, Ptr{Int32}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}) disc Int64(n_owned) Int32(length(neighbors)) neighbors Int64.(sp) reduce(vcat, send; init = Int64[]) Int64.(rp) reduce(vcat, recv; init = Int64[])
    # Optional (ranks of one node, mailboxes enabled): the push halo for the exchanges inside the Krylov loop.  All-gather
    # (jh_halo_ipc_export handle, neighbors, length.(recv)) of every rank; per neighbour q pass its handle, the offset of this
    # rank's segment in q's receive order and q's total receive count to jh_halo_ipc_attach; verify with jh_halo_ipc_selftest
    # (a vector holding global cell ids); jh_halo_ipc_enable with the AND over all ranks.  The ctypes twin of this sequence is
    # jutul.jl_amd/dd.py:setup_push_halo.
end

end # module
