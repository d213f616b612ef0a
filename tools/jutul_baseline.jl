# Baseline B of BASELINE.md / SURVEY.md §8(d): Jutul.jl ITSELF on the host cores -- the time-dependent VariablePoisson system
# (src/applications/test_systems/variable_poisson/variable_poisson.jl) on a CartesianMesh, assembled into StaticCSR by
# ParallelCSRContext (src/core_types/contexts/csr.jl:3-23) and solved with GenericKrylov(:bicgstab) + ILUZeroPreconditioner
# (src/linsolve/krylov.jl:27-58, src/linsolve/precond/ilu.jl), timed with Jutul's own timing_breakdown (src/utils.jl:893-951).
# bench.py runs it when `julia` and an installed Jutul are found (cpu_baseline.jutul_itself); it cannot run in the build image
# (no julia, no network).  usage: julia -t <threads> tools/jutul_baseline.jl <cells> <timesteps> <dt> <rtol>
# Prints ONE JSON line: Newton iterations (linear solves) per second and the assembly / solve split.
using Jutul

function main(args)
    cells = length(args) >= 1 ? parse(Int, args[1]) : 1_000_000
    nstep = length(args) >= 2 ? parse(Int, args[2]) : 10
    dt    = length(args) >= 3 ? parse(Float64, args[3]) : 5.0
    rtol  = length(args) >= 4 ? parse(Float64, args[4]) : 1e-3
    n = max(2, round(Int, cbrt(cells)))
    g = CartesianMesh((n, n, n), (1.0, 1.0, 1.0))
    nc = number_of_cells(g)
    sys = VariablePoissonSystem(time_dependent = true)
    domain = DataDomain(g, poisson_coefficient = 1.0)
    nt = Threads.nthreads()
    ctx = ParallelCSRContext(nt)     # CSR backend, block-Jacobi partition of the rows over the threads
    model = SimulationModel(domain, sys, context = ctx)
    state0 = setup_state(model, U = 1.0)
    param = setup_parameters(model)
    forces = setup_forces(model, sources = [PoissonSource(1, 1.0), PoissonSource(nc, -1.0)])
    case = JutulCase(model, fill(dt, nstep), forces, state0 = state0, parameters = param)
    lsolve = GenericKrylov(:bicgstab, preconditioner = ILUZeroPreconditioner(), relative_tolerance = rtol, max_iterations = 100)
    # one warm-up run (compilation), then the timed one
    simulate(JutulCase(model, [dt], forces, state0 = state0, parameters = param), linear_solver = lsolve, info_level = -1)
    t0 = time()
    states, reports = simulate(case, linear_solver = lsolve, info_level = -1)
    wall = time() - t0
    tb = Jutul.timing_breakdown(reports)
    lin_its = 0
    for r in reports, ms in r[:ministeps]
        haskey(ms, :steps) || continue
        for s in ms[:steps]
            if haskey(s, :linear_iterations)
                lin_its += s[:linear_iterations]
            end
        end
    end
    println("{\"kind\": \"reference\", \"cells\": $nc, \"mesh\": \"CartesianMesh($n,$n,$n)\", \"threads\": $nt, \"timesteps\": $nstep, ",
            "\"newton_iterations\": $(tb.its), \"assemblies\": $(tb.no_asm), \"linear_iterations\": $lin_its, \"wall_s\": $wall, ",
            "\"assembly_s\": $(tb.assembly), \"solve_s\": $(tb.solve), \"newton_iterations_per_s\": $(tb.its / wall)}")
end

main(ARGS)
