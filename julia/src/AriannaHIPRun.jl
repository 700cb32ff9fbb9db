# AriannaHIPRun.jl -- the reference's run! loop with look-ahead over the schedulers (part of module AriannaHIP).
#
# NOT EXECUTED IN THIS REPOSITORY'S ENVIRONMENT (no Julia in the build image or on the GPU box).  It restates, call for call,
# what the Python host mirror does and what the GPU tests measure (montecarlo_amd/simulation.py `run(fuse=True)`,
# policy_guided.py `make_steps_grouped`, simulation.py StoreCallbacks / StoreParameters with `defer`), against the reference's
# own loop:
#   run!                                  src/simulation.jl:175-204  (initialise all; for t: make_step! of whoever is due; finalise)
#   StoreCallbacks                        src/algorithms.jl:62-109   ("$(t) $(callback(simulation))" per callback and scheduled t)
#   StoreParameters                       src/metropolis.jl:380-450  ("$(t) $(collect(parameters))" per stored move)
#   PolicyGradientEstimator / Update      src/PolicyGuided/estimator.jl:111-134, update.jl:50-57
# Arianna's run! calls make_step!(::HIPMetropolis) once per t even when nobody looks at the state in between (config 1: callbacks
# every 10 t after a burn-in of 1000).  run_fused! issues
#   * a stretch of sweeps nobody observes as ONE amc_sweep(h, n)                               (the state stays in registers)
#   * the sweep a callback observes as amc_sweep_reduce_begin                                  (its sums are formed in that launch)
#   * stretches of [HIPMetropolis, HIPPolicyGradientEstimator(, HIPPolicyGradientUpdate)] as ONE amc_pgmc_steps: one launch per
#     time step, gradients_data and the learning step on the device, no host call per step
# and its own HIPStoreCallbacks / HIPStoreParameters write the row of time t when the NEXT scheduled time comes: the sums / the
# parameters of t are formed (copied) on the device in stream order at t, the host reads them a period later and never drains
# the queue of launches.  Rows, values and their order in the files are those of the reference's algorithms.
# tests/test_julia_binding_static.py checks every ccall of this file against include/amc.h like those of AriannaHIP.jl.

# ---- reductions in flight --------------------------------------------------------------------------------------------------
# The engine keeps up to two reductions in flight and hands them back oldest first (amc_reduce_begin .. amc_reduce_end_exact).
"A queued reduction: `fetch!(ticket)` returns [Σe, Σx, Σx², count, Σ acc/tot per move] summed over the shards."
mutable struct ReductionTicket
    metropolis::HIPMetropolis
    t::Int
    value::Union{Nothing,Vector{Float64}}
end

const REDUCE_E, REDUCE_X, REDUCE_XX, REDUCE_ALL = Cint(1), Cint(2), Cint(4), Cint(7)

"Which of Σe / Σx / Σx² the reductions begun from now on form (amc_set_reduce_columns); a sum that is not formed reads NaN."
function set_reduce_columns!(metropolis::HIPMetropolis, columns::Integer)
    check(ccall((:amc_set_reduce_columns, libamc), Cint, (Ptr{Cvoid}, Cint), metropolis.handle, columns))
    return nothing
end

# tickets whose sums still sit in the engine, oldest first, per handle
const INFLIGHT = IdDict{HIPMetropolis,Vector{ReductionTicket}}()
inflight(m::HIPMetropolis) = get!(() -> ReductionTicket[], INFLIGHT, m)

function fetch!(ticket::ReductionTicket)
    ticket.value === nothing || return ticket.value
    q = inflight(ticket.metropolis)
    while !isempty(q)                                   # the engine hands reductions back oldest first
        head = popfirst!(q)
        records = Matrix{Float64}(undef, XSUM_WORDS, 4 + head.metropolis.K)
        steps = Ref{UInt64}(0)
        check(ccall((:amc_reduce_end_exact, libamc), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ref{UInt64}), head.metropolis.handle, records, steps))
        head.value = finish_records!(head.metropolis, records, steps[])
        head === ticket && break
    end
    return ticket.value
end

"Fetch the oldest tickets until at most `keep` reductions are left in the engine (it takes two)."
function settle!(m::HIPMetropolis, keep::Int=0)
    q = inflight(m)
    while length(q) > keep
        fetch!(q[1])
    end
    return nothing
end

# the reduction of the state at simulation.t as a ticket; `pending_t`: a launch has formed the sums of that t already
# (make_step_observed! / fuse_pgmc!), so the ticket only claims them
const PENDING_T = IdDict{HIPMetropolis,Int}()

function reduction_ticket(simulation, m::HIPMetropolis)
    q = inflight(m)
    !isempty(q) && q[end].t == simulation.t && return q[end]          # every callback of this t shares one reduction
    ticket = ReductionTicket(m, simulation.t, nothing)
    if get(PENDING_T, m, -1) == simulation.t
        delete!(PENDING_T, m)
    else
        drop_pending!(m)
        settle!(m, 1)
        check(ccall((:amc_reduce_begin, libamc), Cint, (Ptr{Cvoid},), m.handle))
    end
    push!(q, ticket)
    return ticket
end

# sums that a launch formed and nobody claimed (a callback list that changed its mind): fetch and discard
function drop_pending!(m::HIPMetropolis)
    haskey(PENDING_T, m) || return nothing
    settle!(m)
    records = Matrix{Float64}(undef, XSUM_WORDS, 4 + m.K)
    steps = Ref{UInt64}(0)
    check(ccall((:amc_reduce_end_exact, libamc), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ref{UInt64}), m.handle, records, steps))
    delete!(PENDING_T, m)
    return nothing
end

# ---- callbacks that know how to wait ----------------------------------------------------------------------------------------
# A deferred callback is a pair (name, needs, pick): `pick(r)` turns the fetched reduction into the callback's value.
struct DeferredCallback{F}
    name::String
    needs::Cint                 # REDUCE_* bits of the sums over x it reads
    pick::F
end
const deferred_energy = DeferredCallback("energy", REDUCE_E, r -> r[1] / r[4])                      # particle_1d.jl:68-70
const deferred_acceptance = DeferredCallback("acceptance", Cint(0), r -> r[5:end] ./ r[4])          # metropolis.jl:319-321
const deferred_moments = DeferredCallback("moments", REDUCE_X | REDUCE_XX, r -> [r[2] / r[4], r[3] / r[4]])   # distribution_test.jl:36-37

"""
    HIPStoreCallbacks(chains; callbacks=(deferred_energy, deferred_acceptance), path, store_first=true, store_last=false)

StoreCallbacks (src/algorithms.jl:62-109) for the engine-backed callbacks: same files, same rows, written one scheduled time
late -- the reduction of time t is queued at t and fetched when the next row is due (or at finalise).
"""
mutable struct HIPStoreCallbacks <: Arianna.AriannaAlgorithm
    callbacks::Vector{DeferredCallback}
    paths::Vector{String}
    files::Vector{IOStream}
    store_first::Bool
    store_last::Bool
    pending::Union{Nothing,Tuple{Int,ReductionTicket}}
end

function HIPStoreCallbacks(chains; path=missing, callbacks=(deferred_energy, deferred_acceptance), store_first=true, store_last=false, extras...)
    mkpath(path)
    cbs = collect(DeferredCallback, callbacks)
    paths = [joinpath(path, cb.name * ".dat") for cb in cbs]
    return HIPStoreCallbacks(cbs, paths, Vector{IOStream}(undef, length(paths)), store_first, store_last, nothing)
end

reduction_needs(alg::HIPStoreCallbacks) = foldl(|, (cb.needs for cb in alg.callbacks); init=Cint(0))

function flush_row!(alg::HIPStoreCallbacks)
    alg.pending === nothing && return nothing
    t, ticket = alg.pending
    alg.pending = nothing
    r = fetch!(ticket)
    for (cb, file) in zip(alg.callbacks, alg.files)
        println(file, "$(t) $(cb.pick(r))")                                    # algorithms.jl:99
        flush(file)
    end
    return nothing
end

function initialise(alg::HIPStoreCallbacks, simulation::Simulation)
    alg.files .= open.(alg.paths, "w")
    alg.store_first && make_step!(simulation, alg)                             # algorithms.jl:93 (the t = 0 row: acceptance NaN)
    return nothing
end

function make_step!(simulation::Simulation, alg::HIPStoreCallbacks)
    ticket = reduction_ticket(simulation, hip_algorithm(simulation))           # claims the sums of the state at this t
    flush_row!(alg)                                                            # the previous row: its sums are long there
    alg.pending = (simulation.t, ticket)
    return nothing
end

function finalise(alg::HIPStoreCallbacks, simulation::Simulation)
    alg.store_last && make_step!(simulation, alg)
    flush_row!(alg)
    close.(alg.files)
    return nothing
end

function write_algorithm(io, alg::HIPStoreCallbacks, scheduler)
    println(io, "\tStoreCallbacks (rows written one scheduled time late)")
    println(io, "\t\tCalls: $(length(filter(x -> 0 < x ≤ scheduler[end], scheduler)))")
end

"""
    HIPStoreParameters(chains; dependencies=(HIPMetropolis,), path, ids, store_first=true, store_last=false)

StoreParameters (src/metropolis.jl:380-450) beside device-resident learning steps: the read of time t is a copy queued in stream
order (amc_parameters_begin), its row "\$(t) \$(collect(parameters))" is written when the next scheduled time comes.
"""
mutable struct HIPStoreParameters <: Arianna.AriannaAlgorithm
    metropolis::HIPMetropolis
    ids::Vector{Int}
    paths::Vector{String}
    files::Vector{IOStream}
    store_first::Bool
    store_last::Bool
    pending_t::Int              # -1: no read in flight
end

function HIPStoreParameters(chains; dependencies=missing, path=missing, ids=missing, store_first=true, store_last=false, extras...)
    @assert length(dependencies) == 1
    @assert isa(dependencies[1], HIPMetropolis)
    metropolis = dependencies[1]
    ids = ismissing(ids) ? collect(1:metropolis.K) : collect(Int, ids)
    dirs = joinpath.(path, "parameters", ["$k" for k in ids])
    mkpath.(dirs)
    paths = joinpath.(dirs, "parameters.dat")
    return HIPStoreParameters(metropolis, ids, paths, Vector{IOStream}(undef, length(paths)), store_first, store_last, -1)
end

function flush_row!(alg::HIPStoreParameters)
    alg.pending_t < 0 && return nothing
    m = alg.metropolis
    θ = Matrix{Float64}(undef, m.n_params, m.K)                               # parameters[k * P + p]: column k holds move k's vector
    check(ccall((:amc_parameters_end_all, libamc), Cint, (Ptr{Cvoid}, Ptr{Float64}, Cint), m.handle, θ, length(θ)))
    for (k, file) in zip(alg.ids, alg.files)
        println(file, "$(alg.pending_t) $(θ[:, k])")                           # metropolis.jl:440
        flush(file)
    end
    alg.pending_t = -1
    return nothing
end

function initialise(alg::HIPStoreParameters, simulation::Simulation)
    alg.files .= open.(alg.paths, "w")
    alg.store_first && make_step!(simulation, alg)
    return nothing
end

function make_step!(simulation::Simulation, alg::HIPStoreParameters)
    flush_row!(alg)                                                            # one read in flight per handle
    check(ccall((:amc_parameters_begin, libamc), Cint, (Ptr{Cvoid},), alg.metropolis.handle))
    alg.pending_t = simulation.t
    return nothing
end

function finalise(alg::HIPStoreParameters, simulation::Simulation)
    alg.store_last && make_step!(simulation, alg)
    flush_row!(alg)
    close.(alg.files)
    pull_parameters!(alg.metropolis)                                           # the shared Move.parameters objects, up to date at the end
    return nothing
end

function write_algorithm(io, alg::HIPStoreParameters, scheduler)
    println(io, "\tStoreParameters (rows written one scheduled time late)")
    println(io, "\t\tCalls: $(length(filter(x -> 0 < x ≤ scheduler[end], scheduler)))")
end

# ---- device-resident estimator / update as algorithms of the list ----------------------------------------------------------
optimiser_code(::PolicyGuided.Static) = (Cint(0), 0.0, 0.0)
optimiser_code(o::PolicyGuided.VPG) = (Cint(1), Float64(o.η), 0.0)
optimiser_code(o::PolicyGuided.BLPG) = (Cint(2), Float64(o.η), 0.0)
optimiser_code(o::PolicyGuided.BLAPG) = (Cint(3), Float64(o.δ), Float64(o.ϵid))
optimiser_code(o::PolicyGuided.NPG) = (Cint(4), Float64(o.η), Float64(o.ϵid))
optimiser_code(o::PolicyGuided.ANPG) = (Cint(5), Float64(o.δ), Float64(o.ϵid))
optimiser_code(o::PolicyGuided.BLANPG) = (Cint(6), Float64(o.δ), Float64(o.ϵid))

"""
    HIPDeviceEstimator(chains; dependencies=(HIPMetropolis,), optimisers, q_batch_size=1)
    HIPDeviceUpdate(chains; dependencies=(HIPDeviceEstimator,))

PolicyGradientEstimator / PolicyGradientUpdate (estimator.jl:103-134, update.jl:43-57) with gradients_data and the learning
step on the device: taken one by one they are amc_pg_accumulate / amc_pg_update; run_fused! turns whole stretches of
[HIPMetropolis, HIPDeviceEstimator(, HIPDeviceUpdate)] into ONE amc_pgmc_steps.
"""
struct HIPDeviceEstimator{O} <: Arianna.AriannaAlgorithm
    metropolis::HIPMetropolis
    optimisers::O
    learn_ids::Vector{Int}
    q_batch_size::Int
end

function HIPDeviceEstimator(chains; dependencies=missing, optimisers=missing, q_batch_size=1, extras...)
    @assert length(dependencies) == 1
    @assert isa(dependencies[1], HIPMetropolis)
    @assert length(optimisers) == dependencies[1].K                                                 # estimator.jl:70
    learn_ids = [k for k in eachindex(optimisers) if !isa(optimisers[k], PolicyGuided.Static)]      # :72
    return HIPDeviceEstimator(dependencies[1], optimisers, learn_ids, q_batch_size)
end

function make_step!(::Simulation, alg::HIPDeviceEstimator)
    ids = Cint[k - 1 for k in alg.learn_ids]
    check(ccall((:amc_pg_accumulate, libamc), Cint, (Ptr{Cvoid}, Cint, Ptr{Cint}, Cint), alg.metropolis.handle, length(ids), ids, alg.q_batch_size))
    return nothing
end

struct HIPDeviceUpdate{E} <: Arianna.AriannaAlgorithm
    estimator::E
end

function HIPDeviceUpdate(chains; dependencies=missing, extras...)
    @assert length(dependencies) == 1
    @assert isa(dependencies[1], HIPDeviceEstimator)
    return HIPDeviceUpdate(dependencies[1])
end

function optimiser_arrays(est::HIPDeviceEstimator)
    codes = [optimiser_code(est.optimisers[lid]) for lid in est.learn_ids]
    return Cint[c[1] for c in codes], Float64[c[2] for c in codes], Float64[c[3] for c in codes]
end

function make_step!(::Simulation, alg::HIPDeviceUpdate)
    est = alg.estimator
    ids = Cint[k - 1 for k in est.learn_ids]
    kind, h0, h1 = optimiser_arrays(est)
    check(ccall((:amc_pg_update, libamc), Cint, (Ptr{Cvoid}, Cint, Ptr{Cint}, Ptr{Cint}, Ptr{Float64}, Ptr{Float64}),
                est.metropolis.handle, length(ids), ids, kind, h0, h1))
    return nothing
end

finalise(alg::HIPDeviceUpdate, ::Simulation) = pull_parameters!(alg.estimator.metropolis)

# ---- the loop ----------------------------------------------------------------------------------------------------------------
# when algorithm k is due next (nothing: never again) -- run! indexes schedulers[k][counters[k]] without a guard (every
# scheduler of the reference ends at `steps`); the look-ahead has to be told the end
function due_at(simulation, k)
    c = simulation.counters[k]
    (c === nothing || c > length(simulation.schedulers[k])) && return nothing
    return simulation.schedulers[k][c]
end

# how many of t, t + 1, ... (<= t_last) are consecutive entries of the scheduler from its counter
function consecutive(scheduler, counter::Int, t::Int, t_last::Int)
    n = 0
    while counter + n <= length(scheduler) && scheduler[counter + n] == t + n && t + n <= t_last
        n += 1
    end
    return n
end

reads_reductions(alg) = alg isa HIPStoreCallbacks
moves_chains(alg) = alg isa HIPDeviceEstimator || alg isa HIPPolicyGradientEstimator       # every sample leaves x at (x + δ) - δ (gradients.jl:98,103)

# among the algorithms still due at this time step: does one that reads the reductions come before any that moves the chains?
function observed_next(simulation, later)
    for k in later
        alg = simulation.algorithms[k]
        moves_chains(alg) && return false
        reads_reductions(alg) && return true
    end
    return false
end

# Tell the sampler which sums over x this run's callbacks read (callback_energy: Σe alone).  Any consumer that does not say
# (a stock StoreCallbacks with user functions goes through `reduce`, which asks for everything) keeps all three.
function declare_reduction_needs!(simulation)
    m = hip_algorithm(simulation)
    stores = filter(a -> a isa HIPStoreCallbacks, collect(simulation.algorithms))
    stock = any(a -> a isa Arianna.StoreCallbacks, simulation.algorithms)
    cols = (isempty(stores) || stock) ? REDUCE_ALL : foldl(|, (reduction_needs(a) for a in stores); init=Cint(0))
    set_reduce_columns!(m, cols)
    return nothing
end

# one sweep whose state a callback of the same t observes: the sums are formed inside its launch
function make_step_observed!(simulation, alg::HIPMetropolis)
    drop_pending!(alg)
    settle!(alg, 1)                              # two reductions in flight per engine: the one before the last is fetched now
    check(ccall((:amc_sweep_reduce_begin, libamc), Cint, (Ptr{Cvoid}, Int64), alg.handle, 1))
    PENDING_T[alg] = simulation.t
    alg.red_t = -1
    return nothing
end

# a stretch in which only the sampler is due: ONE launch for its n sweeps.  Returns n (0: nothing to fuse)
function fuse_sweeps!(simulation, k::Int, t::Int)
    alg = simulation.algorithms[k]
    others = [d for j in eachindex(simulation.algorithms) if j != k for d in (due_at(simulation, j),) if d !== nothing]
    horizon = isempty(others) ? simulation.steps + 1 : minimum(others)
    n = consecutive(simulation.schedulers[k], simulation.counters[k], t, min(horizon - 1, simulation.steps))
    n > 1 || return 0
    drop_pending!(alg)
    check(ccall((:amc_sweep, libamc), Cint, (Ptr{Cvoid}, Int64), alg.handle, n))
    alg.red_t = -1
    simulation.counters[k] += n
    simulation.t = t + n - 1
    return n
end

# [HIPMetropolis, HIPDeviceEstimator(, HIPDeviceUpdate)] due together: the next n time steps with that pattern as ONE engine call,
# up to and including the first step at which anybody else is due, provided everybody due there comes AFTER the pattern in the
# list (they then observe the state the group leaves, as when stepping one by one: src/simulation.jl:185-190).  Returns n.
function fuse_pgmc!(simulation, due::Vector{Int}, t::Int)
    algs = simulation.algorithms
    (algs[due[1]] isa HIPMetropolis && algs[due[2]] isa HIPDeviceEstimator && algs[due[2]].metropolis === algs[due[1]]) || return 0
    m, est = algs[due[1]], algs[due[2]]
    isempty(est.learn_ids) && return 0
    head = due[1:2]
    update = length(due) >= 3 && algs[due[3]] isa HIPDeviceUpdate && algs[due[3]].estimator === est
    update && (head = due[1:3])
    others = [(j, d) for j in eachindex(algs) if !(j in head) for d in (due_at(simulation, j),) if d !== nothing]
    s_other = isempty(others) ? simulation.steps + 1 : minimum(last.(others))
    after = all(j > head[end] for (j, d) in others if d == s_other)
    t_last = min(after ? s_other : s_other - 1, simulation.steps)
    n = minimum(consecutive(simulation.schedulers[k], simulation.counters[k], t, t_last) for k in head)
    n >= 1 || return 0
    at_last = sort([j for (j, d) in others if d == s_other])
    observed = after && t + n - 1 == s_other && observed_next(simulation, at_last)
    ids = Cint[k - 1 for k in est.learn_ids]
    kind, h0, h1 = optimiser_arrays(est)
    drop_pending!(m)
    if observed
        # the previous callback's sums are fetched AFTER the n - 1 steps in front of the observed one have been queued: the device
        # works through them while the host reads
        n > 1 && check(ccall((:amc_pgmc_steps, libamc), Cint, (Ptr{Cvoid}, Int64, Cint, Ptr{Cint}, Cint, Cint, Ptr{Cint}, Ptr{Float64}, Ptr{Float64}),
                             m.handle, n - 1, length(ids), ids, est.q_batch_size, update ? 1 : 0, kind, h0, h1))
        settle!(m, 1)
        check(ccall((:amc_pgmc_steps_reduce_begin, libamc), Cint, (Ptr{Cvoid}, Int64, Cint, Ptr{Cint}, Cint, Cint, Ptr{Cint}, Ptr{Float64}, Ptr{Float64}),
                    m.handle, 1, length(ids), ids, est.q_batch_size, update ? 1 : 0, kind, h0, h1))
        PENDING_T[m] = t + n - 1
    else
        check(ccall((:amc_pgmc_steps, libamc), Cint, (Ptr{Cvoid}, Int64, Cint, Ptr{Cint}, Cint, Cint, Ptr{Cint}, Ptr{Float64}, Ptr{Float64}),
                    m.handle, n, length(ids), ids, est.q_batch_size, update ? 1 : 0, kind, h0, h1))
    end
    m.red_t = -1
    for k in head
        simulation.counters[k] += n
    end
    simulation.t = t + n - 1
    for j in eachindex(algs)                     # the others due at the group's last step run after it, in list order
        if !(j in head) && due_at(simulation, j) == simulation.t
            make_step!(simulation, algs[j])
            simulation.counters[j] += 1
        end
    end
    return n
end

"""
    run_fused!(simulation)

`Arianna.run!` (src/simulation.jl:175-204) for a simulation whose sampler is a `HIPMetropolis`: the same algorithms in the same
order at the same times, every observable state the same, with the look-ahead described at the top of this file.
"""
function run_fused!(simulation::Simulation)
    try
        declare_reduction_needs!(simulation)
        for algorithm in simulation.algorithms                                   # :179-181
            initialise(algorithm, simulation)
        end
        Arianna.write_summary(simulation)                                        # :182
        sim_time = @elapsed begin
            t = 1
            while t <= simulation.steps                                          # :184
                simulation.t = t
                due = [k for k in eachindex(simulation.algorithms) if due_at(simulation, k) == t]
                n = 0
                if length(due) == 1 && simulation.algorithms[due[1]] isa HIPMetropolis
                    n = fuse_sweeps!(simulation, due[1], t)
                elseif length(due) >= 2
                    n = fuse_pgmc!(simulation, due, t)
                end
                if n == 0
                    for (i, k) in enumerate(due)                                 # :185-190
                        alg = simulation.algorithms[k]
                        if alg isa HIPMetropolis && observed_next(simulation, due[i+1:end])
                            make_step_observed!(simulation, alg)
                        else
                            alg isa HIPMetropolis && drop_pending!(alg)
                            make_step!(simulation, alg)
                        end
                        simulation.counters[k] += 1
                    end
                    n = 1
                end
                t += n
            end
            m = hip_algorithm(simulation)
            check(ccall((:amc_sync, libamc), Cint, (Ptr{Cvoid},), m.handle))
        end
        Arianna.update_summary(simulation, sim_time)                             # :193
    finally
        for algorithm in simulation.algorithms                                   # :196-198
            finalise(algorithm, simulation)
        end
        m = hip_algorithm(simulation)
        drop_pending!(m)
        settle!(m)
        set_reduce_columns!(m, REDUCE_ALL)           # the run's narrowing ends with the run
        Arianna.finalise_summary(simulation)
    end
    return nothing
end
