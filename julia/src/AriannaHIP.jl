# AriannaHIP.jl -- the reference-side binding of libamc.so (include/amc.h).
#
# NOT EXECUTED IN THIS REPOSITORY'S ENVIRONMENT: the build image and the GPU box have no Julia.
# A package (julia/Project.toml; tests: julia/test/runtests.jl, `julia --project=julia -e 'using Pkg; Pkg.test()'` where Julia,
# Arianna and -- for the device half -- libamc.so and an MI355X exist).  It is what a maintainer of
# TheDisorderedOrganization/MonteCarlo (Arianna.jl) would add, written against the reference's own interfaces:
#   AriannaAlgorithm protocol            src/algorithms.jl:6-37
#   Metropolis(chains; pool, ...)        src/metropolis.jl:288-291
#   callback_acceptance                  src/metropolis.jl:319-321
#   callback_energy                      example/particle_1d/particle_1d.jl:68-70
#   PolicyGradientEstimator / Update     src/PolicyGuided/estimator.jl:103-134, update.jl:43-57
# Every ccall below names the C entry point and its argument types exactly as include/amc.h
# declares them; tests/test_julia_binding_static.py parses this file and checks the AmcConfig field
# list and every ccall signature against the header (names, order, widths, argument counts).
module AriannaHIP

using Arianna
using Arianna.PolicyGuided
import Arianna: initialise, make_step!, finalise, write_algorithm

using Libdl

# LIBAMC names the library file; default: the one built in this repository (montecarlo_amd/libamc.so, two levels up), else the
# loader's search path.  (ccall needs a constant: read once, when the module is loaded.)
const libamc = let built = normpath(joinpath(@__DIR__, "..", "..", "montecarlo_amd", "libamc.so"))
    get(ENV, "LIBAMC", isfile(built) ? built : "libamc.so")
end

"""
    available() -> Bool

`true` when libamc.so can be loaded and sees at least one HIP device.  The engine has no CPU path (amc_create answers
AMC_ERR_NO_DEVICE without a gfx950 GPU): callers that want to run without a GPU use stock `Metropolis`.
"""
function available()
    Libdl.dlopen(libamc; throw_error=false) === nothing && return false
    n = Ref{Cint}(0)
    return ccall((:amc_device_count, libamc), Cint, (Ref{Cint},), n) == 0 && n[] > 0
end

# the engine's draw schedule as an AbstractRNG for the reference's R= hook (src/metropolis.jl:245,263): stock Metropolis then
# consumes the random numbers the device path uses
include("PhiloxRNG.jl")
using .PhiloxRNGs: PhiloxRNG, begin_estimator_step!

# struct amc_config (include/amc.h) -- field order and types are the ABI
struct AmcConfig
    struct_size::UInt32
    device::Int32
    n_chains::Int64
    chain_offset::Int64
    n_chains_global::Int64
    potential::Int32
    n_moves::Int32
    beta::Float64
    sigma::Ptr{Float64}
    weight::Ptr{Float64}
    seed::UInt64
    sweepstep::Int32
    per_chain_counters::Int32
    stream::Ptr{Cvoid}
    state_dtype::Int32            # 0: Float64, 1: Float32 (Particle{Float32}); host buffers stay Float64 either way
    reserved::Int32
end

const POTENTIAL_HARMONIC = Int32(0)
const POTENTIAL_DOUBLE_WELL = Int32(1)
const POTENTIAL_CUSTOM = Int32(2)          # potential given as a C expression in x (amc_create_custom)

# Particle{T}: the state type the engine keeps on the device (DESIGN.md section 3.7)
eltype_of_state(chains) = typeof(chains[1].x)

function check(rc::Cint)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:amc_last_error, libamc), Cstring, ()))
    error("libamc: $msg (status $rc)")      # the reference's convention: error("No ... is defined"), metropolis.jl:26
end

"""
    LazyPools(pool, M)

`Metropolis.pools` (src/metropolis.jl:233) without M deep copies.  The reference deep-copies the pool once per chain
(:289) and then makes every chain's `policy` / `parameters` refer to the objects of `pools[1]` (:252-260); what differs
between chains is `action` (scratch) and the two counters.  Here ONE copy of the pool (`template`) holds the shared
policy / parameters objects, the per-chain counters live in two M x K matrices filled by `finalise` (or
`refresh_counters!`), and `pools[c]` materialises chain c's pool on demand: same `parameters` OBJECT for every c, so an
in-place `learning_step!` on `pools[1][k].parameters` is seen through every `pools[c]`, as in the reference.  Dependants
that only read `pools[1]` (StoreParameters, src/metropolis.jl:423) get the template itself.
"""
struct LazyPools{P} <: AbstractVector{P}
    template::P
    n_chains::Int
    accepted::Matrix{Int64}       # (M, K) after finalise / refresh_counters!; (0, K) before
    total::Matrix{Int64}
end

Base.size(p::LazyPools) = (p.n_chains,)
Base.IndexStyle(::Type{<:LazyPools}) = IndexLinear()

function Base.getindex(p::LazyPools, c::Int)
    @boundscheck checkbounds(p, c)
    c == 1 && size(p.accepted, 1) == 0 && return p.template
    pool = map(enumerate(p.template)) do (k, move)
        m = Move(deepcopy(move.action), move.policy, move.parameters, move.weight)      # shared policy / parameters
        if size(p.accepted, 1) == p.n_chains
            m.accepted_calls = p.accepted[c, k]
            m.total_calls = p.total[c, k]
        end
        m
    end
    return pool
end

"""
    HIPMetropolis(chains; pool, sweepstep=1, seed=1, device=0, potential=:harmonic | :double_well | "C expression in x",
                  chain_offset=0, n_chains_global=length(chains), rank=0, n_ranks=1, unique_id=nothing, ...)

Drop-in for `Metropolis` (src/metropolis.jl:232-291) on one MI355X.  `chains` is the usual
`Vector{Particle}`; the pool must hold `Displacement` moves with a `StandardGaussian` policy.
Sharded runs (one process per GPU): `chains` is this rank's slice, `chain_offset` the global id of its first chain
(even), `n_chains_global` the ensemble size, and `unique_id` the 128 bytes rank 0 obtained from `comm_unique_id()` and
sent to the other ranks (MPI.jl / Distributed / a file): the callbacks and the estimator then sum over all shards.
"""
mutable struct HIPMetropolis{P} <: Arianna.AriannaAlgorithm
    handle::Ptr{Cvoid}
    pools::LazyPools{P}     # kept so dependants (StoreParameters, estimator) find `.pools`, `.seed`
    sweepstep::Int
    seed::Int
    n_chains::Int
    K::Int
    n_ranks::Int
    per_chain_counters::Bool
    n_params::Int           # parameters of the moves' policy (1: sigma; amc_create_vector_policy_model otherwise)
    red_t::Int              # simulation.t the cached reduction belongs to (-1: none)
    red::Vector{Float64}
end

# The particle_1d Gaussian displacement (StandardGaussian: example/particle_1d/particle_1d.jl:48-59, Displacement :26-40) as a CLASS of
# a mixed pool (`classes = (GAUSS_CLASS, (sample, logq, dlogq, perform, invert), ...)`).  Passed in exactly this text the engine knows
# the class for what it is: its sweep takes 2 sigma^2 and log(2 pi sigma^2)/2 from the move's table row, its estimator launch does not
# form the backward density again (same bits either way; a third less time per sweep of a two-class pool).
const GAUSS_CLASS = ("sigma*z", "-(delta*delta)/(2.0*(sigma*sigma)) - amc_log(6.283185307179586*(sigma*sigma))/2.0",
                     "(delta*delta)/(sigma*sigma*sigma) - 1.0/sigma", nothing, nothing)

# proposal = (sample, logq, dlogq): a script-defined policy as C expressions (amc_create_proposal_model); dlogq may be `nothing`: the
# engine then differentiates logq itself (dual numbers in the kernel), as ForwardDiff does for the reference (gradients.jl:28-33); with a policy of
# SEVERAL parameters (Move.parameters of length P > 1) dlogq is the vector of the P partial derivatives and the expressions
# say theta0 .. theta{P-1} (amc_create_vector_policy_model).
function HIPMetropolis(chains; pool=missing, sweepstep=1, seed=1, device=0, potential=:harmonic, reward=nothing, scale=nothing,
                       proposal=nothing, classes=nothing, class_of_move=nothing, chain_offset=0, n_chains_global=length(chains), per_chain_counters=true,
                       rank=0, n_ranks=1, unique_id=nothing, extras...)
    template = deepcopy(pool)                                      # ONE copy (metropolis.jl:289 makes M), see LazyPools
    K = length(template)
    pools = LazyPools(template, length(chains), Matrix{Int64}(undef, 0, K), Matrix{Int64}(undef, 0, K))
    n_params = length(template[1].parameters)
    thetas = [Float64.(collect(move.parameters)) for move in template]
    # parameter 0 at creation must pass sigma's range check; the full vectors follow through amc_set_parameters
    sigma = n_params == 1 ? Float64[th[1] for th in thetas] : ones(Float64, K)
    weight = Float64[move.weight for move in template]
    # the asserts of metropolis.jl:249-251 (identical parameters / weights across chains) hold by construction
    handle = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve sigma weight begin
        pot_id = potential isa AbstractString ? POTENTIAL_CUSTOM :
                 potential === :double_well ? POTENTIAL_DOUBLE_WELL : POTENTIAL_HARMONIC
        cfg = AmcConfig(UInt32(sizeof(AmcConfig)), Int32(device), length(chains), chain_offset, n_chains_global,
                        pot_id,
                        Int32(K), chains[1].β, pointer(sigma), pointer(weight), UInt64(seed),
                        Int32(sweepstep), Int32(per_chain_counters), C_NULL,
                        Int32(eltype_of_state(chains) === Float32 ? 1 : 0), Int32(0))
        if classes !== nothing
            # a pool that mixes policy / action types: one (sample, logq, dlogq, perform, invert) per class (nothing: not given),
            # class_of_move[k] (1-based here) the class of move k -- amc_create_mixed_model
            pot = potential isa AbstractString ? potential : C_NULL
            rew = reward isa AbstractString ? reward : C_NULL
            col(i) = Cstring[(c[i] === nothing ? Cstring(C_NULL) : Base.unsafe_convert(Cstring, Base.cconvert(Cstring, c[i]))) for c in classes]
            texts = [String(t) for c in classes for t in c if t !== nothing]          # keeps the strings alive across the call
            have_d = any(c -> c[3] !== nothing, classes)          # a class without a derivative expression: the engine differentiates its logq (NULL entry)
            com = Cint[k - 1 for k in class_of_move]
            GC.@preserve texts classes check(ccall((:amc_create_mixed_model, libamc), Cint,
                        (Ref{AmcConfig}, Cint, Ptr{Cint}, Cstring, Cstring, Ptr{Cstring}, Ptr{Cstring}, Ptr{Cstring}, Ptr{Cstring}, Ptr{Cstring}, Ref{Ptr{Cvoid}}),
                        cfg, length(classes), com, pot, rew, col(1), col(2), have_d ? col(3) : C_NULL, col(4), col(5), handle))
        elseif proposal !== nothing
            pot = potential isa AbstractString ? potential : C_NULL
            rew = reward isa AbstractString ? reward : C_NULL
            sample, logq, dlogq = proposal
            if n_params == 1
                check(ccall((:amc_create_proposal_model, libamc), Cint,
                            (Ref{AmcConfig}, Cstring, Cstring, Cstring, Cstring, Cstring, Ref{Ptr{Cvoid}}),
                            cfg, pot, rew, sample, logq, dlogq === nothing ? C_NULL : dlogq, handle))
            else
                partials = dlogq === nothing ? C_NULL : Base.cconvert(Ptr{Cstring}, collect(String, dlogq))
                GC.@preserve partials check(ccall((:amc_create_vector_policy_model, libamc), Cint,
                            (Ref{AmcConfig}, Cint, Cstring, Cstring, Cstring, Cstring, Ptr{Cstring}, Cstring, Cstring, Ref{Ptr{Cvoid}}),
                            cfg, n_params, pot, rew, sample, logq, partials === C_NULL ? C_NULL : Base.unsafe_convert(Ptr{Cstring}, partials),
                            C_NULL, C_NULL, handle))
                for k in 1:K
                    check(ccall((:amc_set_parameters, libamc), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Cint), handle[], k - 1, thetas[k], n_params))
                end
            end
        elseif scale isa AbstractString
            # a policy whose width depends on the state: delta ~ Normal(0, sigma * scale(system.x)); the expression is the
            # C restatement of what the script's sample_action! / log_proposal_density do with `system`
            pot = potential isa AbstractString ? potential : C_NULL
            rew = reward isa AbstractString ? reward : C_NULL
            check(ccall((:amc_create_policy_model, libamc), Cint, (Ref{AmcConfig}, Cstring, Cstring, Cstring, Ref{Ptr{Cvoid}}),
                        cfg, pot, rew, scale, handle))
        elseif reward isa AbstractString
            # script-defined reward(action, system) (particle_1d.jl:42-44) as an expression in delta and the new x
            pot = potential isa AbstractString ? potential : C_NULL
            check(ccall((:amc_create_model, libamc), Cint, (Ref{AmcConfig}, Cstring, Cstring, Ref{Ptr{Cvoid}}),
                        cfg, pot, reward, handle))
        elseif potential isa AbstractString
            # the script's `potential(x) = ...` (MC_harmonic_oscillator.jl:4) restated as a C expression in x
            check(ccall((:amc_create_custom, libamc), Cint, (Ref{AmcConfig}, Cstring, Ref{Ptr{Cvoid}}),
                        cfg, potential, handle))
        else
            check(ccall((:amc_create, libamc), Cint, (Ref{AmcConfig}, Ref{Ptr{Cvoid}}), cfg, handle))
        end
    end
    np_ref = Ref{Cint}(0); stride_ref = Ref{Cint}(0)
    check(ccall((:amc_n_params, libamc), Cint, (Ptr{Cvoid}, Ref{Cint}, Ref{Cint}), handle[], np_ref, stride_ref))
    @assert np_ref[] == n_params && stride_ref[] == 2 + 2n_params + n_params^2          # AMC_GD_STRIDE_P
    alg = HIPMetropolis(handle[], pools, sweepstep, seed, length(chains), K, n_ranks, per_chain_counters || K > 1, n_params, -1, Float64[])
    finalizer(a -> ccall((:amc_destroy, libamc), Cint, (Ptr{Cvoid},), a.handle), alg)
    n_ranks > 1 && comm_init!(alg, rank, n_ranks, unique_id)
    return alg
end

# ncclUniqueId for the shards' communicator: call on rank 0, ship the 128 bytes to every rank
function comm_unique_id()
    id = Vector{UInt8}(undef, 128)
    check(ccall((:amc_comm_unique_id, libamc), Cint, (Ptr{Cvoid},), id))
    return id
end

# RCCL communicator over the shards (one rank per GPU); without it amc_allreduce_sum is the identity
function comm_init!(alg::HIPMetropolis, rank::Integer, n_ranks::Integer, unique_id)
    unique_id isa AbstractVector{UInt8} && length(unique_id) == 128 ||
        error("HIPMetropolis: n_ranks > 1 needs the 128-byte unique_id of comm_unique_id() (made on rank 0)")
    id = Vector{UInt8}(unique_id)
    check(ccall((:amc_comm_init, libamc), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}), alg.handle, rank, n_ranks, id))
    alg.n_ranks = n_ranks
    return nothing
end

# initialise: upload chains[c].x (and per-chain beta)                         src/algorithms.jl:13
function initialise(alg::HIPMetropolis, simulation::Simulation)
    x = Float64[s.x for s in simulation.chains]
    β = Float64[s.β for s in simulation.chains]
    βptr = all(==(β[1]), β) ? Ptr{Float64}(C_NULL) : pointer(β)
    GC.@preserve x β check(ccall((:amc_upload_state, libamc), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}),
                                 alg.handle, x, βptr))
    alg.red_t = -1
    return nothing
end

# make_step!: one sweep of every chain                                        src/metropolis.jl:302-309
function make_step!(::Simulation, alg::HIPMetropolis)
    check(ccall((:amc_sweep, libamc), Cint, (Ptr{Cvoid}, Int64), alg.handle, 1))
    alg.red_t = -1
    return nothing
end

# pools[c][k].accepted_calls / total_calls of every chain, as two (M, K) matrices behind `alg.pools` (two bulk copies,
# no per-move Julia objects)
function refresh_counters!(alg::HIPMetropolis)
    M, K = alg.n_chains, alg.K
    acc = Matrix{Int64}(undef, M, K); tot = Matrix{Int64}(undef, M, K)       # move-major == column-major (M, K)
    check(ccall((:amc_download_counters, libamc), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}), alg.handle, acc, tot))
    alg.pools = LazyPools(alg.pools.template, M, acc, tot)
    return nothing
end

# finalise: chains[c].x / .e back into the reference's objects (one pass over the M Particles, which the caller owns
# anyway), the counters behind alg.pools
function finalise(alg::HIPMetropolis, simulation::Simulation)
    M = alg.n_chains
    x = Vector{Float64}(undef, M); e = Vector{Float64}(undef, M)
    check(ccall((:amc_download_state, libamc), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), alg.handle, x, e))
    chains = simulation.chains
    @inbounds for c in 1:M
        chains[c].x = x[c]; chains[c].e = e[c]
    end
    refresh_counters!(alg)
    return nothing
end

function write_algorithm(io, alg::HIPMetropolis, scheduler)
    println(io, "\tHIPMetropolis (libamc, gfx950)")
    println(io, "\t\tCalls: $(length(filter(x -> 0 < x ≤ scheduler[end], scheduler)))")
    println(io, "\t\tMC steps per simulation step: $(alg.sweepstep)")
    println(io, "\t\tSeed: $(alg.seed)")
    println(io, "\t\tShards: $(alg.n_ranks)")
end

hip_algorithm(simulation) = only(filter(a -> isa(a, HIPMetropolis), simulation.algorithms))

# out = [Σe, Σx, Σx², count, Σ_c acc_ck/tot_ck ...]   (AMC_RED_* in amc.h), summed over the shards; ONE reduction and
# ONE all-reduce per simulation.t however many callbacks read it (StoreCallbacks calls them back to back, algorithms.jl:97-102)
# The sums cross the shards as RECORDS (AMC_XSUM_WORDS doubles per sum: integer limbs, amc.h "reproducible sums"): merged
# exactly and rounded once, so every rank -- and a single GPU holding all the chains -- gets the same bits.
const XSUM_WORDS = 12
function finish_records!(alg::HIPMetropolis, records::Matrix{Float64}, steps_counted::UInt64)
    n = size(records, 2)
    check(ccall((:amc_allreduce_xsum, libamc), Cint, (Ptr{Cvoid}, Ptr{Float64}, Cint), alg.handle, records, n))
    out = Vector{Float64}(undef, n)
    check(ccall((:amc_xsum_round, libamc), Cint, (Ptr{Float64}, Cint, Ptr{Float64}), records, n, out))
    # K = 1 without per-chain counters: the ratio record holds the pool-wide accepted TOTAL (amc_reduce_end_exact)
    alg.K == 1 && !alg.per_chain_counters && (out[5] = out[5] / steps_counted)
    return out
end

function reduce(alg::HIPMetropolis, t::Int)
    alg.red_t == t && return alg.red
    drop_pending!(alg); settle!(alg)             # reductions queued by run_fused! come back first (AriannaHIPRun.jl): oldest first
    records = Matrix{Float64}(undef, XSUM_WORDS, 4 + alg.K)
    steps = Ref{UInt64}(0)
    check(ccall((:amc_reduce_begin, libamc), Cint, (Ptr{Cvoid},), alg.handle))
    check(ccall((:amc_reduce_end_exact, libamc), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ref{UInt64}), alg.handle, records, steps))
    out = finish_records!(alg, records, steps[])
    alg.red_t = t; alg.red = out
    return out
end

# callbacks with the reference's names/values                                 particle_1d.jl:68-70, metropolis.jl:319-321
callback_energy(simulation) = (r = reduce(hip_algorithm(simulation), simulation.t); r[1] / r[4])
callback_acceptance(simulation) = (r = reduce(hip_algorithm(simulation), simulation.t); r[5:end] ./ r[4])

"""
    HIPPolicyGradientEstimator(chains; dependencies=(HIPMetropolis,), optimisers, q_batch_size=1)

Mirror of PolicyGradientEstimator (estimator.jl:103-134): the per-sample arithmetic of
gradients.jl:93-121 runs in the kernel, the `+` fold is its reduction (+ all-reduce across shards).
`gradients_data`, `objectives`, `learn_ids`, `parameters_list` keep the reference's meaning, so the stock
`PolicyGradientUpdate`'s `learning_step!` (learning.jl) can be reused; after it, push sigma with
`set_parameters!`.
"""
mutable struct HIPPolicyGradientEstimator{O,VP,VG} <: Arianna.AriannaAlgorithm
    metropolis::HIPMetropolis
    optimisers::O
    learn_ids::Vector{Int}
    q_batch_size::Int
    parameters_list::VP           # the shared Move.parameters objects (estimator.jl:74), aliased with metropolis.pools
    gradients_data::VG
    objectives::Vector{Float64}
end

function make_step!(::Simulation, alg::HIPPolicyGradientEstimator)
    n = length(alg.learn_ids)
    ids = Cint[k - 1 for k in alg.learn_ids]                                  # C side is 0-based
    P = alg.metropolis.n_params
    stride = 2 + 2P + P^2                                                     # (j, ∇j[P], ∇logq[P], g[P,P], n) per move: AMC_GD_STRIDE_P
    records = Array{Float64}(undef, XSUM_WORDS, stride, n)                    # exact records: merged across shards, rounded once
    check(ccall((:amc_pg_estimate_exact, libamc), Cint, (Ptr{Cvoid}, Cint, Ptr{Cint}, Cint, Ptr{Float64}),
                alg.metropolis.handle, n, ids, alg.q_batch_size, records))
    check(ccall((:amc_allreduce_xsum, libamc), Cint, (Ptr{Cvoid}, Ptr{Float64}, Cint), alg.metropolis.handle, records, stride * n))
    out = Matrix{Float64}(undef, stride, n)
    check(ccall((:amc_xsum_round, libamc), Cint, (Ptr{Float64}, Cint, Ptr{Float64}), records, stride * n, out))
    alg.metropolis.red_t = -1                                                 # every sample moves x to (x + δ) - δ
    for k in 1:n
        # g arrives row by row and is symmetric bit for bit, so Julia's column-major reshape is the same matrix
        gd = PolicyGuided.GradientData(out[1, k], out[2:1+P, k], out[2+P:1+2P, k], reshape(out[2+2P:1+2P+P^2, k], P, P),
                                       Int(out[stride, k]))
        alg.gradients_data[k] = alg.gradients_data[k] + gd                    # estimator.jl:130
        alg.objectives[k] = alg.gradients_data[k].j / alg.gradients_data[k].n # estimator.jl:131
    end
    return nothing
end

function set_parameters!(alg::HIPMetropolis, k::Int, parameters)
    p = Float64.(collect(parameters))                    # ComponentArray(σ = ...) or a longer parameter array: P doubles
    check(ccall((:amc_set_parameters, libamc), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Cint), alg.handle, k - 1, p, length(p)))
end

# kw-constructor in the shape Simulation builds algorithms with (src/simulation.jl:76-83; estimator.jl:103-109):
# `dependencies=(HIPMetropolis,)` arrives as the previously built instance.
function HIPPolicyGradientEstimator(chains; dependencies=missing, optimisers=missing, q_batch_size=1, extras...)
    @assert length(dependencies) == 1
    @assert isa(dependencies[1], HIPMetropolis)
    metropolis = dependencies[1]
    @assert length(optimisers) == metropolis.K                                   # estimator.jl:70
    learn_ids = [k for k in eachindex(optimisers) if !isa(optimisers[k], PolicyGuided.Static)]   # :72
    parameters_list = [move.parameters for move in metropolis.pools.template]    # :74 -- the shared objects themselves
    gradients_data = map(k -> PolicyGuided.initialise_gradient_data(parameters_list[k]), learn_ids)   # :84
    return HIPPolicyGradientEstimator(metropolis, optimisers, learn_ids, q_batch_size, parameters_list, gradients_data,
                                      zeros(Float64, length(learn_ids)))
end

"""
    HIPPolicyGradientUpdate(chains; dependencies=(HIPPolicyGradientEstimator,))

Mirror of PolicyGradientUpdate (update.jl:43-57): `average` -> the STOCK `learning_step!` (learning.jl:32-164) on the
shared `parameters` arrays -> reset, then the new sigma is pushed to the device copy (`amc_set_parameters`).
"""
struct HIPPolicyGradientUpdate{E} <: Arianna.AriannaAlgorithm
    estimator::E
end

function HIPPolicyGradientUpdate(chains; dependencies=missing, extras...)
    @assert length(dependencies) == 1
    @assert isa(dependencies[1], HIPPolicyGradientEstimator)
    return HIPPolicyGradientUpdate(dependencies[1])
end

function make_step!(::Simulation, alg::HIPPolicyGradientUpdate)
    est = alg.estimator
    parameters_list = est.parameters_list          # ONE object per move, seen through every pools[c] (metropolis.jl:252-260)
    for (k, lid) in enumerate(est.learn_ids)
        gd = PolicyGuided.average(est.gradients_data[k])                         # update.jl:52
        PolicyGuided.learning_step!(parameters_list[lid], gd, est.optimisers[lid])   # :53, in place
        est.gradients_data[k] = PolicyGuided.initialise_gradient_data(parameters_list[lid])   # :54
        set_parameters!(est.metropolis, lid, parameters_list[lid])
    end
    return nothing
end

# Device-resident alternative for long PGMC runs: n x [sweep; estimator; update] from ONE ccall (amc_pgmc_steps):
# gradients_data and the learning step stay on the GPU (all-reduced over the shards when comm_init! was called).
# optimiser ids / hyper-parameters as in amc.h (amc_optimiser).  pull_parameters! afterwards refreshes the host objects.
function pgmc_steps!(metropolis::HIPMetropolis, n::Integer, learn_ids::Vector{Int}, q_batch::Integer,
                     optimiser::Vector{Cint}, hyper0::Vector{Float64}, hyper1::Vector{Float64})
    ids = Cint[k - 1 for k in learn_ids]
    check(ccall((:amc_pgmc_steps, libamc), Cint,
                (Ptr{Cvoid}, Int64, Cint, Ptr{Cint}, Cint, Cint, Ptr{Cint}, Ptr{Float64}, Ptr{Float64}),
                metropolis.handle, n, length(ids), ids, q_batch, 1, optimiser, hyper0, hyper1))
    metropolis.red_t = -1
    return nothing
end

# The same n time steps when callbacks are scheduled at the LAST of them (simulation.t = t): the fused launch of that step
# also forms the callback sums of the state it leaves (amc_pgmc_steps_reduce_begin) and callback_energy / callback_acceptance at
# t then read them without another pass over the chains (run! calls the callbacks after the three algorithms,
# src/simulation.jl:185-190).
function pgmc_steps_observed!(metropolis::HIPMetropolis, n::Integer, learn_ids::Vector{Int}, q_batch::Integer,
                              optimiser::Vector{Cint}, hyper0::Vector{Float64}, hyper1::Vector{Float64}, t::Int)
    ids = Cint[k - 1 for k in learn_ids]
    drop_pending!(metropolis); settle!(metropolis)
    check(ccall((:amc_pgmc_steps_reduce_begin, libamc), Cint,
                (Ptr{Cvoid}, Int64, Cint, Ptr{Cint}, Cint, Cint, Ptr{Cint}, Ptr{Float64}, Ptr{Float64}),
                metropolis.handle, n, length(ids), ids, q_batch, 1, optimiser, hyper0, hyper1))
    records = Matrix{Float64}(undef, XSUM_WORDS, 4 + metropolis.K)
    steps = Ref{UInt64}(0)
    check(ccall((:amc_reduce_end_exact, libamc), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ref{UInt64}), metropolis.handle, records, steps))
    metropolis.red_t = t; metropolis.red = finish_records!(metropolis, records, steps[])
    return nothing
end

# How an estimator step over `n_learn` learnable moves would run on this handle (amc_pg_route): (one_launch, why).  one_launch:
# every learnable move in ONE launch -- with `fused`, one launch per whole time step --, as the reference's
# make_step!(::PolicyGradientEstimator) loops over all of them in one step (estimator.jl:111-134).  false only for a pool of several
# classes whose several-move kernel form the run-time compiler fails on (`why`: the compiler's last words; the calls then take one
# launch per move, same bits), and for policies with several parameters and more than one learnable move.
function pg_route(alg::HIPMetropolis, n_learn::Integer; q_batch::Integer=1, fused::Bool=false)
    why = zeros(UInt8, 2048)
    rc = ccall((:amc_pg_route, libamc), Cint, (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{UInt8}, Cint),
               alg.handle, n_learn, q_batch, fused ? 1 : 0, why, length(why))
    rc < 0 && check(rc)
    return ((fused ? rc == 2 : rc >= 1), unsafe_string(pointer(why)))      # amc_pg_route: 2 one launch per time step, 1 one estimator launch, 0 one per move
end

# Compile-only check of a script-defined policy, without a GPU (amc_model_check): (sample, logq[, dlogq]) as for `proposal=`;
# dlogq `nothing`: the engine differentiates logq.  Returns the compiler's log; a script that does not compile raises with the
# first diagnostic, a compiler that dies raises with status -7 (AMC_ERR_COMPILE) -- the Julia session lives either way.
function model_check(sample::AbstractString, logq::AbstractString, dlogq=nothing; n_params::Integer=1, potential=nothing, reward=nothing)
    partials = dlogq === nothing ? String[] : dlogq isa AbstractString ? String[dlogq] : collect(String, dlogq)
    isempty(partials) || length(partials) == n_params || error("model_check: dlogq must list the $n_params partial derivatives of logq")
    log = zeros(UInt8, 8192)
    s, l = String[sample], String[logq]
    GC.@preserve s l partials begin
        ps = Cstring[Base.unsafe_convert(Cstring, s[1])]
        pl = Cstring[Base.unsafe_convert(Cstring, l[1])]
        pd = Cstring[Base.unsafe_convert(Cstring, d) for d in partials]
        check(ccall((:amc_model_check, libamc), Cint,
                    (Cint, Cint, Cstring, Cstring, Ptr{Cstring}, Ptr{Cstring}, Ptr{Cstring}, Ptr{Cstring}, Ptr{Cstring}, Ptr{UInt8}, Cint),
                    n_params, 1, potential === nothing ? C_NULL : potential, reward === nothing ? C_NULL : reward, ps, pl,
                    isempty(pd) ? C_NULL : pointer(pd), C_NULL, C_NULL, log, length(log)))
    end
    return unsafe_string(pointer(log))
end

# What the shards' communicator reports about itself (ncclCommCount, ncclCommUserRank, RCCL version, library file); the
# HIP runtime libamc.so is bound to.  For result files that say what they really ran on.
function comm_info(alg::HIPMetropolis)
    n = Ref{Cint}(0); r = Ref{Cint}(0); v = Ref{Cint}(0)
    path = zeros(UInt8, 1024)
    check(ccall((:amc_comm_info, libamc), Cint, (Ptr{Cvoid}, Ref{Cint}, Ref{Cint}, Ref{Cint}, Ptr{UInt8}, Cint),
                alg.handle, n, r, v, path, length(path)))
    return (n_ranks=Int(n[]), rank=Int(r[]), rccl_version=Int(v[]), librccl=unsafe_string(pointer(path)))
end

function runtime_info()
    v = Ref{Cint}(0)
    path = zeros(UInt8, 1024)
    check(ccall((:amc_runtime_info, libamc), Cint, (Ref{Cint}, Ptr{UInt8}, Cint), v, path, length(path)))
    return (hip_runtime_version=Int(v[]), hip_runtime=unsafe_string(pointer(path)))
end

# the device copy of every move's parameters back into the shared Move.parameters objects (after pgmc_steps!)
function pull_parameters!(metropolis::HIPMetropolis)
    p = Vector{Float64}(undef, metropolis.n_params)
    for (k, move) in enumerate(metropolis.pools.template)
        check(ccall((:amc_get_parameters, libamc), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Cint), metropolis.handle, k - 1, p, length(p)))
        move.parameters .= p
    end
    return nothing
end

# The pooled-position histogram accumulated on the device over the sample times (the density plot of
# MC_harmonic_oscillator.jl:40-51 without trajectories): `histogram_accumulate!` queues one pass, `histogram_fetch!` returns
# the running counts (n_bins bins, then below lo, at / above hi, NaN) and optionally resets them.
function histogram_accumulate!(metropolis::HIPMetropolis, lo::Float64, hi::Float64, n_bins::Int)
    check(ccall((:amc_histogram_accumulate, libamc), Cint, (Ptr{Cvoid}, Cdouble, Cdouble, Cint), metropolis.handle, lo, hi, n_bins))
    return nothing
end

function histogram_fetch!(metropolis::HIPMetropolis, n_bins::Int; reset::Bool=true)
    counts = Vector{UInt64}(undef, n_bins + 3)
    check(ccall((:amc_histogram_fetch, libamc), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Cint, Cint), metropolis.handle, counts, n_bins, reset ? 1 : 0))
    return counts
end

# The same read without waiting for the queued steps: `parameters_begin!` queues a copy of every σ as of this point of the
# stream, `parameters_end!` returns them later (a StoreParameters that writes the row of time t when the next one is due,
# src/metropolis.jl:433-440).  One read in flight per handle.
function parameters_begin!(metropolis::HIPMetropolis)
    check(ccall((:amc_parameters_begin, libamc), Cint, (Ptr{Cvoid},), metropolis.handle))
    return nothing
end

function parameters_end!(metropolis::HIPMetropolis)
    if metropolis.n_params > 1                   # all P parameters of every move: column k holds move k's vector
        θ = Matrix{Float64}(undef, metropolis.n_params, metropolis.K)
        check(ccall((:amc_parameters_end_all, libamc), Cint, (Ptr{Cvoid}, Ptr{Float64}, Cint), metropolis.handle, θ, length(θ)))
        return θ
    end
    σ = Vector{Float64}(undef, metropolis.K)
    check(ccall((:amc_parameters_end, libamc), Cint, (Ptr{Cvoid}, Ptr{Float64}), metropolis.handle, σ))
    return σ
end

# running (j, ∇j, ∇logq, g, n) of the device-resident estimator, 5 x n_learn; and its counterpart for a resume
function pg_get_accumulated(metropolis::HIPMetropolis, learn_ids::Vector{Int})
    ids = Cint[k - 1 for k in learn_ids]
    out = Matrix{Float64}(undef, 5, length(ids))
    check(ccall((:amc_pg_get_accumulated, libamc), Cint, (Ptr{Cvoid}, Cint, Ptr{Cint}, Ptr{Float64}),
                metropolis.handle, length(ids), ids, out))
    return out
end

function pg_set_accumulated!(metropolis::HIPMetropolis, learn_ids::Vector{Int}, rows::Matrix{Float64})
    ids = Cint[k - 1 for k in learn_ids]
    @assert size(rows) == (5, length(ids))
    check(ccall((:amc_pg_set_accumulated, libamc), Cint, (Ptr{Cvoid}, Cint, Ptr{Cint}, Ptr{Float64}),
                metropolis.handle, length(ids), ids, rows))
    return nothing
end

# Arianna's run! loop with look-ahead over the schedulers, deferred StoreCallbacks / StoreParameters, the device-resident
# estimator and update as algorithms of the list (what `north_star` asks of the Julia host: the gains of the fused forms)
include("AriannaHIPRun.jl")

export HIPMetropolis, HIPPolicyGradientEstimator, HIPPolicyGradientUpdate, HIPStoreCallbacks, HIPStoreParameters, run_fused!, PhiloxRNG

end # module
