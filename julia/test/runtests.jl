# runtests.jl -- the tests of the AriannaHIP package.
#
#     julia --project=julia -e 'using Pkg; Pkg.test()'            (Arianna must be resolvable: see INTEGRATION.md)
#
# NOT EXECUTED IN THIS REPOSITORY'S ENVIRONMENT (no Julia in the build image or on the GPU box).  Written for the machine
# that has Julia and Arianna; every ccall in it is checked against include/amc.h by tests/test_julia_binding_static.py.
#
# What it pins, and why it exists.  The engine's parity with the reference is proven here only up to the reference's own
# statistical tests: the CPU oracle restates the algorithm on a counter-based draw schedule, the HIP path equals the oracle
# bit for bit, but nothing in this repository's environment can run the Julia path itself.  The reference has a public plug
# point for the random number generator -- `Metropolis(chains; ..., R=...)`, src/metropolis.jl:245,263; the estimator's is
# src/PolicyGuided/estimator.jl:63,92 -- and `PhiloxRNG{seed,1}` serves the engine's draw schedule through it.  So:
#
#   (i)   STOCK `Metropolis` on example/particle_1d/particle_1d.jl with R=PhiloxRNG replays the committed trajectories of
#         tests/golden/oracle_trajectories.json (cases 0-3: harmonic K = 1, the PGMC pool K = 2, double well K = 2 with
#         sweepstep 2, a shard at chain offset 1e7): positions and energies to the bit, accept counts exactly.
#         The one residual: Julia's exp / log differ from the spec's by <= 2 ulp, and a decision `alpha > u` whose two sides
#         agree to ~1e-16 can fall the other way (expected rate ~4e-16 per update; 8 chains x 256 sweeps: ~1e-12 per case).
#         A mismatch is reported chain by chain with the sweep window it happened in -- never hidden.
#         Then the estimator's hook: stock `PolicyGradientEstimator(...; R=PhiloxRNG{seed,2})` on the state the sweeps left meets the
#         fixture's GradientData sums (to the rounding the two operation orders allow) and leaves the positions of `x_after_pg`,
#         bit for bit -- with the reference's aliased scratch buffer (SURVEY App. A.6) accounted for, not papered over.
#   (ii)  `HIPMetropolis` against the same fixtures where libamc.so and a GPU exist (skipped, and said so, elsewhere): the
#         device path through the Julia binding, bits and counts equal, callbacks equal to the oracle's reproducible sums.
#   (iii) test/ad_backends_test.jl's closed forms (logq = 0.6904993792294276, d logq / d sigma = -5 at delta = 0,
#         sigma = 0.2; atol 1e-10) for the stock Julia density, for PhiloxRNG-independent engine arithmetic through
#         amc_selftest_math where a GPU exists, and the two against each other.
using Test
using Random
using Statistics
using Arianna
using Arianna.PolicyGuided
import Arianna: initialise, make_step!, finalise            # the plugin protocol (src/algorithms.jl:13-27) is not exported
using ComponentArrays
using Distributions
import JSON
import AriannaHIP
using AriannaHIP: PhiloxRNG

const libamc = AriannaHIP.libamc             # (ccall wants a constant in this module)
const REPO = normpath(joinpath(@__DIR__, "..", ".."))
const GOLDEN = JSON.parsefile(joinpath(REPO, "tests", "golden", "oracle_trajectories.json"))
const KATS = JSON.parsefile(joinpath(REPO, "tests", "golden", "reference_kats.json"))

# `potential` is a free function of the driver script in the reference (MC_harmonic_oscillator.jl:4) and particle_1d.jl's
# Particle constructor calls it: one global whose body the cases below switch
const POTENTIAL = Ref{Function}(x -> x^2)
potential(x) = POTENTIAL[](x)
harmonic(x) = x^2                        # example/particle_1d/harmonic_oscillator/MC_harmonic_oscillator.jl:4
double_well(x) = (x * x - 1)^2           # BASELINE config 3 (not in the reference): q = x*x - 1; q*q

# the reference's own model file, as its tests include it (test/distribution_test.jl:5)
include(joinpath(pkgdir(Arianna), "example", "particle_1d", "particle_1d.jl"))

"C99 hex float text (Python's float.hex(), what the fixtures hold) -> Float64, exactly."
function hexfloat(s::AbstractString)
    s == "nan" && return NaN
    s == "inf" && return Inf
    s == "-inf" && return -Inf
    m = match(r"^(-?)0x([01])\.?([0-9a-f]*)p([+-]?[0-9]+)$", s)
    m === nothing && error("not a C99 hex float: $s")
    frac = m.captures[3]
    mantissa = parse(UInt64, m.captures[2] * frac; base=16)          # at most 1 + 13 hex digits: below 2^53, exact
    value = ldexp(Float64(mantissa), parse(Int, m.captures[4]) - 4 * length(frac))
    return m.captures[1] == "-" ? -value : value
end

bits(v) = reinterpret(UInt64, Float64(v))
samebits(a, b) = all(bits.(a) .== bits.(b))

function case_setup(case)
    spec = case["spec"]
    M = Int(spec["M"])
    β = Float64(spec["beta"])
    POTENTIAL[] = spec["potential"] == "double_well" ? double_well : harmonic
    snaps = case["snapshots"]
    @assert Int(snaps[1]["sweep"]) == 0
    x0 = hexfloat.(snaps[1]["x"])
    chains = [System(x0[c], β) for c in 1:M]
    @test samebits([c.e for c in chains], hexfloat.(snaps[1]["e"]))             # e = potential(x): same operations, same bits
    move_of(σ, w) = Move(Displacement(0.0), StandardGaussian(), ComponentArray(σ=Float64(σ)), Float64(w))
    pool = Tuple(map(move_of, spec["sigma"], spec["weight"]))
    return spec, M, chains, pool, Dict(Int(s["sweep"]) => s for s in snaps[2:end])
end

"accepted_calls / total_calls of every chain and move as the fixtures hold them: [k][c]."
counters_of(pools, K, M) = ([[pools[c][k].accepted_calls for c in 1:M] for k in 1:K], [[pools[c][k].total_calls for c in 1:M] for k in 1:K])

function compare_snapshot(name, sweep, snap, chains, accepted, total, energy, acceptance; last_good)
    x = [c.x for c in chains]
    e = [c.e for c in chains]
    want_x, want_e = hexfloat.(snap["x"]), hexfloat.(snap["e"])
    off = findall(bits.(x) .!= bits.(want_x))
    if !isempty(off)
        # reported, not hidden: which chains, and between which two snapshots the trajectories parted
        @info "case $name: chains $(off) differ from the fixture at sweep $sweep (equal at sweep $(last_good)); " *
              "a flipped accept decision (Julia's exp / log vs the spec's, <= 2 ulp; expected ~4e-16 per update) or a defect" x[off] want_x[off]
    end
    @test isempty(off)
    @test samebits(e, want_e)
    @test accepted == [Int.(row) for row in snap["accepted"]]
    @test total == [Int.(row) for row in snap["total"]]
    # callbacks: the fixture holds the engine's reproducible sums, the reference adds left to right -- equal to rounding
    @test isapprox(energy, hexfloat(snap["energy"]); rtol=1e-14)
    want_acc = hexfloat.(snap["acceptance"])
    @test all(isnan(a) == isnan(w) && (isnan(a) || isapprox(a, w; rtol=1e-14)) for (a, w) in zip(acceptance, want_acc))
    return isempty(off)
end

@testset "AriannaHIP" begin

@testset "(i) stock Metropolis with R=PhiloxRNG replays the golden trajectories" begin
    for case in GOLDEN["cases"][1:4]
        spec, M, chains, pool, snap_at = case_setup(case)
        name, K = spec["name"], length(pool)
        seed, offset = Int(spec["seed"]), Int(spec["offset"])
        # rngs[c] = R(seed + c - 1) (src/metropolis.jl:262-263); PhiloxRNG{SEED,1}(s) takes s - SEED as the zero-based GLOBAL
        # chain id and SEED as the Philox key: a shard whose first chain has global id `offset` passes seed + offset
        algorithm_list = ((algorithm=Metropolis, pool=pool, seed=seed + offset, sweepstep=Int(spec["sweepstep"]),
                           R=PhiloxRNG{seed,1}, parallel=false),)
        simulation = Simulation(chains, algorithm_list, 256; path=mktempdir(), verbose=false)
        metropolis = simulation.algorithms[1]
        @test eltype(metropolis.rngs) === PhiloxRNG{seed,1} && metropolis.rngs[1].chain == UInt64(offset)
        last_good = 0
        @testset "$name" begin
            for sweep in 1:256
                simulation.t = sweep
                make_step!(simulation, metropolis)                        # src/metropolis.jl:302-309
                haskey(snap_at, sweep) || continue
                accepted, total = counters_of(metropolis.pools, K, M)
                ok = compare_snapshot(name, sweep, snap_at[sweep], chains, accepted, total, callback_energy(simulation),
                                      callback_acceptance(simulation); last_good=last_good)
                ok && (last_good = sweep)
            end
            # every generator served exactly three draws per MH step: categorical, normal, accept (metropolis.jl:206,
            # particle_1d.jl:57, metropolis.jl:184)
            @test all(rng -> rng.calls == UInt64(3 * 256 * Int(spec["sweepstep"])), metropolis.rngs)
        end
        # The estimator's hook (src/PolicyGuided/estimator.jl:63,92): stock PolicyGradientEstimator with R=PhiloxRNG{seed,2} on the
        # state the sweeps left, every move learnable, q_batch_size = 3 -- the fixture's `pg_estimate_q3` (j, grad j, grad logq, g, n
        # per move) and `x_after_pg` (every sample leaves x at (x + delta) + (-delta), gradients.jl:98,103).
        @testset "$name: estimator" begin
            q = 3
            estimator = PolicyGradientEstimator(chains; dependencies=(metropolis,), optimisers=Tuple(VPG(1e-3) for _ in 1:K),
                                                q_batch_size=q, R=PhiloxRNG{seed,2}, parallel=false)
            @test estimator.learn_ids == collect(1:K) && estimator.rngs[1].chain == UInt64(offset)
            make_step!(simulation, estimator)                             # estimator step 0 of every generator (est_step = 0)
            @test samebits([c.x for c in chains], hexfloat.(case["x_after_pg"]))
            want = reshape(hexfloat.(case["pg_estimate_q3"]), 5, K)       # row-major (K, 5) in the fixture
            σs = Float64.(spec["sigma"])
            for k in 1:K
                gd = estimator.gradients_data[k]
                @test gd.n == M * q && gd.n == Int(want[5, k])
                # the summands' operation order differs from the engine's (DESIGN.md 3.6b: <= 64 ulp of j per sample) and the
                # reference folds left to right where the fixture holds the exactly rounded sum
                @test isapprox(gd.j, want[1, k]; rtol=1e-12)
                @test isapprox(gd.∇j[1], want[2, k]; rtol=1e-11, atol=1e-11)
                @test isapprox(gd.g[1, 1], want[4, k]; rtol=1e-12)
                # grad logq_forward: the reference hands its shared scratch buffer out inside every sample's GradientData
                # (gradients.jl:108 with estimator.jl:122), so under foldxl the first sample's array holds the SECOND sample's
                # gradient by the time the first `+` runs (SURVEY Appendix A.6): its sum is the true one minus d1 plus d2, where
                # d1, d2 are the gradients of chain 1's first two samples of this move -- formed here from the same draws
                replay = PhiloxRNG{seed,2}(seed + offset)
                replay.calls = UInt64((k - 1) * q)
                d = map(1:2) do _
                    δ = 0.0 + σs[k] * randn(replay, Float64)
                    δ^2 / σs[k]^3 - 1 / σs[k]
                end
                @test isapprox(gd.∇logq_forward[1], want[3, k] - d[1] + d[2]; rtol=1e-11, atol=1e-11)
            end
        end
    end
end

@testset "(ii) HIPMetropolis replays the same fixtures on the device" begin
    if !AriannaHIP.available()
        @info "libamc.so ($(AriannaHIP.libamc)) not loadable or no HIP device: the device half is skipped"
        @test_skip false
    else
        for case in GOLDEN["cases"][1:4]
            spec, M, chains, pool, snap_at = case_setup(case)
            name, K = spec["name"], length(pool)
            offset = Int(spec["offset"])
            algorithm_list = ((algorithm=AriannaHIP.HIPMetropolis, pool=pool, seed=Int(spec["seed"]), sweepstep=Int(spec["sweepstep"]),
                               potential=spec["potential"] == "double_well" ? :double_well : :harmonic,
                               chain_offset=offset, n_chains_global=offset + M),)
            simulation = Simulation(chains, algorithm_list, 256; path=mktempdir(), verbose=false)
            hip = simulation.algorithms[1]
            initialise(hip, simulation)
            last_good = 0
            @testset "$name" begin
                for sweep in 1:256
                    simulation.t = sweep
                    make_step!(simulation, hip)
                    haskey(snap_at, sweep) || continue
                    energy, acceptance = AriannaHIP.callback_energy(simulation), AriannaHIP.callback_acceptance(simulation)
                    finalise(hip, simulation)                         # x, e into the Particles, the counters behind hip.pools
                    accepted, total = counters_of(hip.pools, K, M)
                    snap = snap_at[sweep]
                    ok = compare_snapshot(name, sweep, snap, chains, accepted, total, energy, acceptance; last_good=last_good)
                    ok && (last_good = sweep)
                    # the device's callbacks ARE the fixture's: reproducible sums (integer records, rounded once)
                    @test bits(energy) == bits(hexfloat(snap["energy"]))
                    @test all(isnan(a) ? isnan(w) : bits(a) == bits(w) for (a, w) in zip(acceptance, hexfloat.(snap["acceptance"])))
                end
            end
        end
    end
end

@testset "(iii) closed forms of test/ad_backends_test.jl" begin
    kat = KATS["ad_backends"]                      # delta = 0, sigma = 0.2: logq = 0.6904993792294276, d logq / d sigma = -5, atol 1e-10
    δ, σ, atol = Float64(kat["delta"]), Float64(kat["sigma"]), Float64(kat["atol"])
    POTENTIAL[] = harmonic
    system = System(0.3, 2.0)
    action, policy, parameters = Displacement(δ), StandardGaussian(), ComponentArray(σ=σ)
    ∇logq = zero(parameters)
    logq = Arianna.PolicyGuided.withgrad_log_proposal_density!(∇logq, action, policy, parameters, system,
                                                               Arianna.PolicyGuided.ForwardDiff_Backend())
    @test isapprox(logq, kat["logq"]; atol=atol)
    @test isapprox(∇logq.σ, kat["grad_sigma"]; atol=atol)
    if !AriannaHIP.available()
        @info "no HIP device: the engine's log_proposal_density / derivative are not evaluated"
        @test_skip false
    else
        # amc_selftest_math fn 9 / 10: log_proposal_density and its sigma-derivative in the reference's operation order
        # (ForwardDiff's dual rules written out), evaluated by the device arithmetic the kernels use
        a, b, out = [δ, 0.05, -0.3, 1.7], [σ, 0.2, 0.1, 1.2], zeros(4)
        for (fn, julia_value) in ((9, (d, s) -> Arianna.log_proposal_density(Displacement(d), policy, ComponentArray(σ=s), system)),
                                  (10, (d, s) -> d^2 / s^3 - 1 / s))
            AriannaHIP.check(ccall((:amc_selftest_math, libamc), Cint, (Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64),
                                   0, fn, a, b, out, length(a)))
            @test isapprox(out[1], fn == 9 ? kat["logq"] : kat["grad_sigma"]; atol=atol)
            @test all(isapprox(out[i], julia_value(a[i], b[i]); rtol=1e-13, atol=1e-13) for i in eachindex(a))
        end
    end
end

end
