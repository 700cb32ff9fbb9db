# PhiloxRNG.jl -- the engine's draw schedule (DESIGN.md §3) as a Julia AbstractRNG, for the reference's
# public `R=` plug point (src/metropolis.jl:245,263; src/PolicyGuided/estimator.jl:63,92).
#
# NOT LOADED BY JULIA IN THIS REPOSITORY'S ENVIRONMENT (no Julia in the image or on the GPU box); its arithmetic is executed
# from this text by tests/julia_subset.py (a Julia-subset interpreter) and compared with the oracle bit for bit
# (tests/test_julia_philox.py) -- keep to the subset that interpreter models when editing.  Purpose: on a
# machine that has Julia, stock `Metropolis(chains; pool, seed, R=PhiloxRNG{seed,1})` consumes exactly the
# random numbers libamc.so / the oracle use, so the unmodified reference and the GPU path can be compared
# accept-for-accept (residual: Julia's exp/log vs the spec's, <= 2 ulp, expected flip rate ~1e-16 per update).
#
# The reference builds `rngs[c] = R(seed + c - 1)` (metropolis.jl:262-263): the constructor argument minus the
# seed (a type parameter here, because `R` must be a DataType) is the zero-based GLOBAL chain id.
# Per mc_step! the reference draws, in this order (metropolis.jl:206, particle_1d.jl:57, metropolis.jl:184):
#   rand(rng, Categorical(w)) -> one Float64 uniform      = 12 spare bits of draw 0 + low 24 bits of the chain's word of
#                                                           draw 1 (uniform_pick, spec v5: 36 bits)
#   rand(rng, Normal(0, s))   -> one randn(rng, Float64)  = the chain's half of draw 0 (Box-Muller pair: 52-bit radius
#                                                           uniform from words (x, y), 28-bit angle from word w)
#   rand(rng)                 -> one Float64 uniform      = 12 (other) spare bits of draw 0 + top 40 bits of the chain's
#                                                           word of draw 1 (uniform_accept)
# so the n-th call of a chain's generator is (step, kind) = divrem(n, 3).  The estimator stream (STREAM = 2)
# draws one randn per sample: call n is draw n of estimator step `est_step` (set it before each make_step!).
module PhiloxRNGs

using Random
include("amc_tables.jl")

export PhiloxRNG

mutable struct PhiloxRNG{SEED,STREAM} <: Random.AbstractRNG
    chain::UInt64      # zero-based global chain id
    calls::UInt64      # draws served so far
    est_step::UInt64   # estimator call index (STREAM == 2 only)
end
PhiloxRNG{SEED,STREAM}(s::Integer) where {SEED,STREAM} = PhiloxRNG{SEED,STREAM}(UInt64(s - SEED), 0, 0)

# Philox4x32-10 (Salmon et al., SC'11; same rounds/constants as rocRAND)
function philox4x32_10(c::NTuple{4,UInt32}, k0::UInt32, k1::UInt32)
    c0, c1, c2, c3 = c
    for _ in 1:10
        m0 = UInt64(0xD2511F53) * c0
        m1 = UInt64(0xCD9E8D57) * c2
        c0, c1, c2, c3 = (m1 >> 32) % UInt32 ⊻ c1 ⊻ k0, m1 % UInt32, (m0 >> 32) % UInt32 ⊻ c3 ⊻ k1, m0 % UInt32
        k0 += 0x9E3779B9
        k1 += 0xBB67AE85
    end
    return (c0, c1, c2, c3)
end

# counter of draw (pair, t, draw, stream): x = t[31:0], y = t[47:32] | draw<<16 | stream<<28, (z,w) = pair
function draw_words(seed::UInt64, pair::UInt64, t::UInt64, draw::Integer, stream::Integer)
    y = UInt32((t >> 32) & 0xFFFF) | (UInt32(draw & 0xFFF) << 16) | (UInt32(stream & 0xF) << 28)
    return philox4x32_10((t % UInt32, y, pair % UInt32, (pair >> 32) % UInt32), seed % UInt32, (seed >> 32) % UInt32)
end

bits12(lo::UInt32, hi::UInt32, expo::UInt64) = reinterpret(Float64, expo | (((UInt64(hi) << 32) | lo) >> 12))
uniform_co(lo, hi) = bits12(lo, hi, 0x3ff0000000000000) - 1.0      # [0,1): Julia's own rand(Float64) construction
uniform_oc(lo, hi) = 2.0 - bits12(lo, hi, 0x3ff0000000000000)      # (0,1]
angle28(w::UInt32) = 2.0 - Float64(w >> 4) * 2.0^-27               # (0,2]: the top 28 bits of ONE word (spec v5)
# spec v5: the spare bits of a normal draw (x[11:0], z, w[3:0]) lead the accept and the pick uniform of the pair's chains
spare_accept12(v::NTuple{4,UInt32}, odd) = odd == 0 ? v[1] & 0x00000fff : (v[3] >> 12) & 0x00000fff
spare_pick12(v::NTuple{4,UInt32}, odd) = odd == 0 ? v[3] & 0x00000fff : (v[3] >> 24) | ((v[4] & 0x0000000f) << 8)
# move pick = pick12 on top of the low 24 bits of the chain's accept-draw word (36 bits); accept uniform = 52-bit
# significand with accept12 on top and the top 40 bits of the accept-draw word below
uniform_pick(pick12::UInt32, lo::UInt32) = Float64((UInt64(pick12 & 0x00000fff) << 24) | UInt64(lo & 0x00ffffff)) * 2.0^-36
function uniform_accept(accept12::UInt32, lo::UInt32, hi::UInt32)
    m = (UInt64(accept12 & 0x00000fff) << 40) | (((UInt64(hi) << 32) | lo) >> 24)
    return reinterpret(Float64, 0x3ff0000000000000 | m) - 1.0
end

# table-driven log for the Box-Muller radius (DESIGN.md §3.4); fma() must be a true fused multiply-add
function logbm(u::Float64)
    ux = reinterpret(UInt64, u)
    hx = (ux >> 32) % UInt32
    k = Int32(hx >> 20) - Int32(1023)
    hx &= 0x000fffff
    i = (hx + 0x00095f64) & 0x00100000
    k += Int32(i >> 20)
    hx |= i ⊻ 0x3ff00000
    m = reinterpret(Float64, (UInt64(hx) << 32) | (ux & 0xffffffff))
    idx = Int((hx >> 13) & 0xff) - AMC_TAB_LOG_IDX_MIN + 1
    r = fma(m, AMC_TAB_LOG_INVC[idx], -1.0)
    p = 1 / 7
    p = fma(p, r, -1 / 6); p = fma(p, r, 0.2); p = fma(p, r, -0.25)
    p = fma(p, r, 1 / 3); p = fma(p, r, -0.5); p = fma(p, r, 1.0)
    dk = Float64(k)
    hi = fma(dk, 0x1.62e42fee00000p-1, AMC_TAB_LOG_LOGC[idx])
    return fma(p, r, fma(dk, 0x1.a39ef35793c76p-33, hi))
end

function sincospi_tab(w::Float64)
    shift = 0x1.8p52
    t = fma(w, 64.0, shift)
    nd = t - shift
    j = Int(reinterpret(UInt64, t) % UInt32 & 0x7f) + 1
    r = fma(nd, -0x1p-6, w)
    z = r * r
    ps = -0x1.32d2cce62bd86p-1
    ps = fma(ps, z, 0x1.466bc6775aae2p+1); ps = fma(ps, z, -0x1.4abbce625be53p+2); ps = fma(ps, z, 0x1.921fb54442d18p+1)
    sr = ps * r
    pc = -0x1.55d3c7e3cbffap+0
    pc = fma(pc, z, 0x1.03c1f081b5ac4p+2); pc = fma(pc, z, -0x1.3bd3cc9be45dep+2)
    cr = fma(pc, z, 1.0)
    S, C = AMC_TAB_SINPI[j], AMC_TAB_COSPI[j]
    return fma(S, cr, C * sr), fma(C, cr, -(S * sr))
end

function box_muller(v::NTuple{4,UInt32})
    s = sqrt(-2.0 * logbm(uniform_oc(v[1], v[2])))
    sn, cs = sincospi_tab(angle28(v[4]))
    return sn * s, cs * s
end

function next_call!(rng::PhiloxRNG)
    n = rng.calls
    rng.calls += 1
    return n
end

# rand(rng) / rand(rng, Float64): categorical uniform (kind 0) or accept uniform (kind 2) of the sampler stream
function Random.rand(rng::PhiloxRNG{SEED,1}, ::Random.SamplerTrivial{Random.CloseOpen01{Float64}}) where {SEED}
    t, kind = divrem(next_call!(rng), UInt64(3))
    pair, odd = rng.chain >> 1, rng.chain & 1
    if kind == 0 || kind == 2
        vn = draw_words(UInt64(SEED), pair, t, 0, 1)
        va = draw_words(UInt64(SEED), pair, t, 1, 1)
        lo, hi = odd == 0 ? (va[1], va[2]) : (va[3], va[4])
        return kind == 0 ? uniform_pick(spare_pick12(vn, odd), lo) : uniform_accept(spare_accept12(vn, odd), lo, hi)
    end
    error("PhiloxRNG: rand() called where the draw schedule expects randn() (call $(rng.calls - 1))")
end

function Random.randn(rng::PhiloxRNG{SEED,1}, ::Type{Float64}) where {SEED}
    t, kind = divrem(next_call!(rng), UInt64(3))
    kind == 1 || error("PhiloxRNG: randn() called where the draw schedule expects rand() (call $(rng.calls - 1))")
    z = box_muller(draw_words(UInt64(SEED), rng.chain >> 1, t, 0, 1))
    return rng.chain & 1 == 0 ? z[1] : z[2]
end

# estimator stream: one randn per sample, draw index = number of samples drawn in this estimator step
function Random.randn(rng::PhiloxRNG{SEED,2}, ::Type{Float64}) where {SEED}
    z = box_muller(draw_words(UInt64(SEED), rng.chain >> 1, rng.est_step, next_call!(rng), 2))
    return rng.chain & 1 == 0 ? z[1] : z[2]
end

"Start estimator make_step! number `t` (0-based): resets the per-step draw counter."
function begin_estimator_step!(rng::PhiloxRNG{SEED,2}, t::Integer) where {SEED}
    rng.est_step = UInt64(t)
    rng.calls = 0
    return rng
end

end # module
