"""ctypes wrapper of the CPU oracle (oracle/libamc_oracle.so) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
``OracleEngine`` exposes the same methods as ``montecarlo_amd._capi.HipEngine`` so the
host logic (Simulation loop, callbacks, sharding, all-reduce) can be exercised on a
CPU-only box through the package's ``engine_factory`` test seam.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Sequence

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libamc_oracle.so")

POTENTIALS = {"harmonic": 0, "double_well": 1}
POT_CUSTOM = 2
OPTIMISERS = {"Static": 0, "VPG": 1, "BLPG": 2, "BLAPG": 3, "NPG": 4, "ANPG": 5, "BLANPG": 6}
STREAM_INIT, STREAM_METROPOLIS, STREAM_ESTIMATOR = 0, 1, 2
DRAW_NORMAL, DRAW_ACCEPT = 0, 1

_lib = None


def build(force: bool = False) -> str:
    src_newer = (not os.path.exists(LIB_PATH)) or any(
        os.path.getmtime(os.path.join(ORACLE_DIR, f)) > os.path.getmtime(LIB_PATH)
        for f in ("amc_oracle.c", "amc_oracle.h", "amc_tables.inc", "Makefile"))
    if force or src_newer:
        subprocess.run(["make", "-C", ORACLE_DIR, "-B"], check=True, capture_output=True)
    return LIB_PATH


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    lib = C.CDLL(LIB_PATH)
    dp = C.POINTER(C.c_double)
    u32p = C.POINTER(C.c_uint32)
    i64p = C.POINTER(C.c_int64)
    S = C.c_void_p
    sig = {
        "amo_philox4x32_10": (None, [u32p, u32p, u32p]),
        "amo_counter": (None, [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, u32p]),
        "amo_set_custom_potential": (None, [C.c_void_p]),
        "amo_set_custom_reward": (None, [C.c_void_p]),
        "amo_set_custom_potential_f32": (None, [C.c_void_p]),
        "amo_set_custom_reward_f32": (None, [C.c_void_p]),
        "amo_set_state_f32": (None, [C.c_void_p, C.c_int]),
        "amo_set_custom_scale": (None, [C.c_void_p]),
        "amo_set_custom_scale_f32": (None, [C.c_void_p]),
        "amo_potential_f32": (C.c_float, [C.c_int, C.c_float]),
        "amo_mc_step_explicit_f32": (C.c_int, [C.c_int, C.c_float, C.c_double, C.c_double, C.c_double,
                                               C.POINTER(C.c_float), C.POINTER(C.c_float)]),
        "amo_exp": (C.c_double, [C.c_double]),
        "amo_log": (C.c_double, [C.c_double]),
        "amo_sincospi": (None, [C.c_double, dp, dp]),
        "amo_box_muller": (None, [u32p, dp]),
        "amo_logbm": (C.c_double, [C.c_double]),
        "amo_uniform_co": (C.c_double, [C.c_uint32, C.c_uint32]),
        "amo_uniform_oc": (C.c_double, [C.c_uint32, C.c_uint32]),
        "amo_angle28": (C.c_double, [C.c_uint32]),
        "amo_spare_accept12": (C.c_uint32, [u32p, C.c_int]),
        "amo_spare_pick12": (C.c_uint32, [u32p, C.c_int]),
        "amo_uniform_pick": (C.c_double, [C.c_uint32, C.c_uint32]),
        "amo_uniform_accept": (C.c_double, [C.c_uint32, C.c_uint32, C.c_uint32]),
        "amo_potential": (C.c_double, [C.c_int, C.c_double]),
        "amo_log_proposal_density": (C.c_double, [C.c_double, C.c_double]),
        "amo_grad_log_proposal_density": (C.c_double, [C.c_double, C.c_double]),
        "amo_categorical": (C.c_int, [dp, C.c_int, C.c_double]),
        "amo_mc_step_explicit": (C.c_int, [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, dp, dp]),
        "amo_create": (S, [C.c_int64, C.c_int64, C.c_int, C.c_double, C.c_int, dp, dp, C.c_uint64, C.c_int]),
        "amo_destroy": (None, [S]),
        "amo_set_x": (None, [S, dp]),
        "amo_set_beta": (None, [S, dp]),
        "amo_init_uniform": (None, [S, C.c_double, C.c_double]),
        "amo_get_state": (None, [S, dp, dp]),
        "amo_get_counters": (None, [S, i64p, i64p]),
        "amo_set_sigma": (None, [S, C.c_int, C.c_double]),
        "amo_get_sigma": (C.c_double, [S, C.c_int]),
        "amo_get_step": (C.c_uint64, [S]),
        "amo_set_step": (None, [S, C.c_uint64]),
        "amo_make_step": (None, [S, C.c_int]),
        "amo_make_steps": (None, [S, C.c_int64, C.c_int]),
        "amo_callback_energy": (C.c_double, [S]),
        "amo_callback_acceptance": (None, [S, dp]),
        "amo_moments": (None, [S, dp]),
        "amo_pg_estimate": (None, [S, C.c_int, C.POINTER(C.c_int), C.c_int, dp]),
        "amo_pg_estimate_records": (None, [S, C.c_int, C.POINTER(C.c_int), C.c_int, dp]),
        "amo_pg_estimate_plain": (None, [S, C.c_int, C.POINTER(C.c_int), C.c_int, dp]),
        "amo_pg_summands_spec": (None, [C.c_int, C.c_double, C.c_double, C.c_double, dp, dp]),
        "amo_pg_summands_reference": (None, [C.c_int, C.c_double, C.c_double, C.c_double, dp, dp]),
        "amo_xsum_q": (None, [dp, C.c_int64, C.c_int, dp]),
        "amo_xsum_q_product": (None, [dp, dp, C.c_int64, C.c_int, dp]),
        "amo_xsum_r": (None, [dp, C.c_int64, dp]),
        "amo_xsum_merge": (None, [dp, dp]),
        "amo_xsum_round": (C.c_double, [dp]),
        "amo_gd_exponents": (None, [C.c_double, C.POINTER(C.c_int)]),
        "amo_callback_records": (None, [S, dp]),
        "amo_callback_energy_plain": (C.c_double, [S]),
        "amo_callback_acceptance_plain": (None, [S, dp]),
        "amo_learning_step": (C.c_double, [C.c_int, C.c_double, C.c_double, C.c_double, dp]),
        "amo_build_schedule_linear": (C.c_int64, [C.c_int64, C.c_int64, C.c_int64, i64p, C.c_int64]),
        "amo_build_schedule_block": (C.c_int64, [C.c_int64, C.c_int64, i64p, C.c_int, i64p, C.c_int64]),
        "amo_build_schedule_log": (C.c_int64, [C.c_int64, C.c_int64, C.c_double, i64p, C.c_int64]),
        "amo_run_pooled_moments": (None, [S, C.c_int64, C.c_int64, C.c_int64, C.c_int, dp]),
        "amo_set_counters": (None, [S, i64p, i64p]),
        "amo_get_estimator_step": (C.c_uint64, [S]),
        "amo_set_estimator_step": (None, [S, C.c_uint64]),
        "amo_max_threads": (C.c_int, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _dptr(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


# ---- reproducible sums (oracle/amc_oracle.c "Reproducible sums"): records of XS_WORDS doubles ----------------------
XS_WORDS = 12


def xsum_q(values, e):
    v = np.ascontiguousarray(values, dtype=np.float64)
    rec = np.zeros(XS_WORDS)
    load().amo_xsum_q(_dptr(v), v.size, int(e), _dptr(rec))
    return rec


def xsum_q_product(x, y, e):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    rec = np.zeros(XS_WORDS)
    load().amo_xsum_q_product(_dptr(x), _dptr(y), x.size, int(e), _dptr(rec))
    return rec


def xsum_r(values):
    v = np.ascontiguousarray(values, dtype=np.float64)
    rec = np.zeros(XS_WORDS)
    load().amo_xsum_r(_dptr(v), v.size, _dptr(rec))
    return rec


def xsum_merge(into, other):
    a = np.ascontiguousarray(into, dtype=np.float64).reshape(-1, XS_WORDS).copy()
    b = np.ascontiguousarray(other, dtype=np.float64).reshape(-1, XS_WORDS)
    for i in range(a.shape[0]):
        load().amo_xsum_merge(_dptr(a[i]), _dptr(np.ascontiguousarray(b[i])))
    return a


def xsum_round(records):
    a = np.ascontiguousarray(records, dtype=np.float64).reshape(-1, XS_WORDS)
    return np.array([load().amo_xsum_round(_dptr(np.ascontiguousarray(r))) for r in a])


def gd_exponents(sigma):
    e = (C.c_int * 4)()
    load().amo_gd_exponents(float(sigma), e)
    return list(e)


def pg_summands(pot, beta, sigma, z, x, form="spec"):
    """One estimator sample of the Gaussian policy: (summands[4], new x); form 'spec'  or 'reference'."""
    xx = np.array([float(x)])
    out = np.zeros(4)
    fn = load().amo_pg_summands_spec if form == "spec" else load().amo_pg_summands_reference
    fn(POTENTIALS[pot] if isinstance(pot, str) else int(pot), float(beta), float(sigma), float(z), _dptr(xx), _dptr(out))
    return out, float(xx[0])


# ---- primitive helpers ---------------------------------------------------------------
def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    load().amo_philox4x32_10(c, k, o)
    return [int(v) for v in o]


def counter(pair, t, draw, stream):
    o = (C.c_uint32 * 4)()
    load().amo_counter(pair, t, draw, stream, o)
    return [int(v) for v in o]


def draw_words(seed, pair, t, draw, stream):
    return philox(counter(pair, t, draw, stream), [seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF])


def sincospi(w):
    s, c = C.c_double(), C.c_double()
    load().amo_sincospi(w, C.byref(s), C.byref(c))
    return s.value, c.value


def spare12(v, half):
    """(accept12, pick12) of one chain of the pair from the words of its step's normal draw (spec v5)."""
    a = (C.c_uint32 * 4)(*[int(x) for x in v])
    return int(load().amo_spare_accept12(a, int(half))), int(load().amo_spare_pick12(a, int(half)))


def box_muller(v):
    a = (C.c_uint32 * 4)(*v)
    z = (C.c_double * 2)()
    load().amo_box_muller(a, z)
    return z[0], z[1]


def mc_step_explicit(pot, beta, sigma, z, u, x, e):
    xx, ee = C.c_double(x), C.c_double(e)
    a = load().amo_mc_step_explicit(pot, beta, sigma, z, u, C.byref(xx), C.byref(ee))
    return a, xx.value, ee.value


def build_schedule(steps, burn, spec):
    lib = load()
    cap = int(steps) + 8
    out = (C.c_int64 * cap)()
    if isinstance(spec, float):
        n = lib.amo_build_schedule_log(steps, burn, spec, out, cap)
    elif isinstance(spec, (list, tuple)):
        b = (C.c_int64 * len(spec))(*spec)
        n = lib.amo_build_schedule_block(steps, burn, b, len(spec), out, cap)
    else:
        n = lib.amo_build_schedule_linear(steps, burn, int(spec), out, cap)
    if n < 0:
        raise ValueError("InexactError")
    return [int(out[i]) for i in range(n)]


def learning_step(opt: str, h0: float, h1: float, theta: float, gd4) -> float:
    a = (C.c_double * 4)(*gd4)
    return load().amo_learning_step(OPTIMISERS[opt], h0, h1, theta, a)


_custom_libs = {}
# The expressions are compiled as C++ like the kernels' (hiprtc): with Float32 state an overloaded call such as sqrt(x) or
# fabs(x) then takes its float form on both sides -- Julia's sqrt(x::Float32) is Float32 too; C would promote to double.
# Dual numbers for the oracle's twins of policies whose derivative the script does not give (the engine differentiates logq itself,
# montecarlo_amd/csrc/amc_dual.h; the reference: ForwardDiff.gradient over log_proposal_density, src/PolicyGuided/gradients.jl:28-33).
# The oracle's OWN statement of ForwardDiff 0.10's rules (src/dual.jl, partials.jl): value and partials per operation, in that
# order of IEEE operations -- x*y -> (vy*px) + (vx*py); x/y -> (inv(vy)*px) + (-(vx/(vy*vy))*py); c/y -> (-( (c/vy) / vy ))*py;
# x/c -> px/c; log -> inv(v)*p; exp -> e*p; sqrt -> inv(s+s)*p; fabs -> signbit ? -x : x; fma -> ((vy*px) + (vx*py)) + pz.
_DUAL_PROLOGUE = r"""
template <int N> struct Dn { double v; double d[N]; };
template <int N> static inline Dn<N> dn_const(double v) { Dn<N> r; r.v = v; for (int i = 0; i < N; ++i) r.d[i] = 0.0; return r; }
template <int N> static inline Dn<N> dn_var(double v, int p) { Dn<N> r = dn_const<N>(v); if (p < N) r.d[p] = 1.0; return r; }
template <int N> static inline Dn<N> dn_of(const Dn<N>& a) { return a; }
template <int N> static inline Dn<N> dn_of(double a) { return dn_const<N>(a); }
#define DN template <int N> static inline
DN Dn<N> operator+(const Dn<N>& a) { return a; }
DN Dn<N> operator-(const Dn<N>& a) { Dn<N> r; r.v = -a.v; for (int i = 0; i < N; ++i) r.d[i] = -a.d[i]; return r; }
DN Dn<N> operator+(const Dn<N>& a, const Dn<N>& b) { Dn<N> r; r.v = a.v + b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
DN Dn<N> operator+(const Dn<N>& a, double c) { Dn<N> r = a; r.v = a.v + c; return r; }
DN Dn<N> operator+(double c, const Dn<N>& a) { Dn<N> r = a; r.v = c + a.v; return r; }
DN Dn<N> operator-(const Dn<N>& a, const Dn<N>& b) { Dn<N> r; r.v = a.v - b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
DN Dn<N> operator-(const Dn<N>& a, double c) { Dn<N> r = a; r.v = a.v - c; return r; }
DN Dn<N> operator-(double c, const Dn<N>& a) { Dn<N> r; r.v = c - a.v; for (int i = 0; i < N; ++i) r.d[i] = -a.d[i]; return r; }
DN Dn<N> operator*(const Dn<N>& a, const Dn<N>& b) { Dn<N> r; r.v = a.v * b.v; for (int i = 0; i < N; ++i) r.d[i] = (b.v * a.d[i]) + (a.v * b.d[i]); return r; }
DN Dn<N> operator*(const Dn<N>& a, double c) { Dn<N> r; r.v = a.v * c; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * c; return r; }
DN Dn<N> operator*(double c, const Dn<N>& a) { Dn<N> r; r.v = c * a.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * c; return r; }
DN Dn<N> operator/(const Dn<N>& a, const Dn<N>& b) { Dn<N> r; r.v = a.v / b.v; const double fa = 1.0 / b.v, fb = -(a.v / (b.v * b.v));
    for (int i = 0; i < N; ++i) r.d[i] = (fa * a.d[i]) + (fb * b.d[i]); return r; }
DN Dn<N> operator/(const Dn<N>& a, double c) { Dn<N> r; r.v = a.v / c; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] / c; return r; }
DN Dn<N> operator/(double c, const Dn<N>& b) { Dn<N> r; const double q = c / b.v; r.v = q; const double f = -(q / b.v);
    for (int i = 0; i < N; ++i) r.d[i] = f * b.d[i]; return r; }
DN bool operator<(const Dn<N>& a, const Dn<N>& b) { return a.v < b.v; }
DN bool operator<(const Dn<N>& a, double b) { return a.v < b; }
DN bool operator<(double a, const Dn<N>& b) { return a < b.v; }
DN bool operator>(const Dn<N>& a, const Dn<N>& b) { return a.v > b.v; }
DN bool operator>(const Dn<N>& a, double b) { return a.v > b; }
DN bool operator>(double a, const Dn<N>& b) { return a > b.v; }
DN bool operator<=(const Dn<N>& a, const Dn<N>& b) { return a.v <= b.v; }
DN bool operator<=(const Dn<N>& a, double b) { return a.v <= b; }
DN bool operator<=(double a, const Dn<N>& b) { return a <= b.v; }
DN bool operator>=(const Dn<N>& a, const Dn<N>& b) { return a.v >= b.v; }
DN bool operator>=(const Dn<N>& a, double b) { return a.v >= b; }
DN bool operator>=(double a, const Dn<N>& b) { return a >= b.v; }
DN Dn<N> amo_log(const Dn<N>& a) { Dn<N> r; r.v = amo_log(a.v); const double f = 1.0 / a.v; for (int i = 0; i < N; ++i) r.d[i] = f * a.d[i]; return r; }
DN Dn<N> amo_exp(const Dn<N>& a) { Dn<N> r; r.v = amo_exp(a.v); for (int i = 0; i < N; ++i) r.d[i] = r.v * a.d[i]; return r; }
DN Dn<N> sqrt(const Dn<N>& a) { Dn<N> r; r.v = std::sqrt(a.v); const double f = 1.0 / (r.v + r.v); for (int i = 0; i < N; ++i) r.d[i] = f * a.d[i]; return r; }
DN Dn<N> fabs(const Dn<N>& a) { return std::signbit(a.v) ? -a : a; }
DN Dn<N> fma(const Dn<N>& a, const Dn<N>& b, const Dn<N>& c) { Dn<N> r; r.v = std::fma(a.v, b.v, c.v);
    for (int i = 0; i < N; ++i) r.d[i] = ((b.v * a.d[i]) + (a.v * b.d[i])) + c.d[i]; return r; }
DN Dn<N> fma(const Dn<N>& a, const Dn<N>& b, double c) { Dn<N> r; r.v = std::fma(a.v, b.v, c); for (int i = 0; i < N; ++i) r.d[i] = (b.v * a.d[i]) + (a.v * b.d[i]); return r; }
DN Dn<N> fma(const Dn<N>& a, double b, const Dn<N>& c) { Dn<N> r; r.v = std::fma(a.v, b, c.v); for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] * b) + c.d[i]; return r; }
DN Dn<N> fma(double a, const Dn<N>& b, const Dn<N>& c) { return fma(b, a, c); }
DN Dn<N> fma(const Dn<N>& a, double b, double c) { Dn<N> r; r.v = std::fma(a.v, b, c); for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b; return r; }
DN Dn<N> fma(double a, const Dn<N>& b, double c) { return fma(b, a, c); }
DN Dn<N> fma(double a, double b, const Dn<N>& c) { Dn<N> r = c; r.v = std::fma(a, b, c.v); return r; }
#undef DN
"""
_CUSTOM_PROLOGUE = ("#include <cmath>\nusing std::sqrt; using std::fabs; using std::fma;\n"
                    "extern \"C\" { double amo_exp(double); double amo_log(double); }\n" + _DUAL_PROLOGUE +
                    "extern \"C\" {\n"
                    "#define amc_exp(v) amo_exp(v)\n#define amc_log(v) amo_log(v)\n")


_TH0 = "const double theta0 = sigma; (void)theta0; "      # theta0 is another name of sigma (amc_model.h AMC_USER_THETAS)
# Float32 twins: the hook's double arguments hold Float32 values; the expression sees them as floats
_F32_X = "const float x = (float)x_; (void)x;"
_F32_DX = "const float delta = (float)delta_; const float x = (float)x_; (void)delta; (void)x;"


def _dual_dlogq_body(logq: str, n_params: int = 1) -> str:
    """C++ statements that leave d logq / d theta_p in out[p]: `logq` evaluated over dual numbers (ForwardDiff's rules above) with
    unit partials on the parameters; `sigma` is theta0, delta and x are constants."""
    P = int(n_params)
    names = "".join(f"const Dn<{P}> theta{i} = dn_var<{P}>({'theta[%d]' % i if i < P else '0.0'}, {i}); (void)theta{i}; " for i in range(4))
    return f"{names}const Dn<{P}> sigma = theta0; (void)sigma; const Dn<{P}> r_ = dn_of<{P}>({logq}); for (int p_ = 0; p_ < {P}; ++p_) out[p_] = r_.d[p_];"


def install_custom_potential(expr: str) -> None:
    """gcc-compile `double f(double x) { return <expr>; }` with the oracle's flags (no contraction) and make it the
    oracle's global `potential` -- the CPU twin of amc_create_custom's hiprtc build.  amc_exp / amc_log resolve to
    the oracle's own amo_exp / amo_log."""
    import hashlib
    import tempfile
    lib = load()
    key = hashlib.sha1(expr.encode()).hexdigest()[:16]
    if key not in _custom_libs:
        d = tempfile.mkdtemp(prefix="amo_pot_")
        src, so = os.path.join(d, "pot.cpp"), os.path.join(d, f"pot_{key}.so")
        with open(src, "w") as f:
            f.write(_CUSTOM_PROLOGUE +
                    f"double amo_user_potential(double x) {{ return ({expr}); }}\n"
                    # Float32 state: the same text with x::Float32 (C's usual arithmetic conversions promote against
                    # double literals like Julia's do); the value is converted to Float32 on return (Particle.e::T)
                    f"float amo_user_potential_f32(float x) {{ return (float)({expr}); }}\n}}\n")
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-mfma", "-fno-math-errno",
                        src, "-o", so, LIB_PATH, "-lm", f"-Wl,-rpath,{ORACLE_DIR}"], check=True, capture_output=True)
        _custom_libs[key] = C.CDLL(so)
    fn = C.cast(_custom_libs[key].amo_user_potential, C.c_void_p)
    lib.amo_set_custom_potential(fn)
    lib.amo_set_custom_potential_f32(C.cast(_custom_libs[key].amo_user_potential_f32, C.c_void_p))


def install_custom_reward(expr: Optional[str]) -> None:
    """The oracle's global reward(action, system): None restores delta^2; else `double f(double delta, double x)`
    compiled by gcc like install_custom_potential."""
    import hashlib
    import tempfile
    lib = load()
    if expr is None:
        lib.amo_set_custom_reward(None)
        lib.amo_set_custom_reward_f32(None)
        return
    key = "r" + hashlib.sha1(expr.encode()).hexdigest()[:16]
    if key not in _custom_libs:
        d = tempfile.mkdtemp(prefix="amo_rew_")
        src, so = os.path.join(d, "rew.cpp"), os.path.join(d, f"rew_{key}.so")
        with open(src, "w") as f:
            f.write(_CUSTOM_PROLOGUE +
                    f"double amo_user_reward(double delta, double x) {{ return ({expr}); }}\n"
                    f"double amo_user_reward_f32(float delta, float x) {{ return (double)({expr}); }}\n}}\n")
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-mfma", "-fno-math-errno",
                        src, "-o", so, LIB_PATH, "-lm", f"-Wl,-rpath,{ORACLE_DIR}"], check=True, capture_output=True)
        _custom_libs[key] = C.CDLL(so)
    lib.amo_set_custom_reward(C.cast(_custom_libs[key].amo_user_reward, C.c_void_p))
    lib.amo_set_custom_reward_f32(C.cast(_custom_libs[key].amo_user_reward_f32, C.c_void_p))


def install_custom_scale(expr: Optional[str]) -> None:
    """The oracle's global proposal-width scale(x) (ScaledGaussian policy): None restores scale == 1."""
    import hashlib
    import tempfile
    lib = load()
    if expr is None:
        lib.amo_set_custom_scale(None)
        lib.amo_set_custom_scale_f32(None)
        return
    key = "s" + hashlib.sha1(expr.encode()).hexdigest()[:16]
    if key not in _custom_libs:
        d = tempfile.mkdtemp(prefix="amo_scale_")
        src, so = os.path.join(d, "scale.cpp"), os.path.join(d, f"scale_{key}.so")
        with open(src, "w") as f:
            f.write(_CUSTOM_PROLOGUE +
                    f"double amo_user_scale(double x) {{ return ({expr}); }}\n"
                    f"float amo_user_scale_f32(float x) {{ return (float)({expr}); }}\n}}\n")
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-mfma", "-fno-math-errno",
                        src, "-o", so, LIB_PATH, "-lm", f"-Wl,-rpath,{ORACLE_DIR}"], check=True, capture_output=True)
        _custom_libs[key] = C.CDLL(so)
    lib.amo_set_custom_scale(C.cast(_custom_libs[key].amo_user_scale, C.c_void_p))
    lib.amo_set_custom_scale_f32(C.cast(_custom_libs[key].amo_user_scale_f32, C.c_void_p))


def install_custom_proposal(proposal, f32: bool = False) -> None:
    """The oracle's global script-defined proposal: None restores the particle_1d Gaussian displacement; else
    (sample, logq, dlogq or None) as C expressions in (z, x, sigma) / (delta, x, sigma), compiled by gcc.  f32: the twins of a
    Float32 simulation -- the same text with x and delta as floats (C's usual arithmetic conversions restate Julia's promotion
    rules), a Float32 value returned where the result lands in a field of type T (delta, x); the hooks keep double signatures."""
    import hashlib
    import tempfile
    lib = load()
    lib.amo_set_custom_proposal.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.amo_set_custom_proposal.restype = None
    lib.amo_set_custom_action.argtypes = [C.c_void_p, C.c_void_p]
    lib.amo_set_custom_action.restype = None
    if proposal is None:
        lib.amo_set_custom_proposal(None, None, None)
        lib.amo_set_custom_action(None, None)
        return
    sample, logq, dlogq, perform, invert = (list(proposal) + [None] * 4)[:5]
    key = "p" + hashlib.sha1(repr((sample, logq, dlogq, perform, invert)).encode()).hexdigest()[:16]
    if key not in _custom_libs:
        d = tempfile.mkdtemp(prefix="amo_prop_")
        src, so = os.path.join(d, "prop.cpp"), os.path.join(d, f"prop_{key}.so")
        with open(src, "w") as f:
            f.write(_CUSTOM_PROLOGUE +
                    f"double amo_user_sample(double z, double x, double sigma) {{ {_TH0}return ({sample}); }}\n"
                    f"double amo_user_logq(double delta, double x, double sigma) {{ {_TH0}return ({logq}); }}\n" +
                    (f"double amo_user_dlogq(double delta, double x, double sigma) {{ {_TH0}return ({dlogq}); }}\n" if dlogq else
                     "double amo_user_dlogq(double delta, double x, double sigma_in) { const double theta[1] = {sigma_in}; double out[1]; "
                     + _dual_dlogq_body(logq, 1) + " return out[0]; }\n") +
                    (f"double amo_user_perform(double x, double delta) {{ return ({perform}); }}\n"
                     f"double amo_user_invert(double delta, double x) {{ return ({invert}); }}\n" if perform else "") +
                    # Float32 state: x and delta are floats inside
                    f"double amo_user_sample_f32(double z, double x_, double sigma) {{ {_F32_X} {_TH0}return (double)(float)({sample}); }}\n"
                    f"double amo_user_logq_f32(double delta_, double x_, double sigma) {{ {_F32_DX} {_TH0}return (double)({logq}); }}\n" +
                    (f"double amo_user_dlogq_f32(double delta_, double x_, double sigma) {{ {_F32_DX} {_TH0}return (double)({dlogq}); }}\n" if dlogq else
                     "double amo_user_dlogq_f32(double delta_, double x_, double sigma_in) { " + _F32_DX + " const double theta[1] = {sigma_in}; double out[1]; "
                     + _dual_dlogq_body(logq, 1) + " return out[0]; }\n") +
                    (f"double amo_user_perform_f32(double x_, double delta_) {{ {_F32_DX} return (double)(float)({perform}); }}\n"
                     f"double amo_user_invert_f32(double delta_, double x_) {{ {_F32_DX} return (double)(float)({invert}); }}\n" if perform else "") + "}\n")
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-mfma", "-fno-math-errno",
                        src, "-o", so, LIB_PATH, "-lm", f"-Wl,-rpath,{ORACLE_DIR}"], check=True, capture_output=True)
        _custom_libs[key] = C.CDLL(so)
    L = _custom_libs[key]
    sfx = "_f32" if f32 else ""
    lib.amo_set_custom_proposal(C.cast(getattr(L, "amo_user_sample" + sfx), C.c_void_p), C.cast(getattr(L, "amo_user_logq" + sfx), C.c_void_p),
                                C.cast(getattr(L, "amo_user_dlogq" + sfx), C.c_void_p))     # (the script's expression, or logq over dual numbers)
    if perform:
        lib.amo_set_custom_action(C.cast(getattr(L, "amo_user_perform" + sfx), C.c_void_p), C.cast(getattr(L, "amo_user_invert" + sfx), C.c_void_p))
    else:
        lib.amo_set_custom_action(None, None)


def install_vector_policy(n_params, proposal, f32: bool = False) -> None:
    """The oracle's global policy with SEVERAL parameters (amo_set_vector_policy): proposal = (sample, logq, [dlogq_0 .. dlogq_{P-1}]
    or None[, perform, invert]) as C expressions in z / delta, x and theta0 .. theta{P-1} (`sigma` names theta0), compiled by
    gcc.  None restores the one-parameter forms."""
    import hashlib
    import tempfile
    lib = load()
    lib.amo_set_vector_policy.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.amo_set_vector_policy.restype = None
    lib.amo_set_custom_action.argtypes = [C.c_void_p, C.c_void_p]
    lib.amo_set_custom_action.restype = None
    if proposal is None:
        lib.amo_set_vector_policy(1, None, None, None)
        return
    P = int(n_params)
    sample, logq, dlogq, perform, invert = (list(proposal) + [None] * 4)[:5]
    key = "v" + hashlib.sha1(repr((P, sample, logq, dlogq, perform, invert)).encode()).hexdigest()[:16]
    if key not in _custom_libs:
        d = tempfile.mkdtemp(prefix="amo_vec_")
        src, so = os.path.join(d, "vec.cpp"), os.path.join(d, f"vec_{key}.so")
        names = "".join(f"const double theta{i} = {'theta[%d]' % i if i < P else '0.0'}; (void)theta{i}; " for i in range(4))
        names += "const double sigma = theta0; (void)sigma; "
        with open(src, "w") as f:
            f.write(_CUSTOM_PROLOGUE +
                    f"double amo_vec_sample(double z, double x, const double* theta) {{ {names} return ({sample}); }}\n"
                    f"double amo_vec_logq(double delta, double x, const double* theta) {{ {names} return ({logq}); }}\n" +
                    ("void amo_vec_dlogq(double delta, double x, const double* theta, double* out) { " + names +
                     " ".join(f"out[{i}] = ({e});" for i, e in enumerate(dlogq)) + " }\n" if dlogq else
                     "void amo_vec_dlogq(double delta, double x, const double* theta, double* out) { " + _dual_dlogq_body(logq, P) + " }\n") +
                    (f"double amo_user_perform(double x, double delta) {{ return ({perform}); }}\n"
                     f"double amo_user_invert(double delta, double x) {{ return ({invert}); }}\n" if perform else "") +
                    # Float32 state (see install_custom_proposal)
                    f"double amo_vec_sample_f32(double z, double x_, const double* theta) {{ {_F32_X} {names} return (double)(float)({sample}); }}\n"
                    f"double amo_vec_logq_f32(double delta_, double x_, const double* theta) {{ {_F32_DX} {names} return (double)({logq}); }}\n" +
                    ("void amo_vec_dlogq_f32(double delta_, double x_, const double* theta, double* out) { " + _F32_DX + " " + names +
                     " ".join(f"out[{i}] = (double)({e});" for i, e in enumerate(dlogq)) + " }\n" if dlogq else
                     "void amo_vec_dlogq_f32(double delta_, double x_, const double* theta, double* out) { " + _F32_DX + " " + _dual_dlogq_body(logq, P) + " }\n") +
                    (f"double amo_user_perform_f32(double x_, double delta_) {{ {_F32_DX} return (double)(float)({perform}); }}\n"
                     f"double amo_user_invert_f32(double delta_, double x_) {{ {_F32_DX} return (double)(float)({invert}); }}\n" if perform else "") + "}\n")
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-mfma", "-fno-math-errno",
                        src, "-o", so, LIB_PATH, "-lm", f"-Wl,-rpath,{ORACLE_DIR}"], check=True, capture_output=True)
        _custom_libs[key] = C.CDLL(so)
    L = _custom_libs[key]
    sfx = "_f32" if f32 else ""
    lib.amo_set_vector_policy(P, C.cast(getattr(L, "amo_vec_sample" + sfx), C.c_void_p), C.cast(getattr(L, "amo_vec_logq" + sfx), C.c_void_p),
                              C.cast(getattr(L, "amo_vec_dlogq" + sfx), C.c_void_p))
    if perform:
        lib.amo_set_custom_action(C.cast(getattr(L, "amo_user_perform" + sfx), C.c_void_p), C.cast(getattr(L, "amo_user_invert" + sfx), C.c_void_p))
    else:
        lib.amo_set_custom_action(None, None)


def install_policy_classes(classes, class_of_move, f32: bool = False) -> None:
    """The oracle's global policy CLASSES of a pool that mixes policy / action types (amo_set_policy_classes): classes = list of
    (sample, logq, dlogq or None, perform or None, invert or None) as C expressions; class_of_move[k] = the class of move k.
    None restores the one-policy forms."""
    import hashlib
    import tempfile
    lib = load()
    VP = C.POINTER(C.c_void_p)
    lib.amo_set_policy_classes.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_int, VP, VP, VP, VP, VP]
    lib.amo_set_policy_classes.restype = None
    if classes is None:
        lib.amo_set_policy_classes(1, None, 0, None, None, None, None, None)
        return
    cl = [tuple((list(c) + [None] * 4)[:5]) for c in classes]
    key = "m" + hashlib.sha1(repr(cl).encode()).hexdigest()[:16]
    if key not in _custom_libs:
        d = tempfile.mkdtemp(prefix="amo_cls_")
        src, so = os.path.join(d, "cls.cpp"), os.path.join(d, f"cls_{key}.so")
        body = ""
        for i, (sample, logq, dlogq, perform, invert) in enumerate(cl):
            th = "const double theta0 = sigma; (void)theta0; "
            body += f"double amo_cls_sample_{i}(double z, double x, double sigma) {{ {th}return ({sample}); }}\n"
            body += f"double amo_cls_logq_{i}(double delta, double x, double sigma) {{ {th}return ({logq}); }}\n"
            if dlogq:
                body += f"double amo_cls_dlogq_{i}(double delta, double x, double sigma) {{ {th}return ({dlogq}); }}\n"
            else:
                body += (f"double amo_cls_dlogq_{i}(double delta, double x, double sigma_in) {{ const double theta[1] = {{sigma_in}}; double out[1]; "
                         + _dual_dlogq_body(logq, 1) + " return out[0]; }\n")
            if perform:
                body += f"double amo_cls_perform_{i}(double x, double delta) {{ return ({perform}); }}\n"
                body += f"double amo_cls_invert_{i}(double delta, double x) {{ return ({invert}); }}\n"
            # Float32 state (see install_custom_proposal)
            body += f"double amo_cls_sample_f32_{i}(double z, double x_, double sigma) {{ {_F32_X} {th}return (double)(float)({sample}); }}\n"
            body += f"double amo_cls_logq_f32_{i}(double delta_, double x_, double sigma) {{ {_F32_DX} {th}return (double)({logq}); }}\n"
            if dlogq:
                body += f"double amo_cls_dlogq_f32_{i}(double delta_, double x_, double sigma) {{ {_F32_DX} {th}return (double)({dlogq}); }}\n"
            else:
                body += (f"double amo_cls_dlogq_f32_{i}(double delta_, double x_, double sigma_in) {{ {_F32_DX} const double theta[1] = {{sigma_in}}; double out[1]; "
                         + _dual_dlogq_body(logq, 1) + " return out[0]; }\n")
            if perform:
                body += f"double amo_cls_perform_f32_{i}(double x_, double delta_) {{ {_F32_DX} return (double)(float)({perform}); }}\n"
                body += f"double amo_cls_invert_f32_{i}(double delta_, double x_) {{ {_F32_DX} return (double)(float)({invert}); }}\n"
        with open(src, "w") as f:
            f.write(_CUSTOM_PROLOGUE + body + "}\n")
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-mfma", "-fno-math-errno",
                        src, "-o", so, LIB_PATH, "-lm", f"-Wl,-rpath,{ORACLE_DIR}"], check=True, capture_output=True)
        _custom_libs[key] = C.CDLL(so)
    L = _custom_libs[key]
    n = len(cl)

    def ptrs(name, present):
        sfx = "_f32" if f32 else ""
        return (C.c_void_p * n)(*[C.cast(getattr(L, f"amo_cls_{name}{sfx}_{i}"), C.c_void_p) if present(cl[i]) else None for i in range(n)])
    have_d = True            # a class without a derivative expression has its logq differentiated (dual numbers)
    com = (C.c_int * len(class_of_move))(*[int(v) for v in class_of_move])
    lib.amo_set_policy_classes(n, com, len(class_of_move), ptrs("sample", lambda c: True), ptrs("logq", lambda c: True),
                               ptrs("dlogq", lambda c: True) if have_d else None, ptrs("perform", lambda c: c[3]),
                               ptrs("invert", lambda c: c[3]))


def learning_step_vec(opt: str, h0: float, h1: float, theta, gd) -> "np.ndarray | None":
    """amo_learning_step_vec on the averaged GradientData [j, grad j, grad logq, g]; None: singular metric."""
    lib = load()
    lib.amo_learning_step_vec.argtypes = [C.c_int, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.amo_learning_step_vec.restype = C.c_int
    th = np.ascontiguousarray(theta, dtype=np.float64).copy()
    g = np.ascontiguousarray(gd, dtype=np.float64)
    ok = lib.amo_learning_step_vec(OPTIMISERS[opt], float(h0), float(h1), int(th.size), _dptr(g), _dptr(th))
    return th if ok else None


def _potential_id(potential) -> int:
    expr = getattr(potential, "expr", None)
    if expr is not None:
        install_custom_potential(expr)
        return POT_CUSTOM
    return POTENTIALS[potential]


class OracleSim:
    """amo_sim: reference-shaped (AoS) ensemble + Metropolis on the CPU."""

    def __init__(self, n_chains, *, chain_offset=0, potential="harmonic", beta=1.0, sigma=(1.0,),
                 weight=(1.0,), seed=1, sweepstep=1, reward_expr=None, dtype="f64", scale_expr=None, proposal=None, n_params=1,
                 classes=None, class_of_move=None):
        self.lib = load()
        self.dtype = dtype
        self.n_params = int(n_params)
        theta = None
        if self.n_params > 1:
            theta = [np.ascontiguousarray(v, dtype=np.float64).reshape(-1) for v in sigma]
            sigma = [float(v[0]) for v in theta]
        install_custom_scale(scale_expr)            # process-global like the potential: one simulation at a time
        f32 = dtype == "f32"
        install_custom_proposal(None if self.n_params > 1 else proposal, f32)
        install_vector_policy(self.n_params, proposal if self.n_params > 1 else None, f32)
        install_policy_classes(classes, class_of_move, f32)       # after the one-policy installer: it sets the script switches
        install_custom_reward(reward_expr)          # process-global, like the reference's script-level definition
        self.M = int(n_chains)
        self.K = len(sigma)
        s = (C.c_double * self.K)(*[float(v) for v in sigma])
        w = (C.c_double * self.K)(*[float(v) for v in weight])
        self.h = self.lib.amo_create(self.M, int(chain_offset), _potential_id(potential), float(beta), self.K, s, w,
                                     int(seed) & 0xFFFFFFFFFFFFFFFF, int(sweepstep))
        if not self.h:
            raise ValueError("amo_create failed")
        if dtype == "f32":
            self.lib.amo_set_state_f32(self.h, 1)
        elif dtype != "f64":
            raise ValueError(f"dtype {dtype!r}")
        self.lib.amo_set_theta.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double]
        self.lib.amo_set_theta.restype = None
        self.lib.amo_get_theta.argtypes = [C.c_void_p, C.c_int, C.c_int]
        self.lib.amo_get_theta.restype = C.c_double
        if theta is not None:
            for k, v in enumerate(theta):
                self.set_theta(k, v)

    def close(self):
        if self.h:
            self.lib.amo_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_x(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        assert x.shape == (self.M,)
        self.lib.amo_set_x(self.h, _dptr(x))

    def set_beta(self, b):
        b = np.ascontiguousarray(b, dtype=np.float64)
        assert b.shape == (self.M,)
        self.lib.amo_set_beta(self.h, _dptr(b))

    def init_uniform(self, lo, hi):
        self.lib.amo_init_uniform(self.h, float(lo), float(hi))

    def state(self):
        x = np.empty(self.M)
        e = np.empty(self.M)
        self.lib.amo_get_state(self.h, _dptr(x), _dptr(e))
        return x, e

    def counters(self):
        acc = np.empty((self.K, self.M), dtype=np.int64)
        tot = np.empty((self.K, self.M), dtype=np.int64)
        p = C.POINTER(C.c_int64)
        self.lib.amo_get_counters(self.h, acc.ctypes.data_as(p), tot.ctypes.data_as(p))
        return acc, tot

    def make_steps(self, n=1, threads=1):
        self.lib.amo_make_steps(self.h, int(n), int(threads))

    def energy(self):
        return self.lib.amo_callback_energy(self.h)

    def acceptance(self):
        out = np.empty(self.K)
        self.lib.amo_callback_acceptance(self.h, _dptr(out))
        return out

    def moments(self):
        out = np.empty(2)
        self.lib.amo_moments(self.h, _dptr(out))
        return out

    def pg_estimate(self, learn_ids, q_batch):
        n = len(learn_ids)
        ids = (C.c_int * max(n, 1))(*[int(i) for i in learn_ids])
        out = np.zeros((n, 5))
        self.lib.amo_pg_estimate(self.h, n, ids, int(q_batch), _dptr(out))
        return out

    def pg_estimate_records(self, learn_ids, q_batch):
        """The fold as records, shape (n_learn, 5, XS_WORDS)."""
        n = len(learn_ids)
        ids = (C.c_int * max(n, 1))(*[int(i) for i in learn_ids])
        out = np.zeros((n, 5, XS_WORDS))
        self.lib.amo_pg_estimate_records(self.h, n, ids, int(q_batch), _dptr(out))
        return out

    def pg_estimate_plain(self, learn_ids, q_batch):
        """Reference-ordered summands folded left to right in Float64 (one order the reference's reducer may take)."""
        n = len(learn_ids)
        ids = (C.c_int * max(n, 1))(*[int(i) for i in learn_ids])
        out = np.zeros((n, 5))
        self.lib.amo_pg_estimate_plain(self.h, n, ids, int(q_batch), _dptr(out))
        return out

    def callback_records(self):
        """The callbacks' sums as records, shape (4 + K, XS_WORDS): sum e, sum x, sum x^2, count, sum_c acc/tot per move."""
        out = np.zeros((4 + self.K, XS_WORDS))
        self.lib.amo_callback_records(self.h, _dptr(out))
        return out

    def energy_plain(self):
        return self.lib.amo_callback_energy_plain(self.h)

    def acceptance_plain(self):
        out = np.empty(self.K)
        self.lib.amo_callback_acceptance_plain(self.h, _dptr(out))
        return out

    def run_pooled_moments(self, steps, burn, dt, threads=1):
        out = np.zeros(4)
        self.lib.amo_run_pooled_moments(self.h, int(steps), int(burn), int(dt), int(threads), _dptr(out))
        return out

    def set_sigma(self, k, s):
        self.lib.amo_set_sigma(self.h, int(k), float(s))

    def set_theta(self, k, v):
        for p, t in enumerate(np.asarray(v, dtype=np.float64).reshape(-1)):
            self.lib.amo_set_theta(self.h, int(k), p, float(t))

    def get_theta(self, k):
        return np.array([self.lib.amo_get_theta(self.h, int(k), p) for p in range(self.n_params)])

    def pg_estimate_records_vec(self, learn_ids, q_batch):
        """A policy with several parameters: records (n_learn, 2 + 2P + P^2, XS_WORDS) -- j, grad j, grad logq, g, n."""
        n = len(learn_ids)
        P = self.n_params
        ids = (C.c_int * max(n, 1))(*[int(i) for i in learn_ids])
        out = np.zeros((n, 2 + 2 * P + P * P, XS_WORDS))
        self.lib.amo_pg_estimate_records_vec.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_double)]
        self.lib.amo_pg_estimate_records_vec.restype = None
        self.lib.amo_pg_estimate_records_vec(self.h, n, ids, int(q_batch), _dptr(out))
        return out

    def get_sigma(self, k):
        return self.lib.amo_get_sigma(self.h, int(k))

    @property
    def step(self):
        return self.lib.amo_get_step(self.h)

    @step.setter
    def step(self, t):
        self.lib.amo_set_step(self.h, int(t))


class OracleEngine:
    """Same surface as montecarlo_amd._capi.HipEngine, computed by the CPU oracle.

    A test double for CPU-only runs of the host logic; never shipped, never a fallback."""

    def __init__(self, *, n_chains, chain_offset=0, n_chains_global=None, potential="harmonic", beta=1.0,
                 sigma=(1.0,), weight=(1.0,), seed=1, sweepstep=1, per_chain_counters=True, device=0,
                 stream=None, reward_expr=None, dtype="f64", scale_expr=None, proposal=None, n_params=1, classes=None,
                 class_of_move=None):
        self.n_chains = int(n_chains)
        self.n_moves = len(sigma)
        self.n_params = int(n_params)
        self.gd_stride = 2 + 2 * self.n_params + self.n_params ** 2
        self.per_chain_counters = bool(per_chain_counters) or self.n_moves > 1
        self.dtype = dtype
        self.sim = OracleSim(n_chains, chain_offset=chain_offset, potential=potential, beta=beta, sigma=sigma,
                             weight=weight, seed=seed, sweepstep=sweepstep, reward_expr=reward_expr, dtype=dtype,
                             scale_expr=scale_expr, proposal=proposal, n_params=n_params, classes=classes, class_of_move=class_of_move)
        self.sim.set_x(np.zeros(self.n_chains))
        self.threads = 1
        self.reduce_columns = 7

    def close(self):
        self.sim.close()

    def set_reduce_columns(self, columns):
        """HipEngine.set_reduce_columns: the sums over x that reductions form (1 sum e, 2 sum x, 4 sum x^2); the others' records
        stay empty."""
        assert 0 <= int(columns) <= 7
        self.reduce_columns = int(columns)

    def upload_state(self, x, beta=None):
        self.sim.set_x(x)
        if beta is not None:
            self.sim.set_beta(beta)

    def init_uniform(self, lo, hi):
        self.sim.init_uniform(lo, hi)

    def download_state(self, want_e=True):
        x, e = self.sim.state()
        return x, (e if want_e else None)

    def download_counters(self):
        return self.sim.counters()

    def counter_totals(self):
        acc, tot = self.sim.counters()
        return acc.sum(axis=1), tot.sum(axis=1)

    def sweep(self, n_sweeps=1):
        self.sim.make_steps(n_sweeps, self.threads)

    @property
    def step(self):
        return self.sim.step

    @step.setter
    def step(self, t):
        self.sim.step = t

    def reduce_exact(self):
        """(records, steps counted): what HipEngine.reduce_end_exact returns.  A K = 1 engine without per-chain counters
        carries the pool-wide accepted TOTAL in the ratio record (amc_reduce_end_exact)."""
        rec = self.sim.callback_records()
        for c in range(3):
            if not self.reduce_columns & (1 << c):
                rec[c] = 0.0
        acc, tot = self.sim.counters()
        if self.n_moves == 1 and not self.per_chain_counters:
            rec[4] = xsum_q([float(acc.sum())], 0)
        # the MH steps counted per chain (every chain has taken the same number: the total_calls of any one add up to it)
        return rec, int(tot[:, 0].sum()) if tot.size else 0

    def reduce_records_value(self, records, steps_counted):
        out = xsum_round(records)
        out[np.asarray(records).reshape(-1, XS_WORDS)[:, 0] == 0.0] = np.nan           # a sum nobody asked for
        if self.n_moves == 1 and not self.per_chain_counters:
            with np.errstate(invalid="ignore", divide="ignore"):
                out[4] = out[4] / np.float64(steps_counted)          # 0/0 = NaN before the first step, like the reference
        return out

    def reduce(self):
        return self.reduce_records_value(*self.reduce_exact())

    def upload_counters(self, accepted, total=None):
        a = np.ascontiguousarray(accepted, dtype=np.int64).reshape(self.n_moves, self.n_chains)
        t = a * 0 + self.sim.step if total is None else np.ascontiguousarray(total, dtype=np.int64).reshape(a.shape)
        p = C.POINTER(C.c_int64)
        self.sim.lib.amo_set_counters(self.sim.h, a.ctypes.data_as(p), t.ctypes.data_as(p))

    def histogram(self, lo, hi, n_bins):
        x, _ = self.sim.state()
        out = np.zeros(n_bins + 3, dtype=np.uint64)
        inv_w = n_bins / (hi - lo)
        nan = np.isnan(x)
        below, above = (x < lo) & ~nan, (x >= hi) & ~nan
        inside = ~(nan | below | above)
        b = np.minimum(((x[inside] - lo) * inv_w).astype(np.int64), n_bins - 1)
        out[:n_bins] = np.bincount(b, minlength=n_bins).astype(np.uint64)
        out[n_bins], out[n_bins + 1], out[n_bins + 2] = below.sum(), above.sum(), nan.sum()
        return out

    def histogram_accumulate(self, lo, hi, n_bins):
        key = (float(lo), float(hi), int(n_bins))
        assert getattr(self, "_hist_key", key) == key, "the running histogram has other bins"
        self._hist_key = key
        self._hist = getattr(self, "_hist", 0) + self.histogram(lo, hi, n_bins)

    def histogram_fetch(self, n_bins, reset=True):
        out = np.asarray(self._hist, dtype=np.uint64).copy()
        assert out.shape[0] == n_bins + 3
        if reset:
            del self._hist, self._hist_key
        return out

    def download_strided(self, first, stride, count):
        x, _ = self.sim.state()
        return x[first:first + count * stride:stride][:count].copy()

    @property
    def estimator_step(self):
        return self.sim.lib.amo_get_estimator_step(self.sim.h)

    @estimator_step.setter
    def estimator_step(self, t):
        self.sim.lib.amo_set_estimator_step(self.sim.h, int(t))

    def reduce_begin(self):
        q = self.__dict__.setdefault("_pending", [])
        assert len(q) < 2, "two reductions are already in flight"       # amc_reduce_begin: AMC_ERR_STATE
        q.append(self.reduce_exact())

    def sweep_reduce_begin(self, n_sweeps=1):
        self.sweep(n_sweeps)
        self.reduce_begin()

    def reduce_end_exact(self):
        return self._pending.pop(0)           # the oldest

    def reduce_end(self):
        return self.reduce_records_value(*self.reduce_end_exact())

    def set_parameters(self, k, p):
        if self.n_params > 1:
            self.sim.set_theta(k, p)
        else:
            self.sim.set_sigma(k, float(np.asarray(p).reshape(-1)[0]))

    def get_parameters(self, k):
        return self.sim.get_theta(k) if self.n_params > 1 else np.array([self.sim.get_sigma(k)])

    def parameters_begin(self):           # amc_parameters_begin: sigma as of now, fetched later
        assert getattr(self, "_params_pending", None) is None, "a parameter read was begun while another was in flight"
        self._params_pending = (np.array([self.sim.get_theta(k) for k in range(self.n_moves)]) if self.n_params > 1 else
                                np.array([self.sim.get_sigma(k) for k in range(self.n_moves)]))

    def parameters_end(self):
        out, self._params_pending = self._params_pending, None
        assert out is not None, "no parameter read in flight"
        return out

    def pg_estimate(self, learn_ids, q_batch):
        if self.n_params > 1:
            rec = self.sim.pg_estimate_records_vec(learn_ids, q_batch)
            return xsum_round(rec).reshape(rec.shape[0], rec.shape[1])
        return self.sim.pg_estimate(learn_ids, q_batch)

    def pg_estimate_exact(self, learn_ids, q_batch):
        if self.n_params > 1:
            return self.sim.pg_estimate_records_vec(learn_ids, q_batch)
        return self.sim.pg_estimate_records(learn_ids, q_batch)

    # device-resident estimator / update of the HIP engine (amc_pg_accumulate / amc_pg_update / amc_pgmc_steps),
    # restated on the host: running sums per move, learning step by the oracle's amo_learning_step
    def pg_accumulate(self, learn_ids, q_batch):
        gd = self.pg_estimate(learn_ids, q_batch)
        acc = self.__dict__.setdefault("_gd_acc", {})
        for row, k in zip(gd, learn_ids):
            acc[k] = acc.get(k, np.zeros(self.gd_stride)) + row
        self.pgmc_calls = getattr(self, "pgmc_calls", 0)

    def pg_update(self, learn_ids, kinds, hyper0, hyper1):
        acc = self.__dict__.setdefault("_gd_acc", {})
        names = {v: k for k, v in OPTIMISERS.items()}
        for k, kind, h0, h1 in zip(learn_ids, kinds, hyper0, hyper1):
            a = acc.get(k, np.zeros(self.gd_stride))
            if self.n_params > 1:
                with np.errstate(invalid="ignore", divide="ignore"):
                    gd = a[:-1] / a[-1]                                   # average (gradients.jl:83-85)
                nxt = learning_step_vec(names[int(kind)], float(h0), float(h1), self.sim.get_theta(k), gd)
                if nxt is not None and np.all(np.isfinite(nxt)):         # as the engine: a step that cannot be taken is not applied
                    self.sim.set_theta(k, nxt)
                acc[k] = np.zeros(self.gd_stride)
                continue
            gd4 = [a[i] / a[4] for i in range(4)]
            self.sim.set_sigma(k, learning_step(names[int(kind)], float(h0), float(h1), self.sim.get_sigma(k), gd4))
            acc[k] = np.zeros(5)

    def pg_get_accumulated(self, learn_ids):
        acc = self.__dict__.setdefault("_gd_acc", {})
        return np.array([acc.get(k, np.zeros(self.gd_stride)) for k in learn_ids]).reshape(len(learn_ids), self.gd_stride)

    def pg_set_accumulated(self, learn_ids, rows):
        acc = self.__dict__.setdefault("_gd_acc", {})
        for k, row in zip(learn_ids, np.asarray(rows, dtype=np.float64).reshape(len(learn_ids), self.gd_stride)):
            acc[k] = row.copy()

    def pgmc_steps(self, n_steps, learn_ids, q_batch, kinds=None, hyper0=(), hyper1=(), reduce_begin=False):
        self.pgmc_calls = getattr(self, "pgmc_calls", 0) + 1
        for _ in range(int(n_steps)):
            self.sweep(1)
            self.pg_accumulate(learn_ids, q_batch)
            if kinds is not None:
                self.pg_update(learn_ids, kinds, hyper0, hyper1)
        if reduce_begin:                      # amc_pgmc_steps_reduce_begin: the sums of the state the last step leaves
            self.reduce_begin()

    def sync(self):
        pass

    def timing_begin(self):
        pass

    def timing_end(self):
        return 0.0
