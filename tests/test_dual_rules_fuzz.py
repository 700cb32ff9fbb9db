"""The product's dual numbers (montecarlo_amd/csrc/amc_dual.h: what the run-time compiled estimator differentiates a script's
logq with) against the oracle twin's (tests/oracle_lib.py: Dn<N>) on RANDOM expressions -- both compiled for the host by g++ in
one translation unit, the vocabulary's log / exp bound to the oracle's on both sides, so what is compared is the differentiation
RULES and their operation order (ForwardDiff 0.10's, /root/reference/src/PolicyGuided/gradients.jl:28-33 takes its gradients
from that package): values and all partials must be the same bits at every sample point.  No GPU: the device build of the same
header is held to the twin by tests/test_autodiff.py on five policies; this one covers the operator table."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "montecarlo_amd", "csrc")
N_EXPR, N_POINTS, P = 160, 64, 3


def _expressions(seed=20261005, n=N_EXPR):
    rng = np.random.default_rng(seed)
    leaves = ["theta0", "theta1", "theta2", "sigma", "delta", "x", "0.5", "2.0", "1.25", "3.0", "(-0.75)"]

    def lit():
        return str(rng.choice(["0.5", "2.0", "1.25", "3.0", "0.1", "7.0"]))

    def gen(depth):
        if depth == 0 or rng.random() < 0.12:
            return str(rng.choice(leaves))
        a, b, c = gen(depth - 1), gen(depth - 1), gen(depth - 1)
        forms = [f"({a} + {b})", f"({a} - {b})", f"({a} * {b})", f"(-{a})", f"(+{a})",
                 f"({a} / (fabs({b}) + 0.75))", f"({a} / {lit()})", f"({lit()} / (fabs({a}) + 1.0))",
                 f"({lit()} * {a})", f"({a} * {lit()})", f"({lit()} + {a})", f"({a} - {lit()})", f"({lit()} - {a})",
                 f"amc_log(fabs({a}) + 0.5)", f"amc_exp(-fabs({a}))", f"sqrt(fabs({a}) + 0.25)", f"fabs({a})",
                 f"fma({a}, {b}, {c})", f"fma({a}, {lit()}, {c})", f"fma({lit()}, {a}, {c})", f"fma({a}, {b}, {lit()})",
                 f"fma({a}, {lit()}, {lit()})", f"fma({lit()}, {a}, {lit()})", f"fma({lit()}, {lit()}, {a})",
                 # a comparison looks at the values; both arms are forced to the same (dual) type by the parameter in each
                 f"(({a}) < ({b}) ? (theta0 * {a}) : (theta1 + {b}))", f"(({a}) >= {lit()} ? (theta2 - {a}) : (theta0 / (fabs({b}) + 1.5)))"]
        return str(rng.choice(forms))

    return [gen(4) for _ in range(n)]


def _build(tmp_path, exprs):
    src = tmp_path / "dual_fuzz.cpp"
    so = tmp_path / "dual_fuzz.so"
    prod_cases = "\n".join(f"    case {k}: {{ const Dual<{P}> r_ = as_dual<{P}>({e}); out[0] = r_.v; for (int i = 0; i < {P}; ++i) out[1 + i] = r_.d[i]; break; }}"
                           for k, e in enumerate(exprs))
    twin_cases = "\n".join(f"    case {k}: {{ const Dn<{P}> r_ = dn_of<{P}>({e}); out[0] = r_.v; for (int i = 0; i < {P}; ++i) out[1 + i] = r_.d[i]; break; }}"
                           for k, e in enumerate(exprs))
    src.write_text(f"""
#include <cmath>
extern "C" {{ double amo_exp(double); double amo_log(double); }}
// the device header on the host: its attributes mean nothing here, its log / exp of plain numbers are the oracle's
#define __device__
#define __forceinline__ inline
namespace amc {{
static inline double log_f64(double v) {{ return amo_log(v); }}
static inline double exp_f64(double v, const double*) {{ return amo_exp(v); }}
}}
#include "amc_dual.h"
namespace amc {{
#define amc_log(v) (::amc::log_f64((v)))
#define amc_exp(v) (::amc::exp_f64((v), (const double*)0))
void product_eval(int k, const double* th, double delta, double x, double* out)
{{
    const Dual<{P}> theta0 = dual_var<{P}>(th[0], 0), theta1 = dual_var<{P}>(th[1], 1), theta2 = dual_var<{P}>(th[2], 2), sigma = theta0;
    (void)sigma; (void)theta1; (void)theta2;
    switch (k) {{
{prod_cases}
    }}
}}
#undef amc_log
#undef amc_exp
}}
using std::sqrt; using std::fabs; using std::fma;
{oracle_lib._DUAL_PROLOGUE}
#define amc_log(v) amo_log(v)
#define amc_exp(v) amo_exp(v)
static void twin_eval(int k, const double* th, double delta, double x, double* out)
{{
    const Dn<{P}> theta0 = dn_var<{P}>(th[0], 0), theta1 = dn_var<{P}>(th[1], 1), theta2 = dn_var<{P}>(th[2], 2), sigma = theta0;
    (void)sigma; (void)theta1; (void)theta2;
    switch (k) {{
{twin_cases}
    }}
}}
extern "C" void fuzz_eval(int k, int n, const double* pts, double* prod, double* twin)
{{
    for (int i = 0; i < n; ++i) {{
        const double* p = pts + 5 * i;
        amc::product_eval(k, p, p[3], p[4], prod + {P + 1} * i);
        twin_eval(k, p, p[3], p[4], twin + {P + 1} * i);
    }}
}}
""")
    oracle_lib.load()
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-mfma", "-fno-math-errno", "-Wno-unknown-pragmas",
                        f"-I{CSRC}", str(src), "-o", str(so), oracle_lib.LIB_PATH, "-lm", f"-Wl,-rpath,{oracle_lib.ORACLE_DIR}"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    lib = C.CDLL(str(so))
    lib.fuzz_eval.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    return lib


def test_dual_number_rules_equal_the_twins_on_random_expressions(tmp_path):
    exprs = _expressions()
    assert len(set(exprs)) > N_EXPR * 0.9
    lib = _build(tmp_path, exprs)
    rng = np.random.default_rng(7)
    pts = rng.normal(size=(N_POINTS, 5)) * np.array([1.5, 1.0, 2.0, 1.0, 3.0])
    pts[:4] = [[0.0, 1.0, -1.0, 0.0, 0.0], [-0.0, 0.0, 0.0, -0.0, 1.0], [1e-300, 1e300, -1e300, 1e-310, 2.0], [np.inf, 1.0, -2.0, 0.5, -np.inf]]
    pts = np.ascontiguousarray(pts)
    used = {"dual results": 0, "finite partials": 0, "nonzero partials": 0}
    for k, e in enumerate(exprs):
        prod = np.full((N_POINTS, P + 1), 7.0)
        twin = np.full((N_POINTS, P + 1), -7.0)
        lib.fuzz_eval(k, N_POINTS, pts.ctypes.data, prod.ctypes.data, twin.ctypes.data)
        same = (prod.view(np.uint64) == twin.view(np.uint64)) | (np.isnan(prod) & np.isnan(twin))
        assert same.all(), (k, e, pts[np.argwhere(~same)[0][0]], prod[~same][:3], twin[~same][:3])
        used["dual results"] += 1
        used["finite partials"] += int(np.isfinite(prod[:, 1:]).sum())
        used["nonzero partials"] += int((prod[4:, 1:] != 0).sum())
    # the comparison is not vacuous: the partials are finite, and a good share of them non-zero (an expression need not mention every parameter)
    assert used["finite partials"] > 0.9 * N_EXPR * N_POINTS * P and used["nonzero partials"] > 0.25 * N_EXPR * (N_POINTS - 4) * P, used


def test_dual_partials_agree_with_central_differences(tmp_path):
    """Independent of both rule tables: the partials are derivatives (central differences of the value part, smooth expressions
    only: no fabs kink or comparison near the sample points)."""
    exprs = ["-((delta-theta0)*(delta-theta0))/(2.0*theta1*theta1) - amc_log(theta1) + theta2*x",
             "amc_exp(-theta0*theta0) * sqrt(theta1*theta1 + 1.0) / (theta2*theta2 + 2.0)",
             "fma(theta0, theta1, theta2) / (1.0 + amc_exp(delta*theta2))",
             "amc_log(1.0 + theta0*theta0 + theta1*theta1*x*x) - 3.0/(2.0 + theta2*theta2)"]
    lib = _build(tmp_path, exprs)
    rng = np.random.default_rng(11)
    pts = np.ascontiguousarray(rng.normal(size=(16, 5)) * 0.7 + np.array([0.3, 1.5, -0.4, 0.2, 0.5]))
    for k in range(len(exprs)):
        out = np.zeros((16, P + 1)); twin = np.zeros_like(out)
        lib.fuzz_eval(k, 16, pts.ctypes.data, out.ctypes.data, twin.ctypes.data)
        for p in range(P):
            h = 1e-6
            hi, lo = pts.copy(), pts.copy()
            hi[:, p] += h; lo[:, p] -= h
            vh = np.zeros((16, P + 1)); vl = np.zeros((16, P + 1))
            lib.fuzz_eval(k, 16, hi.ctypes.data, vh.ctypes.data, twin.ctypes.data)
            lib.fuzz_eval(k, 16, lo.ctypes.data, vl.ctypes.data, twin.ctypes.data)
            fd = (vh[:, 0] - vl[:, 0]) / (2 * h)
            assert np.allclose(out[:, 1 + p], fd, rtol=2e-6, atol=2e-8), (k, p, out[:, 1 + p], fd)


def _policy_expressions(seed, n, n_params):
    """Random log-densities over the vocabulary, finite for every (delta, x, theta) the chains can reach: they need not normalise --
    parity is about operations, not about sampling the right distribution."""
    rng = np.random.default_rng(seed)
    thetas = [f"theta{i}" for i in range(n_params)]
    leaves = thetas + ["sigma", "delta", "x", "0.5", "2.0", "1.25"]

    def lit():
        return str(rng.choice(["0.5", "2.0", "1.25", "3.0"]))

    def gen(depth):
        if depth == 0 or rng.random() < 0.15:
            return str(rng.choice(leaves))
        a, b, c = gen(depth - 1), gen(depth - 1), gen(depth - 1)
        forms = [f"({a} + {b})", f"({a} - {b})", f"({a} * {b})", f"(-{a})", f"({a} / (fabs({b}) + 0.75))", f"({lit()} / (fabs({a}) + 1.0))",
                 f"({lit()} * {a})", f"({a} - {lit()})", f"amc_log(fabs({a}) + 0.5)", f"amc_exp(-fabs({a}))", f"sqrt(fabs({a}) + 0.25)",
                 f"fma({a}, {b}, {c})", f"fma({a}, {lit()}, {c})", f"(({a}) < ({b}) ? (theta0 * {a}) : (theta0 + {b}))"]
        return str(rng.choice(forms))

    out = []
    while len(out) < n:
        e = gen(3)
        # a density the acceptance can work with: bounded above by the Gaussian term, every parameter and delta in it
        body = " + ".join(f"0.01*{t}" for t in thetas)
        out.append(f"-(delta*delta)/(2.0*(sigma*sigma)) - amc_log(fabs(sigma) + 0.1) + 0.05*amc_exp(-fabs({e})) + 0.02*sqrt(fabs({e}) + 1.0) + {body}")
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("n_params", [1, 2])
def test_random_densities_differentiated_on_the_device(gpu, n_params):
    """The DEVICE build of the dual numbers (hiprtc, the estimator kernel) against the twin on random log-densities nobody wrote a
    derivative for: sweeps, counters and GradientData records equal to the oracle's, bit for bit -- Float64 and Float32 state."""
    for i, logq in enumerate(_policy_expressions(100 + n_params, 3, n_params)):
        dtype = "f32" if i == 2 else "f64"
        sample = "sigma*z" if n_params == 1 else "theta1*0.1 + sigma*z"
        kw = dict(n_chains=3001, potential="harmonic", beta=2.0, weight=[0.5, 0.5], seed=31 + i, proposal=(sample, logq, None), dtype=dtype)
        if n_params == 1:
            kw["sigma"] = [0.4, 0.9]
        else:
            kw.update(sigma=[[0.4, 0.2], [0.9, -0.3]], n_params=2)
        eng, ref = gpu.HipEngine(**kw), oracle_lib.OracleEngine(**kw)
        for e in (eng, ref):
            e.init_uniform(-2.0, 2.0)
            e.sweep(12)
        assert np.array_equal(eng.download_state()[0].view(np.uint64), ref.download_state()[0].view(np.uint64)), logq
        a, t = eng.download_counters()
        ao, to = ref.download_counters()
        assert np.array_equal(a, ao) and np.array_equal(t, to) and 0.05 < a.sum() / t.sum() < 0.999, logq
        got, want = eng.pg_estimate_exact([0, 1], 2), ref.pg_estimate_exact([0, 1], 2)
        assert np.array_equal(got, want, equal_nan=True), logq
        vals = eng.pg_estimate([0, 1], 1)
        assert np.all(np.isfinite(vals)) and np.any(vals[:, 1:1 + n_params] != 0.0), logq
        eng.close()
    oracle_lib.install_custom_proposal(None)
    oracle_lib.install_vector_policy(1, None)
