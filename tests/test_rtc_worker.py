"""The run-time compiler runs in a child process: a compiler that dies is an error code, not a dead host.

Reference convention: a model that cannot be used RAISES -- error("No ... is defined"), src/metropolis.jl:35 -- and run! closes
its files in `finally` (src/simulation.jl:176,194-199).  hiprtcCompileProgram inside the engine's process turned an LLVM fatal
error (hipcc 7.2: "illegal VGPR to SGPR copy" on some class-pool estimator forms) into abort() of the host -- a Julia session.
montecarlo_amd/amc_rtc_worker (amc_rtc_worker.cpp) now holds the compiler; libamc.so starts it per instantiation
(amc_rtc.hip build_in_child).  The CPU half of this file provokes the failures with the worker's fault hook; the GPU half asks
for the kernel form that really kills hipcc 7.2 and checks that the call falls back to the form that builds, bit for bit."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _in_fresh_process(code, env=None, timeout=300):
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r)\n" % ROOT + code],
                       env=dict(os.environ, **(env or {})), capture_output=True, text=True, timeout=timeout)
    return r


def test_worker_is_built_beside_the_library_and_is_not_a_gpu_program():
    exe = os.path.join(ROOT, "montecarlo_amd", "amc_rtc_worker")
    assert os.access(exe, os.X_OK), "make -C montecarlo_amd/csrc builds it (__graft_entry__.build())"
    needed = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "amdhip64" not in needed and "hsa-runtime" not in needed         # no HIP runtime: hiprtc (comgr) alone, through dlopen
    r = subprocess.run([exe], input=b"not a request", capture_output=True, timeout=60)
    assert r.returncode == 2 and b"started by libamc.so" in r.stderr


@pytest.mark.parametrize("fault,needle", [
    ("abort", "the compiler died (signal 6"),                 # what LLVM's report_fatal_error ends in
    ("hang", "did not come back within 2 s"),                 # AMC_RTC_TIMEOUT_S
    ("garbage", "not an answer"),
])
def test_a_compiler_that_dies_hangs_or_babbles_is_amc_err_compile(fault, needle):
    code = ("from montecarlo_amd import _capi as A\n"
            "try:\n"
            "    A.potential_check('x*x*x*x + 0.5*x')\n"
            "    print('NO ERROR')\n"
            "except A.AmcError as e:\n"
            "    print('ERR', e)\n"
            "import os; del os.environ['AMC_RTC_WORKER_FAULT']\n"
            "assert A.potential_check('x*x*x*x + 0.25*x') == ''\n"         # the process lives, and the next build works
            "print('ALIVE')\n")
    r = _in_fresh_process(code, env=dict(AMC_RTC_WORKER_FAULT=fault, AMC_RTC_TIMEOUT_S="2"))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert lines[-1] == "ALIVE" and lines[0].startswith("ERR amc error -7:"), r.stdout       # AMC_ERR_COMPILE
    assert needle in lines[0]
    if fault == "abort":
        assert "LLVM ERROR: injected fatal error" in r.stdout                  # the compiler's last words reach amc_last_error()


def test_a_missing_worker_is_said_so_and_the_in_process_knob_still_compiles(amc):
    code = ("from montecarlo_amd import _capi as A\n"
            "try:\n"
            "    A.potential_check('x*x + 0.25*x*x*x*x')\n"
            "except A.AmcError as e:\n"
            "    print('ERR', e)\n"
            "import os; os.environ['AMC_RTC_IN_PROCESS'] = '1'\n"
            "assert A.potential_check('x*x + 0.25*x*x*x*x') == ''\n"
            "print('ALIVE')\n")
    r = _in_fresh_process(code, env=dict(AMC_RTC_WORKER="/nonexistent/amc_rtc_worker"))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "amc error -7" in r.stdout and "missing or not executable" in r.stdout and r.stdout.strip().endswith("ALIVE")


def test_a_death_is_remembered_in_the_process_and_in_the_cache_directory(tmp_path):
    """The compiler is not asked twice for what killed it: the answer is kept per process and, with AMC_RTC_CACHE_DIR, in a
    `.broken` note whose name carries the compiler release (a new release is asked again)."""
    code = ("import time\n"
            "from montecarlo_amd import _capi as A\n"
            "for i in range(2):\n"
            "    try:\n"
            "        A.potential_check('x*x*x*x - x')\n"
            "    except A.AmcError as e:\n"
            "        print('ERR', e)\n")
    env = dict(AMC_RTC_CACHE_DIR=str(tmp_path), AMC_RTC_WORKER_FAULT="abort")
    r = _in_fresh_process(code, env=env)
    assert r.returncode == 0 and r.stdout.count("the compiler died") == 2, r.stdout + r.stderr[-1000:]
    notes = [f for f in os.listdir(tmp_path) if f.endswith(".broken")]
    assert len(notes) == 1 and "signal 6" in open(tmp_path / notes[0]).read()
    # another process, a healthy compiler: the note answers (nobody is started), until it is removed
    r = _in_fresh_process(code, env=dict(AMC_RTC_CACHE_DIR=str(tmp_path), AMC_RTC_WORKER="/nonexistent/worker"))
    assert r.stdout.count("the compiler died") == 2 and "missing or not executable" not in r.stdout
    os.remove(tmp_path / notes[0])
    r = _in_fresh_process("from montecarlo_amd import _capi as A\nassert A.potential_check('x*x*x*x - x') == ''\nprint('OK')\n",
                          env=dict(AMC_RTC_CACHE_DIR=str(tmp_path)))
    assert r.stdout.strip() == "OK", r.stderr[-1000:]


def test_a_script_that_does_not_compile_is_still_the_callers_error(amc):
    with pytest.raises(amc.AmcError, match=r"amc error -1: custom potential does not compile.*undeclared identifier"):
        amc.potential_check("x*undefined_symbol")


def test_no_sigpipe_when_the_child_dies_before_reading_its_request():
    """The request is some hundred kilobytes of kernel source; a child that is gone before it has read them must cost an
    EPIPE on a socket (MSG_NOSIGNAL), not a SIGPIPE that ends the host."""
    code = ("import signal\n"
            "signal.signal(signal.SIGPIPE, signal.SIG_DFL)\n"          # as in a C or Julia host (Python ignores SIGPIPE by default)
            "from montecarlo_amd import _capi as A\n"
            "for i in range(5):\n"
            "    try:\n"
            "        A.potential_check('x*x + %d.0' % i)\n"
            "    except A.AmcError as e:\n"
            "        pass\n"
            "print('ALIVE')\n")
    r = _in_fresh_process(code, env=dict(AMC_RTC_WORKER_FAULT="abort"))
    assert r.returncode == 0 and r.stdout.strip() == "ALIVE", (r.returncode, r.stderr[-1000:])


# ---------------------------------------------------------------- GPU -------------------------------------------------
from test_mixed_pool import CLASSES, CLASS_OF_MOVE, _kw, bits      # noqa: E402  (the four-move, three-class pool)


@pytest.mark.gpu
def test_the_form_that_kills_hipcc_is_an_error_code_and_the_call_falls_back(gpu, oracle):
    """All four moves of the three-class pool learn: the estimator's one-launch form is NL = 4 with a class switch inside the
    unrolled loop over moves -- the form hipcc 7.2 dies on ("illegal VGPR to SGPR copy", NOTES_r05.md section 3).  The engine
    asks for it (reference: one pass over all learnable moves, estimator.jl:111-134), gets AMC_ERR_COMPILE from the child,
    keeps the compiler's words, and takes one launch per move: records EQUAL to the oracle's.  Should a later compiler build the
    form, the route is the one-launch one and the records are the same."""
    M = 4099
    eng = gpu.HipEngine(**_kw(M))
    ref = oracle.OracleEngine(**_kw(M))
    eng.init_uniform(0.1, 2.0)
    ref.init_uniform(0.1, 2.0)
    eng.sweep(5)
    ref.sweep(5)
    one_launch, why = eng.pg_route(4, 2)
    if not one_launch:
        assert "the compiler" in why and len(why) > 40, why           # the child's stderr tail, not an empty shrug
        print("\nNL = 4 class form:", why[:300])
    g = eng.pg_estimate_exact([0, 1, 2, 3], 2)
    go = ref.pg_estimate_exact([0, 1, 2, 3], 2)
    assert np.array_equal(g, go)
    assert np.array_equal(bits(eng.download_state()[0]), bits(ref.download_state()[0]))
    # the process is alive and the engine usable: two learnable moves take the one-launch form (it builds)
    two, why2 = eng.pg_route(2, 2)
    assert two, why2
    # amc_pg_route's own answers: 2 = the whole time step in one launch, 1 = one estimator launch for all moves, 0 = one per move
    assert eng.pg_route_code(2, 2, fused=True)[0] == 2 and eng.pg_route_code(2, 2)[0] == 1
    assert eng.pg_route_code(4, 2, fused=True)[0] == (1 if one_launch else 0)          # more than two moves never fuse with the sweep
    assert np.array_equal(eng.pg_estimate_exact([1, 2], 2), ref.pg_estimate_exact([1, 2], 2))
    eng.close()
    oracle.install_policy_classes(None, None)


@pytest.mark.gpu
def test_class_pool_routes_agree_bit_for_bit(gpu, oracle):
    """One launch for all learnable moves (and the fused time step) against AMC_CLASS_PER_MOVE=1, the round-5 route: sigma,
    positions, counters after device-resident PGMC steps are the same bits, and both equal the oracle."""
    M = 6001
    out = []
    for forced in ("0", "1"):
        os.environ["AMC_CLASS_PER_MOVE"] = forced
        try:
            eng = gpu.HipEngine(**_kw(M, per_chain_counters=True))
        finally:
            del os.environ["AMC_CLASS_PER_MOVE"]
        eng.init_uniform(0.1, 2.0)
        one_launch, _ = eng.pg_route(2, 2, fused=True)
        assert one_launch == (forced == "0")
        eng.pgmc_steps(6, [1, 2], 2, [1, 2], [0.05, 0.02], [0.0, 0.0])
        eng.pg_accumulate([1, 2], 3)
        acc = eng.pg_get_accumulated([1, 2])
        out.append((eng.download_state()[0], [eng.get_parameters(k)[0] for k in range(4)], eng.download_counters(), acc))
        eng.close()
    assert np.array_equal(bits(out[0][0]), bits(out[1][0])) and out[0][1] == out[1][1]
    assert np.array_equal(out[0][2][0], out[1][2][0]) and np.array_equal(out[0][2][1], out[1][2][1])
    assert np.array_equal(out[0][3], out[1][3])
    o = oracle.OracleEngine(**_kw(M))
    o.init_uniform(0.1, 2.0)
    for _ in range(6):
        o.sweep(1)
        o.pg_accumulate([1, 2], 2)
        o.pg_update([1, 2], [1, 2], [0.05, 0.02], [0.0, 0.0])
    o.pg_accumulate([1, 2], 3)
    assert np.array_equal(out[0][3], o.pg_get_accumulated([1, 2]))
    assert np.array_equal(bits(out[0][0]), bits(o.download_state()[0]))
    assert out[0][1] == [o.get_parameters(k)[0] for k in range(4)]
    oracle.install_policy_classes(None, None)
