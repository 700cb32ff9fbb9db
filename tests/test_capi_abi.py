"""C-ABI checks that need no GPU: libamc.so loads, exports every symbol include/amc.h declares,
agrees with the Python struct layout, and fails loudly (no CPU fallback) without a device."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "amc.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(amc_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(amc):
    lib = amc.load()
    names = declared_functions()
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), f"libamc.so does not export {name} declared in include/amc.h"


def test_library_exports_nothing_but_the_c_abi():
    """The boundary is the header: every function in the dynamic symbol table of libamc.so is an entry point of include/amc.h
    (the library is built with hidden visibility) -- beside them only what hipcc keeps visible of the HIP kernels in namespace
    amc (their handle objects and the launch stubs of explicit instantiations)."""
    so = os.path.join(ROOT, "montecarlo_amd", "libamc.so")
    out = subprocess.run(["nm", "-D", "--defined-only", so], check=True, capture_output=True, text=True).stdout
    functions = sorted(ln.split()[2] for ln in out.splitlines()
                       if len(ln.split()) == 3 and ln.split()[1] in ("T", "W")
                       and not ln.split()[2].startswith(("_ZN3amc", "_ZNSt", "_ZNKSt", "_ZSt", "_ZN9__gnu_cxx")))     # kernels; libstdc++ templates
    assert functions == declared_functions()


def test_python_binding_covers_the_header(amc):
    lib = amc.load()
    for name in declared_functions():
        fn = getattr(lib, name)
        assert fn.argtypes is not None or name in ("amc_last_error", "amc_version"), f"{name} has no ctypes signature"


def test_struct_layout_matches_c(amc, tmp_path):
    """sizeof / offsets of amc_config as the C compiler sees them == the ctypes mirror."""
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "amc.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu\\n",'
                   "sizeof(amc_config), offsetof(amc_config, n_chains), offsetof(amc_config, beta),"
                   "offsetof(amc_config, sigma), offsetof(amc_config, seed), offsetof(amc_config, stream));return 0;}\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    S = amc.AmcConfig
    assert got == [C.sizeof(S), S.n_chains.offset, S.beta.offset, S.sigma.offset, S.seed.offset, S.stream.offset]


def test_header_is_plain_c(tmp_path):
    src = tmp_path / "plain.c"
    src.write_text('#include "amc.h"\nint main(void){return AMC_OK;}\n')
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    "-c", str(src), "-o", str(tmp_path / "plain.o")], check=True)


def test_version(amc):
    assert amc.load().amc_version() >= 1


def test_argument_validation_needs_no_gpu(amc):
    """Bad arguments are rejected before any device is touched (reference asserts: metropolis.jl:249-251)."""
    lib = amc.load()
    h = C.c_void_p()
    cfg = amc.AmcConfig()
    assert lib.amc_create(C.byref(cfg), C.byref(h)) == -1          # struct_size = 0: ABI guard
    assert b"struct_size" in lib.amc_last_error()
    sig = (C.c_double * 2)(0.1, 0.2)
    wgt = (C.c_double * 2)(0.5, 0.6)
    cfg.struct_size = C.sizeof(amc.AmcConfig)
    cfg.n_chains, cfg.n_chains_global, cfg.n_moves, cfg.sweepstep = 10, 10, 2, 1
    cfg.sigma, cfg.weight = sig, wgt
    assert lib.amc_create(C.byref(cfg), C.byref(h)) == -1 and b"sum to 1" in lib.amc_last_error()
    wgt[1] = 0.5
    cfg.chain_offset = 3
    cfg.n_chains_global = 13
    assert lib.amc_create(C.byref(cfg), C.byref(h)) == -1 and b"even" in lib.amc_last_error()
    cfg.chain_offset = 0
    sig[0] = -1.0
    assert lib.amc_create(C.byref(cfg), C.byref(h)) == -1 and b"sigma" in lib.amc_last_error()
    assert lib.amc_sweep(None, 1) == -1 and lib.amc_reduce(None, None) == -1


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_fails_loudly_without_a_gpu(amc):
    """No silent CPU path: constructing the engine without a device raises with a clear message."""
    assert amc.device_count() == 0
    with pytest.raises(amc.AmcError, match="no HIP device"):
        amc.HipEngine(n_chains=10, sigma=[0.1], weight=[1.0])
    import montecarlo_amd as ma
    chains = ma.ParticleChains.uniform(10, 2.0)
    pool = (ma.Move(ma.Displacement(), ma.StandardGaussian(), [0.1], 1.0),)
    with pytest.raises(ma.AmcError):
        ma.Metropolis(chains, pool=pool)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under montecarlo_amd/ may mention it."""
    pkg = os.path.join(ROOT, "montecarlo_amd")
    for dirpath, _d, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "oracle_lib" not in text and "libamc_oracle" not in text and "amo_" not in text, fn


@pytest.mark.gpu
def test_plain_c_client_links_and_runs(tmp_path):
    """tests/aux/c_usage.c: a C99 program linked against libamc.so only (the boundary a Julia ccall / cgo / JNI binding
    sees), built-in and run-time compiled potentials, checked against quadrature."""
    import math
    from scipy import integrate
    lib_dir = os.path.join(ROOT, "montecarlo_amd")
    exe = tmp_path / "c_usage"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "aux", "c_usage.c"), "-o", str(exe), os.path.join(lib_dir, "libamc.so"),
                    "-lm", f"-Wl,-rpath,{lib_dir}"], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().splitlines()
    dw = dict(zip(lines[0].split()[1::2], map(float, lines[0].split()[2::2])))
    assert dw["mean_U"] == pytest.approx(0.272864, abs=5e-3) and dw["mean_x2"] == pytest.approx(0.852136, abs=8e-3)
    assert dw["acc0"] > dw["acc1"] > 0.2
    U = lambda x: 0.5 * x * x + 0.25 * x ** 4
    z = integrate.quad(lambda x: math.exp(-2 * U(x)), -5, 5)[0]
    mean_u = integrate.quad(lambda x: U(x) * math.exp(-2 * U(x)), -5, 5)[0] / z
    cu = dict(zip(lines[1].split()[1::2], map(float, lines[1].split()[2::2])))
    assert cu["mean_U"] == pytest.approx(mean_u, abs=5e-3) and cu["mean_x"] == pytest.approx(0.0, abs=8e-3)
    f32 = dict(zip(lines[2].split()[2::2], map(float, lines[2].split()[3::2])))     # "f32 harmonic mean_U .. acc .."
    assert f32["mean_U"] == pytest.approx(0.25, abs=5e-3) and f32["acc"] == pytest.approx(0.9365, abs=5e-3)
    assert "n_moves must be in" in lines[3]
    pg = dict(zip(lines[4].split()[1::2], map(float, lines[4].split()[2::2])))     # "pgmc sigma2 .. mean_e .. acc0 .. acc1 .. hip .."
    assert pg["sigma2"] == pytest.approx(1.2, abs=0.2) and pg["mean_e"] == pytest.approx(0.25, abs=1e-2)      # pgmc_test.jl:45,50
    assert pg["acc0"] > pg["acc1"] > 0.3 and pg["hip"] > 60000000


def test_every_entry_survives_null_arguments():
    """"Errors: negative amc_status + amc_last_error(); nothing throws" (INTEGRATION.md): every entry point of the ABI called with a
    NULL handle and zeros / NULLs for every other argument -- in a process of its own, so that a fault would show up as the name
    it died on.  No GPU involved: the argument checks come first."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "aux", "null_handle_calls.py")], capture_output=True, text=True, timeout=120)
    lines = r.stdout.split("\n")
    assert r.returncode == 0 and "done" in lines, (r.returncode, lines[-3:], r.stderr[-500:])
    got = dict(ln.split() for ln in lines if len(ln.split()) == 2)
    assert len(got) >= 65, len(got)          # (every entry that takes an argument)
    ok_with_nothing = {"amc_destroy", "amc_runtime_info"}          # destroying nothing and asking for no answer are no errors
    assert {n for n, rc in got.items() if rc == "0"} == ok_with_nothing
    assert all(rc == "-1" for n, rc in got.items() if n not in ok_with_nothing), {n: rc for n, rc in got.items() if rc not in ("-1", "0")}


@pytest.mark.gpu
def test_every_entry_survives_null_arguments_on_a_live_handle(gpu):
    """The same with a LIVE handle (three kinds): NULL arrays and zero counts are refused or mean "nothing", never a fault, and the
    handle sweeps on afterwards."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "aux", "valid_handle_null_calls.py")], capture_output=True, text=True, timeout=300)
    lines = r.stdout.split("\n")
    assert r.returncode == 0 and "done" in lines, (r.returncode, lines[-3:], r.stderr[-800:])
    got = [ln.split() for ln in lines if len(ln.split()) == 3]
    assert len(got) >= 3 * 45 and all(int(rc) <= 0 or name == "amc_pg_route" for _, name, rc in got), [g for g in got if int(g[2]) > 0]
