"""ctypes binding of libamc.so (include/amc.h) -- the only way this package computes.

There is no CPU path: if the shared library is missing, or no gfx950 device is
usable, construction raises.  The library is built in-tree by
``__graft_entry__.build()`` / ``make -C montecarlo_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libamc.so")

AMC_MAX_MOVES = 64
AMC_MAX_LEARN = 8
AMC_RED_HEADER = 4
AMC_GD_STRIDE = 5
AMC_XSUM_WORDS = 12      # doubles per record of a reproducible sum (include/amc.h)

POTENTIALS = {"harmonic": 0, "double_well": 1}
AMC_POTENTIAL_CUSTOM = 2


class AmcError(RuntimeError):
    """Non-zero amc_status from the C ABI (the Julia binding calls error(...) likewise)."""


class AmcConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("device", C.c_int32),
        ("n_chains", C.c_int64),
        ("chain_offset", C.c_int64),
        ("n_chains_global", C.c_int64),
        ("potential", C.c_int32),
        ("n_moves", C.c_int32),
        ("beta", C.c_double),
        ("sigma", C.POINTER(C.c_double)),
        ("weight", C.POINTER(C.c_double)),
        ("seed", C.c_uint64),
        ("sweepstep", C.c_int32),
        ("per_chain_counters", C.c_int32),
        ("stream", C.c_void_p),
        ("state_dtype", C.c_int32),
        ("reserved", C.c_int32),
    ]


# amc_state_dtype: Particle{T} (particle_1d.jl:9).  "f64" is the parity type; "f32" keeps x, beta, e, delta in Float32.
STATE_DTYPES = {"f64": 0, "f32": 1, "float64": 0, "float32": 1}


_lib = None
SIGNATURES: dict = {}          # name -> (restype, argtypes) of every bound entry point (filled by load(); tests walk it)


def load() -> C.CDLL:
    """Load libamc.so; fail loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AmcError(
            f"{LIB_PATH} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C montecarlo_amd/csrc). "
            "montecarlo_amd has no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    H = C.c_void_p
    dp = C.POINTER(C.c_double)
    i64p = C.POINTER(C.c_int64)
    sig = {
        "amc_last_error": (C.c_char_p, []),
        "amc_version": (C.c_int, []),
        "amc_device_count": (C.c_int, [C.POINTER(C.c_int)]),
        "amc_create": (C.c_int, [C.POINTER(AmcConfig), C.POINTER(H)]),
        "amc_create_custom": (C.c_int, [C.POINTER(AmcConfig), C.c_char_p, C.POINTER(H)]),
        "amc_create_model": (C.c_int, [C.POINTER(AmcConfig), C.c_char_p, C.c_char_p, C.POINTER(H)]),
        "amc_create_policy_model": (C.c_int, [C.POINTER(AmcConfig), C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(H)]),
        "amc_create_proposal_model": (C.c_int, [C.POINTER(AmcConfig), C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p,
                                                C.POINTER(H)]),
        "amc_create_action_model": (C.c_int, [C.POINTER(AmcConfig), C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p,
                                              C.c_char_p, C.c_char_p, C.POINTER(H)]),
        "amc_create_vector_policy_model": (C.c_int, [C.POINTER(AmcConfig), C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p,
                                                     C.POINTER(C.c_char_p), C.c_char_p, C.c_char_p, C.POINTER(H)]),
        "amc_n_params": (C.c_int, [H, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "amc_create_mixed_model": (C.c_int, [C.POINTER(AmcConfig), C.c_int, C.POINTER(C.c_int), C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p),
                                             C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_char_p),
                                             C.POINTER(H)]),
        "amc_potential_check": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int]),
        "amc_destroy": (C.c_int, [H]),
        "amc_upload_state": (C.c_int, [H, dp, dp]),
        "amc_init_uniform": (C.c_int, [H, C.c_double, C.c_double]),
        "amc_download_state": (C.c_int, [H, dp, dp]),
        "amc_download_counters": (C.c_int, [H, i64p, i64p]),
        "amc_counter_totals": (C.c_int, [H, i64p, i64p]),
        "amc_upload_counters": (C.c_int, [H, i64p, i64p]),
        "amc_set_counter_totals": (C.c_int, [H, i64p, C.c_uint64]),
        "amc_histogram": (C.c_int, [H, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_uint64)]),
        "amc_histogram_accumulate": (C.c_int, [H, C.c_double, C.c_double, C.c_int]),
        "amc_histogram_fetch": (C.c_int, [H, C.POINTER(C.c_uint64), C.c_int, C.c_int]),
        "amc_download_strided": (C.c_int, [H, C.c_int64, C.c_int64, C.c_int64, dp]),
        "amc_get_estimator_step": (C.c_int, [H, C.POINTER(C.c_uint64)]),
        "amc_set_estimator_step": (C.c_int, [H, C.c_uint64]),
        "amc_sweep": (C.c_int, [H, C.c_int64]),
        "amc_sweep_launches": (C.c_int, [H, C.c_int64]),
        "amc_get_step": (C.c_int, [H, C.POINTER(C.c_uint64)]),
        "amc_set_step": (C.c_int, [H, C.c_uint64]),
        "amc_reduce": (C.c_int, [H, dp]),
        "amc_reduce_begin": (C.c_int, [H]),
        "amc_sweep_reduce_begin": (C.c_int, [H, C.c_int64]),
        "amc_reduce_end": (C.c_int, [H, dp]),
        "amc_reduce_end_exact": (C.c_int, [H, dp, C.POINTER(C.c_uint64)]),
        "amc_xsum_merge": (C.c_int, [dp, dp, C.c_int]),
        "amc_xsum_round": (C.c_int, [dp, C.c_int, dp]),
        "amc_pg_estimate_exact": (C.c_int, [H, C.c_int, C.POINTER(C.c_int), C.c_int, dp]),
        "amc_pg_route": (C.c_int, [H, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int]),
        "amc_model_check": (C.c_int, [C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p),
                                      C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_char_p, C.c_int]),
        "amc_allreduce_xsum": (C.c_int, [H, dp, C.c_int]),
        "amc_comm_library_forced": (C.c_int, [C.POINTER(C.c_int)]),
        "amc_set_parameters": (C.c_int, [H, C.c_int, dp, C.c_int]),
        "amc_get_parameters": (C.c_int, [H, C.c_int, dp, C.c_int]),
        "amc_parameters_begin": (C.c_int, [H]),
        "amc_parameters_end": (C.c_int, [H, dp]),
        "amc_pg_estimate": (C.c_int, [H, C.c_int, C.POINTER(C.c_int), C.c_int, dp]),
        "amc_pg_accumulate": (C.c_int, [H, C.c_int, C.POINTER(C.c_int), C.c_int]),
        "amc_pg_update": (C.c_int, [H, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), dp, dp]),
        "amc_pg_get_accumulated": (C.c_int, [H, C.c_int, C.POINTER(C.c_int), dp]),
        "amc_pg_set_accumulated": (C.c_int, [H, C.c_int, C.POINTER(C.c_int), dp]),
        "amc_pgmc_steps": (C.c_int, [H, C.c_int64, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_int),
                                     dp, dp]),
        "amc_pgmc_steps_reduce_begin": (C.c_int, [H, C.c_int64, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_int),
                                                  dp, dp]),
        "amc_sync": (C.c_int, [H]),
        "amc_get_stream": (C.c_int, [H, C.POINTER(C.c_void_p)]),
        "amc_timing_begin": (C.c_int, [H]),
        "amc_timing_end": (C.c_int, [H, dp]),
        "amc_timing_mark": (C.c_int, [H]),
        "amc_comm_unique_id": (C.c_int, [C.c_void_p]),
        "amc_comm_init": (C.c_int, [H, C.c_int, C.c_int, C.c_void_p]),
        "amc_allreduce_sum": (C.c_int, [H, dp, C.c_int]),
        "amc_comm_destroy": (C.c_int, [H]),
        "amc_comm_info": (C.c_int, [H, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_int]),
        "amc_runtime_info": (C.c_int, [C.POINTER(C.c_int), C.c_char_p, C.c_int]),
        "amc_selftest_math": (C.c_int, [C.c_int, C.c_int, dp, dp, dp, C.c_int64]),
        "amc_selftest_accept_filter": (C.c_int, [C.c_int, C.c_float, C.c_float, dp]),
        "amc_selftest_philox": (C.c_int, [C.c_int, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                          C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.c_int64]),
        "amc_set_reduce_columns": (C.c_int, [H, C.c_int]),
        "amc_parameters_end_all": (C.c_int, [H, dp, C.c_int]),
        "amc_selftest_wave_totals": (C.c_int, [C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    SIGNATURES.update(sig)
    _lib = lib
    return lib


def runtime_info() -> dict:
    """The HIP runtime libamc.so is bound to in this process: hipRuntimeGetVersion and the file it came from."""
    v = C.c_int(0)
    path = C.create_string_buffer(1024)
    _check(load().amc_runtime_info(C.byref(v), path, 1024))
    return {"hip_runtime_version": v.value, "hip_runtime": path.value.decode()}


def _check(rc: int) -> None:
    if rc != 0:
        msg = load().amc_last_error()
        raise AmcError(f"amc error {rc}: {msg.decode() if msg else '?'}")


def model_check(sample, logq, dlogq=None, perform=None, invert=None, *, n_params: int = 1, potential: Optional[str] = None,
                reward: Optional[str] = None) -> str:
    """Compile-only check of a script-defined model (no GPU needed; amc_model_check): one policy -- strings, `dlogq` a string, a
    list of n_params strings or None (forward-mode differentiation of logq) -- or a pool of classes -- lists with one entry per
    class, None entries where a class has no expression of that kind.  Returns the compiler log."""
    many = isinstance(sample, (list, tuple))
    n_classes = len(sample) if many else 1

    def col(v, n):
        if v is None:
            return None
        items = list(v) if isinstance(v, (list, tuple)) else [v]
        assert len(items) == n, (items, n)
        return (C.c_char_p * n)(*[None if e is None else str(e).encode() for e in items])
    n_d = n_classes if many else int(n_params)
    buf = C.create_string_buffer(8192)
    lib = load()
    _check(lib.amc_model_check(int(n_params), n_classes, None if potential is None else str(potential).encode(),
                               None if reward is None else str(reward).encode(), col(sample, n_classes), col(logq, n_classes),
                               col(dlogq, n_d), col(perform, n_classes), col(invert, n_classes), buf, len(buf)))
    return buf.value.decode(errors="replace")


def device_count() -> int:
    n = C.c_int(0)
    rc = load().amc_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def potential_check(expr: str) -> str:
    """Compile-only check of a custom potential expression (no GPU needed); returns the compiler log, raises
    AmcError with the first diagnostics when the expression does not compile."""
    buf = C.create_string_buffer(8192)
    _check(load().amc_potential_check(str(expr).encode(), buf, len(buf)))
    return buf.value.decode(errors="replace")


def _dptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


# ---- reproducible sums (include/amc.h "Reproducible sums"): records are rows of AMC_XSUM_WORDS doubles -----------------
def xsum_merge(into: np.ndarray, other: np.ndarray) -> np.ndarray:
    """into[i] += other[i] for records of shape (n, AMC_XSUM_WORDS); exact (integers), so the order of merges is immaterial."""
    a = np.ascontiguousarray(into, dtype=np.float64).reshape(-1, AMC_XSUM_WORDS).copy()
    b = np.ascontiguousarray(other, dtype=np.float64).reshape(-1, AMC_XSUM_WORDS)
    if a.shape != b.shape:
        raise AmcError(f"xsum_merge: record arrays of shapes {a.shape} and {b.shape}")
    _check(load().amc_xsum_merge(_dptr(a), _dptr(b), a.shape[0]))
    return a


def xsum_round(records: np.ndarray) -> np.ndarray:
    """The Float64 of each record: the integer total rounded once."""
    a = np.ascontiguousarray(records, dtype=np.float64).reshape(-1, AMC_XSUM_WORDS)
    out = np.empty(a.shape[0], dtype=np.float64)
    _check(load().amc_xsum_round(_dptr(a), a.shape[0], _dptr(out)))
    return out


def xsum_plain(values) -> np.ndarray:
    """Records that hold plain numbers (counts: integers, exact under +)."""
    v = np.atleast_1d(np.asarray(values, dtype=np.float64))
    rec = np.zeros((v.size, AMC_XSUM_WORDS))
    rec[:, 0] = 3.0
    rec[:, 11] = v
    return rec


def comm_library_forced() -> bool:
    """True when AMC_RCCL_LIBRARY replaced librccl in this process (a site's own build, or the tests' stand-in)."""
    f = C.c_int(0)
    _check(load().amc_comm_library_forced(C.byref(f)))
    return bool(f.value)


class HipEngine:
    """One amc_handle: the device-resident shard of the chain ensemble.

    Mirrors what the reference's ``Metropolis`` object owns (pools, seed, rngs,
    sweepstep: src/metropolis.jl:232-239), SoA on one MI355X.
    """

    def __init__(self, *, n_chains: int, chain_offset: int = 0, n_chains_global: Optional[int] = None,
                 potential="harmonic", beta: float = 1.0, sigma: Sequence[float] = (1.0,),
                 weight: Sequence[float] = (1.0,), seed: int = 1, sweepstep: int = 1,
                 per_chain_counters: bool = True, device: int = 0, stream: Optional[int] = None,
                 reward_expr: Optional[str] = None, dtype: str = "f64", scale_expr: Optional[str] = None,
                 proposal: Optional[Sequence[Optional[str]]] = None, n_params: int = 1,
                 classes: Optional[Sequence[Sequence[Optional[str]]]] = None, class_of_move: Optional[Sequence[int]] = None):
        """``n_params`` > 1 (with ``proposal``): a policy with several parameters (amc_create_vector_policy_model) -- the
        expressions see theta0 .. theta{P-1}, ``proposal[2]`` is the list of the P partials of logq (or None), and ``sigma``
        holds one parameter VECTOR per move.
        ``classes`` (with ``class_of_move``): a pool that MIXES policy / action types (amc_create_mixed_model) -- one
        (sample, logq, dlogq or None[, perform, invert]) per class, ``class_of_move[k]`` the class move k uses."""
        lib = load()
        if str(dtype) not in STATE_DTYPES:
            raise AmcError(f"unknown state dtype {dtype!r}; one of {sorted(STATE_DTYPES)}")
        self.dtype = "f32" if STATE_DTYPES[str(dtype)] else "f64"
        expr = getattr(potential, "expr", None)        # system.CustomPotential: a C expression in x
        if expr is None and potential not in POTENTIALS:
            raise AmcError(f"unknown potential {potential!r}; the HIP engine offers {sorted(POTENTIALS)} "
                           "and CustomPotential(expr)")
        self.n_chains = int(n_chains)
        self.n_moves = len(sigma)
        self.n_params = int(n_params)
        self.gd_stride = 2 + 2 * self.n_params + self.n_params ** 2          # AMC_GD_STRIDE_P
        if self.n_params != 1 and proposal is None:
            raise AmcError("n_params > 1 needs a script-defined proposal (sample, logq, [dlogq ...])")
        theta = None
        if self.n_params > 1:
            theta = [np.ascontiguousarray(v, dtype=np.float64).reshape(-1) for v in sigma]
            if any(v.size != self.n_params for v in theta):
                raise AmcError(f"sigma must hold one vector of {self.n_params} parameters per move")
            sigma = [1.0] * self.n_moves          # parameter 0 at creation; the vectors follow through amc_set_parameters
        self.per_chain_counters = bool(per_chain_counters) or self.n_moves > 1
        self._sigma = (C.c_double * self.n_moves)(*[float(s) for s in sigma])
        self._weight = (C.c_double * self.n_moves)(*[float(w) for w in weight])
        cfg = AmcConfig()
        cfg.struct_size = C.sizeof(AmcConfig)
        cfg.device = int(device)
        cfg.n_chains = self.n_chains
        cfg.chain_offset = int(chain_offset)
        cfg.n_chains_global = int(n_chains_global if n_chains_global is not None else chain_offset + n_chains)
        cfg.potential = AMC_POTENTIAL_CUSTOM if expr is not None else POTENTIALS[potential]
        cfg.n_moves = self.n_moves
        cfg.beta = float(beta)
        cfg.sigma = self._sigma
        cfg.weight = self._weight
        cfg.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        cfg.sweepstep = int(sweepstep)
        cfg.per_chain_counters = 1 if per_chain_counters else 0
        cfg.stream = stream
        cfg.state_dtype = STATE_DTYPES[self.dtype]
        self._lib = lib
        self._h = C.c_void_p()
        enc = lambda t: None if t is None else str(t).encode()
        if classes is not None:
            if proposal is not None or scale_expr is not None or self.n_params != 1:
                raise AmcError("classes cannot be combined with proposal / scale_expr / n_params > 1")
            if class_of_move is None or len(class_of_move) != self.n_moves:
                raise AmcError("class_of_move must name one class per move")
            cl = [tuple((list(c) + [None] * 4)[:5]) for c in classes]
            n = len(cl)
            arr = lambda i, need: ((C.c_char_p * n)(*[enc(c[i]) for c in cl]) if need else None)
            have_d = any(c[2] is not None for c in cl)       # a class without one has its logq differentiated by the engine (NULL entry)
            com = (C.c_int * self.n_moves)(*[int(v) for v in class_of_move])
            if expr is None:
                cfg.potential = POTENTIALS[potential]
            _check(lib.amc_create_mixed_model(C.byref(cfg), n, com, enc(expr), enc(reward_expr), arr(0, True), arr(1, True), arr(2, have_d),
                                              arr(3, any(c[3] is not None for c in cl)), arr(4, any(c[4] is not None for c in cl)),
                                              C.byref(self._h)))
        elif proposal is not None:
            # script-defined sample_action! / log_proposal_density (/ its sigma-derivative) and, optionally, the action's
            # perform_action! / invert_action!: (sample, logq, dlogq or None[, perform, invert])
            if scale_expr is not None:
                raise AmcError("a script-defined proposal and a ScaledGaussian scale cannot be combined")
            sample, logq, dlogq, perform, invert = (list(proposal) + [None] * 4)[:5]
            if expr is None:
                cfg.potential = POTENTIALS[potential]
            if self.n_params > 1:
                partials = None
                if dlogq is not None:
                    if isinstance(dlogq, (str, bytes)) or len(dlogq) != self.n_params:
                        raise AmcError(f"proposal[2] must list the {self.n_params} partial derivatives of logq (or be None)")
                    partials = (C.c_char_p * self.n_params)(*[enc(d) for d in dlogq])
                _check(lib.amc_create_vector_policy_model(C.byref(cfg), self.n_params, enc(expr), enc(reward_expr), enc(sample),
                                                          enc(logq), partials, enc(perform), enc(invert), C.byref(self._h)))
                for k, v in enumerate(theta):
                    self.set_parameters(k, v)
            else:
                _check(lib.amc_create_action_model(C.byref(cfg), enc(expr), enc(reward_expr), enc(sample), enc(logq), enc(dlogq),
                                                   enc(perform), enc(invert), C.byref(self._h)))
        elif scale_expr is not None:
            # script-defined policy of the Gaussian-displacement family: proposal width sigma * scale(x)
            if expr is None:
                cfg.potential = POTENTIALS[potential]
            _check(lib.amc_create_policy_model(C.byref(cfg), None if expr is None else str(expr).encode(),
                                               None if reward_expr is None else str(reward_expr).encode(),
                                               str(scale_expr).encode(), C.byref(self._h)))
        elif reward_expr is not None:
            # script-defined reward(action, system) (gradients.jl:20): an expression in delta and the new position x
            if expr is None:
                cfg.potential = POTENTIALS[potential]
            _check(lib.amc_create_model(C.byref(cfg), None if expr is None else str(expr).encode(),
                                        str(reward_expr).encode(), C.byref(self._h)))
        elif expr is not None:
            _check(lib.amc_create_custom(C.byref(cfg), str(expr).encode(), C.byref(self._h)))
        else:
            _check(lib.amc_create(C.byref(cfg), C.byref(self._h)))

    # -- lifetime --------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.amc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- state -----------------------------------------------------------------
    def upload_state(self, x: np.ndarray, beta: Optional[np.ndarray] = None) -> None:
        x = np.ascontiguousarray(x, dtype=np.float64)
        if x.shape != (self.n_chains,):
            raise AmcError(f"upload_state: x has shape {x.shape}, expected ({self.n_chains},)")
        if beta is not None:
            beta = np.ascontiguousarray(beta, dtype=np.float64)
            if beta.shape != (self.n_chains,):
                raise AmcError("upload_state: beta must have one entry per chain")
        _check(self._lib.amc_upload_state(self._h, _dptr(x), _dptr(beta)))

    def init_uniform(self, lo: float, hi: float) -> None:
        _check(self._lib.amc_init_uniform(self._h, float(lo), float(hi)))

    def download_state(self, want_e: bool = True):
        x = np.empty(self.n_chains, dtype=np.float64)
        e = np.empty(self.n_chains, dtype=np.float64) if want_e else None
        _check(self._lib.amc_download_state(self._h, _dptr(x), _dptr(e)))
        return x, e

    def download_counters(self):
        acc = np.empty((self.n_moves, self.n_chains), dtype=np.int64)
        tot = np.empty((self.n_moves, self.n_chains), dtype=np.int64)
        p = C.POINTER(C.c_int64)
        _check(self._lib.amc_download_counters(self._h, acc.ctypes.data_as(p), tot.ctypes.data_as(p)))
        return acc, tot

    def counter_totals(self):
        acc = np.zeros(self.n_moves, dtype=np.int64)
        tot = np.zeros(self.n_moves, dtype=np.int64)
        p = C.POINTER(C.c_int64)
        _check(self._lib.amc_counter_totals(self._h, acc.ctypes.data_as(p), tot.ctypes.data_as(p)))
        return acc, tot

    def upload_counters(self, accepted: np.ndarray, total: Optional[np.ndarray] = None) -> None:
        p = C.POINTER(C.c_int64)
        a = np.ascontiguousarray(accepted, dtype=np.int64).reshape(self.n_moves, self.n_chains)
        t = None if total is None else np.ascontiguousarray(total, dtype=np.int64).reshape(self.n_moves, self.n_chains)
        _check(self._lib.amc_upload_counters(self._h, a.ctypes.data_as(p), None if t is None else t.ctypes.data_as(p)))

    def set_counter_totals(self, accepted: int, steps_counted: int) -> None:
        a = np.array([int(accepted)], dtype=np.int64)
        _check(self._lib.amc_set_counter_totals(self._h, a.ctypes.data_as(C.POINTER(C.c_int64)), int(steps_counted)))

    def histogram(self, lo: float, hi: float, n_bins: int) -> np.ndarray:
        """counts[n_bins + 3]: the bins of [lo, hi), then below lo, at/above hi, NaN (this shard only)."""
        out = np.zeros(int(n_bins) + 3, dtype=np.uint64)
        _check(self._lib.amc_histogram(self._h, float(lo), float(hi), int(n_bins), out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return out

    def histogram_accumulate(self, lo: float, hi: float, n_bins: int) -> None:
        """Add the histogram of the positions as of this point of the stream to the running one on the device (asynchronous)."""
        _check(self._lib.amc_histogram_accumulate(self._h, float(lo), float(hi), int(n_bins)))

    def histogram_fetch(self, n_bins: int, reset: bool = True) -> np.ndarray:
        out = np.zeros(int(n_bins) + 3, dtype=np.uint64)
        _check(self._lib.amc_histogram_fetch(self._h, out.ctypes.data_as(C.POINTER(C.c_uint64)), int(n_bins), 1 if reset else 0))
        return out

    def download_strided(self, first: int, stride: int, count: int) -> np.ndarray:
        out = np.empty(int(count), dtype=np.float64)
        _check(self._lib.amc_download_strided(self._h, int(first), int(stride), int(count), _dptr(out)))
        return out

    @property
    def estimator_step(self) -> int:
        t = C.c_uint64(0)
        _check(self._lib.amc_get_estimator_step(self._h, C.byref(t)))
        return t.value

    @estimator_step.setter
    def estimator_step(self, t: int) -> None:
        _check(self._lib.amc_set_estimator_step(self._h, int(t)))

    # -- hot path ----------------------------------------------------------------
    def sweep(self, n_sweeps: int = 1) -> None:
        _check(self._lib.amc_sweep(self._h, int(n_sweeps)))

    @property
    def step(self) -> int:
        t = C.c_uint64(0)
        _check(self._lib.amc_get_step(self._h, C.byref(t)))
        return t.value

    @step.setter
    def step(self, t: int) -> None:
        _check(self._lib.amc_set_step(self._h, int(t)))

    def reduce(self) -> np.ndarray:
        out = np.empty(AMC_RED_HEADER + self.n_moves, dtype=np.float64)
        _check(self._lib.amc_reduce(self._h, _dptr(out)))
        return out

    REDUCE_E, REDUCE_X, REDUCE_XX, REDUCE_ALL = 1, 2, 4, 7

    def set_reduce_columns(self, columns: int) -> None:
        """Which of sum e / sum x / sum x^2 the reductions begun from now on form (amc_set_reduce_columns; REDUCE_* bits).  A
        sum that is not formed reads NaN."""
        _check(self._lib.amc_set_reduce_columns(self._h, int(columns)))

    def reduce_begin(self) -> None:
        """Enqueue the reduction; sweeps queued afterwards keep running while the host does other work."""
        _check(self._lib.amc_reduce_begin(self._h))

    def sweep_reduce_begin(self, n_sweeps: int = 1) -> None:
        """n sweeps, then the callback reduction of the resulting state (fused into the last launch when possible)."""
        _check(self._lib.amc_sweep_reduce_begin(self._h, int(n_sweeps)))

    def reduce_end(self) -> np.ndarray:
        out = np.empty(AMC_RED_HEADER + self.n_moves, dtype=np.float64)
        _check(self._lib.amc_reduce_end(self._h, _dptr(out)))
        return out

    def reduce_end_exact(self):
        """The oldest reduction in flight as records, shape (AMC_RED_HEADER + K, AMC_XSUM_WORDS), and the MH steps counted per
        chain when it was begun.  What shards exchange (sharding.allreduce_xsum); ``reduce_records_value`` turns merged
        records into the numbers reduce_end() returns."""
        rec = np.zeros((AMC_RED_HEADER + self.n_moves, AMC_XSUM_WORDS), dtype=np.float64)
        steps = C.c_uint64(0)
        _check(self._lib.amc_reduce_end_exact(self._h, _dptr(rec), C.byref(steps)))
        return rec, int(steps.value)

    def reduce_exact(self):
        self.reduce_begin()
        return self.reduce_end_exact()

    def reduce_records_value(self, records: np.ndarray, steps_counted: int) -> np.ndarray:
        """Merged records -> the sums (layout of amc_reduce).  A K = 1 engine without per-chain counters carries the pool-wide
        accepted TOTAL in the ratio record: every chain has the same total_calls, the ratio sum is that total / steps."""
        out = xsum_round(records)
        out[np.asarray(records).reshape(-1, AMC_XSUM_WORDS)[:, 0] == 0.0] = np.nan      # an empty record: a sum nobody asked for
        if self.n_moves == 1 and not self.per_chain_counters:
            with np.errstate(invalid="ignore", divide="ignore"):
                out[AMC_RED_HEADER] = out[AMC_RED_HEADER] / np.float64(steps_counted)   # 0/0 = NaN before the first step
        return out

    def set_parameters(self, k: int, p: Sequence[float]) -> None:
        a = np.ascontiguousarray(p, dtype=np.float64).reshape(-1)
        _check(self._lib.amc_set_parameters(self._h, int(k), _dptr(a), int(a.size)))

    def get_parameters(self, k: int) -> np.ndarray:
        a = np.empty(self.n_params, dtype=np.float64)
        _check(self._lib.amc_get_parameters(self._h, int(k), _dptr(a), self.n_params))
        return a

    def parameters_begin(self) -> None:
        """Queue a read of every move's sigma as of this point of the stream (amc_parameters_begin); fetch with parameters_end()."""
        _check(self._lib.amc_parameters_begin(self._h))

    def parameters_end(self) -> np.ndarray:
        """sigma[K] -- for a policy with several parameters all of them, shape (K, P)."""
        if self.n_params > 1:
            a = np.empty((self.n_moves, self.n_params), dtype=np.float64)
            _check(self._lib.amc_parameters_end_all(self._h, _dptr(a), a.size))
            return a
        a = np.empty(self.n_moves, dtype=np.float64)
        _check(self._lib.amc_parameters_end(self._h, _dptr(a)))
        return a

    def pg_estimate(self, learn_ids: Sequence[int], q_batch: int) -> np.ndarray:
        n = len(learn_ids)
        ids = (C.c_int * max(n, 1))(*[int(i) for i in learn_ids])
        out = np.zeros((n, self.gd_stride), dtype=np.float64)
        _check(self._lib.amc_pg_estimate(self._h, n, ids, int(q_batch), _dptr(out)))
        return out

    def pg_estimate_exact(self, learn_ids: Sequence[int], q_batch: int) -> np.ndarray:
        """The same fold as records, shape (n_learn, AMC_GD_STRIDE, AMC_XSUM_WORDS): what shards exchange."""
        n = len(learn_ids)
        ids = (C.c_int * max(n, 1))(*[int(i) for i in learn_ids])
        out = np.zeros((n, self.gd_stride, AMC_XSUM_WORDS), dtype=np.float64)
        _check(self._lib.amc_pg_estimate_exact(self._h, n, ids, int(q_batch), _dptr(out)))
        return out

    def sweep_launches(self, n_launches: int) -> None:
        """n make_step!s as n launches of one sweep each, queued by one call (amc_sweep_launches)."""
        _check(self._lib.amc_sweep_launches(self._h, int(n_launches)))

    def pg_route(self, n_learn: int, q_batch: int = 1, fused: bool = False):
        """(one_launch, why): whether an estimator call over n_learn learnable moves takes them all in ONE launch -- with
        ``fused``: whether the whole time step (sweep + estimator + update) is one launch -- and, for a pool of several classes
        whose several-move kernel form the run-time compiler fails on, what the compiler said (the calls then take one launch per
        move: same bits).  ``pg_route_code`` returns amc_pg_route's own answer (2 / 1 / 0)."""
        code, why = self.pg_route_code(n_learn, q_batch, fused)
        return (code == 2 if fused else code >= 1), why

    def pg_route_code(self, n_learn: int, q_batch: int = 1, fused: bool = False):
        why = C.create_string_buffer(2048)
        rc = self._lib.amc_pg_route(self._h, int(n_learn), int(q_batch), int(bool(fused)), why, len(why))
        if rc < 0:
            _check(rc)
        return int(rc), why.value.decode(errors="replace")

    def pg_accumulate(self, learn_ids: Sequence[int], q_batch: int) -> None:
        """Estimator step kept on the device: gradients_data[k] += gd (asynchronous)."""
        n = len(learn_ids)
        ids = (C.c_int * max(n, 1))(*[int(i) for i in learn_ids])
        _check(self._lib.amc_pg_accumulate(self._h, n, ids, int(q_batch)))

    def pg_update(self, learn_ids: Sequence[int], kinds: Sequence[int], hyper0: Sequence[float],
                  hyper1: Sequence[float]) -> None:
        """average -> learning_step! -> reset on the device; sigma and its table are refreshed in place."""
        n = len(learn_ids)
        ids = (C.c_int * max(n, 1))(*[int(i) for i in learn_ids])
        kd = (C.c_int * max(n, 1))(*[int(k) for k in kinds])
        h0 = (C.c_double * max(n, 1))(*[float(v) for v in hyper0])
        h1 = (C.c_double * max(n, 1))(*[float(v) for v in hyper1])
        _check(self._lib.amc_pg_update(self._h, n, ids, kd, h0, h1))

    def pgmc_steps(self, n_steps: int, learn_ids: Sequence[int], q_batch: int, kinds: Optional[Sequence[int]] = None,
                   hyper0: Sequence[float] = (), hyper1: Sequence[float] = (), reduce_begin: bool = False) -> None:
        """n x [sweep(1); pg_accumulate; pg_update if kinds is given] enqueued by one call (asynchronous).
        ``reduce_begin``: followed by reduce_begin() of the state the last step leaves -- the callback sums then ride in that
        step's launch (amc_pgmc_steps_reduce_begin); fetch them with reduce_end()."""
        n = len(learn_ids)
        ids = (C.c_int * max(n, 1))(*[int(i) for i in learn_ids])
        do_update = kinds is not None
        kd = (C.c_int * max(n, 1))(*[int(k) for k in (kinds or [0] * n)])
        h0 = (C.c_double * max(n, 1))(*[float(v) for v in (hyper0 if do_update else [0.0] * n)])
        h1 = (C.c_double * max(n, 1))(*[float(v) for v in (hyper1 if do_update else [0.0] * n)])
        fn = self._lib.amc_pgmc_steps_reduce_begin if reduce_begin else self._lib.amc_pgmc_steps
        _check(fn(self._h, int(n_steps), n, ids, int(q_batch), 1 if do_update else 0, kd, h0, h1))

    def pg_get_accumulated(self, learn_ids: Sequence[int]) -> np.ndarray:
        n = len(learn_ids)
        ids = (C.c_int * max(n, 1))(*[int(i) for i in learn_ids])
        out = np.zeros((n, self.gd_stride), dtype=np.float64)
        _check(self._lib.amc_pg_get_accumulated(self._h, n, ids, _dptr(out)))
        return out

    def pg_set_accumulated(self, learn_ids: Sequence[int], rows: np.ndarray) -> None:
        """Resume: replace the device-resident gradients_data of the moves learn_ids by rows[n][5] (j, grad_j, grad_logq, g, n)."""
        n = len(learn_ids)
        ids = (C.c_int * max(n, 1))(*[int(i) for i in learn_ids])
        a = np.ascontiguousarray(rows, dtype=np.float64).reshape(n, self.gd_stride)
        _check(self._lib.amc_pg_set_accumulated(self._h, n, ids, _dptr(a)))

    def sync(self) -> None:
        _check(self._lib.amc_sync(self._h))

    @property
    def stream(self) -> int:
        s = C.c_void_p()
        _check(self._lib.amc_get_stream(self._h, C.byref(s)))
        return s.value or 0

    def timing_begin(self) -> None:
        _check(self._lib.amc_timing_begin(self._h))

    def timing_end(self) -> float:
        ms = C.c_double(0.0)
        _check(self._lib.amc_timing_end(self._h, C.byref(ms)))
        return ms.value

    def timing_mark(self) -> None:
        """Record the end event now (asynchronous); timing_end then waits for it and returns the elapsed device time."""
        _check(self._lib.amc_timing_mark(self._h))

    # -- RCCL through the C ABI (what the Julia binding uses) ----------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        _check(load().amc_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, rank: int, n_ranks: int, unique_id: bytes) -> None:
        buf = C.create_string_buffer(unique_id, 128)
        _check(self._lib.amc_comm_init(self._h, int(rank), int(n_ranks), buf))

    def allreduce_sum(self, a: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(a, dtype=np.float64).copy()
        _check(self._lib.amc_allreduce_sum(self._h, _dptr(a), int(a.size)))
        return a

    def allreduce_xsum(self, records: np.ndarray) -> np.ndarray:
        """The merged records of all shards (amc_allreduce_xsum: one RCCL all-reduce used as a gather, then the integer merge)."""
        a = np.ascontiguousarray(records, dtype=np.float64).copy()
        _check(self._lib.amc_allreduce_xsum(self._h, _dptr(a), int(a.size // AMC_XSUM_WORDS)))
        return a

    def comm_destroy(self) -> None:
        """Drop the communicator: a single shard again."""
        _check(self._lib.amc_comm_destroy(self._h))

    def comm_info(self) -> dict:
        """What RCCL reports about this engine's communicator: ranks it spans, this rank, RCCL version, the librccl file."""
        n, r, v = C.c_int(0), C.c_int(0), C.c_int(0)
        path = C.create_string_buffer(1024)
        _check(self._lib.amc_comm_info(self._h, C.byref(n), C.byref(r), C.byref(v), path, 1024))
        return {"n_ranks": n.value, "rank": r.value, "rccl_version": v.value, "librccl": path.value.decode()}


class SplitEngine:
    """The shard of one GPU as `n_parts` sub-shards, each a HipEngine with its own stream.

    A single-sweep launch spends ~3 us in its launch boundary and ~2 us waiting for far memory at its ends; sub-shards
    on separate streams overlap one's ends with the other's body (31.3 -> 28.7 us per sweep of 1e7 chains measured with
    two).  Chains keep their GLOBAL ids, so every per-chain result is identical to the unsplit engine's; sums
    (reductions, gradient data) are merged over the parts on the host as exact integer records (reproducible sums), i.e. they
    are bit-identical to the unsplit engine's too.
    The device-resident estimator / update path (pg_accumulate, pg_update, pgmc_steps) is not offered: those keep
    per-engine state; PolicyGradientEstimator falls back to the host path (pg_estimate) on a split engine."""

    def __init__(self, *, n_chains: int, chain_offset: int = 0, n_chains_global: Optional[int] = None, n_parts: int = 2,
                 **kw):
        if chain_offset % 2:
            raise AmcError("chain_offset must be even")
        self.n_chains = int(n_chains)
        n_glob = int(n_chains_global if n_chains_global is not None else chain_offset + n_chains)
        n_pairs = (self.n_chains + 1) // 2
        n_parts = max(1, min(int(n_parts), n_pairs))
        bounds = [min(2 * ((i * n_pairs) // n_parts), self.n_chains) for i in range(n_parts)] + [self.n_chains]
        self.bounds = bounds
        self.parts = [HipEngine(n_chains=bounds[i + 1] - bounds[i], chain_offset=chain_offset + bounds[i],
                                n_chains_global=n_glob, **kw) for i in range(n_parts)]
        self.n_moves = self.parts[0].n_moves

    def close(self) -> None:
        for p in self.parts:
            p.close()

    def _slices(self):
        return [slice(self.bounds[i], self.bounds[i + 1]) for i in range(len(self.parts))]

    def upload_state(self, x, beta=None) -> None:
        x = np.ascontiguousarray(x, dtype=np.float64)
        for p, sl in zip(self.parts, self._slices()):
            p.upload_state(x[sl], None if beta is None else np.ascontiguousarray(beta, dtype=np.float64)[sl])

    def init_uniform(self, lo, hi) -> None:
        for p in self.parts:
            p.init_uniform(lo, hi)

    def download_state(self, want_e: bool = True):
        xs, es = zip(*[p.download_state(want_e) for p in self.parts])
        return np.concatenate(xs), (np.concatenate(es) if want_e else None)

    def download_counters(self):
        a, t = zip(*[p.download_counters() for p in self.parts])
        return np.concatenate(a, axis=1), np.concatenate(t, axis=1)

    def counter_totals(self):
        a, t = zip(*[p.counter_totals() for p in self.parts])
        return np.sum(a, axis=0), np.sum(t, axis=0)

    def upload_counters(self, accepted, total=None) -> None:
        a = np.ascontiguousarray(accepted, dtype=np.int64).reshape(self.n_moves, self.n_chains)
        t = None if total is None else np.ascontiguousarray(total, dtype=np.int64).reshape(self.n_moves, self.n_chains)
        for p, sl in zip(self.parts, self._slices()):
            p.upload_counters(a[:, sl], None if t is None else t[:, sl])

    def set_counter_totals(self, accepted: int, steps_counted: int) -> None:
        for i, p in enumerate(self.parts):          # the pool-wide total lives in the first part, the step count in all
            p.set_counter_totals(int(accepted) if i == 0 else 0, steps_counted)

    def histogram(self, lo, hi, n_bins):
        return np.sum([p.histogram(lo, hi, n_bins) for p in self.parts], axis=0).astype(np.uint64)

    def download_strided(self, first: int, stride: int, count: int) -> np.ndarray:
        idx = int(first) + int(stride) * np.arange(int(count), dtype=np.int64)
        out = np.empty(int(count), dtype=np.float64)
        for p, sl in zip(self.parts, self._slices()):
            sel = np.nonzero((idx >= sl.start) & (idx < sl.stop))[0]
            if sel.size:
                out[sel] = p.download_strided(int(idx[sel[0]] - sl.start), int(stride), int(sel.size))
        return out

    def sweep(self, n_sweeps: int = 1) -> None:
        for p in self.parts:          # asynchronous: the parts' launches overlap
            p.sweep(n_sweeps)

    @property
    def step(self) -> int:
        return self.parts[0].step

    @step.setter
    def step(self, t: int) -> None:
        for p in self.parts:
            p.step = t

    @property
    def estimator_step(self) -> int:
        return self.parts[0].estimator_step

    @estimator_step.setter
    def estimator_step(self, t: int) -> None:
        for p in self.parts:
            p.estimator_step = t

    def reduce(self) -> np.ndarray:
        rec, steps = self.reduce_exact()
        return self.parts[0].reduce_records_value(rec, steps)

    def reduce_exact(self):
        self.reduce_begin()
        return self.reduce_end_exact()

    def reduce_end_exact(self):
        rec, steps = None, 0
        for p in self.parts:          # integer merges: the split does not enter the result
            r, steps = p.reduce_end_exact()
            rec = r if rec is None else xsum_merge(rec, r)
        return rec, steps

    def reduce_records_value(self, records, steps_counted):
        return self.parts[0].reduce_records_value(records, steps_counted)

    @property
    def per_chain_counters(self) -> bool:
        return self.parts[0].per_chain_counters

    def reduce_begin(self) -> None:
        for p in self.parts:
            p.reduce_begin()

    def sweep_reduce_begin(self, n_sweeps: int = 1) -> None:
        for p in self.parts:
            p.sweep_reduce_begin(n_sweeps)

    def reduce_end(self) -> np.ndarray:
        rec, steps = self.reduce_end_exact()
        return self.parts[0].reduce_records_value(rec, steps)

    def set_parameters(self, k, p_) -> None:
        for p in self.parts:
            p.set_parameters(k, p_)

    def get_parameters(self, k):
        return self.parts[0].get_parameters(k)

    def pg_estimate_exact(self, learn_ids, q_batch) -> np.ndarray:
        rec = None
        for p in self.parts:
            r = p.pg_estimate_exact(learn_ids, q_batch)
            rec = r if rec is None else xsum_merge(rec, r).reshape(r.shape)
        return rec

    def pg_estimate(self, learn_ids, q_batch) -> np.ndarray:
        rec = self.pg_estimate_exact(learn_ids, q_batch)
        return xsum_round(rec).reshape(rec.shape[0], rec.shape[1])

    def sync(self) -> None:
        for p in self.parts:
            p.sync()

    def timing_begin(self) -> None:
        self.sync()
        import time
        self._t0 = time.perf_counter()

    def timing_end(self) -> float:
        """Wall-clock ms between timing_begin and the completion of everything queued since (the parts run on
        different streams: there is no single pair of stream events that brackets them)."""
        self.sync()
        import time
        return (time.perf_counter() - self._t0) * 1e3


def selftest_math(fn: str, a: np.ndarray, b: Optional[np.ndarray] = None, device: int = 0) -> np.ndarray:
    """Evaluate one arithmetic-spec primitive on the GPU (parity tests only)."""
    ids = {"exp": 0, "log": 1, "sinpi": 2, "cospi": 3, "sqrt": 4, "div": 5, "div_by_const": 6, "logbm": 7,
           "sqrt_radius": 8, "log_proposal_density": 9, "grad_log_proposal_density": 10,
           "grad_log_proposal_density_kernel_form": 11}
    a = np.ascontiguousarray(a, dtype=np.float64)
    if b is not None:
        b = np.ascontiguousarray(b, dtype=np.float64)
    out = np.empty_like(a)
    _check(load().amc_selftest_math(int(device), ids[fn], _dptr(a), _dptr(b), _dptr(out), a.size))
    return out


def selftest_accept_filter(t_from: float, t_to: float, device: int = 0) -> float:
    """Largest relative deviation of the accept filter's float estimate from the spec's f64 exp over EVERY float in
    [t_to, t_from] (t_to <= t_from <= 0), measured on the GPU (parity tests only)."""
    out = C.c_double(0.0)
    _check(load().amc_selftest_accept_filter(int(device), C.c_float(t_from), C.c_float(t_to), C.byref(out)))
    return out.value


def selftest_philox(seed: int, pair: np.ndarray, t: np.ndarray, draw: int, stream: int,
                    device: int = 0) -> np.ndarray:
    """Raw Philox4x32-10 words of draw (pair, t, draw, stream) on the GPU (parity tests only)."""
    pair = np.ascontiguousarray(pair, dtype=np.uint64)
    t = np.ascontiguousarray(t, dtype=np.uint64)
    out = np.empty((pair.size, 4), dtype=np.uint32)
    u64p = C.POINTER(C.c_uint64)
    _check(load().amc_selftest_philox(int(device), C.c_uint64(int(seed)), pair.ctypes.data_as(u64p),
                                      t.ctypes.data_as(u64p), int(draw), int(stream),
                                      out.ctypes.data_as(C.POINTER(C.c_uint32)), pair.size))
    return out


def selftest_wave_totals(values: np.ndarray, device: int = 0):
    """The kernels' wave-wide integer totals of values[6][64] (int64) on the GPU: (totals[13], totals_plain[6]), see
    include/amc.h (parity tests only)."""
    values = np.ascontiguousarray(values, dtype=np.int64)
    assert values.shape == (6, 64)
    out = np.zeros(13, dtype=np.int64)
    ref = np.zeros(6, dtype=np.int64)
    i64p = C.POINTER(C.c_int64)
    _check(load().amc_selftest_wave_totals(int(device), values.ctypes.data_as(i64p), out.ctypes.data_as(i64p),
                                           ref.ctypes.data_as(i64p)))
    return out, ref
