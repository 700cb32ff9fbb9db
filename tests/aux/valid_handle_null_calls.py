"""Test helper (own process): every entry of the C ABI that takes a handle, called on a LIVE handle with zeros / NULLs for every other
argument -- for three kinds of handle.  Prints `name status` per call (the last line before a fault names it), then shows that the
handle still works."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from montecarlo_amd import _capi as A

lib = A.load()
drift = ("theta0 + theta1*z", "-((delta-theta0)*(delta-theta0))/(2.0*theta1*theta1) - amc_log(theta1)", None)
kinds = {"builtin_k2": dict(sigma=[0.2, 0.4], weight=[0.5, 0.5]),
         "builtin_k1": dict(sigma=[0.3], weight=[1.0], per_chain_counters=False),
         "vector_policy": dict(sigma=[[0.0, 0.5], [0.1, 0.9]], weight=[0.5, 0.5], proposal=drift, n_params=2)}
for kind, kw in kinds.items():
    e = A.HipEngine(n_chains=1001, potential="harmonic", beta=2.0, seed=3, **kw)
    e.init_uniform(-2, 2)
    for name, (res, args) in A.SIGNATURES.items():
        if not args or args[0] is not C.c_void_p or name in ("amc_destroy", "amc_comm_unique_id"):
            continue
        vals = [e._h]
        for t in args[1:]:
            vals.append(0 if t in (C.c_int, C.c_int64, C.c_uint64, C.c_uint32) else 0.0 if t in (C.c_double, C.c_float) else None)
        print(kind, name, end=" ", flush=True)
        print(getattr(lib, name)(*vals), flush=True)
    # whatever the calls above started (a reduction, a parameter read, a timing bracket) is finished or abandoned: the handle works
    out = (C.c_double * 64)()
    while lib.amc_reduce_end(e._h, out) == 0:
        pass
    lib.amc_parameters_end(e._h, out)
    assert lib.amc_set_step(e._h, 0) == 0 and lib.amc_set_reduce_columns(e._h, 7) == 0
    e.init_uniform(-2, 2)
    e.sweep(3)
    x = e.download_state()[0]
    t = C.c_uint64(0)
    assert lib.amc_get_step(e._h, C.byref(t)) == 0 and t.value == 3, kind
    assert np.all(np.isfinite(x)) and np.ptp(x) > 1.0, kind
    red = e.reduce()
    assert np.isfinite(red["energy"]) if isinstance(red, dict) else True
    e.close()
print("done")
