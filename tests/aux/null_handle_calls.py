"""Test helper (own process: a fault must not take the test runner down): every entry of the C ABI called with a NULL handle and
zeros / NULLs for everything else.  Prints `name status` per call; the last line printed before a crash names it."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from montecarlo_amd import _capi as A

lib = A.load()
names = [n for n in A.SIGNATURES] if hasattr(A, "SIGNATURES") else None
if names is None:
    raise SystemExit("montecarlo_amd._capi.SIGNATURES missing")
for name in names:
    res, args = A.SIGNATURES[name]
    if not args:
        continue
    vals = []
    for t in args:
        if t in (C.c_int, C.c_int64, C.c_uint64, C.c_uint32):
            vals.append(0)
        elif t in (C.c_double, C.c_float):
            vals.append(0.0)
        else:
            vals.append(None)
    print(name, end=" ", flush=True)
    rc = getattr(lib, name)(*vals)
    print(rc, flush=True)
print("done")
