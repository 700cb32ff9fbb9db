import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import oracle_lib as O
from montecarlo_amd import _capi as A
for M in (7, 300, 50001):
    for pcc in (False, True):
        kw = dict(potential="harmonic", beta=2.0, sigma=[0.1], weight=[1.0], seed=1)
        eng = A.HipEngine(n_chains=M, device=0, dtype="f32", per_chain_counters=pcc, **kw)
        sim = O.OracleSim(M, dtype="f32", **kw)
        eng.init_uniform(-2, 2); sim.init_uniform(-2, 2)
        eng.sweep(5); sim.make_steps(5)
        for mode in ("fused", "separate"):
            if mode == "fused":
                eng.sweep_reduce_begin(1); sim.make_steps(1)
                rec, st = eng.reduce_end_exact()
            else:
                rec, st = eng.reduce_exact()
            want = sim.callback_records()
            x, e = eng.download_state(); xo, eo = sim.state()
            print(M, pcc, mode, "state eq", np.array_equal(x, xo), np.array_equal(e, eo))
            for i in range(3):
                if not np.array_equal(rec[i], want[i]):
                    print("  col", i, rec[i], want[i])
            # independent: sum of downloaded e through oracle xsum
            r2 = O.xsum_r(e)
            print("  xsum_r(e) == oracle rec:", np.array_equal(r2, want[0]), " == device:", np.array_equal(r2, rec[0]))
