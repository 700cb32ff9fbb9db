#!/usr/bin/env python3
"""Developer tool: same-box A/B timing of library variants.

  tools/gpu_ab.py snapshot NAME      copy the built package to tools/_variants/NAME (run on the build host)
  tools/gpu_ab.py run [ROUNDS]       time every variant, interleaved, ROUNDS times (run on the GPU box)
Devices of one type differ by several percent on VALU-bound kernels, so variants are only comparable inside one call.
"""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VAR = os.path.join(ROOT, "tools", "_variants")

CHILD = r"""
import os, sys, time
sys.path.insert(0, sys.argv[1])
from montecarlo_amd import _capi as A
M = int(os.environ.get("M", 10_000_000))
def timed(e, f, n=400, reps=4):
    t0 = time.time()
    while time.time() - t0 < 0.5:
        f(100); e.sync()
    best = 1e9
    for rep in range(reps):
        e.timing_begin(); f(n); best = min(best, e.timing_end() / n * 1e3)
    return best
out = []
e = A.HipEngine(n_chains=M, potential="harmonic", beta=2.0, sigma=[0.1], weight=[1.0], seed=1, per_chain_counters=False)
e.init_uniform(-2, 2); out.append(timed(e, lambda n: [e.sweep(1) for _ in range(n)])); e.close()
e = A.HipEngine(n_chains=M, potential="double_well", beta=2.0, sigma=[0.1, 1.0], weight=[0.5, 0.5], seed=1)
e.init_uniform(-2, 2); out.append(timed(e, lambda n: [e.sweep(1) for _ in range(n)])); e.close()
e = A.HipEngine(n_chains=M, potential="harmonic", beta=2.0, sigma=[0.2, 0.1], weight=[0.6, 0.4], seed=1)
e.init_uniform(-2, 2); out.append(timed(e, lambda n: e.pgmc_steps(n, [1], 1, [1], [1e-3], [0.0]), 200))
out.append(timed(e, lambda n: [e.pg_accumulate([1], 1) for _ in range(n)], 200)); e.close()
e = A.HipEngine(n_chains=M, potential="harmonic", beta=2.0, sigma=[0.2], weight=[1.0], seed=1, per_chain_counters=False)
e.init_uniform(-2, 2); out.append(timed(e, lambda n: e.pgmc_steps(n, [0], 1, [1], [1e-3], [0.0]), 200)); e.close()
print(" ".join(f"{v:6.2f}" for v in out))
"""

if sys.argv[1] == "snapshot":
    dst = os.path.join(VAR, sys.argv[2])
    shutil.rmtree(dst, ignore_errors=True)
    shutil.copytree(os.path.join(ROOT, "montecarlo_amd"), os.path.join(dst, "montecarlo_amd"),
                    ignore=shutil.ignore_patterns("__pycache__", "csrc"))
    print("snapshot ->", dst)
else:
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    names = sorted(os.listdir(VAR))
    res = {n: [] for n in names}
    for _ in range(rounds):
        for n in names:
            out = subprocess.run([sys.executable, "-c", CHILD, os.path.join(VAR, n)], capture_output=True, text=True)
            res[n].append(out.stdout.strip() or out.stderr.strip()[-120:])
    print("us per launch at 1e7 chains: K=1 sweep, K=2 double-well sweep, fused PGMC step (config 5), estimator launch alone, fused PGMC step with K=1; one column group per round")
    for n in names:
        print(f"{n:24s} " + "  |  ".join(res[n]), flush=True)
