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
M = int(os.environ.get("M", 10_000_000)); K = int(os.environ.get("K", 1))
sigma = [0.1, 1.0][:K]; weight = [[1.0], [0.5, 0.5]][K - 1]
e = A.HipEngine(n_chains=M, potential="harmonic" if K == 1 else "double_well", beta=2.0, sigma=sigma, weight=weight,
                seed=1, per_chain_counters=(K > 1))
e.init_uniform(-2, 2)
t0 = time.time()
while time.time() - t0 < 0.6:
    for _ in range(200): e.sweep(1)
    e.sync()
best = 1e9
for rep in range(4):
    e.timing_begin()
    for _ in range(1000): e.sweep(1)
    best = min(best, e.timing_end())
print(f"{best:.2f}")
e.close()
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
    for n in names:
        print(f"{n:24s} " + "  ".join(res[n]), flush=True)
