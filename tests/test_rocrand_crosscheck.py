"""The engine's Philox4x32-10 against the REAL rocRAND device API on the GPU (third-party anchor of the draw
schedule): tests/aux/rocrand_words.hip calls rocrand_init / rocrand4 from
/opt/rocm/include/rocrand/rocrand_kernel.h; its words must equal the oracle's and the product kernels' for the same
(seed, chain pair, step, draw, stream)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_philox_words_equal_rocrand_device_api(gpu, oracle, tmp_path):
    exe = tmp_path / "rocrand_words"
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "--offload-arch=gfx950", "-w",
                        os.path.join(ROOT, "tests", "aux", "rocrand_words.hip"), "-o", str(exe)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    rng = np.random.default_rng(12)
    cases = []
    for seed in (0, 1, 42, 2 ** 63 + 9, 2 ** 64 - 1):
        for _ in range(40):
            pair = int(rng.integers(0, 2 ** 63)) if rng.random() < 0.5 else int(rng.integers(0, 10 ** 7))
            t = int(rng.integers(0, 2 ** 48)) if rng.random() < 0.5 else int(rng.integers(0, 10 ** 6))
            draw, stream = int(rng.integers(0, 4096)), int(rng.integers(0, 3))
            cases.append((seed, pair, t, draw, stream))
    cases += [(1, 0, 0, 0, 1), (1, 4_999_999, 0, 1, 1), (7, 2 ** 40 + 1, 2 ** 48 - 1, 4095, 2)]
    lines = []
    for seed, pair, t, draw, stream in cases:
        c = oracle.counter(pair, t, draw, stream)
        assert c[2] == pair & 0xFFFFFFFF and c[3] == pair >> 32
        lines.append(f"{seed} {pair} {c[0] | (c[1] << 32)}")
    out = subprocess.run([str(exe)], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    got = [[int(v) for v in ln.split()] for ln in out.stdout.strip().splitlines()]
    assert len(got) == len(cases)
    for (seed, pair, t, draw, stream), words in zip(cases, got):
        assert words == oracle.draw_words(seed, pair, t, draw, stream), (seed, pair, t, draw, stream)
    # and the product's own device Philox (amc_selftest_philox) agrees with both
    for seed, draw, stream in ((42, 0, 1), (2 ** 64 - 1, 1, 1)):
        sel = [(p, t) for s, p, t, d, st in cases if s == seed][:20]
        pairs = np.array([p for p, _ in sel], dtype=np.uint64)
        ts = np.array([t for _, t in sel], dtype=np.uint64)
        dev = gpu.selftest_philox(seed, pairs, ts, draw, stream)
        for i, (p, t) in enumerate(sel):
            assert list(dev[i]) == oracle.draw_words(seed, p, t, draw, stream)
