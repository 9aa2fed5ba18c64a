"""Every corpus address the scan computes lies inside the scan copy — also the prefetches of k-steps that do not exist.

Why this test exists: on 2026-10-04 a round-1 run of tests/test_gpu_parity.py::test_fast_path_other_dims (d = 768/200/64, i.e.
12/4/1 k-steps per tile, 9000 rows = 36 tiles for 256 streams) aborted inside rdx_search on an uncommitted build of the scan
(DESIGN.md §10). The scan prefetches corpus fragments two k-steps ahead, also past a stream's last step; the build that aborted
computed those addresses from the NEXT schedule entry (tile + n_streams), far outside a 1 MB scan copy, and the GPU faulted. The
committed kernel re-reads the stream's first step instead. A later green run does not prove that: a stray read faults only when
it leaves the mapped range. This test builds the library with -DRDX_CHECK_BOUNDS (scan_kernel.hpp: every fragment address is
checked on the device and reported through the search's error code), loads it in a child process through RDX_LIB_PATH and runs
exactly the shapes that stress the prefetch: one k-step per tile, fewer tiles than streams, one tile in all, odd k-step counts,
the XCD-balanced schedule, a row bitmap, the second-chance pass."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent("""
    import numpy as np, sys
    sys.path.insert(0, '@ROOT@')
    from rag_dpo_amd import engine
    rng = np.random.default_rng(11)
    shapes = [(64, 9000, 70), (200, 9000, 70), (768, 9000, 70), (64, 9000, 40), (128, 3000, 300), (1024, 5000, 33), (64, 200, 3),
              (320, 70001, 257), (64, 300000, 600), (1024, 257, 1), (192, 66000, 130)]
    for d, n, b in shapes:
        corpus = rng.standard_normal((n, d)).astype(np.float32)
        q = rng.standard_normal((b, d)).astype(np.float32)
        ix = engine.HipIndex(d); ix.add(corpus); ix.set_option("force_fast", 1)
        for opts in ({}, {"sib_sync": 1}, {"cand_cap": 8}):
            for name, v in opts.items():
                ix.set_option(name, v)
            s, r, c = ix.search(q, 10)                      # raises if the device flagged an out-of-range address
            allow = np.packbits(np.pad(rng.random(n) < 0.5, (0, (-n) % 32)).reshape(-1, 32), axis=1, bitorder="little").view(np.uint32).reshape(-1)
            ix.search(q, 5, allow)
            for name in opts:
                ix.set_option(name, 0)
        ix.close()
    print("bounds ok", len(shapes))
""")


def test_scan_addresses_stay_inside_the_scan_copy(tmp_path):
    from rag_dpo_amd.build import build_lib
    lib = build_lib(extra_flags=("-DRDX_CHECK_BOUNDS",), out=str(tmp_path / "librdx_bounds.so"))
    env = dict(os.environ, RDX_LIB_PATH=lib)
    p = subprocess.run([sys.executable, "-c", CHILD.replace("@ROOT@", ROOT)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "bounds ok" in p.stdout, (p.stdout[-2000:], p.stderr[-4000:])
