"""Developer A/B of the small-batch (HBM-bound) legs: python tools/ab_small.py NAME|base [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rag_dpo_amd import _lib
if sys.argv[1] != "base":
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), f"librdx_{sys.argv[1]}.so")
from tools.quick_bench import build, run
ix = build(int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000)
for b in (64, 128, 1, 64, 128):
    o = run(ix, b, 10, iters=10)
    print("  ", sys.argv[1], "b", b, "scan_ms", o["scan_ms"], "GB/s", o["scan_GBs"], flush=True)
