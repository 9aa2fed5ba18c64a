"""Developer experiment: which resource bounds the main scan? Builds a -DRDX_ABLATE copy of the library
(tools/librdx_dev.so) whose k_scan can skip the corpus DMA / query DMA / MFMAs / fragment reads, and times each."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DEV = os.path.join(ROOT, "tools", "librdx_dev.so")

def build():
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DRDX_ABLATE",
                           os.path.join(ROOT, "rag_dpo_amd", "csrc", "rdx_api.hip"), "-o", DEV])

if __name__ == "__main__":
    if sys.argv[1:] == ["build"]:
        build(); sys.exit(0)
    from rag_dpo_amd import _lib
    _lib.LIB_PATH = DEV
    import torch
    from tools.quick_bench import build as build_index, run
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    b = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    ix = build_index(n)
    names = {0: "full", 4: "no-mfma", 12: "no-mfma,no-ldsread", 1: "no-corpus-dma", 2: "no-query-dma", 3: "no-dma", 7: "no-dma,no-mfma", 8: "no-ldsread,no-mfma(=12?)", 16: "dma-not-waited", 19: "no-dma, raw barrier"}
    for abl in (0, 16, 3, 19, 12, 0):
        ix.set_option("ablate", abl)
        out = run(ix, b, 10, iters=5)
        print("ABL", abl, names[abl], "scan_ms", out["scan_ms"], flush=True)
