"""Developer: main-scan time of one shape for several library variants, alternating inside ONE process group of child runs.
   python tools/ab_shape.py ROWS B K ITERS name1 name2 ...   (names: base or tools/librdx_<name>.so built by tools/ab_lib.py build)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, json
sys.path.insert(0, %r)
from rag_dpo_amd import _lib
name = sys.argv[1]
if name != "base":
    _lib.LIB_PATH = os.path.join(%r, "tools", "librdx_" + name + ".so")
from tools.quick_bench import build, run
rows, b, k, iters = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
ix = build(rows)
best = None
for rep in range(3):
    o = run(ix, b, k, iters=iters)
    if best is None or o["scan_ms"] < best["scan_ms"]: best = o
print("RESULT", name, json.dumps(best))
''' % (ROOT, ROOT)
rows, b, k, iters = sys.argv[1:5]
names = sys.argv[5:]
for rnd in range(2):
    for n in names:
        p = subprocess.run([sys.executable, "-c", CHILD, n, rows, b, k, iters], capture_output=True, text=True)
        for ln in p.stdout.splitlines():
            if ln.startswith("RESULT"):
                d = json.loads(ln.split(" ", 2)[2])
                print(f"round {rnd} {n:>10}: scan {d['scan_ms']:.4f} ms  {d['scan_GBs']} GB/s  {d['scan_TFLOPs']} TF  wall {d['wall_ms']} ms  sample {d['ms_scan_sample']} tau {d['ms_tau']} refine {d['ms_refine']}", flush=True)
        if p.returncode:
            print(n, "FAILED", p.stderr[-800:], flush=True)
