import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.quick_bench import build, run
n = int(sys.argv[1]); shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[2:]]
ix = build(n)
for b, k in shapes:
    run(ix, b, k, iters=10)
