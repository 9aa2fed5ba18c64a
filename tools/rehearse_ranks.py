"""Rehearsal of the N>1 bench path on a one-GPU box (gloo through host memory, every rank on cuda:0) + an exactness
check of the merged result against a single-shard search. Usage: python tools/rehearse_ranks.py [world] [rows]"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
rows = sys.argv[2] if len(sys.argv) > 2 else "600000"
env = dict(os.environ, RDX_BENCH_REHEARSAL="1")
cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
       "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1", "--rows", rows,
       "--check-merged"]
p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
print(p.stdout[-3000:]); print(p.stderr[-1500:])
sys.exit(p.returncode)
