"""Developer: cost of the exchange step alone (all-gather of the packed partials + merge kernel) with a one-rank RCCL group."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
import torch, torch.distributed as dist
real_out = os.dup(1); os.dup2(2, 1)
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
from rag_dpo_amd.sharded import HipShard, ShardedSearcher
nq, k = 1024, 10
sh = HipShard(1024, 0)
ss = ShardedSearcher(sh, always_exchange=True)
per_pad, local, allb, out = ss._buffers(nq, k)
s, r, c, _f = ss.views(local, nq, k); c.fill_(k); r.copy_(torch.arange(nq * k, device=dev).view(nq, k)); s.copy_(torch.rand(nq, k, device=dev).sort(dim=1, descending=True).values)
def run(n, what):
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n):
        if what in ("both", "gather"): dist.all_gather_into_tensor(allb, local)
        if what in ("both", "merge"): sh.merge_packed(allb, per_pad, 1, nq, k, out[0], out[1], out[2])
    e1.record(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return e0.elapsed_time(e1) / n * 1e3, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6
for what in ("gather", "merge", "both"):
    run(20, what)
    g, h, w = run(200, what)
    os.write(real_out, f"{what}: gpu {g:.1f} us/iter, host enqueue {h:.1f} us/iter, wall {w:.1f} us/iter\n".encode())
dist.destroy_process_group()
