"""Developer: the encoder layer's four GEMMs (F.linear, fp16) by token count — which canonical padded shapes the BLAS library likes."""
import sys, torch
dev = "cuda:0"
g = torch.Generator(device=dev); g.manual_seed(0)
def bench(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
shapes = ((3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096))
ws = [((torch.randn((N, K), device=dev, dtype=torch.float16, generator=g) * K ** -0.5), torch.randn((N,), device=dev, dtype=torch.float16, generator=g)) for N, K in shapes]
for T in [int(a) for a in sys.argv[1:]] or [20480, 20649, 20736, 20992, 21504, 22528, 29696, 29700, 30720]:
    xs = [torch.randn((T, K), device=dev, dtype=torch.float16, generator=g) for _, K in shapes]
    ts = [bench(lambda: torch.nn.functional.linear(x, w, b)) for x, (w, b) in zip(xs, ws)]
    print(f"T {T:6d}: " + " ".join(f"{t*1e3:6.1f}" for t in ts) + f" us  sum {sum(ts)*1e3:6.1f} us = {sum(ts)*1e6/T:5.2f} ns/token", flush=True)
