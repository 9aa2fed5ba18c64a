"""Developer: what PyTorch's TunableOp (run-time choice among the BLAS libraries' solutions per GEMM shape) would buy the encoder's four
projection shapes: default heuristic vs tuned, TFLOP/s by HIP events. usage: gemm_tunable.py [T]"""
import os, sys, time, torch
T = int(sys.argv[1]) if len(sys.argv) > 1 else 21504
dev = "cuda:0"
g = torch.Generator(device=dev); g.manual_seed(0)
def bench(f, n=30):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
shapes = ((3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096))
data = []
for N, K in shapes:
    x = torch.randn((T, K), device=dev, dtype=torch.float16, generator=g)
    w = (torch.randn((N, K), device=dev, dtype=torch.float16, generator=g) * K ** -0.5)
    b = torch.randn((N,), device=dev, dtype=torch.float16, generator=g)
    data.append((x, w, b))
base = [bench(lambda: torch.nn.functional.linear(x, w, b)) for x, w, b in data]
import torch.cuda.tunable as tun
tun.set_filename(os.path.join(os.environ.get("TMPDIR", "/tmp"), "rdx_tunableop.csv"))
tun.enable(True)
tun.tuning_enable(True)
tun.set_max_tuning_duration(int(os.environ.get("TUNE_MS", "60")))
tun.set_max_tuning_iterations(20)
t0 = time.time()
for x, w, b in data:
    torch.nn.functional.linear(x, w, b)
torch.cuda.synchronize()
t_tune = time.time() - t0
tun.tuning_enable(False)
tuned = [bench(lambda: torch.nn.functional.linear(x, w, b)) for x, w, b in data]
for (N, K), a, c in zip(shapes, base, tuned):
    fl = 2.0 * T * N * K
    print(f"T {T} N {N} K {K}: default {a*1e3:.0f} us = {fl / a / 1e9:.0f} TF/s | tuned {c*1e3:.0f} us = {fl / c / 1e9:.0f} TF/s ({(a / c - 1) * 100:+.1f} %)", flush=True)
print(f"layer sum: default {sum(base)*1e3:.0f} us, tuned {sum(tuned)*1e3:.0f} us; tuning the four shapes took {t_tune:.1f} s")
for r in tun.get_results():
    print("  ", r)
