"""Developer: the encoder's four projection shapes through the BLAS library in its layouts — F.linear(x, W[N][K]) (what the module calls),
addmm with a pre-transposed weight W^T[K][N], and the fused-GELU epilogue variants torch exposes — TFLOP/s by HIP events, random data."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
T = int(sys.argv[1]) if len(sys.argv) > 1 else 20480
dev = "cuda:0"
g = torch.Generator(device=dev); g.manual_seed(0)
def bench(f, n=30):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for N, K in ((3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)):
    x = torch.randn((T, K), device=dev, dtype=torch.float16, generator=g)
    w = (torch.randn((N, K), device=dev, dtype=torch.float16, generator=g) * K ** -0.5)
    b = torch.randn((N,), device=dev, dtype=torch.float16, generator=g)
    wt = w.t().contiguous()
    fl = 2.0 * T * N * K
    r = {"linear W[N][K]": bench(lambda: torch.nn.functional.linear(x, w, b)),
         "addmm W^T[K][N]": bench(lambda: torch.addmm(b, x, wt)),
         "matmul W^T (no bias)": bench(lambda: x @ wt)}
    if N == 4096:
        r["linear + gelu (erf, two kernels)"] = bench(lambda: torch.nn.functional.gelu(torch.nn.functional.linear(x, w, b)))
        from rag_dpo_amd import _lib
        L = _lib.load()
        def lin_gelu_inplace():
            y = torch.nn.functional.linear(x, w, b)
            L.rdx_enc_gelu_f16(0, y.data_ptr(), y.numel(), torch.cuda.current_stream().cuda_stream)
            return y
        r["linear + librdx gelu in place (E13)"] = bench(lin_gelu_inplace)
        try:
            r["_addmm_activation gelu (tanh epilogue)"] = bench(lambda: torch._addmm_activation(b, x, wt, use_gelu=True))
        except Exception as e:
            r["_addmm_activation"] = float("nan")
    print(f"T {T} N {N} K {K}: " + " | ".join(f"{k} {v*1e3:.0f} us = {fl / v / 1e9:.0f} TF/s" for k, v in r.items()), flush=True)
