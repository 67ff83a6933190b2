"""Dev aid: what the vendor fp32 GEMM reaches on this GPU (context for the tile engine's TFLOP/s)."""
import time, torch
dev = torch.device("cuda:0")
for n in (4096, 8192):
    a = torch.randn(n, n, device=dev); b = torch.randn(n, n, device=dev)
    for tr in ("NN", "TN"):
        f = (lambda: a @ b) if tr == "NN" else (lambda: a.t() @ b)
        for _ in range(3): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): f()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print("torch fp32 %s n=%d: %.1f TFLOP/s" % (tr, n, 2 * n ** 3 / dt / 1e12), flush=True)
# is the vendor path true fp32?  error vs fp64 reference on a 2048 problem
n = 2048
a = torch.randn(n, n, device=dev); b = torch.randn(n, n, device=dev)
ref = a.double() @ b.double()
err = ((a @ b).double() - ref).abs().max() / ref.abs().max()
print("max rel err of torch fp32 matmul vs fp64: %.3e (fp32 accumulate ~1e-6, tf32/bf16 split ~1e-3)" % float(err))
print("allow_tf32:", torch.backends.cuda.matmul.allow_tf32, torch.get_float32_matmul_precision())
