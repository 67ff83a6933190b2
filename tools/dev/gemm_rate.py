"""dev: rate of the library's TN tile engine (plmc_gemm_tn) on a plain product, fp32 and fp64."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "projected-lmc_amd")]
import torch
from projectedlmc._dense import gemm_tn
dev = torch.device("cuda:0")
for dt, n in ((torch.float32, 8192), (torch.float64, 8192), (torch.float64, 4096)):
    A = torch.randn(1, n, n, dtype=dt, device=dev); B = torch.randn(1, n, n, dtype=dt, device=dev)
    for _ in range(2): C = gemm_tn(A, B)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 5
    for _ in range(K): C = gemm_tn(A, B)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / K
    print("%s n=%d: %.2f ms, %.1f TF" % (dt, n, 1e3 * t, 2.0 * n ** 3 / t / 1e12), flush=True)
    t0 = time.perf_counter()
    for _ in range(K): C = torch.matmul(A.transpose(-1, -2), B)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / K
    print("   torch.matmul(A^T, B): %.2f ms, %.1f TF" % (1e3 * t, 2.0 * n ** 3 / t / 1e12), flush=True)
