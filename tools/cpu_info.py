import os, torch, time
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads())
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
for nt in (8, 16, 32, torch.get_num_threads()):
    torch.set_num_threads(nt)
    a = torch.randn(4096, 4096)
    t0 = time.time(); torch.linalg.cholesky(a @ a.T + 4096 * torch.eye(4096)); t1 = time.time()
    print("threads", nt, "matmul+chol 4096: %.2f s" % (t1 - t0), flush=True)
