import os, time, torch
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    if os.path.exists(f): print(f, open(f).read().strip())
a = torch.randn(2048, 1024); b = torch.randn(1024, 4096)
for n in (8, 16, 32, 64, 128, 256):
    torch.set_num_threads(n)
    a @ b
    t = time.time()
    for _ in range(5): a @ b
    print(n, "threads:", (time.time() - t) / 5 * 1e3, "ms per 2048x1024x4096 sgemm")
