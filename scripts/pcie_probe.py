import torch, time
for mb in (1, 11, 64, 256):
    n = mb << 20
    d = torch.empty(n, dtype=torch.uint8, device="cuda")
    h = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    for _ in range(3):
        h.copy_(d, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        h.copy_(d, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    dt2 = (time.perf_counter() - t0) / reps
    print(f"{mb:4d} MB  D2H {n/dt/1e9:6.1f} GB/s ({dt*1e6:8.1f} us)   H2D {n/dt2/1e9:6.1f} GB/s")
