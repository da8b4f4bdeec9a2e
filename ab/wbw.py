import torch, time
x = torch.empty(1000000 * 1024, dtype=torch.float64, device="cuda:0")
for name, fn in (("fill_", lambda: x.fill_(1.5)), ("zero_", lambda: x.zero_()), ("mul_ (r+w)", lambda: x.mul_(1.0001))):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"{name:12s} {dt*1e3:.3f} ms  {x.numel()*8/dt/1e12:.2f} TB/s written")
