import torch, time
n = 67108864  # 268 MB
a = torch.randn(n, device="cuda"); b = torch.randn(n, device="cuda"); c = torch.empty_like(a)
def t(f, bytes_, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    return bytes_ / dt / 1e12, dt * 1e6
print("copy  (1R+1W)  %.2f TB/s  %.0f us" % t(lambda: c.copy_(a), 2 * 4 * n))
print("add   (2R+1W)  %.2f TB/s  %.0f us" % t(lambda: torch.add(a, b, out=c), 3 * 4 * n))
print("relu_ (1R+1W)  %.2f TB/s  %.0f us" % t(lambda: torch.relu_(a), 2 * 4 * n))
print("sum   (1R)     %.2f TB/s  %.0f us" % t(lambda: a.sum(), 4 * n))
s = torch.randn(n // 4, device="cuda"); o = torch.empty(n, device="cuda")
print("fill  (1W)     %.2f TB/s  %.0f us" % t(lambda: o.fill_(1.0), 4 * n))
