import torch, time
def bench(nbytes, iters=20):
    n = nbytes // 8
    a = torch.empty(n, dtype=torch.float64, device="cuda")
    b = torch.empty(n, dtype=torch.float64, device="cuda")
    a.fill_(1.0); torch.cuda.synchronize()
    # write then read-back pattern: b = a * 2 (read a, write b), then a = b * 2 (read b just written)
    for _ in range(3): torch.mul(a, 2.0, out=b); torch.mul(b, 0.5, out=a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): torch.mul(a, 2.0, out=b); torch.mul(b, 0.5, out=a)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    tot = iters * 2 * 2 * nbytes
    return tot / (ms * 1e-3) / 1e12
for mb in (16, 32, 64, 96, 128, 192, 256, 512, 2048, 8192):
    print("buffers 2 x %5d MiB: %.2f TB/s (read+write)" % (mb, bench(mb << 20)))
