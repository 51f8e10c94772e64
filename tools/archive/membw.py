"""On-box probe: what pure-write / copy streams sustain on this MI355X (torch kernels), to read the witness
kernel's 86%-writes traffic against."""
import torch, time
dev = torch.device("cuda", 0)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (448, 1024, 4096):
    x = torch.empty(mb * 1024 * 1024, dtype=torch.uint8, device=dev)
    y = torch.empty_like(x)
    s = t(lambda: x.fill_(7)); print("fill  %5d MB: %.1f us  %.2f TB/s written" % (mb, s * 1e6, x.numel() / s / 1e12))
    s = t(lambda: y.copy_(x)); print("copy  %5d MB: %.1f us  %.2f TB/s read+written" % (mb, s * 1e6, 2 * x.numel() / s / 1e12))
    xi = x.view(torch.int32)
    s = t(lambda: xi.sum()); print("read  %5d MB: %.1f us  %.2f TB/s read" % (mb, s * 1e6, x.numel() / s / 1e12))
