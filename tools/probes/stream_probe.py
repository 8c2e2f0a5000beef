"""Practical HBM rates of the box with PyTorch's own elementwise kernels (2.5 GB arrays, far beyond the 256 MB Infinity
Cache): the yardstick beside the 8 TB/s peak that bench.py's roofline fractions are priced against.
    python tools/probes/stream_probe.py"""
import torch, time
n = 640*1024*1024  # floats = 2.5 GB
a = torch.rand(n, device='cuda'); b = torch.empty_like(a)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = t(lambda: b.copy_(a)); print('copy  (R+W) %.3f ms  %.2f TB/s' % (ms, 2 * n * 4 / ms / 1e9))
ms = t(lambda: a.sum());     print('sum   (R)   %.3f ms  %.2f TB/s' % (ms, n * 4 / ms / 1e9))
ms = t(lambda: b.fill_(1.0)); print('fill  (W)   %.3f ms  %.2f TB/s' % (ms, n * 4 / ms / 1e9))
ms = t(lambda: a.mul_(1.0001)); print('scale (RMW) %.3f ms  %.2f TB/s' % (ms, 2 * n * 4 / ms / 1e9))
