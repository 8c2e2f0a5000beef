"""Probe: library FFT (hipFFT via torch.fft) throughput for the plane sizes of the FFT formulation.
Run on the GPU box:  python tools/probes/fft_probe.py"""
import sys
import time
import torch


def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def main():
    dev = torch.device('cuda:0')
    for planes, hy in ((8192, 267), (1024, 267), (256, 267), (2048, 527)):
        H = torch.rand(planes, hy, hy, device=dev)
        for L in ((270, 280, 288, 320, 384, 512) if hy == 267 else (528, 540, 576, 640)):
            gb = planes * L * (L // 2 + 1) * 8 / 1e9
            try:
                f = lambda: torch.fft.rfft2(H, s=(L, L))
                tf = t(f)
                F = f()
                ti = t(lambda: torch.fft.irfft2(F, s=(L, L)))
                tm = t(lambda: F.mul_(F))
                print('planes %5d  H %d  L %d  spectrum %.3f GB  rfft2 %.3f ms  irfft2 %.3f ms  (complex mul in place %.3f ms)'
                      % (planes, hy, L, gb, tf * 1e3, ti * 1e3, tm * 1e3), flush=True)
                del F
            except Exception as e:  # noqa
                print('planes', planes, 'L', L, 'failed', repr(e)[:200], flush=True)
        del H
        torch.cuda.empty_cache()


if __name__ == '__main__':
    sys.exit(main())
