"""Does a stream that waited for another stream's PAST work also wait for work queued on that stream afterwards?  (ROCm 7.2, MI355X.)
main: wait for side's tail (work A, long finished) -> side gets work B (long) -> main launches K.  K's start relative to B's end, for
(1) main.wait_stream(side) issued right before B is queued, (2) an event recorded on side right after A and waited for by main.
Usage: python tools/micro/stream_wait_probe.py"""
import time
import torch

dev = torch.device("cuda:0")
side = torch.cuda.Stream(dev)
main = torch.cuda.current_stream()
x = torch.randn(8192, 8192, device=dev)
y = torch.randn(1024, 1024, device=dev)


def work(n, t):
    for _ in range(n):
        t = t @ t * 1e-4
    return t


for variant in ("wait_stream at take time", "event recorded behind A at submit time"):
    for _ in range(2):
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            a = work(2, x)                                   # A
            doneA = torch.cuda.Event(); doneA.record(side)
        torch.cuda.synchronize()                             # A finished long ago
        k0, k1, b1 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        b0 = torch.cuda.Event(enable_timing=True)
        if variant.startswith("wait_stream"):
            main.wait_stream(side)
        else:
            main.wait_event(doneA)
        with torch.cuda.stream(side):
            b0.record(side)
            b = work(12, x)                                  # B: ~tens of ms
            b1.record(side)
        k0.record(main)
        k = work(1, y)                                       # K on main
        k1.record(main)
        torch.cuda.synchronize()
    print(f"{variant:45s}: B runs {b0.elapsed_time(b1):7.2f} ms; K starts {b0.elapsed_time(k0):7.2f} ms after B starts (K after B ends: {b0.elapsed_time(k0) >= b0.elapsed_time(b1) - 0.05})")
