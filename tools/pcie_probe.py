"""PCIe copy rates of this box for page-locked host memory (development aid for the overlapped host path): one copy of N MB,
the same bytes as 10 copies, two streams at once, H2D and D2H at once.  torch only moves the bytes."""
import time

import torch

dev = torch.device("cuda:0")


def rate(fn, bytes_moved, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return dt * 1e3, bytes_moved / dt / 1e9


for mb in (1, 4, 12, 48):
    n = mb * (1 << 20)
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    h2 = torch.empty(n, dtype=torch.uint8).pin_memory()
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    d2 = torch.empty(n, dtype=torch.uint8, device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def d2h():
        with torch.cuda.stream(s1):
            h.copy_(d, non_blocking=True)

    def h2d():
        with torch.cuda.stream(s1):
            d.copy_(h, non_blocking=True)

    def d2h_10():
        with torch.cuda.stream(s1):
            k = n // 10
            for i in range(10):
                h[i * k:(i + 1) * k].copy_(d[i * k:(i + 1) * k], non_blocking=True)

    def d2h_two_streams():
        with torch.cuda.stream(s1):
            h[:n // 2].copy_(d[:n // 2], non_blocking=True)
        with torch.cuda.stream(s2):
            h[n // 2:].copy_(d[n // 2:], non_blocking=True)

    def both_ways():
        with torch.cuda.stream(s1):
            h.copy_(d, non_blocking=True)
        with torch.cuda.stream(s2):
            d2.copy_(h2, non_blocking=True)

    print(f"{mb:3d} MB: D2H {rate(d2h, n)[1]:6.1f} GB/s ({rate(d2h, n)[0]:.3f} ms)   H2D {rate(h2d, n)[1]:6.1f}   D2H as 10 copies {rate(d2h_10, n)[1]:6.1f}"
          f"   D2H on two streams {rate(d2h_two_streams, n)[1]:6.1f}   D2H + H2D at once {rate(both_ways, 2 * n)[1]:6.1f} (sum)")
