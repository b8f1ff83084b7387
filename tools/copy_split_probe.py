"""How fast do the step's two PCIe copies go, and does splitting one over two streams (two SDMA engines) help?
1.59 MB host-to-device and 1.19 MB device-to-host, pinned host memory, HIP events on the issuing stream.
usage: python tools/copy_split_probe.py"""
import time
import torch

dev = torch.device("cuda", 0)
for name, nbytes, h2d in (("H2D xq 1.59 MB", 99200 * 16, True), ("D2H f 1.19 MB", 99200 * 12, False), ("D2H slots 36 KB", 36864, False), ("D2H 548 B", 548, False)):
    n = nbytes // 4
    host = torch.empty(n, dtype=torch.float32).pin_memory()
    devt = torch.empty(n, dtype=torch.float32, device=dev)
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    src, dst = (host, devt) if h2d else (devt, host)

    def one():
        with torch.cuda.stream(s0):
            dst.copy_(src, non_blocking=True)

    def two():
        h = n // 2
        ev = torch.cuda.Event()
        with torch.cuda.stream(s0):
            ev.record(s0)
        s1.wait_event(ev)
        with torch.cuda.stream(s1):
            dst[h:].copy_(src[h:], non_blocking=True)
            e1 = torch.cuda.Event(); e1.record(s1)
        with torch.cuda.stream(s0):
            dst[:h].copy_(src[:h], non_blocking=True)
            s0.wait_event(e1)

    for label, fn in (("one copy", one), ("two halves on two streams", two)):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(50):
            a.record(s0); fn(); b.record(s0)
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        t0 = time.perf_counter()
        for _ in range(200):
            fn(); torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 200 * 1e6
        ts.sort()
        print(f"{name:16s} {label:26s}: events median {ts[len(ts) // 2]:6.1f} us ({nbytes / ts[len(ts) // 2] / 1e3:5.1f} GB/s), issue + wait on the host {wall:6.1f} us")
