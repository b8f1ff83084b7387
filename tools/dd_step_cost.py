"""CPU cost of enqueueing one domain-decomposed step (dd.SlabRun.step), without the messages.

Two ranks share cuda:0 over gloo for the set-up (as tests/test_gpu_z_dd.py does); then the exchange and
the slot reduction are replaced by no-ops and the step is timed on rank 0 while rank 1 idles: what is
left is the host-side cost of the kernel launches, halo kernels and stream ordering.  Add the cost of the
three RCCL calls (tools/rccl_cost_probe.py measures them on one rank) for the step's CPU floor.
Usage: python tools/dd_step_cost.py [torch|device]"""
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, device_halo):
    import gromacs_cosmo_amd as g
    from gromacs_cosmo_amd import abi, dd
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    ip = g.system.ewald(1.0)
    run = dd.SlabRun((5, 5, 6), "spce", ip, rank, world, 0, abi.FORCE_VIRIAL, device_halo=device_halo)
    run.nb.set_timing(False)
    for _ in range(3):
        run.step()
    torch.cuda.synchronize()
    dist.barrier()
    dd._exchange = lambda *a: None
    dd.reduce_slots = lambda t: None
    if rank == 0:
        for n in (200, 200):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                run.step()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            print(f"device_halo={device_halo}: enqueue {1e6 * (t1 - t0) / n:.1f} us/step, with drain {1e6 * (t2 - t0) / n:.1f} us/step", flush=True)
        nb, h, fl = run.nb, run.halo, abi.FORCE_VIRIAL

        def t(name, fn, n=200):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            print(f"  {name:26s} {1e6 * (t1 - t0) / n:6.1f} us", flush=True)
        t("clear_outputs", lambda: nb.clear_outputs(fl))
        t("launch LOCAL", lambda: nb.launch_kernel_resident(fl, abi.LOCAL))
        t("launch NONLOCAL", lambda: nb.launch_kernel_resident(fl, abi.NONLOCAL))
        if device_halo:
            p = h._p
            t("halo_pack_x", lambda: nb.halo_pack_x(p[0], h.n_send, h.shift, p[2], abi.NONLOCAL))
            t("halo_unpack_x", lambda: nb.halo_unpack_x(p[1], h.n_halo, p[3], abi.NONLOCAL))
            t("halo_pack_f", lambda: nb.halo_pack_f(p[1], h.n_halo, p[4], abi.NONLOCAL))
            t("halo_unpack_f", lambda: nb.halo_unpack_f(p[0], h.n_send, p[5], h.seg[0][3] if h.seg else -1, abi.NONLOCAL))
        t("stream_wait", lambda: nb.stream_wait(abi.NONLOCAL, abi.LOCAL))
        t("_current_views", lambda: run._current_views())

        def ctx():
            with torch.cuda.stream(run.s_nl):
                pass
        t("with torch.cuda.stream", ctx)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "device"
    mp.spawn(worker, args=(2, 29541, mode == "device"), nprocs=2, join=True)
