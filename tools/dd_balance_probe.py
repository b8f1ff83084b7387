"""Load balancing measured on a one-GPU box: two ranks share cuda:0 over gloo, the system is a liquid film under thin
vapour (system.water_film) on two z slabs.  For equal slabs and for the slabs of dd.balance_domain each rank IN TURN
(the other waits at a barrier) times its own local + non-local kernels on its own lists; a decomposed step lasts as
long as its slowest rank, so max over ranks is what the balancing buys.
    python tools/dd_balance_probe.py [tiles_x,tiles_y,tiles_z per rank] [steps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, tiles, steps, ret):
    import torch
    import torch.distributed as dist
    import gromacs_cosmo_amd as g
    from gromacs_cosmo_amd import abi, dd
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    ip = g.system.ewald(1.0)
    fl = abi.FORCE_VIRIAL
    out = {}
    for balance in (0, 24):
        run = dd.SlabRun(tiles, "spce", ip, rank, world, 0, fl, balance=balance, film=(0.4, 8), device_halo=True)
        nb = run.nb
        nb.set_timing(False)
        for r in range(world):
            dist.barrier()
            if r != rank:
                continue

            def one():
                nb.clear_outputs(fl)
                nb.launch_kernel_resident(fl, abi.LOCAL)
                nb.launch_kernel_resident(fl, abi.NONLOCAL)
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.3:      # clock ramp
                for _ in range(50):
                    one()
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                one()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
        dist.barrier()
        d = run.dom
        out["balanced" if balance else "equal slabs"] = dict(
            kernels_ms=round(ms, 4), home_atoms=d.nhome, halo_atoms=d.nhalo, cluster_pairs=d.pl_local.nci_tot + d.pl_nonlocal.nci_tot,
            slab_nm=round(float(d.hi[2] - d.lo[2]), 3), history=[round(v, 3) for v in (d.balance_history or [])])
        del run
    ret[rank] = out
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    tiles = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "5,5,6").split(","))
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(worker, args=(2, 29533, tiles, steps, ret), nprocs=2, join=True)
        res = {f"rank{r}": ret[r] for r in range(2)}
    for k in ("equal slabs", "balanced"):
        res[f"step_bound_ms ({k})"] = max(res[f"rank{r}"][k]["kernels_ms"] for r in range(2))
    print(json.dumps(res, indent=1))
