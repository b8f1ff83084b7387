"""Times the device-side x gather / f scatter on device-resident x and f at the
config-4 size (1,533,168 atoms) and the bench size; run under rocprofv3 --kernel-trace --stats
for the kernel durations.  Algorithmic bytes per atom: gather 4 (a) + 12 (x) + 16 + 16 (xq
read-modify-write) = 48; scatter 4 (cell) + 12 (fnb) + 12 + 12 (f read-modify-write) = 40."""
import sys; sys.path.insert(0, ".")
import numpy as np, torch
import gromacs_cosmo_amd as g
from gromacs_cosmo_amd import abi
torch.cuda.init()
nb = g.gpu.NbnxnHip(0)
nb.set_timing(False)
ip = g.system.ewald(1.0)
for tiles in ((5, 5, 6), (13, 13, 14)):
    s = g.system.water_box(*tiles, model="spce")
    p = g.workload.prepare(s, ip, nb.min_ci_balanced())
    g.workload.upload(nb, p, ip)
    nb.set_atom_order(s.natoms, p.host.atom_order(), p.host.cell_index())
    nb.upload_xq(p.xq)
    x_t = torch.from_numpy(s.x).cuda(); f_t = torch.zeros_like(x_t)
    st = torch.cuda.ExternalStream(nb.stream(abi.LOCAL))
    torch.cuda.synchronize()
    for name, fn in (("gather_x", lambda: nb.copy_x_to_xq(x_t.data_ptr())), ("scatter_f", lambda: nb.add_f(f_t.data_ptr()))):
        for _ in range(5):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 200
        a.record(st)
        for _ in range(n):
            fn()
        b.record(st); torch.cuda.synchronize()
        us = a.elapsed_time(b) / n * 1e3
        bpa = 48 if name == "gather_x" else 40
        print(f"{name} {s.natoms} atoms: {us:.2f} us per call back-to-back (launch gaps included), "
              f"{s.natoms * bpa / us / 1e3:.0f} GB/s algorithmic", flush=True)
