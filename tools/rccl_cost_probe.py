"""Probe: CPU enqueue cost per call of the pieces of the DD step (one rank, RCCL)."""
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
a = torch.zeros(2 * 137 * 64, dtype=torch.float32, device=dev)
s = torch.zeros((12000, 4), device=dev); r = torch.zeros((12000, 4), device=dev)
big = torch.zeros((110000, 4), device=dev)
idx = torch.randint(0, 110000, (12000,), device=dev)
side = torch.cuda.Stream()
def xchg():
    ops = [dist.P2POp(dist.isend, s, 0), dist.P2POp(dist.irecv, r, 0)]
    for w in dist.batch_isend_irecv(ops): w.wait()
def xchg2():
    w1 = dist.isend(s, 0); w2 = dist.irecv(r, 0)
def t(name, fn, n=300):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:28s} enqueue {(t1 - t0) / n * 1e6:7.1f} us   with drain {(t2 - t0) / n * 1e6:7.1f} us", flush=True)
t("batch_isend_irecv(self)", xchg)
t("all_reduce 70KB", lambda: dist.all_reduce(a))
t("index_select out=", lambda: torch.index_select(big, 0, idx, out=s))
t("index_copy_", lambda: big.index_copy_(0, idx, r))
t("index_add_", lambda: big.index_add_(0, idx, r))
t("add_", lambda: s.add_(1.0))
t("wait_stream pair", lambda: (side.wait_stream(torch.cuda.current_stream()), torch.cuda.current_stream().wait_stream(side)))
def ctx():
    with torch.cuda.stream(side):
        pass
t("with stream ctx", ctx)
def a2a():
    dist.all_to_all_single(r, s, output_split_sizes=[12000], input_split_sizes=[12000])
r.zero_(); s.fill_(3.0); a2a(); torch.cuda.synchronize(); print("a2a self ok", bool((r == s).all()), flush=True)
t("all_to_all_single(self)", a2a)
e1 = torch.cuda.Event(); e2 = torch.cuda.Event()
cur = torch.cuda.current_stream()
def evsync():
    e1.record(cur); side.wait_event(e1); e2.record(side); cur.wait_event(e2)
t("event record+wait pair", evsync)
def a2a_async():
    w = dist.all_to_all_single(r, s, output_split_sizes=[12000], input_split_sizes=[12000], async_op=True)
t("all_to_all_single async", a2a_async)
# the module's own RCCL path (nbnxn_hip_dd_*), one rank talking to itself
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gromacs_cosmo_amd as g
from gromacs_cosmo_amd import abi
sysm = g.system.water_box(5, 5, 6, model="spce")
ip = g.system.ewald(1.0)
p = g.workload.prepare(sysm, ip, 0)
nb = g.gpu.NbnxnHip(0, local_and_nonlocal=True)
nb.set_timing(False)
g.workload.upload(nb, p, ip)
nb.upload_xq(p.xq)
nb.dd_init(nb.dd_get_unique_id(), 0, 1)
perm = np.random.default_rng(1).permutation(p.natoms)
nb.dd_set_pulse(0, 0, perm[:8575], perm[8575:17150], (0, 0, 1.0), 17)
nb.clear_outputs(abi.FORCE_VIRIAL)
nb.wait_gpu(0)
t("module dd_move_x (8575 atoms)", nb.dd_move_x)
t("module dd_move_f", nb.dd_move_f)
t("module dd_sum_outputs", lambda: nb.dd_sum_outputs(abi.LOCAL))
nb.close()
dist.destroy_process_group()
