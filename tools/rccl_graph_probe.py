"""Probe: RCCL with one rank on one GPU: eager + HIP-graph capture of all_reduce and self send/recv."""
import os, sys, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
print("init ok", flush=True)
a = torch.arange(1024, dtype=torch.float32, device=dev)
dist.all_reduce(a); torch.cuda.synchronize(); print("all_reduce ok", float(a.sum()), flush=True)
s = torch.arange(256, dtype=torch.float32, device=dev); r = torch.zeros(256, device=dev)
def xchg():
    ops = [dist.P2POp(dist.isend, s, 0), dist.P2POp(dist.irecv, r, 0)]
    for w in dist.batch_isend_irecv(ops): w.wait()
xchg(); torch.cuda.synchronize(); print("self sendrecv ok", bool((r == s).all()), flush=True)
side = torch.cuda.Stream()
def step():
    with torch.cuda.stream(side):
        side.wait_stream(torch.cuda.current_stream())
    s.add_(1.0)
    with torch.cuda.stream(side):
        side.wait_stream(torch.cuda.current_stream())
        xchg()
        dist.all_reduce(a)
    torch.cuda.current_stream().wait_stream(side)
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize(); print("eager us/step", (time.perf_counter() - t0) / 200 * 1e6, flush=True)
g = torch.cuda.CUDAGraph()
cs = torch.cuda.Stream()
with torch.cuda.stream(cs):
    with torch.cuda.graph(g, stream=cs, capture_error_mode="thread_local"):
        step()
print("capture ok", flush=True)
before = s.clone()
g.replay(); torch.cuda.synchronize()
print("replay ok", bool((s == before + 1).all()), bool((r == s).all()), s[:3].tolist(), r[:3].tolist(), before[:3].tolist(), flush=True)
g.replay(); torch.cuda.synchronize()
print("replay2", s[:3].tolist(), r[:3].tolist(), flush=True)
r.zero_(); torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
print("replay3 after r.zero_", s[:3].tolist(), r[:3].tolist(), flush=True)
t0 = time.perf_counter()
for _ in range(200): g.replay()
torch.cuda.synchronize(); print("graph us/step", (time.perf_counter() - t0) / 200 * 1e6, flush=True)
dist.destroy_process_group()
print("done", flush=True)
