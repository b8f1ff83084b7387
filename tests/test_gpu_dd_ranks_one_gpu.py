"""The N > 1 path rehearsed with TWO and FOUR processes sharing the one GPU of the box (gloo carries the messages; RCCL
needs one GPU per rank): SlabRun with the module's halo kernels or with torch indexing, unequal slabs from the load
balancing, a cut along x, four bricks, and bench.py starting its own ranks.

Round 2 kept these in a file collected last because a rank's stream was, now and then, never scheduled while other
processes held the card.  Round 3 found the cause (tools/stall_ab.sh): the module's non-local stream
had the HIGH priority and its local stream the normal one, the two wait for each other through hipStreamWaitEvent (a
barrier packet at the head of a hardware queue), and with several processes on one GPU a higher-priority queue parked
on such a packet starved the lower-priority queue it was waiting for.  Both streams now have the same priority; the
four-process case, which stalled in about half its runs, is a test again and the file needs no place in the order."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_dd_cpu import _free_port  # noqa: E402
from test_gpu_z_dd import _worker, _watch  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("model,device_halo", [("spc", True), ("spce", True), ("spce", False)])
def test_slabrun_two_ranks_one_gpu(model, device_halo):
    with mp.Manager() as m:
        ret = m.dict()
        ctx = mp.spawn(_worker, args=(2, _free_port(), model, device_halo, ret), nprocs=2, join=False)
        _watch(ctx)
        res = [ret[r] for r in range(2)]
    for r in res:
        assert r["err"] < 1e-5 and r["err_moved"] < 1e-5, r
        assert abs(r["e_lj"] - r["e_lj_ref"]) < 1e-5 * abs(r["e_lj_ref"]), r
        assert abs(r["e_el"] - r["e_el_ref"]) < 1e-5 * abs(r["e_el_ref"]), r
        assert r["vir_err"] < 1e-5, r

def test_load_balanced_slabs_two_ranks_one_gpu():
    """A liquid film under thin vapour on two z slabs: the cut plane moved by the reference's balancing rule on the
    ranks' cluster-pair counts (dd.balance_domain; the rule itself and four ranks: tests/test_dd_cpu.py), then the
    decomposed step on the device with the grids and lists of the unequal slabs made by the device search."""
    with mp.Manager() as m:
        ret = m.dict()
        _watch(mp.spawn(_worker, args=(2, _free_port(), "spce", True, ret, "gloo", False, (1, 1, 2), (), "half", 24, (0.4, 6)), nprocs=2, join=False))
        res = [ret[r] for r in range(2)]
    h = res[0]["history"]
    loads = [r["load"] for r in res]
    assert h[0] > 0.5 and max(loads) / (sum(loads) / 2) - 1 < 0.1, (h, loads)
    for r in res:
        assert r["err"] < 1e-5 and r["err_moved"] < 1e-5, r
        assert abs(r["e_lj"] - r["e_lj_ref"]) < 1e-5 * abs(r["e_lj_ref"]), r
        assert abs(r["e_el"] - r["e_el_ref"]) < 1e-5 * abs(r["e_el_ref"]), r
        assert r["vir_err"] < 1e-5, r


def test_brickrun_four_ranks_one_gpu():
    """Four bricks, grid (2, 2, 1), four processes on cuda:0 over gloo: the configuration that stalled in about half
    its runs while the non-local stream had a higher priority than the local one (module docstring)."""
    with mp.Manager() as m:
        ret = m.dict()
        _watch(mp.spawn(_worker, args=(4, _free_port(), "spce", True, ret, "gloo", False, (2, 2, 1), ()), nprocs=4, join=False))
        res = [ret[r] for r in range(4)]
    for r in res:
        assert r["err"] < 1e-5 and r["err_moved"] < 1e-5, r
        assert abs(r["e_lj"] - r["e_lj_ref"]) < 1e-5 * abs(r["e_lj_ref"]), r
        assert abs(r["e_el"] - r["e_el_ref"]) < 1e-5 * abs(r["e_el_ref"]), r
        assert r["vir_err"] < 1e-5, r


def test_brickrun_two_ranks_cut_along_x_one_gpu():
    """A cut along x instead of z, grid (2, 1, 1): two ranks share cuda:0 and talk over gloo; the search keeps its
    periodic shifts in y and z only."""
    with mp.Manager() as m:
        ret = m.dict()
        _watch(mp.spawn(_worker, args=(2, _free_port(), "spce", True, ret, "gloo", False, (2, 1, 1), ()), nprocs=2, join=False))
        res = [ret[r] for r in range(2)]
    for r in res:
        assert r["err"] < 1e-5 and r["err_moved"] < 1e-5, r
        assert abs(r["e_lj"] - r["e_lj_ref"]) < 1e-5 * abs(r["e_lj_ref"]), r
        assert abs(r["e_el"] - r["e_el_ref"]) < 1e-5 * abs(r["e_el_ref"]), r
        assert r["vir_err"] < 1e-5, r

def test_bench_starts_its_own_ranks_and_cuts_one_box():
    """`python bench.py --gpus 2 ... --strong` as the driver types it (no launcher): the script starts its two ranks
    itself, before it touches the GPU, and cuts ONE box into two slabs (BASELINE configs[3] in small); with a
    preconditioning phase, whose length rank 0's clock decides for both ranks.  On this
    one-GPU box both ranks share cuda:0 over gloo (BENCH_REHEARSAL=1): the plumbing is what is checked, not a number."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["BENCH_REHEARSAL"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--tiles", "2,2,4", "--strong", "--steps", "3",
                        "--warmup", "1", "--precondition", "0.2"], env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-800:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 3
    assert "10368 atoms" in d["config"]["workload"] and "cut into 2 z slabs" in d["config"]["workload"]
    # the same box on one rank, no launcher either: the two ranks together hold the same list
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--tiles", "2,2,4", "--strong", "--steps", "3",
                         "--warmup", "1", "--precondition", "0", "--no-cpu"], env=env, capture_output=True, text=True, timeout=280)
    assert r1.returncode == 0, r1.stderr[-800:]
    d1 = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][0])
    assert d1["n_gpus"] == 1 and "10368 atoms" in d1["config"]["workload"]
    assert abs(d["config"]["pair_slots_per_step"] - d1["config"]["pair_slots_per_step"]) < 0.02 * d1["config"]["pair_slots_per_step"]
