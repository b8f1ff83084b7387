"""The decomposed step on the GPU in ONE process: a rank that is its own neighbour in one to three dimensions carries
the whole step -- local + non-local lists on two HIP streams, the halo kernels, real RCCL sends / receives (to itself)
issued by the module or by torch.distributed, the slot reduction, half shell / eighth shell / staged pulses -- and the
module's exchange calls are checked row by row.  The rehearsals with two and four processes sharing the card over gloo
are in test_gpu_dd_ranks_one_gpu.py, which borrows _worker and _watch from here.

Worker processes are watched with a deadline, so that a stuck rank ends as a failure with the ranks' stacks on record
(gpurun_out/dd_hang_rank*.txt) instead of a hang."""
import os
import sys
import time

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_dd_cpu import _free_port  # noqa: E402

pytestmark = pytest.mark.gpu


DEADLINE_S = 100


def _worker(rank, world, port, model, device_halo, ret, backend="gloo", rccl_direct=False, grid=None, self_dims=(2,), scheme="half",
            balance=0, film=None, break_rccl=False):
    import datetime
    import faulthandler
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    hang_log = open(os.path.join(ROOT, "gpurun_out", f"dd_hang_rank{rank}.txt"), "w")
    faulthandler.dump_traceback_later(DEADLINE_S - 20, file=hang_log, exit=False)
    import gromacs_cosmo_amd as g
    from gromacs_cosmo_amd import abi, dd
    import oracle as O
    from util import virial_from
    torch.cuda.set_device(0)
    dist.init_process_group(backend, init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=60), **({"device_id": torch.device("cuda", 0)} if backend == "nccl" else {}))
    try:
        ip = g.system.reaction_field(0.9, rlist=1.0) if model == "spc" else g.system.ewald(0.9, rlist=1.0)
        tiles = (2, 2, 2)
        if break_rccl:
            # the module's first exchange brings one wrong row: SlabRun's self-check has to notice, put the rows back
            # and take the run to torch.distributed
            real_send_x = dd.RcclHalo.send_x

            def bad_send_x(self):
                real_send_x(self)
                d_xq, _, _ = self.nb.device_buffers()
                xq = torch.as_tensor(dd._DevArray(d_xq, (self.dom.npad, 4)), device=torch.device("cuda", 0))
                torch.cuda.synchronize()
                xq[int(self.halo_slots[0]), 0] += 0.25
                torch.cuda.synchronize()
            dd.RcclHalo.send_x = bad_send_x
        run = dd.SlabRun(tiles, model, ip, rank, world, 0, abi.FORCE_VIRIAL, min_ci_balanced=300, device_halo=device_halo,
                         rccl_direct=rccl_direct, grid=grid, self_dims=self_dims, scheme=scheme, balance=balance, film=film)
        dom = run.dom
        # a test that asks for the module's own RCCL path must get it: the fall-back to torch.distributed is for
        # production runs, here it would hide a broken nbnxn_hip_dd_* path behind a line on stderr
        assert run.rccl_direct == (rccl_direct and not break_rccl), "the module's RCCL path fell back to torch.distributed"
        assert len(dom.msgs) == {"half": {1: 1, 2: 4, 3: 13}, "eighth": {1: 1, 2: 3, 3: 7}, "staged": {1: 1, 2: 2, 3: 3}}[scheme][sum(dom.decomp)]
        # wreck the imported coordinates on the device: the step has to bring them back
        run.xq_t[run.halo.halo_slots, :3] = 1e6
        for _ in range(3):
            run.step()
        f_home = run.forces_home()
        e = run.read_energies().cpu().numpy()
        glob = run.glob
        prep = g.workload.prepare(glob, ip, 0)
        o = O.kernel_gpu_list(prep.xq, prep.type, prep.ntype, prep.nbfp, prep.shift_vec, prep.pl, ip)
        f_ref = np.zeros((glob.natoms, 3), dtype=np.float32)
        prep.host.add_f(np.asarray(o["f"], dtype=np.float32), f_ref)
        fr = f_ref[dom.home_ids]
        # virial: home atoms' x (x) f summed over ranks, plus the shift-force term, which includes the forces
        # that came back across the periodic wrap (dd_move_f, domdec.c:837-851)
        xf_t = torch.from_numpy(glob.x[dom.home_ids].astype(np.float64).T @ f_home.astype(np.float64))
        if backend == "nccl":
            xf_t = xf_t.cuda()
        dist.all_reduce(xf_t)
        xf_t = xf_t.cpu()
        vir_dd = -0.5 * (xf_t.numpy() + dom.shift_vec.astype(np.float64).T @ e[2:2 + 135].reshape(45, 3))
        vir_ref = virial_from(np.concatenate([glob.x, np.zeros((glob.natoms, 1), np.float32)], 1), f_ref,
                              prep.shift_vec, np.asarray(o["fshift"]))
        # Atoms move between steps: new home coordinates are gathered on the LOCAL stream (nbnxn_hip_copy_x_to_xq) and the
        # step follows at once; the halo pack on the non-local stream has to wait for that gather.  Forces against the
        # oracle on the moved coordinates (the lists were built with a buffer: rlist = rc + 0.1 nm).
        err_moved = 0.0
        if run.rccl_direct or run.device_halo:
            rng = np.random.default_rng(99)
            for it in range(3):
                x_new = (glob.x + rng.normal(0, 0.004, glob.x.shape)).astype(np.float32)
                x_all = np.ascontiguousarray(np.concatenate([x_new[dom.home_ids], x_new[dom.halo_ids]]))
                run.nb.copy_x_to_xq(x_all, abi.LOCAL)
                run.step()
            f_moved = run.forces_home()
            pm = g.workload.prepare(glob, ip, 0)
            xq_m = pm.xq.copy()
            pm.host.copy_x_to_xq(x_new, xq_m)
            om = O.kernel_gpu_list(xq_m, pm.type, pm.ntype, pm.nbfp, pm.shift_vec, pm.pl, ip)
            f_ref_m = np.zeros((glob.natoms, 3), dtype=np.float32)
            pm.host.add_f(np.asarray(om["f"], dtype=np.float32), f_ref_m)
            frm = f_ref_m[dom.home_ids]
            err_moved = float(np.sqrt(((f_moved - frm) ** 2).sum() / (frm ** 2).sum()))
        ret[rank] = dict(err=float(np.sqrt(((f_home - fr) ** 2).sum() / (fr ** 2).sum())), err_moved=err_moved,
                         e_lj=float(e[0]), e_el=float(e[1]), e_lj_ref=float(o["Vvdw"]), e_el_ref=float(o["Vc"]),
                         vir_err=float(np.abs(vir_dd - vir_ref).max() / np.abs(vir_ref).max()),
                         history=dom.balance_history, load=dom.pl_local.nci_tot + dom.pl_nonlocal.nci_tot)
    finally:
        dist.destroy_process_group()
        faulthandler.cancel_dump_traceback_later()
        hang_log.close()
        if os.path.getsize(hang_log.name) == 0:
            os.remove(hang_log.name)


def _watch(ctx):
    t0 = time.time()
    while not ctx.join(timeout=2):
        if time.time() - t0 > DEADLINE_S:
            alive = [i for i, p in enumerate(ctx.processes) if p.is_alive()]
            for p in ctx.processes:
                if p.is_alive():
                    p.kill()
            pytest.fail(f"ranks {alive} still running after {DEADLINE_S} s; their stacks are in gpurun_out/dd_hang_rank*.txt")


@pytest.mark.parametrize("reserve_cus", ["default", "0"])
@pytest.mark.parametrize("model,rccl_direct", [("spce", True), ("spc", True), ("spce", False)])
def test_slabrun_one_rank_over_rccl(model, rccl_direct, reserve_cus, monkeypatch):
    """The whole decomposed step over the RCCL backend on the one GPU of the box: ONE rank that is its own neighbour
    (periodic in z through a halo of its own atoms), so every message is a real RCCL send / receive (to itself) and the
    reduction a real all-reduce.  rccl_direct: issued by the HIP module (nbnxn_hip_dd_*, what bench.py --gpus N uses);
    otherwise through torch.distributed (all_to_all_single).  Forces, energies and virial against the single-domain
    oracle."""
    # both kinds of local stream: with the CU reservation (a CU-masked stream, the coordinate exchange issued beside the
    # local kernel) and without it (an ordinary non-blocking stream, the exchange issued first)
    # ... and both ways of issuing the module's step: from one call (nbnxn_hip_dd_step, the default) and as seven calls
    if reserve_cus == "0":
        monkeypatch.setenv("GMX_HIP_NB_RESERVE_CUS", "0")
        monkeypatch.setenv("NBNXN_DD_STEP_CALLS", "many")
    else:
        monkeypatch.delenv("GMX_HIP_NB_RESERVE_CUS", raising=False)
        monkeypatch.delenv("NBNXN_DD_STEP_CALLS", raising=False)
    with mp.Manager() as m:
        ret = m.dict()
        _watch(mp.spawn(_worker, args=(1, _free_port(), model, True, ret, "nccl", rccl_direct), nprocs=1, join=False))
        r = ret[0]
    assert r["err"] < 1e-5 and r["err_moved"] < 1e-5, r
    assert abs(r["e_lj"] - r["e_lj_ref"]) < 1e-5 * abs(r["e_lj_ref"]), r
    assert abs(r["e_el"] - r["e_el_ref"]) < 1e-5 * abs(r["e_el_ref"]), r
    assert r["vir_err"] < 1e-5, r


def test_fall_back_from_a_broken_rccl_exchange():
    """The module's first exchange over RCCL brings one wrong coordinate (a patched RcclHalo.send_x): SlabRun's check
    against the coordinates the set-up brought has to catch it, restore the halo rows it had wrecked for the check and
    carry on through torch.distributed -- forces, energies and virial still those of the single domain.  (The tests
    that ask for the module's path assert that they GOT it, so a fall-back cannot hide a broken path there.)"""
    with mp.Manager() as m:
        ret = m.dict()
        _watch(mp.spawn(_worker, args=(1, _free_port(), "spce", True, ret, "nccl", True, None, (2,), "half", 0, None, True), nprocs=1, join=False))
        r = ret[0]
    assert r["err"] < 1e-5 and r["err_moved"] < 1e-5, r
    assert abs(r["e_lj"] - r["e_lj_ref"]) < 1e-5 * abs(r["e_lj_ref"]), r
    assert abs(r["e_el"] - r["e_el_ref"]) < 1e-5 * abs(r["e_el_ref"]), r
    assert r["vir_err"] < 1e-5, r


@pytest.mark.parametrize("self_dims,rccl_direct", [((0, 1, 2), True), ((1, 2), True), ((0, 1, 2), False)])
def test_brickrun_one_rank_over_rccl(self_dims, rccl_direct):
    """The 13-message step of a decomposition in three dimensions (4 messages for two) on the one GPU of the box: one
    rank that is its own neighbour in every cut dimension, so all messages are real RCCL sends / receives to itself,
    issued by the module in ONE grouped call between one pack and one unpack kernel (nbnxn_hip_dd_set_messages), or
    through torch.distributed.  14 grid zones in the device search."""
    with mp.Manager() as m:
        ret = m.dict()
        _watch(mp.spawn(_worker, args=(1, _free_port(), "spce", True, ret, "nccl", rccl_direct, (1, 1, 1), self_dims), nprocs=1, join=False))
        r = ret[0]
    assert r["err"] < 1e-5 and r["err_moved"] < 1e-5, r
    assert abs(r["e_lj"] - r["e_lj_ref"]) < 1e-5 * abs(r["e_lj_ref"]), r
    assert abs(r["e_el"] - r["e_el_ref"]) < 1e-5 * abs(r["e_el_ref"]), r
    assert r["vir_err"] < 1e-5, r


@pytest.mark.parametrize("self_dims", [(0, 1, 2), (0, 1)])
def test_eighth_shell_one_rank_over_rccl(self_dims):
    """The reference's eighth shell (dd_zo / dd_zp3, domdec.c:432-449) on the GPU: 7 (3) imported zones on one rank that is
    its own neighbour in three (two) dimensions, messages over RCCL issued by the module, and a NON-LOCAL list with
    pairs between imported zones -- i-clusters that are not home clusters, as stock mdrun's search gives the kernel
    under domain decomposition.  Grids and lists from the device search (nbnxn_hip_make_pairlist_zones), checked against
    the host builder's by SlabRun."""
    with mp.Manager() as m:
        ret = m.dict()
        _watch(mp.spawn(_worker, args=(1, _free_port(), "spce", True, ret, "nccl", True, (1, 1, 1), self_dims, "eighth"), nprocs=1, join=False))
        r = ret[0]
    assert r["err"] < 1e-5 and r["err_moved"] < 1e-5, r
    assert abs(r["e_lj"] - r["e_lj_ref"]) < 1e-5 * abs(r["e_lj_ref"]), r
    assert abs(r["e_el"] - r["e_el_ref"]) < 1e-5 * abs(r["e_el_ref"]), r
    assert r["vir_err"] < 1e-5, r


@pytest.mark.parametrize("self_dims,backend,rccl_direct", [((0, 1, 2), "nccl", True), ((0, 1), "nccl", True), ((0, 1, 2), "gloo", False)])
def test_staged_pulses_one_rank(self_dims, backend, rccl_direct):
    """The reference's staged pulses (dd_move_x / dd_move_f, domdec.c:667-870): one pulse per cut dimension, the later
    ones forwarding the zones the earlier ones brought, forces back in reverse order -- through the module's phases
    (nbnxn_hip_dd_set_phases) over RCCL, and through its per-pulse pack / unpack kernels with torch.distributed carrying
    the messages.  Same zones and lists as the eighth shell with direct messages; coordinates that moved are forwarded
    in the same step."""
    with mp.Manager() as m:
        ret = m.dict()
        _watch(mp.spawn(_worker, args=(1, _free_port(), "spce", True, ret, backend, rccl_direct, (1, 1, 1), self_dims, "staged"), nprocs=1, join=False))
        r = ret[0]
    assert r["err"] < 1e-5 and r["err_moved"] < 1e-5, r
    assert abs(r["e_lj"] - r["e_lj_ref"]) < 1e-5 * abs(r["e_lj_ref"]), r
    assert abs(r["e_el"] - r["e_el_ref"]) < 1e-5 * abs(r["e_el_ref"]), r
    assert r["vir_err"] < 1e-5, r


def _rccl_worker(rank, port, ret):
    """one rank, the RCCL backend: the exchange and the slot reduction as bench.py --gpus N issues them,
    to itself (RCCL needs one GPU per rank, so one rank is all a one-GPU box can give it)"""
    import datetime
    import faulthandler
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    hang_log = open(os.path.join(ROOT, "gpurun_out", "dd_hang_rccl.txt"), "w")
    faulthandler.dump_traceback_later(DEADLINE_S - 20, file=hang_log, exit=False)
    from gromacs_cosmo_amd import dd
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev,
                            timeout=datetime.timedelta(seconds=60))
    try:
        side = torch.cuda.Stream()
        ok = True
        with torch.cuda.stream(side):
            for n in (1000, 0, 37):
                s = torch.randn((n, 4), device=dev)
                r = torch.zeros((n, 4), device=dev)
                dd._exchange(s, 0, r, 0)
                ok = ok and bool((r == s).all())
            ids = torch.arange(50, dtype=torch.int64, device=dev)
            got = torch.zeros_like(ids)
            dd._exchange(ids, 0, got, 0)
            ok = ok and bool((got == ids).all())
            out = torch.ones(1000, device=dev)
            dd.reduce_slots(out)
            ok = ok and bool((out == 1).all())
        torch.cuda.synchronize()
        ret["ok"] = ok
    finally:
        dist.destroy_process_group()
        faulthandler.cancel_dump_traceback_later()
        hang_log.close()
        if os.path.getsize(hang_log.name) == 0:
            os.remove(hang_log.name)


def test_exchange_over_rccl_single_rank():
    with mp.Manager() as m:
        ret = m.dict()
        _watch(mp.spawn(_rccl_worker, args=(_free_port(), ret), nprocs=1, join=False))
        assert ret["ok"]


def _rccl_direct_worker(rank, port, ret):
    """one rank: the module's own RCCL path (nbnxn_hip_dd_*) sending to itself"""
    import faulthandler
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    hang_log = open(os.path.join(ROOT, "gpurun_out", "dd_hang_rccl_direct.txt"), "w")
    faulthandler.dump_traceback_later(DEADLINE_S - 20, file=hang_log, exit=False)
    import gromacs_cosmo_amd as g
    from gromacs_cosmo_amd import abi
    from gromacs_cosmo_amd.dd import _DevArray
    torch.cuda.set_device(0)
    try:
        s = g.system.water_box(2, 2, 2, model="spc")
        ip = g.system.reaction_field(0.9)
        p = g.workload.prepare(s, ip, 0)
        nb = g.gpu.NbnxnHip(0, local_and_nonlocal=True)
        fresh = g.gpu.NbnxnHip(0)
        refused = 0
        for call in (fresh.dd_move_x, lambda: fresh.dd_sum_outputs(abi.LOCAL)):      # no communicator yet
            try:
                call()
            except g.gpu.HipError:
                refused += 1
        fresh.close()
        g.workload.upload(nb, p, ip)
        nb.upload_xq(p.xq)
        nb.dd_init(nb.dd_get_unique_id(), 0, 1)
        try:
            nb.dd_move_x()                                                          # no pulse yet
        except g.gpu.HipError:
            refused += 1
        try:
            nb.dd_set_pulse(1, 0, [1, 2], [3, 4])                                   # neighbour outside the communicator
        except g.gpu.HipError:
            refused += 1
        rng = np.random.default_rng(9)
        perm = rng.permutation(p.natoms)
        send, recv = perm[:500].astype(np.int32), perm[500:1100].astype(np.int32)[:500]
        shift = np.array([0.1, 0.2, 3.0], dtype=np.float32)
        nb.dd_set_pulse(0, 0, send, recv, shift, 17)
        nb.clear_outputs(abi.FORCE_VIRIAL)
        nb.wait_gpu(0)
        d_xq, d_f, _ = nb.device_buffers()
        d_out, nslots, stride = nb.output_slots()
        xq_t = torch.as_tensor(_DevArray(d_xq, (p.natoms, 4)), device="cuda:0")
        f_t = torch.as_tensor(_DevArray(d_f, (p.natoms, 3)), device="cuda:0")
        out_t = torch.as_tensor(_DevArray(d_out, (nslots * stride,)), device="cuda:0")
        ok = True
        for _ in range(2):
            nb.dd_move_x()
        torch.cuda.synchronize()
        want = p.xq.copy()
        want[recv, :3] = p.xq[send, :3] + shift
        want[recv, 3] = p.xq[send, 3]
        ok = ok and np.array_equal(xq_t.cpu().numpy(), want)
        f0 = rng.normal(0, 300, (p.natoms, 3)).astype(np.float32)
        f_t.copy_(torch.from_numpy(f0))
        torch.cuda.synchronize()
        nb.dd_move_f()
        nb.stream_wait(abi.LOCAL, abi.NONLOCAL)
        nb.dd_sum_outputs(abi.LOCAL)
        torch.cuda.synchronize()
        wantf = f0.copy()
        wantf[send] += f0[recv]
        ok = ok and np.array_equal(f_t.cpu().numpy(), wantf)
        o = out_t.cpu().numpy().astype(np.float64).reshape(nslots, stride).sum(0)      # the shift forces land in any of the slots
        got = o[2 + 3 * 17:5 + 3 * 17]
        ok = ok and np.abs(got - f0[recv].astype(np.float64).sum(0)).max() < 1e-6 * np.abs(f0[recv]).sum()
        o[2 + 3 * 17:5 + 3 * 17] = 0
        ok = ok and not o.any()
        # several messages in one call (nbnxn_hip_dd_set_messages): ragged row counts, an empty message, two periodic
        # crossings and one message without; the rows of a message are padded to a multiple of 64 inside the module
        counts = (130, 0, 77, 64)
        shifts = np.array([[0.5, 0, 0], [0, 0, 0], [0, -1.5, 2.5], [0, 0, 0]], dtype=np.float32)
        sidx = (23, -1, 40, -1)
        perm = rng.permutation(p.natoms)
        n_tot = sum(counts)
        send, recv = perm[:n_tot].astype(np.int32), perm[n_tot:2 * n_tot].astype(np.int32)
        nb.dd_set_messages([(0, 0, c, c, si, sh) for c, sh, si in zip(counts, shifts, sidx)], send, recv)
        nb.upload_xq(p.xq)
        nb.clear_outputs(abi.FORCE_VIRIAL)
        nb.wait_gpu(0)
        nb.dd_move_x()
        torch.cuda.synchronize()
        row_shift = np.repeat(shifts, counts, axis=0)
        want = p.xq.copy()
        want[recv, :3] = p.xq[send, :3] + row_shift
        want[recv, 3] = p.xq[send, 3]
        ok = ok and np.array_equal(xq_t.cpu().numpy(), want)
        f_t.copy_(torch.from_numpy(f0))
        torch.cuda.synchronize()
        nb.dd_move_f()
        torch.cuda.synchronize()
        wantf = f0.copy()
        wantf[send] += f0[recv]
        ok = ok and np.array_equal(f_t.cpu().numpy(), wantf)
        o = torch.as_tensor(_DevArray(nb.output_slots()[0], (nslots * stride,)), device="cuda:0").cpu().numpy().astype(np.float64)
        o = o.reshape(nslots, stride).sum(0)
        start = np.concatenate([[0], np.cumsum(counts)])
        for k, si in enumerate(sidx):
            if si >= 0:
                seg = f0[recv[start[k]:start[k + 1]]].astype(np.float64)
                ok = ok and np.abs(o[2 + 3 * si:5 + 3 * si] - seg.sum(0)).max() < 1e-6 * np.abs(seg).sum()
                o[2 + 3 * si:5 + 3 * si] = 0
        ok = ok and not o.any()
        # phases (nbnxn_hip_dd_set_phases, the reference's staged pulses): message B forwards 50 of the slots message A
        # fills, in the same exchange; forces come back through B first, then through A
        perm = rng.permutation(p.natoms).astype(np.int32)
        s_a, r_a, s_b_home, r_b = perm[:100], perm[100:200], perm[200:230], perm[300:380]
        s_b = np.concatenate([r_a[:50], s_b_home])
        sh_a, sh_b = np.array([0.25, 0, 0], np.float32), np.array([0, 1.75, 0], np.float32)
        nb.dd_set_messages([(0, 0, 100, 100, 23, sh_a), (0, 0, 80, 80, 27, sh_b)], np.concatenate([s_a, s_b]), np.concatenate([r_a, r_b]))
        nb.dd_set_phases([1, 2])
        nb.upload_xq(p.xq)
        nb.clear_outputs(abi.FORCE_VIRIAL)
        nb.wait_gpu(0)
        nb.dd_move_x()
        torch.cuda.synchronize()
        want = p.xq.copy()
        want[r_a, :3] = p.xq[s_a, :3] + sh_a
        want[r_a, 3] = p.xq[s_a, 3]
        want[r_b, :3] = want[s_b, :3] + sh_b
        want[r_b, 3] = want[s_b, 3]
        ok = ok and np.array_equal(xq_t.cpu().numpy(), want)
        f_t.copy_(torch.from_numpy(f0))
        torch.cuda.synchronize()
        nb.dd_move_f()
        torch.cuda.synchronize()
        wantf = f0.copy()
        wantf[s_b] += f0[r_b]
        wantf[s_a] += wantf[r_a]
        ok = ok and np.array_equal(f_t.cpu().numpy(), wantf)
        o = torch.as_tensor(_DevArray(nb.output_slots()[0], (nslots * stride,)), device="cuda:0").cpu().numpy().astype(np.float64)
        o = o.reshape(nslots, stride).sum(0)
        for si, rows in ((27, f0[r_b]), (23, wantf[r_a])):
            ok = ok and np.abs(o[2 + 3 * si:5 + 3 * si] - rows.astype(np.float64).sum(0)).max() < 1e-6 * np.abs(rows).sum()
            o[2 + 3 * si:5 + 3 * si] = 0
        ok = ok and not o.any()
        for bad in ([2], [1, 1], [2, 1], [1, 2, 3], []):
            try:
                nb.dd_set_phases(bad)
            except g.gpu.HipError:
                refused += 1
        try:
            nb.dd_set_messages([(0, 0, 1, 1, -1, (0, 0, 0))] * 14, np.zeros(14, np.int32), np.zeros(14, np.int32))   # too many
        except g.gpu.HipError:
            refused += 1
        nb.close()
        ret["ok"], ret["refused"] = bool(ok), refused
    finally:
        faulthandler.cancel_dump_traceback_later()
        hang_log.close()
        if os.path.getsize(hang_log.name) == 0:
            os.remove(hang_log.name)


def test_module_rccl_path_single_rank():
    """nbnxn_hip_dd_*: communicator from a unique id, pack -> grouped ncclSend/ncclRecv -> unpack on the module's
    stream, reverse path with the shift-force term, ncclAllReduce of the slots; one rank talking to itself is what a
    one-GPU box can show."""
    with mp.Manager() as m:
        ret = m.dict()
        _watch(mp.spawn(_rccl_direct_worker, args=(_free_port(), ret), nprocs=1, join=False))
        assert ret["ok"] and ret["refused"] == 9
