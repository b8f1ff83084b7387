"""GPU (-m gpu): nbnxn_hip_launch_kernel_noalloc, the one entry point that allocates nothing (include/nbnxn_hip.h),
through the C ABI against the existing entry points (launch_kernel -> launch_cpyback -> wait_gpu) on the same handle
state.  Two calls in a row, the second on a SMALLER system, so that anything the first call left in the library's
static scratch array (the 64 partial-sum slots) would show in the second's energies.

Bar: the kernels are the same, only the order of the fp32 atomics differs between two runs, so forces agree to 2e-6
relative RMS (float), energies and shift forces to 1e-6 relative (slots folded in double both ways); the parity of the
existing entry points against the oracle (1e-5) is tests/test_gpu_parity.py's."""
import numpy as np
import pytest

import gromacs_cosmo_amd as g
from gromacs_cosmo_amd import abi
from util import relrms

pytestmark = pytest.mark.gpu


def _one_call(nb, torch, p, ip, flags):
    g.workload.upload(nb, p, ip)                       # allocating set-up calls come BEFORE the entry point under test
    f_ref, (elj, eel, fsh) = nb.compute(p.xq, flags)   # existing entry points; also prunes the fresh list
    xq = torch.from_numpy(p.xq).cuda()
    f = torch.zeros((p.natoms, 3), dtype=torch.float32, device="cuda")
    e = torch.full((137,), 7.0, dtype=torch.float32, device="cuda")     # assigned, not added to
    torch.cuda.synchronize()
    mem0 = torch.cuda.mem_get_info()[0]
    nb.launch_kernel_noalloc(flags, p.natoms, xq.data_ptr(), f.data_ptr(), e.data_ptr() if flags else None)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == mem0        # nothing was allocated on the device
    f, e = f.cpu().numpy(), e.cpu().numpy()
    assert np.isfinite(f).all() and relrms(f, f_ref) < 2e-6
    if flags:
        assert abs(e[0] - elj) <= 1e-6 * abs(elj) and abs(e[1] - eel) <= 1e-6 * abs(eel)
        assert np.abs(e[2:].reshape(45, 3) - fsh).max() <= 1e-6 * np.abs(fsh).max()
    else:
        assert (e == 7.0).all()
    return f, e


def test_no_alloc_entry_point_matches_the_existing_one_twice_in_a_row():
    import torch
    nb = g.gpu.NbnxnHip(0)
    try:
        ip = g.system.ewald(0.9)
        big = g.workload.prepare(g.system.water_box(5, 1, 1, model="spce"), ip, 300)     # 3,240 atoms
        small = g.workload.prepare(g.system.water_box(2, 1, 1, model="spce"), ip, 100)   # 1,296 atoms
        assert small.natoms < big.natoms and small.pl.nsci < big.pl.nsci
        _, e1 = _one_call(nb, torch, big, ip, abi.FORCE_VIRIAL)
        _, e2 = _one_call(nb, torch, small, ip, abi.FORCE_VIRIAL)       # smaller input right after: nothing stale
        assert abs(e2[1]) < 0.6 * abs(e1[1])
        # a second call on the unchanged state gives the same energies again (the scratch array was cleared)
        xq = torch.from_numpy(small.xq).cuda()
        f = torch.zeros((small.natoms, 3), dtype=torch.float32, device="cuda")
        e = torch.zeros(137, dtype=torch.float32, device="cuda")
        nb.launch_kernel_noalloc(abi.FORCE_VIRIAL, small.natoms, xq.data_ptr(), f.data_ptr(), e.data_ptr())
        torch.cuda.synchronize()
        assert np.abs(e.cpu().numpy()[:2] - e2[:2]).max() <= 1e-6 * np.abs(e2[:2]).max()
        # forces only, reaction field, the exclusion-dense mixture
        ip2 = g.system.reaction_field(0.9)
        mix = g.workload.prepare(g.system.lj_chain_mixture(2000), ip2, 100)
        _one_call(nb, torch, mix, ip2, 0)
        # refused, with the handle left usable: wrong atom count, null arrays, a size beyond the stated maximum
        with pytest.raises(g.gpu.HipError):
            nb.launch_kernel_noalloc(0, mix.natoms - 64, xq.data_ptr(), f.data_ptr())
        with pytest.raises(g.gpu.HipError):
            nb.launch_kernel_noalloc(abi.FORCE_VIRIAL, mix.natoms, xq.data_ptr(), f.data_ptr(), None)
        with pytest.raises(g.gpu.HipError, match="exceed the maximum"):
            nb.launch_kernel_noalloc(0, (1 << 28) + 64, xq.data_ptr(), f.data_ptr())
        _one_call(nb, torch, small, ip, abi.FORCE_VIRIAL)
    finally:
        nb.close()


def test_no_alloc_local_and_nonlocal_back_to_back_on_a_two_stream_handle():
    """Both localities of a two-stream handle, issued back to back so that they overlap on the device (each on its own
    stream, each with its own half of the scratch array): energies and shift forces of the two calls together, and the
    forces both add into ONE array, equal what the existing entry points give for the two lists (compute_both)."""
    import torch
    import torch.distributed as dist
    from gromacs_cosmo_amd import dd
    from test_dd_cpu import _free_port
    nb = g.gpu.NbnxnHip(0, local_and_nonlocal=True)
    made_group = not dist.is_initialized()
    if made_group:      # build_domain gathers the halo bookkeeping through the process group, also for one rank
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1)
    try:
        ip = g.system.ewald(0.9, rlist=1.0)
        glob = g.system.water_box(2, 2, 4, model="spce")
        dom = dd.build_domain(glob, ip, 0, 1, torch.device("cuda", 0), 300)      # one rank that is its own neighbour along z
        nb.init_const(ip, dom.ntype, dom.nbfp)
        nb.init_atomdata(dom.npad, dom.npad_local, dom.ntype, dom.type)
        nb.upload_shiftvec(dom.shift_vec)
        nb.init_pairlist(dom.pl_local, abi.LOCAL)
        nb.init_pairlist(dom.pl_nonlocal, abi.NONLOCAL)
        assert dom.pl_nonlocal.nsci > 0
        fl = abi.FORCE_VIRIAL
        f_ref, (elj, eel, fsh) = nb.compute_both(dom.xq, fl)              # existing entry points; prunes both lists
        xq = torch.from_numpy(dom.xq).cuda()
        for _ in range(2):
            f = torch.zeros((dom.npad, 3), dtype=torch.float32, device="cuda")
            e = torch.full((2, 137), 7.0, dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            mem0 = torch.cuda.mem_get_info()[0]
            nb.launch_kernel_noalloc(fl, dom.npad, xq.data_ptr(), f.data_ptr(), e[0].data_ptr(), iloc=abi.LOCAL)
            nb.launch_kernel_noalloc(fl, dom.npad, xq.data_ptr(), f.data_ptr(), e[1].data_ptr(), iloc=abi.NONLOCAL)
            torch.cuda.synchronize()
            assert torch.cuda.mem_get_info()[0] == mem0
            fh, eh = f.cpu().numpy(), e.cpu().numpy().astype(np.float64).sum(0)
            assert np.isfinite(fh).all() and relrms(fh, f_ref) < 2e-6
            assert abs(eh[0] - elj) <= 2e-6 * abs(elj) and abs(eh[1] - eel) <= 2e-6 * abs(eel)
            assert np.abs(eh[2:].reshape(45, 3) - fsh).max() <= 2e-6 * np.abs(fsh).max()
    finally:
        nb.close()
        if made_group:
            dist.destroy_process_group()
