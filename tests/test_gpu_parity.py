"""GPU (-m gpu): the HIP path, called through the C ABI, against the oracle and the golden
vectors.  Forces within 1e-5 relative RMS, energies and virial within 1e-5 relative
(BASELINE.json north_star); list upload and pruning bit-exact."""
import ctypes as C
import os

import numpy as np
import pytest

import gromacs_cosmo_amd as g
from gromacs_cosmo_amd import abi
import oracle as O
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from util import SMALL, LARGE, COINCIDENT, FORCE_RTOL, ENERGY_RTOL, relrms, load_small, load_large, virial_from

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nb():
    h = g.gpu.NbnxnHip(0)
    yield h
    h.close()


def _upload(nb, p, ip):
    g.workload.upload(nb, p, ip)
    assert nb.verify_pairlist(p.pl) == 0          # bit-exact upload (north_star)


def _check_against(f, e, p, ip, o, d=None):
    elj, eel, fsh = e
    assert relrms(f, o["f"]) < FORCE_RTOL
    assert abs(elj - o["Vvdw"]) < ENERGY_RTOL * abs(o["Vvdw"])
    assert abs(eel - o["Vc"]) < ENERGY_RTOL * abs(o["Vc"])
    fsh_o = o["fshift"].copy()
    fsh_o[abi.CENTRAL] = 0                         # the GPU path skips the central shift (virial-neutral)
    vir = virial_from(p.xq, f, p.shift_vec, fsh)
    vir_o = virial_from(p.xq, o["f"], p.shift_vec, fsh_o)
    assert np.abs(vir - vir_o).max() < ENERGY_RTOL * np.abs(vir_o).max()
    if d is not None:
        assert relrms(f, d["ref_f"]) < FORCE_RTOL          # the reference's nbnxn_kernel_ref
        assert relrms(f, d["gpuref_f"]) < FORCE_RTOL       # the reference's nbnxn_kernel_gpu_ref
        assert abs(eel - d["gpuref_Vc"]) < 2e-5 * abs(d["gpuref_Vc"])
        assert abs(elj - d["gpuref_Vvdw"]) < 2e-5 * abs(d["gpuref_Vvdw"])
        # the contract oracle's energies: nbnxn_kernel_ref sums each list in one float and is itself up to 2.1e-5
        # from the double oracle on these <= 3,840-atom fixtures (tests/test_oracle_golden.py)
        assert abs(eel - d["ref_Vc"]) < 3e-5 * abs(d["ref_Vc"])
        assert abs(elj - d["ref_Vvdw"]) < 3e-5 * abs(d["ref_Vvdw"])


@pytest.mark.parametrize("name", SMALL)
def test_parity_small_fixtures(nb, name):
    p, ip, d = load_small(name)
    _upload(nb, p, ip)
    o = O.kernel_gpu_list(p.xq, p.type, p.ntype, p.nbfp, p.shift_vec, p.pl, ip)
    # 1st call after init_pairlist: energy + prune flavour
    f, e = nb.compute(p.xq, abi.FORCE_VIRIAL)
    _check_against(f, e, p, ip, o, d)
    # pruning is bit-exact against the CPU rule and leaves the physics alone
    cj4 = nb.download_cj4(p.pl)
    assert np.array_equal(cj4["imei"]["imask"], O.prune_list(p.xq, p.shift_vec, p.pl, ip.rlist)["imei"]["imask"])
    # force-only flavour on the pruned list
    f2, e2 = nb.compute(p.xq, 0)
    assert relrms(f2, o["f"]) < FORCE_RTOL
    assert e2[0] == 0 and e2[1] == 0 and not e2[2].any()   # no energies/fshift without GMX_FORCE_VIRIAL
    # energy flavour without pruning
    f3, e3 = nb.compute(p.xq, abi.FORCE_VIRIAL)
    _check_against(f3, e3, p, ip, o, d)
    # force-only + prune flavour
    nb.init_pairlist(p.pl)
    f4, _ = nb.compute(p.xq, 0)
    assert relrms(f4, o["f"]) < FORCE_RTOL
    assert np.array_equal(nb.download_cj4(p.pl)["imei"]["imask"], cj4["imei"]["imask"])


@pytest.mark.parametrize("name", LARGE)
def test_parity_97k_against_reference_samples(nb, name):
    system, ip, d = load_large(name)
    p = g.workload.prepare(system, ip, nb.min_ci_balanced())
    _upload(nb, p, ip)
    f, (elj, eel, fsh) = nb.compute(p.xq, abi.FORCE_VIRIAL)
    stride = int(d["stride"])
    assert relrms(f[::stride], d["ref_f_sample"]) < FORCE_RTOL
    assert relrms(f[::stride], d["gpuref_f_sample"]) < FORCE_RTOL
    assert abs(np.sqrt((f.astype(np.float64) ** 2).sum()) - float(d["ref_f_norm"])) < 1e-5 * float(d["ref_f_norm"])
    assert abs(eel - d["gpuref_Vc"]) < 2e-5 * abs(d["gpuref_Vc"])
    assert abs(elj - d["gpuref_Vvdw"]) < 2e-5 * abs(d["gpuref_Vvdw"])
    # nbnxn_kernel_ref's own one-float energy sums are 8e-5 from the double oracle at this size
    assert abs(eel - d["ref_Vc"]) < 2e-4 * abs(d["ref_Vc"])
    assert abs(elj - d["ref_Vvdw"]) < 2e-4 * abs(d["ref_Vvdw"])
    # full-size check against the double-precision oracle (a few seconds on the host)
    o = O.kernel_gpu_list(p.xq, p.type, p.ntype, p.nbfp, p.shift_vec, p.pl, ip)
    _check_against(f, (elj, eel, fsh), p, ip, o)
    # size-independent properties: Newton's third law and idempotence of repeated steps
    assert np.abs(f.astype(np.float64).sum(axis=0)).max() < 1e-3 * np.abs(f).max()
    f2, _ = nb.compute(p.xq, 0)
    assert relrms(f2, f) < 2e-6                     # only the fp32 atomic order differs


def test_struct_entry_points_take_reference_layout_objects(nb):
    """The struct-taking calls (what sim_util.c / forcerec.c would call) with ctypes mirrors of
    nbnxn_atomdata_t, nbnxn_pairlist_t, interaction_const_t, nonbonded_verlet_group_t."""
    s = g.system.water_box(5, 1, 1)
    ip = g.system.ewald(0.9)
    p = g.workload.prepare(s, ip)
    o = O.kernel_gpu_list(p.xq, p.type, p.ntype, p.nbfp, p.shift_vec, p.pl, ip)
    f = np.zeros((p.natoms, 3), dtype=np.float32)
    nbfp = np.ascontiguousarray(p.nbfp, dtype=np.float32)
    out = abi.AtomdataOutput()
    out.f = abi.ptr(f)
    nbat = abi.Atomdata()
    nbat.ntype = p.ntype
    nbat.nbfp = abi.ptr(nbfp)
    nbat.natoms = p.natoms
    nbat.natoms_local = p.natoms
    nbat.type = abi.ptr(p.type)
    nbat.XFormat, nbat.FFormat = 1, 0           # nbatXYZQ, nbatXYZ
    nbat.na_c = 8
    nbat.nenergrp = 1
    nbat.bDynamicBox = 1           # the shared handle has seen other boxes: force the shift-vector upload
    nbat.shift_vec = abi.ptr(p.shift_vec)
    nbat.xstride, nbat.fstride = 4, 3
    nbat.x = abi.ptr(p.xq)
    nbat.nout = 1
    nbat.out = C.pointer(out)
    grp = abi.VerletGroup()
    grp.nbat = C.pointer(nbat)
    ic = ip.to_struct()
    lib = nb.lib
    chk = nb._chk
    chk(lib.nbnxn_hip_init_const(nb.h, C.byref(ic), C.byref(grp)))
    chk(lib.nbnxn_hip_init_atomdata(nb.h, C.byref(nbat)))
    nb.natoms = p.natoms
    chk(lib.nbnxn_hip_upload_shiftvec(nb.h, C.byref(nbat)))
    chk(lib.nbnxn_hip_init_pairlist(nb.h, p.host.pairlist_struct(abi.LOCAL), abi.LOCAL))
    assert nb.is_kernel_ewald_analytical()
    chk(lib.nbnxn_hip_clear_outputs(nb.h, abi.FORCE_VIRIAL))
    chk(lib.nbnxn_hip_launch_kernel(nb.h, C.byref(nbat), abi.FORCE_VIRIAL, abi.LOCAL))
    chk(lib.nbnxn_hip_launch_cpyback(nb.h, C.byref(nbat), abi.FORCE_VIRIAL, abi.LOCAL))
    e_lj, e_el = np.full(1, 10.0, np.float32), np.full(1, -20.0, np.float32)
    fsh = np.zeros((45, 3), dtype=np.float32)
    chk(lib.nbnxn_hip_wait_gpu(nb.h, C.byref(nbat), abi.FORCE_VIRIAL, abi.LOCAL, abi.ptr(e_lj), abi.ptr(e_el), abi.ptr(fsh)))
    assert relrms(f, o["f"]) < FORCE_RTOL
    # wait_gpu ADDS into the caller's accumulators (nbnxn_cuda.cu:662-680)
    assert abs((e_lj[0] - 10.0) - o["Vvdw"]) < 2e-5 * abs(o["Vvdw"])
    assert abs((e_el[0] + 20.0) - o["Vc"]) < 2e-5 * abs(o["Vc"])
    t = nb.timings()
    assert t is not None and t.nb_c >= 1 and t.ktime[1][1].c >= 1 and t.ktime[1][1].t > 0


def test_resident_path_matches_host_path(nb):
    p, ip, d = load_small("spc3240_rf09")
    _upload(nb, p, ip)
    f, e = nb.compute(p.xq, abi.FORCE_VIRIAL)       # uploads xq, prunes
    nb.clear_outputs(abi.FORCE_VIRIAL)
    nb.launch_kernel_resident(abi.FORCE_VIRIAL)     # coordinates already in HBM
    f2 = np.zeros_like(f)
    nb.launch_cpyback(f2, abi.FORCE_VIRIAL)
    nb.wait_gpu(abi.FORCE_VIRIAL)
    elj, eel, fsh = nb.read_outputs()
    assert relrms(f2, f) < 2e-6
    assert abs(elj - e[0]) < 1e-6 * abs(e[0]) and abs(eel - e[1]) < 1e-6 * abs(e[1])
    assert nb.last_kernel_ms() > 0


def test_empty_and_ragged_inputs(nb):
    p, ip, d = load_small("spc216_rf09")
    _upload(nb, p, ip)
    # an empty list is a no-op (nbnxn_cuda.cu:305-309)
    from gromacs_cosmo_amd.host import PairList
    empty = PairList(p.pl.sci[:0], p.pl.cj4[:0], p.pl.excl[:1], 0, ip.rlist)
    nb.init_pairlist(empty)
    f, e = nb.compute(p.xq, abi.FORCE_VIRIAL)
    assert not f.any() and e[0] == 0 and e[1] == 0
    # a list whose every interaction mask is zero: entries are walked, nothing is computed
    cj4 = p.pl.cj4.copy()
    cj4["imei"]["imask"][:] = 0
    nb.init_pairlist(PairList(p.pl.sci, cj4, p.pl.excl, 0, ip.rlist))
    f, e = nb.compute(p.xq, 0)
    assert not f.any()
    # one entry only (ragged tail of the grid: the last super-cluster is mostly fillers)
    last = p.pl.sci[np.argmax(p.pl.sci["sci"])][None].copy()
    one = PairList(last, p.pl.cj4, p.pl.excl, 0, ip.rlist)
    nb.init_pairlist(one)
    f, e = nb.compute(p.xq, abi.FORCE_VIRIAL)
    o = O.kernel_gpu_list(p.xq, p.type, p.ntype, p.nbfp, p.shift_vec, one, ip)
    assert relrms(f, o["f"]) < FORCE_RTOL


def test_bad_inputs_fail_loudly_before_any_launch(nb):
    p, ip, d = load_small("spc216_rf09")
    _upload(nb, p, ip)
    bad = p.pl.cj4.copy()
    bad["cj"][0, 0] = 10 ** 6                      # j-cluster outside the atom arrays
    from gromacs_cosmo_amd.host import PairList
    with pytest.raises(g.gpu.HipError):
        nb.init_pairlist(PairList(p.pl.sci, bad, p.pl.excl, p.pl.nci_tot, ip.rlist))
    bad = p.pl.cj4.copy()
    bad["imei"]["excl_ind"][0, 0] = p.pl.nexcl     # exclusion index out of range
    with pytest.raises(g.gpu.HipError):
        nb.init_pairlist(PairList(p.pl.sci, bad, p.pl.excl, p.pl.nci_tot, ip.rlist))
    with pytest.raises(g.gpu.HipError):
        nb.init_atomdata(100, 100, p.ntype, p.type[:100])   # not a multiple of 64
    ip2 = g.system.reaction_field(0.9)
    ip2.rvdw = 0.8
    with pytest.raises(g.gpu.HipError):
        nb.init_const(ip2, p.ntype, p.nbfp)        # rvdw != rcoulomb exists for Ewald only
    with pytest.raises(g.gpu.HipError):
        nb.wait_gpu(0, aloc=2)                     # eatAll is not valid here (nbnxn_cuda.cu:568-574)


def test_many_atom_types_use_the_global_lj_table(nb):
    s = g.system.lj_chain_mixture(1200, ntype=20, seed=3)   # 21 types > 16: table stays in L2
    ip = g.system.reaction_field(0.8, epsilon_rf=0.0)
    p = g.workload.prepare(s, ip)
    _upload(nb, p, ip)
    o = O.kernel_gpu_list(p.xq, p.type, p.ntype, p.nbfp, p.shift_vec, p.pl, ip)
    f, e = nb.compute(p.xq, abi.FORCE_VIRIAL)
    assert relrms(f, o["f"]) < FORCE_RTOL
    assert abs(e[0] - o["Vvdw"]) < ENERGY_RTOL * abs(o["Vvdw"])
    assert abs(e[1] - o["Vc"]) < ENERGY_RTOL * abs(o["Vc"])


def test_clear_in_kernel_matches_fill(nb):
    """nbnxn_hip_set_clear_in_kernel: ping-pong output buffers zeroed by the previous launch give
    the same forces and energies as a dispatched fill, step after step, also when a force-only
    step sits between two energy steps."""
    p, ip, d = load_small("spce3240_ewald09")
    _upload(nb, p, ip)
    f_ref, e_ref = nb.compute(p.xq, abi.FORCE_VIRIAL)
    nb.set_clear_in_kernel(True)
    try:
        for flags in (abi.FORCE_VIRIAL, 0, abi.FORCE_VIRIAL, abi.FORCE_VIRIAL):
            nb.clear_outputs(flags)
            nb.launch_kernel_resident(flags)
            f = np.zeros_like(f_ref)
            nb.launch_cpyback(f, flags)
            nb.wait_gpu(flags)
            assert relrms(f, f_ref) < 2e-6
            if flags:
                elj, eel, fsh = nb.read_outputs()
                assert abs(elj - e_ref[0]) < 1e-6 * abs(e_ref[0]) and abs(eel - e_ref[1]) < 1e-6 * abs(e_ref[1])
                assert np.abs(fsh - e_ref[2]).max() < 1e-5 * np.abs(e_ref[2]).max()
    finally:
        nb.set_clear_in_kernel(False)
    f2, e2 = nb.compute(p.xq, abi.FORCE_VIRIAL)
    assert relrms(f2, f_ref) < 2e-6 and abs(e2[1] - e_ref[1]) < 1e-6 * abs(e_ref[1])


def test_copy_kernels_move_the_same_bytes_as_the_copy_engine(nb):
    """nbnxn_hip_set_copy_kernels: the step of the reference's calling convention (host xq in at launch_kernel, f and the
    partial sums out at launch_cpyback, wait_gpu) with pinned + mapped host arrays moved by kernels, against the same step
    through hipMemcpyAsync: forces bit for bit, energies and shift forces equal (both fold the same slots in double) --
    with the in-kernel clear on and off, with pageable arrays and with a pinned array that is not 16-byte aligned (both
    fall back to the copy engine), and the coordinates really come from the host array of THAT step."""
    import ctypes as C
    p, ip, d = load_small("spce3240_ewald09")
    _upload(nb, p, ip)
    held = []

    def pinned(nfloat):
        ptr = C.c_void_p()
        nb.lib.nbnxn_hip_pmalloc(C.byref(ptr), 4 * nfloat)
        assert ptr.value
        held.append(ptr)
        return np.frombuffer((C.c_char * (4 * nfloat)).from_address(ptr.value), dtype=np.float32)

    def step(xq, f):
        e_lj, e_el, fsh = np.zeros(1, np.float32), np.zeros(1, np.float32), np.zeros((abi.SHIFTS, 3), np.float32)
        f[:] = 7.0
        nb.clear_outputs(abi.FORCE_VIRIAL)
        nb.launch_kernel(xq, abi.FORCE_VIRIAL)
        nb.launch_cpyback(f, abi.FORCE_VIRIAL)
        nb.wait_gpu(abi.FORCE_VIRIAL, abi.LOCAL, e_lj, e_el, fsh)
        return f.copy(), float(e_lj[0]), float(e_el[0]), fsh.copy()

    n = p.natoms
    xq_pin = pinned(4 * n).reshape(n, 4)
    f_pin = pinned(3 * n).reshape(n, 3)
    f_odd = pinned(3 * n + 4)[1:1 + 3 * n].reshape(n, 3)        # pinned, 4 bytes off a 16-byte boundary
    xq_pag, f_pag = np.empty((n, 4), np.float32), np.empty((n, 3), np.float32)
    rng = np.random.default_rng(5)
    try:
        for clear_in_kernel in (False, True):
            nb.set_clear_in_kernel(clear_in_kernel)
            for rep in range(2):
                xq = p.xq.copy()
                xq[:, :3] += rng.normal(0.0, 0.002, (n, 3)).astype(np.float32) * (rep > 0)     # a different x every step
                xq_pin[:] = xq; xq_pag[:] = xq
                nb.set_copy_kernels(False)
                want = step(xq_pin, f_pin)
                nb.set_copy_kernels(True)
                for xq_h, f_h in ((xq_pin, f_pin), (xq_pag, f_pag), (xq_pin, f_odd), (xq_pag, f_pin)):
                    got = step(xq_h, f_h)
                    # the force atomics of a launch commute only up to rounding: launches of the same step differ in
                    # the last bits, copies of ONE launch do not -- so compare at rounding level here ...
                    assert relrms(got[0], want[0]) < 2e-6
                    assert abs(got[1] - want[1]) < 1e-6 * abs(want[1]) and abs(got[2] - want[2]) < 1e-6 * abs(want[2])
                    assert np.abs(got[3] - want[3]).max() < 1e-5 * np.abs(want[3]).max()
                # ... and bit for bit on what one launch left in device memory: the kernel copy against the module's own
                # device-to-host read of the same buffer
                # (and the sums of the partial-sum slots made by the copy-back kernel against the host's sums of the same slots)
                ea, eb = [np.zeros(1, np.float32) for _ in range(2)], [np.zeros(1, np.float32) for _ in range(2)]
                fa, fb = np.zeros((abi.SHIFTS, 3), np.float32), np.zeros((abi.SHIFTS, 3), np.float32)
                nb.clear_outputs(abi.FORCE_VIRIAL)
                nb.launch_kernel(xq_pin, abi.FORCE_VIRIAL)
                nb.launch_cpyback(f_pin, abi.FORCE_VIRIAL)
                nb.wait_gpu(abi.FORCE_VIRIAL, abi.LOCAL, ea[0], ea[1], fa)
                nb.set_copy_kernels(False)
                nb.launch_cpyback(f_pag, abi.FORCE_VIRIAL)
                nb.wait_gpu(abi.FORCE_VIRIAL, abi.LOCAL, eb[0], eb[1], fb)
                nb.set_copy_kernels(True)
                assert np.array_equal(f_pin, f_pag)
                assert ea[0][0] == eb[0][0] and ea[1][0] == eb[1][0] and np.array_equal(fa, fb) and ea[1][0] != 0.0
        f_ref, e_ref = nb.compute(xq, abi.FORCE_VIRIAL)
        assert relrms(got[0], f_ref) < 2e-6 and abs(got[2] - e_ref[1]) < 1e-6 * abs(e_ref[1])
    finally:
        nb.set_clear_in_kernel(False)
        nb.set_copy_kernels(True)
        del xq_pin, f_pin, f_odd
        for ptr in held:
            nb.lib.nbnxn_hip_pfree(ptr)


def test_two_locality_sequence_with_host_arrays_on_a_two_stream_handle():
    """The reference's step with domain decomposition (sim_util.c:1007-1147, 1385-1428): launch_kernel LOCAL then NONLOCAL with the
    host xq (each copies its own atom range in), launch_cpyback NONLOCAL then LOCAL (each copies its own range of f out; the
    local one waits for the non-local kernel, nbnxn_cuda.cu:483-503), wait NONLOCAL, wait LOCAL -- on a two-stream handle,
    with pinned arrays (copy kernels, the local copy-back folding the slots) and with pageable ones (copy engine).  The
    entries of one list split into those among the first half of the atoms ("local") and the rest: the two launches together
    must give the whole list's forces and energies; each copy must touch its own range only."""
    import ctypes as C
    import torch
    from gromacs_cosmo_amd.dd import _DevArray
    from gromacs_cosmo_amd.host import PairList
    p, ip, d = load_small("spce3240_ewald09")
    one = g.gpu.NbnxnHip(0)
    _upload(one, p, ip)
    f_ref, e_ref = one.compute(p.xq, abi.FORCE_VIRIAL)
    one.close()
    nb = g.gpu.NbnxnHip(0, local_and_nonlocal=True)
    held = []

    def pinned(nfloat):
        ptr = C.c_void_p()
        nb.lib.nbnxn_hip_pmalloc(C.byref(ptr), 4 * nfloat)
        assert ptr.value
        held.append(ptr)
        return np.frombuffer((C.c_char * (4 * nfloat)).from_address(ptr.value), dtype=np.float32)

    n = p.natoms
    nl = (n // 2) // 64 * 64
    try:
        nb.init_const(ip, p.ntype, p.nbfp)
        nb.init_atomdata(n, nl, p.ntype, p.type)
        nb.upload_shiftvec(p.shift_vec)
        # local list: the entries that touch atoms below nl only (what a local list is: the non-local copy-back does not wait
        # for the local kernel); everything else is "non-local" work, which may add to local atoms as well
        im = (p.pl.cj4["imei"]["imask"][:, 0] | p.pl.cj4["imei"]["imask"][:, 1]).astype(np.uint32)
        live = np.stack([((im >> (8 * k)) & 0xff) != 0 for k in range(4)], axis=1)
        gmax = np.where(live, p.pl.cj4["cj"], -1).max(axis=1)
        local = np.array([e["sci"] * 64 + 64 <= nl and gmax[e["cj4_ind_start"]:e["cj4_ind_end"]].max() * 8 + 8 <= nl for e in p.pl.sci])
        assert 0.1 * p.pl.nsci < local.sum() < 0.9 * p.pl.nsci
        nb.init_pairlist(PairList(np.ascontiguousarray(p.pl.sci[local]), p.pl.cj4, p.pl.excl, 0, ip.rlist), abi.LOCAL)
        nb.init_pairlist(PairList(np.ascontiguousarray(p.pl.sci[~local]), p.pl.cj4, p.pl.excl, 0, ip.rlist), abi.NONLOCAL)
        for kind in ("pinned", "pageable"):
            xq_h = pinned(4 * n).reshape(n, 4) if kind == "pinned" else np.empty((n, 4), np.float32)
            f_h = pinned(3 * n).reshape(n, 3) if kind == "pinned" else np.empty((n, 3), np.float32)
            xq_h[:] = p.xq
            # the coordinate copies: a device array of zeros, then the two launches (the forces of THIS pass are meaningless)
            nb.upload_xq(np.zeros((n, 4), np.float32), abi.LOCAL)
            nb.upload_xq(np.zeros((n, 4), np.float32), abi.NONLOCAL)
            d_xq, _, _ = nb.device_buffers()
            xq_t = torch.as_tensor(_DevArray(d_xq, (n, 4)), device="cuda:0")
            nb.clear_outputs(0)
            nb.launch_kernel(xq_h, 0, abi.LOCAL)
            torch.cuda.synchronize()
            got = xq_t.cpu().numpy()
            assert np.array_equal(got[:nl], p.xq[:nl]) and not got[nl:].any()
            nb.launch_kernel(xq_h, 0, abi.NONLOCAL)
            torch.cuda.synchronize()
            assert np.array_equal(xq_t.cpu().numpy(), p.xq)
            nb.wait_gpu(0, abi.NONLOCAL); nb.wait_gpu(0, abi.LOCAL)
            # the step
            for rep in range(2):
                e_lj, e_el, fsh = np.zeros(1, np.float32), np.zeros(1, np.float32), np.zeros((abi.SHIFTS, 3), np.float32)
                f_h[:] = 7.0
                nb.clear_outputs(abi.FORCE_VIRIAL)
                nb.launch_kernel(xq_h, abi.FORCE_VIRIAL, abi.LOCAL)
                nb.launch_kernel(xq_h, abi.FORCE_VIRIAL, abi.NONLOCAL)
                nb.launch_cpyback(f_h, abi.FORCE_VIRIAL, abi.NONLOCAL)
                nb.wait_gpu(abi.FORCE_VIRIAL, abi.NONLOCAL, e_lj, e_el, fsh)
                assert (f_h[:nl] == 7.0).all() and not (f_h[nl:] == 7.0).all()      # the non-local range only, so far
                nb.launch_cpyback(f_h, abi.FORCE_VIRIAL, abi.LOCAL)
                nb.wait_gpu(abi.FORCE_VIRIAL, abi.LOCAL, e_lj, e_el, fsh)
                assert relrms(f_h, f_ref) < 2e-6
                assert abs(e_lj[0] - e_ref[0]) < 1e-6 * abs(e_ref[0]) and abs(e_el[0] - e_ref[1]) < 1e-6 * abs(e_ref[1])
                assert np.abs(fsh - e_ref[2]).max() < 1e-5 * np.abs(e_ref[2]).max()
            del xq_h, f_h
    finally:
        nb.close()


@pytest.mark.parametrize("model", ["spc", "spce"])
def test_config4_size_properties(nb, model):
    """BASELINE config 4 size (13x13x14 tiles, 1,533,168 atoms, >1e9 pair slots) on one GPU.  The
    full oracle cannot run that in seconds, so: (1) a sample of atoms against brute force over ALL
    atoms in double; (2) the box is a periodic tiling of spc216, so the energies are 13*13*14/8
    times those the oracle finds for the 2x2x2 tiling (forces are not compared tile-to-tile: the
    float32 rounding of coordinates near 26 nm moves them by ~2e-5); (3) Newton's third law."""
    from util import brute_force_sample
    ip = g.system.reaction_field(1.0) if model == "spc" else g.system.ewald(1.0)
    small = g.system.water_box(2, 2, 2, model=model)
    ps = g.workload.prepare(small, ip, 0)
    o = O.kernel_gpu_list(ps.xq, ps.type, ps.ntype, ps.nbfp, ps.shift_vec, ps.pl, ip)
    tiles = (13, 13, 14)
    ntile = tiles[0] * tiles[1] * tiles[2]
    big = g.system.water_box(*tiles, model=model)
    p = g.workload.prepare(big, ip, nb.min_ci_balanced())
    assert p.pl.nci_tot * 64 > 1.0e9
    _upload(nb, p, ip)
    assert nb.verify_pairlist(p.pl) == 0
    fnb, (elj, eel, fsh) = nb.compute(p.xq, abi.FORCE_VIRIAL)
    f = np.zeros((big.natoms, 3), dtype=np.float32)
    p.host.add_f(fnb, f)
    idx = np.random.default_rng(4).choice(big.natoms, 96, replace=False)
    assert relrms(f[idx], brute_force_sample(big, ip, idx)) < FORCE_RTOL
    assert abs(elj - o["Vvdw"] * ntile / 8) < ENERGY_RTOL * abs(o["Vvdw"] * ntile / 8)
    assert abs(eel - o["Vc"] * ntile / 8) < ENERGY_RTOL * abs(o["Vc"] * ntile / 8)
    assert np.abs(f.astype(np.float64).sum(axis=0)).max() < 1e-3 * np.abs(f).max()


def test_config5_size_properties(nb):
    """BASELINE config 5 size on one GPU: 3,000,000 atoms of the 8-type, exclusion-dense chain
    mixture (no reference fixture exists for it), analytical Ewald.  A sample of atoms
    against brute force over all atoms in double, Newton's third law, and the virial from the
    kernel's shift forces against the pair virial of the sample's forces is left to the small
    fixtures; here the point is size: 3.0e6 atoms, 1.7e9 pair slots, 5.9e5 exclusion words."""
    from util import brute_force_sample
    ip = g.system.ewald(1.0)
    big = g.system.lj_chain_mixture(3000000)
    p = g.workload.prepare(big, ip, nb.min_ci_balanced())
    assert p.pl.nci_tot * 64 > 1.5e9
    _upload(nb, p, ip)
    assert nb.verify_pairlist(p.pl) == 0
    fnb, (elj, eel, fsh) = nb.compute(p.xq, abi.FORCE_VIRIAL)
    f = np.zeros((big.natoms, 3), dtype=np.float32)
    p.host.add_f(fnb, f)
    idx = np.random.default_rng(5).choice(big.natoms, 64, replace=False)
    assert relrms(f[idx], brute_force_sample(big, ip, idx)) < FORCE_RTOL
    assert np.isfinite(elj) and np.isfinite(eel)
    assert np.abs(f.astype(np.float64).sum(axis=0)).max() < 1e-3 * np.abs(f).max()
    # the pruned list gives the same forces on the next step
    f2, _ = nb.compute(p.xq, 0)
    assert relrms(f2, fnb) < 2e-6


def test_device_atomdata_moves_match_host_library(nb):
    """nbnxn_atomdata_copy_x_to_nbat_x / add_nbat_f_to_f on the device against the
    host library's versions, bit for bit (they are permutations plus one add), for host and
    device-resident x / f, on the 97,200-atom box."""
    import torch
    from gromacs_cosmo_amd.dd import _DevArray
    s = g.system.water_box(5, 5, 6, model="spce")
    ip = g.system.ewald(1.0)
    p = g.workload.prepare(s, ip, nb.min_ci_balanced())
    _upload(nb, p, ip)
    nb.set_atom_order(s.natoms, p.host.atom_order(), p.host.cell_index())
    rng = np.random.default_rng(11)
    x2 = (s.x + rng.normal(0, 0.002, s.x.shape)).astype(np.float32)
    xq_ref = p.xq.copy()
    p.host.copy_x_to_xq(x2, xq_ref)
    d_xq, d_f, _ = nb.device_buffers()
    xq_t = torch.as_tensor(_DevArray(d_xq, (p.natoms, 4)), device="cuda:0")
    # host x
    nb.upload_xq(p.xq)
    nb.copy_x_to_xq(x2)
    nb.wait_gpu(0)
    torch.cuda.synchronize()
    assert np.array_equal(xq_t.cpu().numpy(), xq_ref)
    # device x
    nb.upload_xq(p.xq)
    x_t = torch.from_numpy(x2).cuda()
    torch.cuda.synchronize()
    nb.copy_x_to_xq(x_t.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(xq_t.cpu().numpy(), xq_ref)
    # a step on the moved coordinates, then forces back into caller order three ways
    nb.clear_outputs(0)
    nb.launch_kernel_resident(0)
    fnb = np.zeros((p.natoms, 3), dtype=np.float32)
    nb.launch_cpyback(fnb, 0)
    nb.wait_gpu(0)
    f0 = rng.normal(0, 100, s.x.shape).astype(np.float32)
    f_ref = f0.copy()
    p.host.add_f(fnb, f_ref)
    f_host = f0.copy()
    nb.add_f(f_host)
    assert np.array_equal(f_host, f_ref)
    f_t = torch.from_numpy(f0).cuda()
    torch.cuda.synchronize()
    nb.add_f(f_t.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(f_t.cpu().numpy(), f_ref)
    # a sub-range, as the reference calls it per locality
    f_part = f0.copy()
    nb.add_f(f_part, a0=1000, n=5000)
    assert np.array_equal(f_part[1000:6000], f_ref[1000:6000]) and np.array_equal(f_part[:1000], f0[:1000])
    # and the forces are those of the oracle on the moved coordinates
    o = O.kernel_gpu_list(xq_ref, p.type, p.ntype, p.nbfp, p.shift_vec, p.pl, ip)
    assert relrms(fnb, o["f"]) < FORCE_RTOL
    with pytest.raises(g.gpu.HipError):
        nb.add_f(f_part, a0=s.natoms - 10, n=20)


def test_halo_moves_on_the_device(nb):
    """the pack / unpack kernels of the halo exchange (dd_move_x / dd_move_f,
    domdec.c:651-875) against the same moves in numpy: copies and single float adds, so bit for bit;
    the shift-force sum of a wrapped pulse within float rounding of its terms."""
    import torch
    from gromacs_cosmo_amd.dd import _DevArray
    s = g.system.water_box(2, 2, 2, model="spc")
    ip = g.system.reaction_field(0.9)
    p = g.workload.prepare(s, ip, 0)
    _upload(nb, p, ip)
    nb.upload_xq(p.xq)
    nb.clear_outputs(abi.FORCE_VIRIAL)
    nb.wait_gpu(0)
    d_xq, d_f, _ = nb.device_buffers()
    d_out, nslots, stride = nb.output_slots()
    xq_t = torch.as_tensor(_DevArray(d_xq, (p.natoms, 4)), device="cuda:0")
    f_t = torch.as_tensor(_DevArray(d_f, (p.natoms, 3)), device="cuda:0")
    out_t = torch.as_tensor(_DevArray(d_out, (nslots * stride,)), device="cuda:0")
    rng = np.random.default_rng(5)
    n = 777
    slots = rng.permutation(p.natoms)[:n].astype(np.int32)
    slots_t = torch.from_numpy(slots).cuda()
    shift = np.array([0.5, -1.0, 2.25], dtype=np.float32)
    for iloc in (abi.LOCAL, abi.NONLOCAL):
        buf = torch.zeros((n, 4), device="cuda:0")
        nb.halo_pack_x(slots_t.data_ptr(), n, shift, buf.data_ptr(), iloc)
        torch.cuda.synchronize()
        want = p.xq[slots].copy()
        want[:, :3] += shift
        assert np.array_equal(buf.cpu().numpy(), want)
        nb.halo_pack_x(slots_t.data_ptr(), n, None, buf.data_ptr(), iloc)
        torch.cuda.synchronize()
        assert np.array_equal(buf.cpu().numpy(), p.xq[slots])
    new = rng.normal(0, 1, (n, 4)).astype(np.float32)
    nb.halo_unpack_x(slots_t.data_ptr(), n, torch.from_numpy(new).cuda().data_ptr())
    torch.cuda.synchronize()
    want = p.xq.copy()
    want[slots] = new
    assert np.array_equal(xq_t.cpu().numpy(), want)
    f0 = rng.normal(0, 300, (p.natoms, 3)).astype(np.float32)
    f_t.copy_(torch.from_numpy(f0))
    fbuf = torch.zeros((n, 3), device="cuda:0")
    nb.halo_pack_f(slots_t.data_ptr(), n, fbuf.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(fbuf.cpu().numpy(), f0[slots])
    back = rng.normal(0, 300, (n, 3)).astype(np.float32)
    back_t = torch.from_numpy(back).cuda()
    nb.halo_unpack_f(slots_t.data_ptr(), n, back_t.data_ptr(), -1)
    torch.cuda.synchronize()
    want = f0.copy()
    want[slots] += back
    assert np.array_equal(f_t.cpu().numpy(), want)
    assert float(out_t.abs().max()) == 0.0
    nb.halo_unpack_f(slots_t.data_ptr(), n, back_t.data_ptr(), 17)
    torch.cuda.synchronize()
    want[slots] += back
    assert np.array_equal(f_t.cpu().numpy(), want)
    o = out_t.cpu().numpy().astype(np.float64).reshape(abi.OUT_SLOTS, abi.OUT_STRIDE).sum(0)      # any of the accumulation slots
    got = o[2 + 3 * 17:5 + 3 * 17]
    assert np.abs(got - back.astype(np.float64).sum(0)).max() < 1e-6 * np.abs(back).sum()
    o[2 + 3 * 17:5 + 3 * 17] = 0
    assert not o.any()
    nb.halo_unpack_f(slots_t.data_ptr(), 0, 0, 17)        # an empty halo is fine
    nb.stream_wait(abi.LOCAL, abi.NONLOCAL)
    with pytest.raises(g.gpu.HipError):
        nb.halo_pack_f(slots_t.data_ptr(), p.natoms + 1, fbuf.data_ptr())
    with pytest.raises(g.gpu.HipError):
        nb.halo_unpack_f(slots_t.data_ptr(), n, back_t.data_ptr(), 45)
    with pytest.raises(g.gpu.HipError):
        nb.halo_pack_x(0, n, None, buf.data_ptr())


def test_twin_range_flavour_and_pme_loadbal_update(nb):
    """rvdw < rcoulomb (VDW_CUTOFF_CHECK flavours, nbnxn_cuda_kernel.cuh:405-441) against the oracle,
    reached both ways the reference reaches it: init_const with a twin-range interaction_const_t,
    and nbnxn_cuda_pme_loadbal_update_param scaling rcoulomb and beta under a fixed rvdw
    (pme_loadbal.c:704)."""
    import dataclasses
    import math
    s = g.system.water_box(2, 2, 2, model="spce")
    base = g.system.ewald(0.9, rlist=1.2)
    p = g.workload.prepare(s, base, 300)

    def twin(rc):
        beta = g.system.calc_ewaldcoeff_q(rc, 1e-5)
        return dataclasses.replace(base, rcoulomb=rc, ewaldcoeff_q=beta, sh_ewald=math.erfc(beta * rc))

    def check(ip, f, e):
        o = O.kernel_gpu_list(p.xq, p.type, p.ntype, p.nbfp, p.shift_vec, p.pl, ip)
        assert relrms(f, o["f"]) < FORCE_RTOL
        assert abs(e[0] - o["Vvdw"]) < ENERGY_RTOL * abs(o["Vvdw"]) and abs(e[1] - o["Vc"]) < ENERGY_RTOL * abs(o["Vc"])
        return o

    # equal cut-offs first, then the load balancer moves rcoulomb to 1.1 and on to 1.2 (= rlist)
    _upload(nb, p, base)
    o0 = check(base, *nb.compute(p.xq, abi.FORCE_VIRIAL))
    for rc in (1.1, 1.2):
        ip = twin(rc)
        nb.pme_loadbal_update_param(ip.to_struct())
        o1 = check(ip, *nb.compute(p.xq, abi.FORCE_VIRIAL))
        f_only, _ = nb.compute(p.xq, 0)
        assert relrms(f_only, o1["f"]) < FORCE_RTOL
        assert abs(o1["Vvdw"] - o0["Vvdw"]) < 1e-9 * abs(o0["Vvdw"])     # LJ saw no change: rvdw stayed 0.9
    # a fresh module state straight from a twin-range interaction_const_t, pruning flavour included
    ip = twin(1.0)
    _upload(nb, p, ip)
    check(ip, *nb.compute(p.xq, abi.FORCE_VIRIAL))
    # and back to equal cut-offs
    _upload(nb, p, base)
    check(base, *nb.compute(p.xq, abi.FORCE_VIRIAL))


@pytest.mark.parametrize("el", ["rf", "ewald"])
@pytest.mark.parametrize("mod", ["force-switch", "potential-switch"])
def test_lj_switch_flavours(nb, el, mod):
    """vdw-modifier = force-switch / potential-switch (nbnxn_kernel_ref_inner.h:170-205,
    nbnxn_cuda_kernel_utils.cuh:84-183) against the oracle, whose switch arithmetic is pinned to the
    reference's nbnxn_kernel_ref in tests/test_oracle_golden.py.  Water (O-O LJ only) and the
    8-type chain mixture; F+V, F-only and prune flavours; through the struct entry point as well."""
    base = g.system.reaction_field(0.9) if el == "rf" else g.system.ewald(0.9)
    ip = (g.system.with_force_switch if mod == "force-switch" else g.system.with_potential_switch)(base, 0.7)
    for s in (g.system.water_box(5, 1, 1, model="spce"), g.system.lj_chain_mixture(3000)):
        p = g.workload.prepare(s, ip, 200)
        o = O.kernel_gpu_list(p.xq, p.type, p.ntype, p.nbfp, p.shift_vec, p.pl, ip)
        o_shift = O.kernel_gpu_list(p.xq, p.type, p.ntype, p.nbfp, p.shift_vec, p.pl, base)
        # the modifier is visible well above the parity tolerance (in the LJ energy; the forces of the chain
        # mixture are dominated by close contacts the switch does not touch)
        assert abs(o_shift["Vvdw"] - o["Vvdw"]) > 10 * ENERGY_RTOL * abs(o["Vvdw"])
        _upload(nb, p, ip)
        f, e = nb.compute(p.xq, abi.FORCE_VIRIAL)          # prune + energies
        _check_against(f, e, p, ip, o)
        f2, _ = nb.compute(p.xq, 0)                        # force only, pruned list
        assert relrms(f2, o["f"]) < FORCE_RTOL
        f3, e3 = nb.compute(p.xq, abi.FORCE_VIRIAL)        # energies, pruned list
        _check_against(f3, e3, p, ip, o)
        # back to the plain cut-off on the same handle: no modifier state may leak
        _upload(nb, p, base)
        f4, e4 = nb.compute(p.xq, abi.FORCE_VIRIAL)
        _check_against(f4, e4, p, base, o_shift)


def test_force_switch_with_twin_range(nb):
    """force switch to rvdw = 0.9 under rcoulomb = 1.1 (what PME load balancing does to a CHARMM-style
    set-up): the switch polynomial is not zero beyond rvdw, the cut-off check has to remove it."""
    import dataclasses
    import math
    s = g.system.lj_chain_mixture(4000)
    base = g.system.with_force_switch(g.system.ewald(0.9, rlist=1.2), 0.75)
    p = g.workload.prepare(s, base, 300)
    beta = g.system.calc_ewaldcoeff_q(1.1, 1e-5)
    ip = dataclasses.replace(base, rcoulomb=1.1, ewaldcoeff_q=beta, sh_ewald=math.erfc(beta * 1.1))
    o = O.kernel_gpu_list(p.xq, p.type, p.ntype, p.nbfp, p.shift_vec, p.pl, ip)
    _upload(nb, p, ip)
    _check_against(*nb.compute(p.xq, abi.FORCE_VIRIAL), p, ip, o)
    # and reached through the load balancer's call
    _upload(nb, p, base)
    nb.compute(p.xq, abi.FORCE_VIRIAL)
    nb.pme_loadbal_update_param(ip.to_struct())
    _check_against(*nb.compute(p.xq, abi.FORCE_VIRIAL), p, ip, o)


@pytest.mark.parametrize("name", __import__("util").SWITCH)
def test_lj_switch_fixtures_from_the_reference_kernel(nb, name):
    """the HIP kernels straight against the reference's nbnxn_kernel_ref outputs for the switch flavours"""
    p, ip, d = load_small(name)
    _upload(nb, p, ip)
    f, (elj, eel, fsh) = nb.compute(p.xq, abi.FORCE_VIRIAL)
    assert relrms(f, d["ref_f"]) < FORCE_RTOL
    assert abs(elj - float(d["ref_Vvdw"])) < 2e-5 * abs(float(d["ref_Vvdw"]))
    assert abs(eel - float(d["ref_Vc"])) < 3e-5 * abs(float(d["ref_Vc"]))    # one-float accumulation in the reference


@pytest.mark.parametrize("el", ["rf", "ewald"])
@pytest.mark.parametrize("rule", ["geometric", "lorentz-berthelot"])
def test_ljpme_flavours(nb, el, rule):
    """vdwtype = PME (LJ_EWALD_COMB_GEOM / _LB, nbnxn_cuda_kernel_utils.cuh:186-302) against
    the oracle, whose LJ-PME arithmetic is pinned to the reference's nbnxn_kernel_ref in
    tests/test_oracle_golden.py.  Water and the 8-type chain mixture (where the two rules differ);
    F+V with pruning, F-only, F+V on the pruned list; the plain cut-off afterwards on the same handle."""
    base = g.system.reaction_field(0.9) if el == "rf" else g.system.ewald(0.9)
    ip = g.system.with_ljpme(base, abi.LJCR_GEOM if rule == "geometric" else abi.LJCR_LB)
    for s in (g.system.water_box(5, 1, 1, model="spce"), g.system.lj_chain_mixture(3000)):
        p = g.workload.prepare(s, ip, 200)
        o = O.kernel_gpu_list(p.xq, p.type, p.ntype, p.nbfp, p.shift_vec, p.pl, ip)
        o_cut = O.kernel_gpu_list(p.xq, p.type, p.ntype, p.nbfp, p.shift_vec, p.pl, base)
        assert abs(o_cut["Vvdw"] - o["Vvdw"]) > 10 * ENERGY_RTOL * abs(o["Vvdw"])
        _upload(nb, p, ip)
        f, e = nb.compute(p.xq, abi.FORCE_VIRIAL)
        _check_against(f, e, p, ip, o)
        f2, _ = nb.compute(p.xq, 0)
        assert relrms(f2, o["f"]) < FORCE_RTOL
        f3, e3 = nb.compute(p.xq, abi.FORCE_VIRIAL)
        _check_against(f3, e3, p, ip, o)
        _upload(nb, p, base)
        f4, e4 = nb.compute(p.xq, abi.FORCE_VIRIAL)
        _check_against(f4, e4, p, base, o_cut)


@pytest.mark.parametrize("name", __import__("util").LJPME)
def test_ljpme_fixtures_from_the_reference_kernel(nb, name):
    """the HIP kernels straight against the reference's nbnxn_kernel_ref outputs for LJ-PME"""
    p, ip, d = load_small(name)
    _upload(nb, p, ip)
    f, (elj, eel, fsh) = nb.compute(p.xq, abi.FORCE_VIRIAL)
    assert relrms(f, d["ref_f"]) < FORCE_RTOL
    assert abs(elj - float(d["ref_Vvdw"])) < 2e-5 * abs(float(d["ref_Vvdw"]))
    assert abs(eel - float(d["ref_Vc"])) < 3e-5 * abs(float(d["ref_Vc"]))    # one-float accumulation in the reference


def test_ljpme_twin_range_many_types_and_struct_entry(nb):
    """LJ-PME where the other template axes bite: rvdw < rcoulomb (the grid term has to obey the LJ
    cut-off as well), 21 atom types (LJ and grid tables read from L2 instead of LDS), and the
    struct-taking init_const reading vdwtype / ljpme_comb_rule / nbat->nbfp_comb."""
    import dataclasses
    import math
    s = g.system.lj_chain_mixture(2500, ntype=20, seed=5)
    base = g.system.with_ljpme(g.system.ewald(0.9, rlist=1.15), abi.LJCR_LB)
    beta = g.system.calc_ewaldcoeff_q(1.1, 1e-5)
    ip = dataclasses.replace(base, rcoulomb=1.1, ewaldcoeff_q=beta, sh_ewald=math.erfc(beta * 1.1))
    p = g.workload.prepare(s, ip, 100)
    assert p.ntype > 16
    o = O.kernel_gpu_list(p.xq, p.type, p.ntype, p.nbfp, p.shift_vec, p.pl, ip)
    _upload(nb, p, ip)
    _check_against(*nb.compute(p.xq, abi.FORCE_VIRIAL), p, ip, o)
    f2, _ = nb.compute(p.xq, 0)
    assert relrms(f2, o["f"]) < FORCE_RTOL
    # the struct entry: cut-off LJ first so that a stale grid table cannot pass for the right one
    _upload(nb, p, g.system.ewald(0.9, rlist=1.15))
    nbfp = np.ascontiguousarray(p.nbfp, dtype=np.float32)
    comb = g.system.nbfp_comb(nbfp, p.ntype, abi.LJCR_LB)
    nbat = abi.Atomdata()
    nbat.ntype, nbat.nbfp = p.ntype, abi.ptr(nbfp)
    nbat.comb_rule, nbat.nbfp_comb = abi.LJCR_LB, abi.ptr(comb)
    grp = abi.VerletGroup()
    grp.nbat = C.pointer(nbat)
    nb.init_const_struct(ip.to_struct(), grp)
    _check_against(*nb.compute(p.xq, abi.FORCE_VIRIAL), p, ip, o)
    # a combination rule the atom data does not carry is refused
    nbat.comb_rule = abi.LJCR_GEOM
    with pytest.raises(RuntimeError):
        nb.init_const_struct(ip.to_struct(), grp)


def test_tabulated_ewald_flavour(nb):
    """eelCuEWALD_TAB(_TWIN) (nbnxn_cuda_kernel.cuh:451-458): the module's own force table against the
    oracle's restatement of the reference's table filler, the kernels against the oracle evaluating
    the same interpolation, against the analytical flavour, and the table following rcoulomb / beta
    through nbnxn_cuda_pme_loadbal_update_param."""
    import dataclasses
    import math
    s = g.system.water_box(2, 2, 2, model="spce")
    ana = g.system.ewald(0.9, rlist=1.1)
    ip = g.system.with_tabulated_ewald(ana)
    p = g.workload.prepare(s, ip, 300)
    _upload(nb, p, ip)
    assert not nb.is_kernel_ewald_analytical()
    tab, scale = nb.coulomb_tab()
    tab_o, scale_o = O.ewald_force_table(ip.ewaldcoeff_q, ip.rcoulomb)
    assert len(tab) == len(tab_o) == 1536 and scale == pytest.approx(scale_o, rel=1e-7)
    # same table up to the half-ulp of the round trip through the kernel's units
    assert np.abs(tab - tab_o).max() <= 2.4e-7 * np.abs(tab_o).max()
    o = O.kernel_gpu_list(p.xq, p.type, p.ntype, p.nbfp, p.shift_vec, p.pl, ip)
    o_ana = O.kernel_gpu_list(p.xq, p.type, p.ntype, p.nbfp, p.shift_vec, p.pl, ana)
    f, e = nb.compute(p.xq, abi.FORCE_VIRIAL)
    _check_against(f, e, p, ip, o)
    assert relrms(f, o_ana["f"]) < FORCE_RTOL
    f2, _ = nb.compute(p.xq, 0)
    assert relrms(f2, o["f"]) < FORCE_RTOL
    # PME load balancing: rcoulomb 1.0, new beta -> new table, twin-range kernels
    beta = g.system.calc_ewaldcoeff_q(1.0, 1e-5)
    ip2 = dataclasses.replace(ip, rcoulomb=1.0, ewaldcoeff_q=beta, sh_ewald=math.erfc(beta * 1.0))
    nb.pme_loadbal_update_param(ip2.to_struct())
    tab2, scale2 = nb.coulomb_tab()
    assert scale2 == pytest.approx(1534 / 1.0, rel=1e-6) and np.abs(tab2 - tab).max() > 1e-3
    o2 = O.kernel_gpu_list(p.xq, p.type, p.ntype, p.nbfp, p.shift_vec, p.pl, ip2)
    _check_against(*nb.compute(p.xq, abi.FORCE_VIRIAL), p, ip2, o2)
    # tabulated electrostatics under LJ-PME and under the force switch
    for ipx in (g.system.with_ljpme(ip, abi.LJCR_GEOM), g.system.with_force_switch(ip, 0.7)):
        _upload(nb, p, ipx)
        ox = O.kernel_gpu_list(p.xq, p.type, p.ntype, p.nbfp, p.shift_vec, p.pl, ipx)
        _check_against(*nb.compute(p.xq, abi.FORCE_VIRIAL), p, ipx, ox)
    # back to the analytical flavour on the same handle
    _upload(nb, p, ana)
    assert nb.is_kernel_ewald_analytical()
    _check_against(*nb.compute(p.xq, abi.FORCE_VIRIAL), p, ana, o_ana)


def _flavour_families():
    S = g.system
    out = []
    for el_name, el in (("rf", lambda: S.reaction_field(0.9)), ("ewald", lambda: S.ewald(0.9)),
                        ("ewald-tab", lambda: S.with_tabulated_ewald(S.ewald(0.9))),
                        ("ewald-twin", None)):
        for lj_name, lj in (("cut", lambda ip: ip), ("fswitch", lambda ip: S.with_force_switch(ip, 0.7)),
                            ("pswitch", lambda ip: S.with_potential_switch(ip, 0.7)),
                            ("ljpme-geom", lambda ip: S.with_ljpme(ip, abi.LJCR_GEOM)), ("ljpme-lb", lambda ip: S.with_ljpme(ip, abi.LJCR_LB))):
            out.append((el_name, lj_name, el, lj))
    return out


@pytest.mark.parametrize("el_name,lj_name,el,lj", _flavour_families(), ids=[f"{a}-{b}" for a, b, _, _ in _flavour_families()])
def test_coincident_excluded_atoms_every_flavour(nb, el_name, lj_name, el, lj):
    """Two DISTINCT, mutually excluded atoms on one point -- shell / Drude particles, virtual sites on atoms -- in
    one cluster and across two clusters.  The reference guards the pair with r^2 += (1 - int_bit) * 1e-12
    (nbnxn_consts.h:66-79, nbnxn_cuda_kernel.cuh:371, nbnxn_kernel_ref_inner.h:132, nbnxn_kernel_gpu_ref.c:257);
    without it the switch, LJ-PME and tabulated flavours produce 0 * inf.  Every electrostatics x LJ family,
    F + V with pruning, F only, F + V on the pruned list, against the double oracle (which applies the increment)."""
    import dataclasses
    import math
    from util import coincident_excluded_system, ljpme_grid_c6
    S = g.system
    s, pairs = coincident_excluded_system()
    if el is None:       # twin-range Ewald: rvdw 0.9 < rcoulomb 1.05
        beta = S.calc_ewaldcoeff_q(1.05, 1e-5)
        base = dataclasses.replace(S.ewald(0.9, rlist=1.05), rcoulomb=1.05, ewaldcoeff_q=beta, sh_ewald=math.erfc(beta * 1.05))
    else:
        base = el()
    ip = lj(base)
    p = g.workload.prepare(s, ip, 100)
    cell = p.host.cell_index()
    same = [(a, b) for a, b in pairs if cell[a] // 8 == cell[b] // 8]
    assert len(same) >= 10 and len(pairs) - len(same) >= 3        # inside one cluster, and straddling two
    for a, b in pairs:
        assert np.array_equal(p.xq[cell[a], :3], p.xq[cell[b], :3])
    o = O.kernel_gpu_list(p.xq, p.type, p.ntype, p.nbfp, p.shift_vec, p.pl, ip)
    assert np.isfinite(o["f"]).all()
    # LJ-PME energy: in float the reference's 1 - exp(-x)(1 + x + x^2/2) is exactly 0 at r^2 = 1e-12, where the true value
    # times r^-6 / 6 is c6grid beta_lj^6 / 36 (the double oracle sums that, by series): the float kernels -- the
    # reference's own (tests/golden/*coincident*ljpme*) and this one -- are short of the oracle by that constant per pair
    vv = o["Vvdw"]
    if ip.vdwtype == abi.VDW_PME:
        nbfp = np.ascontiguousarray(p.nbfp, dtype=np.float32)
        vv -= sum(ljpme_grid_c6(nbfp, p.ntype, ip.ljpme_comb_rule, p.type[cell[a]], p.type[cell[b]]) for a, b in pairs) \
            * float(np.float32(ip.ewaldcoeff_lj)) ** 6 / 36.0
    _upload(nb, p, ip)
    for flags in (abi.FORCE_VIRIAL, 0, abi.FORCE_VIRIAL):
        f, (elj, eel, fsh) = nb.compute(p.xq, flags)
        assert np.isfinite(f).all()
        assert relrms(f, o["f"]) < FORCE_RTOL
        if flags:
            assert abs(eel - o["Vc"]) < ENERGY_RTOL * abs(o["Vc"])
            assert abs(elj - vv) < ENERGY_RTOL * abs(vv)
            assert np.isfinite(fsh).all()


@pytest.mark.parametrize("name", COINCIDENT)
def test_coincident_excluded_atoms_against_the_reference_kernels(nb, name):
    """the same case straight against committed outputs of the reference's nbnxn_kernel_ref (and, for the two plain
    cut-off fixtures, nbnxn_kernel_gpu_ref) on the same list"""
    p, ip, d = load_small(name)
    _upload(nb, p, ip)
    f, (elj, eel, fsh) = nb.compute(p.xq, abi.FORCE_VIRIAL)
    assert np.isfinite(f).all() and relrms(f, d["ref_f"]) < FORCE_RTOL
    assert abs(eel - float(d["ref_Vc"])) < 3e-5 * abs(float(d["ref_Vc"]))
    assert abs(elj - float(d["ref_Vvdw"])) < 3e-5 * abs(float(d["ref_Vvdw"]))
    if "gpuref_f" in d.files:
        assert relrms(f, d["gpuref_f"]) < FORCE_RTOL
        assert abs(eel - float(d["gpuref_Vc"])) < 2e-5 * abs(float(d["gpuref_Vc"]))
        assert abs(elj - float(d["gpuref_Vvdw"])) < 2e-5 * abs(float(d["gpuref_Vvdw"]))


def test_list_quirks_of_the_reference_builders(nb):
    """What lists made by the reference carry and this repo's builders do not: (1) cj4.cj[] of slots without an
    interaction bit is uninitialised memory (only imask / excl_ind are cleared, nbnxn_search.c:2460-2472); (2) after a
    GPU prune the two 32-lane halves of a group hold DIFFERENT masks (each warp clears its own,
    nbnxn_cuda_kernel.cuh:332-340, :503).  A wave64 walks the union; both against the oracle on the same list."""
    from gromacs_cosmo_amd.host import PairList
    rng = np.random.default_rng(17)
    for ip, s in ((g.system.ewald(0.9, rlist=1.0), g.system.water_box(2, 2, 2, model="spce")),
                  (g.system.reaction_field(0.9, rlist=1.0), g.system.lj_chain_mixture(3000))):
        p = g.workload.prepare(s, ip, 150)
        cj4 = p.pl.cj4.copy()
        im = cj4["imei"]["imask"][:, 0].copy()
        # (1) garbage where no bit is set: indices far outside the atom arrays, negative ones too
        nq = 0
        for jm in range(4):
            dead = ((im >> (8 * jm)) & 0xff) == 0
            cj4["cj"][dead, jm] = rng.integers(-2 ** 31, 2 ** 31 - 1, size=int(dead.sum()), dtype=np.int64).astype(np.int32)
            nq += int(dead.sum())
        assert nq > 50
        # (2) each half drops the cluster pairs none of ITS j-atoms sees within rlist, as a warp of the CUDA kernel would
        half = [np.zeros_like(im), np.zeros_like(im)]
        x = p.xq[:, :3].astype(np.float32)
        rl2 = np.float32(ip.rlist) ** 2
        for e in p.pl.sci:
            sh = p.shift_vec[e["shift"]].astype(np.float32)
            for j4 in range(e["cj4_ind_start"], e["cj4_ind_end"]):
                for jm in range(4):
                    for ci in range(8):
                        if not (im[j4] >> (8 * jm + ci)) & 1:
                            continue
                        xi = x[(e["sci"] * 8 + ci) * 8:(e["sci"] * 8 + ci) * 8 + 8] + sh
                        xj = x[p.pl.cj4["cj"][j4, jm] * 8:p.pl.cj4["cj"][j4, jm] * 8 + 8]
                        d2 = ((xi[:, None, :] - xj[None, :, :]) ** 2).sum(-1)
                        for w in (0, 1):
                            if (d2[:, 4 * w:4 * w + 4] < rl2).any():
                                half[w][j4] |= np.uint32(1 << (8 * jm + ci))
        cj4["imei"]["imask"][:, 0], cj4["imei"]["imask"][:, 1] = half[0], half[1]
        assert (half[0] != half[1]).sum() > 20
        pl = PairList(p.pl.sci, cj4, p.pl.excl, p.pl.nci_tot, ip.rlist)
        o = O.kernel_gpu_list(p.xq, p.type, p.ntype, p.nbfp, p.shift_vec, pl, ip)
        o_full = O.kernel_gpu_list(p.xq, p.type, p.ntype, p.nbfp, p.shift_vec, p.pl, ip)
        assert relrms(o["f"], o_full["f"]) < 1e-13          # pruning per half loses nothing inside the cut-off
        g.workload.upload(nb, p, ip)
        nb.init_pairlist(pl)
        assert nb.verify_pairlist(pl) == 0                   # the comparison masks the dead slots, the upload does not
        f, e = nb.compute(p.xq, abi.FORCE_VIRIAL)            # with the in-kernel prune on top
        _check_against(f, e, p, ip, o)
        # the in-kernel prune of the union equals the CPU rule on the union, and writes both halves alike
        got = nb.download_cj4(pl)
        want = O.prune_list(p.xq, p.shift_vec, pl, ip.rlist)
        assert np.array_equal(got["imei"]["imask"][:, 0], got["imei"]["imask"][:, 1])
        if ip.el_type == abi.EL_RF:
            assert np.array_equal(got["imei"]["imask"], want["imei"]["imask"])
        else:
            # the Ewald force kernels measure in beta-scaled coordinates: a pair within an ulp of rlist^2 may fall either way
            assert np.count_nonzero(got["imei"]["imask"][:, 0] != want["imei"]["imask"][:, 0]) <= 2
        f2, _ = nb.compute(p.xq, 0)
        assert relrms(f2, o["f"]) < FORCE_RTOL
        # the prune-only kernel on a fresh copy of the quirky list
        nb.init_pairlist(pl)
        nb.prune_only(ip.rlist)
        assert np.array_equal(nb.download_cj4(pl)["imei"]["imask"], want["imei"]["imask"])


def test_device_selection_and_memory_test(nb):
    """the gpu_utils.h surface (detect / pick / check / device id / memtest) on the box's one MI355X"""
    lib = nb.lib
    info, ndev, ncompat = g.gpu.detect_gpus_info()
    assert ndev >= 1 and ncompat >= 1 and lib.nbnxn_hip_sizeof_dev_info() == C.sizeof(abi.GpuDevInfo)
    assert info[0].id == 0 and info[0].stat == abi.GPU_COMPATIBLE and info[0].arch.decode().startswith("gfx950")
    assert info[0].compute_units == 256 and info[0].total_mem > 200 * 2 ** 30
    use = (C.c_int * ndev)()
    nuse = C.c_int(0)
    assert lib.nbnxn_hip_pick_compatible_gpus(info, ndev, use, ndev, C.byref(nuse)) == 0
    assert nuse.value == ncompat and use[0] == 0
    assert lib.nbnxn_hip_get_gpu_device_id(info, ndev, use, nuse.value, 0) == 0
    assert lib.nbnxn_hip_get_gpu_device_id(info, ndev, use, nuse.value, nuse.value) == -1
    # a selection with a device that does not exist
    sel = (C.c_int * 2)(0, ndev + 3)
    res = (C.c_int * 2)()
    assert lib.nbnxn_hip_check_selected_gpus(res, info, ndev, sel, 2) == 0
    assert list(res) == [abi.GPU_COMPATIBLE, abi.GPU_NONEXISTENT]
    assert lib.nbnxn_hip_check_selected_gpus(res, info, ndev, sel, 1) == 1
    # detection must leave the current device alone, and the module still works afterwards
    assert lib.nbnxn_hip_get_current_gpu_device_id() == 0
    assert lib.nbnxn_hip_do_quick_memtest(-1) == 0
    assert lib.nbnxn_hip_do_timed_memtest(0, 1) == 0
    assert lib.nbnxn_hip_do_quick_memtest(ndev + 3) == -1
    p, ip, d = load_small("spc216_rf09")
    _upload(nb, p, ip)
    f, _ = nb.compute(p.xq, abi.FORCE_VIRIAL)
    assert relrms(f, d["ref_f"]) < FORCE_RTOL


_SWITCH_CHILD = r"""
import sys, json
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import numpy as np
import gromacs_cosmo_amd as g
from gromacs_cosmo_amd import abi
from util import load_small
p, ip, d = load_small("spc3240_rf09")
nb = g.gpu.NbnxnHip(0)
g.workload.upload(nb, p, ip)
out = {}
for k, flags in (("first_F", 0), ("second_VF", abi.FORCE_VIRIAL), ("third_F", 0)):
    nb.reset_timings()
    f, e = nb.compute(p.xq, flags)
    t = nb.timings()
    out[k] = {"bins": [[t.ktime[a][b].c for b in range(2)] for a in range(2)], "f": np.asarray(f, dtype=np.float64).tolist()}
print(json.dumps(out))
"""


@pytest.mark.gpu
@pytest.mark.parametrize("env,bins", [
    ({}, {"first_F": (1, 0), "second_VF": (0, 1), "third_F": (0, 0)}),
    ({"GMX_GPU_ALWAYS_ENER": "1"}, {"first_F": (1, 1), "second_VF": (0, 1), "third_F": (0, 1)}),
    ({"GMX_GPU_NEVER_ENER": "1"}, {"first_F": (1, 0), "second_VF": (0, 0), "third_F": (0, 0)}),
    ({"GMX_GPU_ALWAYS_PRUNE": "1"}, {"first_F": (1, 0), "second_VF": (0, 1), "third_F": (0, 0)}),
])
def test_benchmark_switches_of_the_cuda_module(env, bins):
    """GMX_GPU_ALWAYS_ENER / NEVER_ENER / ALWAYS_PRUNE (nbnxn_cuda.cu:116-119) pick the kernel flavour whatever the
    step's flags; they are read once per process, so each case runs in a child.  The timing table's bin
    ktime[prune][ener] says which flavour ran (booked by the list's prune state, as at nbnxn_cuda.cu:633), and the
    forces are the oracle's in every case (an energy or pruning flavour computes the same forces)."""
    import subprocess, sys, json
    p, ip, d = load_small("spc3240_rf09")
    o = O.kernel_gpu_list(p.xq, p.type, p.ntype, p.nbfp, p.shift_vec, p.pl, ip)
    e = dict(os.environ, **env)
    for k in ("GMX_GPU_ALWAYS_ENER", "GMX_GPU_NEVER_ENER", "GMX_GPU_ALWAYS_PRUNE"):
        if k not in env:
            e.pop(k, None)
    r = subprocess.run([sys.executable, "-c", _SWITCH_CHILD, ROOT], env=e, capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    for k, (prune, ener) in bins.items():
        b = out[k]["bins"]
        assert b[prune][ener] == 1 and sum(map(sum, b)) == 1, (k, b)
        assert relrms(np.array(out[k]["f"]), o["f"]) < FORCE_RTOL


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["spc3240_rf09", "spce3240_ewald09"])
def test_reference_named_calls_run_a_step(name, tmp_path):
    """The drop-in boundary at run time: a C program compiled against the REFERENCE's own headers
    (oracle/binding_check.c, built into oracle/_ref where the reference tree exists) builds the reference's objects --
    gmx_gpu_info_t, nonbonded_verlet_t, nbnxn_atomdata_t with pinned x and f, nbnxn_pairlist_t, interaction_const_t --
    and makes mdrun's calls by their reference names (detect_cuda_gpus ... init_gpu, nbnxn_cuda_init, _init_const,
    _init_atomdata, _upload_shiftvec, _init_pairlist, _launch_kernel, _launch_cpyback, _wait_gpu, _clear_outputs,
    _get_timings, _free, free_gpu), all resolved by include/nbnxn_hip_gmx.h over libnbnxn_hip.so.  Forces of an energy
    step and of a force-only step, energies and the virial's shift forces against the oracle."""
    import subprocess
    from util import write_step_file, read_step_result
    exe = os.path.join(ROOT, "oracle", "_ref", "binding_check")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/binding_check not built (needs /root/reference at build time)")
    p, ip, d = load_small(name)
    o = O.kernel_gpu_list(p.xq, p.type, p.ntype, p.nbfp, p.shift_vec, p.pl, ip)
    fin, fout = str(tmp_path / "step.in"), str(tmp_path / "step.out")
    write_step_file(fin, p, ip)
    r = subprocess.run([exe, fin, fout], capture_output=True, text=True, timeout=100)
    assert r.returncode == 0, r.stdout[-500:] + r.stderr[-1500:]
    res = read_step_result(fout, p.natoms)
    assert relrms(res.f1, o["f"]) < FORCE_RTOL and relrms(res.f2, o["f"]) < FORCE_RTOL
    assert abs(res.e_lj - o["Vvdw"]) < ENERGY_RTOL * abs(o["Vvdw"]) and abs(res.e_el - o["Vc"]) < ENERGY_RTOL * abs(o["Vc"])
    vir = virial_from(p.xq, res.f1, p.shift_vec, res.fshift)
    vir_o = virial_from(p.xq, o["f"], p.shift_vec, o["fshift"])
    assert np.abs(vir - vir_o).max() < 1e-5 * np.abs(vir_o).max()
    assert res.analytical == int(ip.el_type != abi.EL_RF) and res.min_ci_balanced > 0 and res.ndev >= 1
    # the timing table books the first launch on the new list as prune + energies, the second as neither (nbnxn_cuda.cu:633)
    assert res.nb_c == 2 and res.c_prune_ener == 1 and res.c_plain == 1
