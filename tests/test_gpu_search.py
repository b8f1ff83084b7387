"""Pair search on the device (nbnxn_put_on_grid + nbnxn_make_pairlist) against the
host-side builder (gromacs-cosmo_amd/csrc/nbnxn_host.cpp, itself checked against brute force in
tests/test_host_list.py): the grid order, the grid-ordered atom data and the list must be THE SAME,
index for index and bit for bit, and the forces computed from the device-built list must meet the
kernel's parity bar against the oracle."""
import numpy as np
import pytest

import gromacs_cosmo_amd as g
from gromacs_cosmo_amd import abi
import oracle as O
from util import FORCE_RTOL, ENERGY_RTOL, relrms

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nb():
    h = g.gpu.NbnxnHip(0)
    yield h
    h.close()


def _excl_words(pl):
    """[ncj4, 2, 32] exclusion words each (group, half) points to"""
    return pl.excl["pair"][pl.cj4["imei"]["excl_ind"]]


def _assert_same_list(dev, host):
    assert (dev.nsci, dev.ncj4, dev.nexcl, dev.nci_tot) == (host.nsci, host.ncj4, host.nexcl, host.nci_tot)
    assert np.array_equal(dev.sci, host.sci)
    assert np.array_equal(dev.cj4["imei"]["imask"], host.cj4["imei"]["imask"])
    used = ((host.cj4["imei"]["imask"][:, 0, None] >> (8 * np.arange(4))) & 0xff) != 0
    assert np.array_equal(dev.cj4["cj"][used], host.cj4["cj"][used])
    # exclusion sets are numbered in a different order on the device; what they hold must agree
    assert np.array_equal(_excl_words(dev), _excl_words(host))
    assert np.all(dev.excl["pair"][0] == 0xffffffff)


def _device_search(nb, s, ip, mcb):
    nb.init_const(ip, s.ntype + 1, s.nbfp())
    npad = nb.put_on_grid(s.box, s.x, s.type, s.q)
    nb.make_pairlist(ip.rlist, s.excl_index, s.excl_a, mcb)
    return npad


def _forces_resident(nb, flags):
    f = np.zeros((nb.natoms, 3), dtype=np.float32)
    e_lj, e_el = np.zeros(1, np.float32), np.zeros(1, np.float32)
    fsh = np.zeros((abi.SHIFTS, 3), dtype=np.float32)
    nb.clear_outputs(flags | abi.FORCE_VIRIAL)
    nb.launch_kernel_resident(flags)
    nb.launch_cpyback(f, flags)
    nb.wait_gpu(flags, abi.LOCAL, e_lj, e_el, fsh)
    return f, float(e_lj[0]), float(e_el[0])


CASES = [
    ("water 5k rf", lambda: g.system.water_box(2, 2, 2), lambda: g.system.reaction_field(0.9), 0),
    ("water 5k ewald split", lambda: g.system.water_box(2, 2, 2, model="spce"), lambda: g.system.ewald(0.9, rlist=1.0), 600),
    ("chains 3k", lambda: g.system.lj_chain_mixture(3000), lambda: g.system.reaction_field(0.9), 0),
    ("chains 6k split", lambda: g.system.lj_chain_mixture(6000, ntype=8, seed=11), lambda: g.system.ewald(0.9, rlist=1.05), 2000),
    ("water 97k", lambda: g.system.water_box(5, 5, 6), lambda: g.system.reaction_field(1.0), None),
    # triclinic cells: "wide" ones need two box lengths in x (nbnxn_search.c:4917-4921)
    ("chains 3k triclinic wide", lambda: g.system.lj_chain_mixture(2744, ntype=5, seed=7, skew=(6, -5, 4)), lambda: g.system.reaction_field(0.9), 0),
    ("chains 3k triclinic wide split", lambda: g.system.lj_chain_mixture(2744, ntype=5, seed=7, skew=(6, -5, 4)),
     lambda: g.system.ewald(0.9, rlist=1.0), 1000),
    ("chains 6k triclinic split", lambda: g.system.lj_chain_mixture(5832, ntype=8, seed=11, skew=(4, 3, -7)), lambda: g.system.ewald(0.9, rlist=1.0), 2000),
]


@pytest.mark.parametrize("name,make_system,make_ip,mcb", CASES, ids=[c[0] for c in CASES])
def test_device_grid_and_list_equal_the_host_builders(nb, name, make_system, make_ip, mcb):
    s, ip = make_system(), make_ip()
    mcb = nb.min_ci_balanced() if mcb is None else mcb
    p = g.workload.prepare(s, ip, mcb)
    npad = _device_search(nb, s, ip, mcb)
    assert npad == p.natoms
    grid = nb.download_grid()
    assert (grid["ncx"], grid["ncy"]) == p.host.ncolumns() and grid["nc"] * 64 == npad
    assert np.array_equal(grid["a"], p.host.atom_order())
    assert np.array_equal(grid["cell"], p.host.cell_index())
    assert np.array_equal(grid["xq"].view(np.uint32), p.xq.view(np.uint32))      # fillers included, bit for bit
    assert np.array_equal(grid["type"], p.type)
    # every cluster's box holds its atoms
    real = grid["a"].reshape(-1, 8) >= 0
    xc = grid["xq"][:, :3].reshape(-1, 8, 3)
    for cl in np.flatnonzero(real.any(axis=1))[::97]:
        pts = xc[cl][real[cl]]
        assert np.array_equal(grid["bb"][cl, :3], pts.min(axis=0)) and np.array_equal(grid["bb"][cl, 3:], pts.max(axis=0))
    dev = nb.download_pairlist()
    _assert_same_list(dev, p.pl)
    if mcb:
        assert dev.nsci > grid["nc"]          # the balancing split really happened
    # and the kernel on the device-built list, coordinates never having left the device
    o = O.kernel_gpu_list(p.xq, p.type, p.ntype, p.nbfp, p.shift_vec, p.pl, ip)
    f, elj, eel = _forces_resident(nb, abi.FORCE_VIRIAL)
    assert relrms(f, o["f"]) < FORCE_RTOL
    assert abs(elj - o["Vvdw"]) < ENERGY_RTOL * abs(o["Vvdw"]) and abs(eel - o["Vc"]) < ENERGY_RTOL * abs(o["Vc"])
    # forces back in caller order through the device-side atom order
    fa = np.zeros((s.natoms, 3), dtype=np.float32)
    nb.add_f(fa)
    fo = np.zeros((s.natoms, 3), dtype=np.float32)
    p.host.add_f(np.asarray(o["f"], dtype=np.float32), fo)
    assert relrms(fa, fo) < FORCE_RTOL


def test_triclinic_cell_against_brute_force(nb):
    """The whole device path on a strongly skewed cell (search on the device, list never on the host, kernel) against
    the list-independent truth: brute force over all atom pairs with the nearest of the 45 shift images, in double.
    Forces, energies and the virial from the shift forces (whose shift vectors are sums of box vectors here)."""
    from util import virial_from
    s = g.system.lj_chain_mixture(2744, ntype=5, seed=7, skew=(6, -5, 4))
    for ip in (g.system.reaction_field(0.9, epsilon_rf=0.0), g.system.ewald(0.9)):
        _device_search(nb, s, ip, 500)
        tx = nb.download_pairlist().sci["shift"] % 5 - 2
        assert tx.min() == -2
        f = np.zeros((nb.natoms, 3), dtype=np.float32)
        e_lj, e_el = np.zeros(1, np.float32), np.zeros(1, np.float32)
        fsh = np.zeros((abi.SHIFTS, 3), dtype=np.float32)
        nb.clear_outputs(abi.FORCE_VIRIAL)
        nb.launch_kernel_resident(abi.FORCE_VIRIAL)
        nb.launch_cpyback(f, abi.FORCE_VIRIAL)
        nb.wait_gpu(abi.FORCE_VIRIAL, abi.LOCAL, e_lj, e_el, fsh)
        fa = np.zeros((s.natoms, 3), dtype=np.float32)
        nb.add_f(fa)
        b = O.brute_force(s.x, s.q, s.type, s.ntype + 1, s.nbfp(), s.box, s.excl_index, s.excl_a, ip)
        assert relrms(fa, b["f"]) < FORCE_RTOL
        assert abs(float(e_lj[0]) - b["Vvdw"]) < ENERGY_RTOL * abs(b["Vvdw"])
        assert abs(float(e_el[0]) - b["Vc"]) < ENERGY_RTOL * abs(b["Vc"])
        grid = nb.download_grid()
        sv = np.array([tx_ * s.box[0] + ty * s.box[1] + tz * s.box[2] for tz in (-1, 0, 1) for ty in (-1, 0, 1) for tx_ in (-2, -1, 0, 1, 2)],
                      dtype=np.float32)
        vir = virial_from(grid["xq"], f, sv, fsh)
        assert np.abs(vir - b["vir"]).max() < ENERGY_RTOL * np.abs(b["vir"]).max()
    with pytest.raises(g.gpu.HipError):
        nb.put_on_grid(np.array([[3, 0, 0], [1.6, 3, 0], [0, 0, 3]], dtype=np.float32), s.x, s.type, s.q)     # too skewed
    with pytest.raises(g.gpu.HipError):
        nb.put_on_grid(s.box, s.x, s.type, s.q, pbc_shift_dim=(1, 1, 0))    # a skewed z vector needs shifts in z


def test_device_search_edge_cases(nb):
    """fewer atoms than one super-cell, a ragged last cluster, no exclusions, a non-periodic dimension"""
    rng = np.random.default_rng(5)
    ip = g.system.reaction_field(0.9)
    for n, pbc in ((37, (1, 1, 1)), (64, (1, 1, 1)), (1001, (1, 1, 1)), (1001, (1, 1, 0))):
        box = np.array([3.0, 3.2, 3.4], dtype=np.float32)
        # jittered lattice (no overlapping atoms), a random subset of its sites
        sites = np.stack(np.meshgrid(np.arange(10), np.arange(10), np.arange(11), indexing="ij"), -1).reshape(-1, 3)
        x = ((sites[rng.permutation(len(sites))[:n]] + 0.5 + rng.uniform(-0.15, 0.15, (n, 3))) * (box / (10, 10, 11))).astype(np.float32)
        x[10:15, 2] = x[15:20, 2]                   # equal z: ties go by atom index
        x[3, 0] = 0.0                               # on the lower edge of the first column
        x[4, :2] = np.nextafter(box[:2], 0)         # the last representable point inside the last column
        typ = rng.integers(0, 2, n).astype(np.int32)
        q = rng.uniform(-0.5, 0.5, n).astype(np.float32)
        nbfp = np.zeros((3, 3, 2), dtype=np.float32)
        nbfp[:2, :2, 0], nbfp[:2, :2, 1] = 6 * 2.6e-3, 12 * 2.6e-6
        h = g.host.NbnxnHost(box, pbc)
        h.put_on_grid(0, (0, 0, 0), box, 0, x)
        xq, t = h.set_atomdata(3, typ, q)
        h.copy_x_to_xq(x, xq)
        pl = h.make_pairlist(abi.LOCAL, ip.rlist, xq, None, None, 0)
        nb.init_const(ip, 3, nbfp)
        nb.forget_exclusions()                      # an earlier test's topology must not leak in
        assert nb.put_on_grid(box, x, typ, q, pbc_shift_dim=pbc) == len(xq)
        nb.make_pairlist(ip.rlist, None, None, 0)
        grid = nb.download_grid()
        assert np.array_equal(grid["a"], h.atom_order()) and np.array_equal(grid["xq"].view(np.uint32), xq.view(np.uint32))
        _assert_same_list(nb.download_pairlist(), pl)
        o = O.kernel_gpu_list(xq, t, 3, nbfp, h.shift_vec(), pl, ip)
        f, elj, eel = _forces_resident(nb, abi.FORCE_VIRIAL)
        assert relrms(f, o["f"]) < FORCE_RTOL


def test_device_search_follows_moving_atoms(nb):
    """a second search on displaced coordinates given as a DEVICE array (the MD loop's case)"""
    import torch
    s = g.system.water_box(2, 2, 2)
    ip = g.system.reaction_field(0.9)
    _device_search(nb, s, ip, 0)
    rng = np.random.default_rng(9)
    x2 = np.mod(s.x + rng.normal(0, 0.05, s.x.shape), s.box).astype(np.float32)
    xd = torch.from_numpy(x2).cuda()
    nb.put_on_grid(s.box, xd, None, None)           # topology held from the first search
    nb.make_pairlist(ip.rlist, None, None, 0)
    import dataclasses
    s2 = dataclasses.replace(s, x=x2)
    p = g.workload.prepare(s2, ip, 0)
    assert np.array_equal(nb.download_grid()["a"], p.host.atom_order())
    _assert_same_list(nb.download_pairlist(), p.pl)


def test_device_search_refusals(nb):
    s = g.system.water_box(1, 1, 1)
    ip = g.system.reaction_field(0.9)
    nb.init_const(ip, s.ntype + 1, s.nbfp())
    nb.put_on_grid(s.box, s.x, s.type, s.q)
    with pytest.raises(RuntimeError):
        nb.make_pairlist(ip.rlist, s.excl_index, s.excl_a, 0, iloc=abi.NONLOCAL)     # no halo zone on the grid
    with pytest.raises(RuntimeError):
        nb.make_pairlist(1.0, s.excl_index, s.excl_a, 0)                             # box 1.86 < 2 rlist
    with pytest.raises(RuntimeError):
        nb.put_on_grid(s.box, s.x, s.type + 7, s.q)                                  # types outside the LJ table
    with pytest.raises(RuntimeError):
        nb.put_on_grid(s.box, s.x[:100], None, None)                                 # held topology is for another atom count
    # a host-side init_atomdata invalidates the device grid
    p = g.workload.prepare(s, ip, 0)
    g.workload.upload(nb, p, ip)
    with pytest.raises(RuntimeError):
        nb.make_pairlist(ip.rlist, s.excl_index, s.excl_a, 0)


@pytest.mark.parametrize("kind", ["water", "chains", "triclinic"])
def test_verlet_loop_with_moving_atoms(nb, kind):
    """The callers either side of the kernel in the reference's order (gromacs-cosmo_amd/verlet.py = the
    device-resident do_force_cutsVERLET): search every 5 steps with a 0.1 nm buffer, atoms drifting in between
    (some out of the box), the list reused and pruned once.  Every step's forces, in caller order, against the
    oracle on a FRESH host-built list for that step's positions."""
    import dataclasses
    import torch
    if kind == "water":
        s, base = g.system.water_box(2, 2, 2, model="spce"), g.system.ewald(0.9)
    elif kind == "triclinic":
        s, base = g.system.lj_chain_mixture(2744, ntype=5, seed=7, skew=(6, -5, 4)), g.system.ewald(0.9)
    else:
        s, base = g.system.lj_chain_mixture(3000), g.system.reaction_field(0.9)
    ip = dataclasses.replace(base, rlist=1.0)
    ip_exact = base                                        # oracle side: list radius = cut-off
    loop = g.verlet.VerletLoop(nb, s, ip, nstlist=5, min_ci_balanced=200)
    rng = np.random.default_rng(3)
    x = s.x.copy()
    vel = rng.normal(0, 0.004, x.shape).astype(np.float32)   # <= ~0.015 nm per step, 4 steps between searches
    for step in range(11):
        if step % 5 == 0:
            x = g.system.put_atoms_in_box(x, s.box)         # as the reference does on search steps
        xd = torch.from_numpy(x).cuda()
        fd = torch.zeros_like(xd)
        e = loop.force(xd, fd, abi.FORCE_VIRIAL)
        torch.cuda.synchronize()
        f = fd.cpu().numpy()
        p = g.workload.prepare(dataclasses.replace(s, x=g.system.put_atoms_in_box(x, s.box)), ip_exact, 0)
        o = O.kernel_gpu_list(p.xq, p.type, p.ntype, p.nbfp, p.shift_vec, p.pl, ip_exact)
        fo = np.zeros((s.natoms, 3), dtype=np.float32)
        p.host.add_f(np.asarray(o["f"], dtype=np.float32), fo)
        assert relrms(f, fo) < FORCE_RTOL, step
        assert abs(e[0] - o["Vvdw"]) < ENERGY_RTOL * abs(o["Vvdw"]) and abs(e[1] - o["Vc"]) < ENERGY_RTOL * abs(o["Vc"]), step
        x = x + vel
    assert loop.nsearch == 3


def test_prune_only_kernel_equals_the_pruning_rule(nb):
    """dynamic pruning: the outer list (rlist 1.1) re-pruned to 1.0 by the prune-only kernel, bit for bit the
    masks of the reference's pruning rule (oracle_prune_list = nbnxn_cuda_kernel.cuh:332-340 at that radius);
    then with moved atoms, one slice of three at a time"""
    from gromacs_cosmo_amd.host import PairList
    s = g.system.water_box(2, 2, 2)
    ip = g.system.reaction_field(0.9, rlist=1.1)
    p = g.workload.prepare(s, ip, 300)
    g.workload.upload(nb, p, ip)
    nb.upload_xq(p.xq)
    nb.prune_only(1.0)                                       # first call on a new list: the whole list
    got = nb.download_cj4(p.pl)["imei"]["imask"]
    want = O.prune_list(p.xq, p.shift_vec, p.pl, 1.0)["imei"]["imask"]
    assert np.array_equal(got, want) and (want != p.pl.cj4["imei"]["imask"]).any()
    # the force kernel now runs on the inner list, without pruning flavour, and gives the same physics
    o = O.kernel_gpu_list(p.xq, p.type, p.ntype, p.nbfp, p.shift_vec, p.pl, ip)
    f, e = nb.compute(p.xq, abi.FORCE_VIRIAL)
    assert relrms(f, o["f"]) < FORCE_RTOL
    assert np.array_equal(nb.download_cj4(p.pl)["imei"]["imask"], want)
    # atoms move a little; rolling prune, slice by slice, from the OUTER masks (pruned at 1.1 with the old x)
    rng = np.random.default_rng(1)
    xq2 = p.xq.copy()
    xq2[:, :3] += rng.normal(0, 0.02, (len(xq2), 3)).astype(np.float32)
    nb.upload_xq(xq2)
    outer = PairList(p.pl.sci, O.prune_list(p.xq, p.shift_vec, p.pl, 1.1), p.pl.excl, p.pl.nci_tot, 1.1)
    want2 = O.prune_list(xq2, p.shift_vec, outer, 1.0)["imei"]["imask"]
    assert (want2 != want).any()
    nsci = p.pl.nsci
    for part in range(3):
        nb.prune_only(1.0, 3)
        got2 = nb.download_cj4(p.pl)["imei"]["imask"]
        done = np.zeros(p.pl.ncj4, dtype=bool)
        for e_ in p.pl.sci[:nsci * (part + 1) // 3]:
            done[e_["cj4_ind_start"]:e_["cj4_ind_end"]] = True
        assert np.array_equal(got2[done], want2[done]) and np.array_equal(got2[~done], want[~done])
    o2 = O.kernel_gpu_list(xq2, p.type, p.ntype, p.nbfp, p.shift_vec, p.pl, ip)
    f2, _ = nb.compute(xq2, 0)
    assert relrms(f2, o2["f"]) < FORCE_RTOL
    with pytest.raises(RuntimeError):
        nb.prune_only(1.2)                                   # beyond the radius the list was built with


def test_verlet_loop_with_dynamic_pruning(nb):
    """outer list 1.06 nm every 10 steps, rolling re-prune to 0.97 nm in 4 slices, cut-off 0.9: per-step
    displacements bounded so that both buffers hold; every step's forces against a fresh exact list"""
    import dataclasses
    import torch
    s = g.system.lj_chain_mixture(3000)
    base = g.system.ewald(0.9)
    ip = dataclasses.replace(base, rlist=1.06)
    loop = g.verlet.VerletLoop(nb, s, ip, nstlist=10, min_ci_balanced=200, rlist_inner=0.97, prune_parts=4)
    rng = np.random.default_rng(8)
    x = s.x.copy()
    vel = rng.uniform(-0.004, 0.004, x.shape).astype(np.float32)      # |v| <= 0.007 nm / step
    for step in range(21):
        if step % 10 == 0:
            x = np.mod(x, s.box).astype(np.float32)
        xd = torch.from_numpy(x).cuda()
        fd = torch.zeros_like(xd)
        loop.force(xd, fd, 0)
        torch.cuda.synchronize()
        p = g.workload.prepare(dataclasses.replace(s, x=np.mod(x, s.box).astype(np.float32)), base, 0)
        o = O.kernel_gpu_list(p.xq, p.type, p.ntype, p.nbfp, p.shift_vec, p.pl, base, calc_ener=False)
        fo = np.zeros((s.natoms, 3), dtype=np.float32)
        p.host.add_f(np.asarray(o["f"], dtype=np.float32), fo)
        assert relrms(fd.cpu().numpy(), fo) < FORCE_RTOL, step
        x = x + vel


def _slab_domain(glob, rank, world, rlist):
    """the arrays dd.build_domain would hand to the grid for rank `rank` of `world` z-slabs (no messages needed here:
    every rank of a test holds the global system)"""
    box_search = np.asarray(glob.box, dtype=np.float32)      # [3], or the matrix of a cell skewed in xy
    box = np.ascontiguousarray(np.diag(box_search)) if box_search.shape == (3, 3) else box_search
    lz = float(box[2]) / world
    owner = np.clip(np.floor(glob.x[:, 2].astype(np.float64) / lz).astype(np.int64), 0, world - 1)
    home = np.nonzero(owner == rank)[0]
    up = (rank + 1) % world
    cand = np.nonzero(owner == up)[0]
    zrel = glob.x[cand, 2] - np.float32(up * lz)
    halo = cand[zrel < rlist]
    x_halo = glob.x[halo].copy()
    if up == 0:
        x_halo[:, 2] += box[2]                       # across the periodic wrap
    ids = np.concatenate([home, halo])
    x_all = np.concatenate([glob.x[home], x_halo]).astype(np.float32)
    z0 = rank * lz
    zones = [((0.0, 0.0, z0), (box[0], box[1], z0 + lz), 0, len(home)),
             ((0.0, 0.0, z0 + lz), (box[0], box[1], z0 + lz + rlist), len(home), len(halo))]
    # topology exclusions in local numbering; partners that are not on this rank drop out
    g2l = np.full(glob.natoms, -1, dtype=np.int64)
    g2l[ids] = np.arange(len(ids))
    ei, ea = [0], []
    for a in ids:
        part = g2l[glob.excl_a[glob.excl_index[a]:glob.excl_index[a + 1]]]
        ea.extend(int(v) for v in part if v >= 0)
        ei.append(len(ea))
    return box_search, ids, x_all, zones, np.array(ei, dtype=np.int32), np.array(ea, dtype=np.int32)


@pytest.mark.parametrize("kind,rank", [("water", 0), ("water", 1), ("chains", 1), ("chains skewed in xy", 0), ("chains skewed in xy", 1)])
def test_device_search_with_a_halo_zone(nb, kind, rank):
    """home zone + one halo zone (a z-slab decomposition's rank, the upper one wrapping around the box): both grids,
    the local and the non-local list equal to the host builder's; forces of both kernels against the oracle"""
    if kind == "water":
        glob, ip = g.system.water_box(2, 2, 4, model="spce"), g.system.ewald(0.9)
    elif kind == "chains":
        glob, ip = g.system.lj_chain_mixture(6000, ntype=8, seed=21), g.system.reaction_field(0.9, rlist=1.0)
    else:
        # box[YY][XX] != 0 with box[ZZ] along z: what a z-slab decomposition can take (dd.build_domain)
        glob, ip = g.system.lj_chain_mixture(5832, ntype=8, seed=21, skew=(7, 0, 0)), g.system.reaction_field(0.9, rlist=1.0)
    box, ids, x_all, zones, ei, ea = _slab_domain(glob, rank, 2, ip.rlist)
    mcb = 150
    h = g.host.NbnxnHost(box, (1, 1, 0))
    for zi, (c0, c1, a0, n) in enumerate(zones):
        h.put_on_grid(zi, c0, c1, a0, x_all[a0:a0 + n])
    ntype = glob.ntype + 1
    xq, typ = h.set_atomdata(ntype, glob.type[ids], glob.q[ids])
    h.copy_x_to_xq(x_all, xq)
    pl_l = h.make_pairlist(abi.LOCAL, ip.rlist, xq, ei, ea, mcb)
    pl_n = h.make_pairlist(abi.NONLOCAL, ip.rlist, xq, ei, ea, mcb)
    assert pl_n.nsci > 0

    nb.init_const(ip, ntype, glob.nbfp())
    nb.forget_exclusions()
    npad = nb.put_on_grid_zones(box, zones, x_all, glob.type[ids], glob.q[ids], pbc_shift_dim=(1, 1, 0))
    assert npad == len(xq)
    grid = nb.download_grid()
    assert np.array_equal(grid["a"], h.atom_order()) and np.array_equal(grid["cell"], h.cell_index())
    assert np.array_equal(grid["xq"].view(np.uint32), xq.view(np.uint32)) and np.array_equal(grid["type"], typ)
    nb.make_pairlist(ip.rlist, ei, ea, mcb, iloc=abi.LOCAL)
    nb.make_pairlist(ip.rlist, None, None, mcb, iloc=abi.NONLOCAL)
    _assert_same_list(nb.download_pairlist(abi.LOCAL), pl_l)
    _assert_same_list(nb.download_pairlist(abi.NONLOCAL), pl_n)
    # both kernels on the device-built lists (on this one-stream handle they simply run in order)
    o_l = O.kernel_gpu_list(xq, typ, ntype, glob.nbfp(), h.shift_vec(), pl_l, ip)
    o_n = O.kernel_gpu_list(xq, typ, ntype, glob.nbfp(), h.shift_vec(), pl_n, ip)
    f = np.zeros((npad, 3), dtype=np.float32)
    nb.clear_outputs(abi.FORCE_VIRIAL)
    nb.launch_kernel_resident(abi.FORCE_VIRIAL, abi.LOCAL)
    nb.launch_kernel_resident(abi.FORCE_VIRIAL, abi.NONLOCAL)
    nb.launch_cpyback(f, abi.FORCE_VIRIAL, abi.NONLOCAL)
    nb.launch_cpyback(f, abi.FORCE_VIRIAL, abi.LOCAL)
    nb.wait_gpu(abi.FORCE_VIRIAL, abi.NONLOCAL)
    e_lj, e_el = np.zeros(1, np.float32), np.zeros(1, np.float32)
    fsh = np.zeros((abi.SHIFTS, 3), dtype=np.float32)
    nb.wait_gpu(abi.FORCE_VIRIAL, abi.LOCAL, e_lj, e_el, fsh)
    fo = np.asarray(o_l["f"]) + np.asarray(o_n["f"])
    assert relrms(f, fo) < FORCE_RTOL
    assert abs(e_lj[0] - (o_l["Vvdw"] + o_n["Vvdw"])) < ENERGY_RTOL * abs(o_l["Vvdw"] + o_n["Vvdw"])
    assert abs(e_el[0] - (o_l["Vc"] + o_n["Vc"])) < ENERGY_RTOL * abs(o_l["Vc"] + o_n["Vc"])
    if ip.rlist > ip.rcoulomb:
        # dynamic pruning of the NON-LOCAL list too: a fresh list pruned to an inner radius by the prune-only kernel,
        # its masks against the reference's pruning rule on the host list
        nb.make_pairlist(ip.rlist, None, None, mcb, iloc=abi.NONLOCAL)
        inner = 0.5 * (ip.rlist + ip.rcoulomb)
        nb.prune_only(inner, 1, iloc=abi.NONLOCAL)
        want = O.prune_list(xq, h.shift_vec(), pl_n, inner)["imei"]["imask"]
        got = nb.download_pairlist(abi.NONLOCAL).cj4["imei"]["imask"]
        assert np.array_equal(got, want) and (want != pl_n.cj4["imei"]["imask"]).any()


@pytest.mark.parametrize("self_dims", [(2,), (1, 2), (0, 2), (0, 1, 2)])
def test_device_search_with_zone_pairs(nb, self_dims, grid=(1, 1, 1), rank=0):
    """the reference's eighth shell (dd_zo / dd_zp1-3, domdec.c:433-449): 1 / 3 / 7 imported zones, the non-local list
    with imported i-clusters; the device search makes the host builder's lists, whole arrays"""
    import tempfile
    import torch.distributed as dist
    from gromacs_cosmo_amd import dd
    glob, ip = g.system.water_box(3, 3, 3, model="spce"), g.system.ewald(0.9, rlist=1.0)
    world = grid[0] * grid[1] * grid[2]
    # build_domain sends the halo bookkeeping through torch.distributed, also to itself
    with tempfile.TemporaryDirectory() as tmp:
        dist.init_process_group("gloo", init_method=f"file://{tmp}/store", rank=rank, world_size=world)
        try:
            dom = dd.build_domain(glob, ip, rank, world, "cpu", 600, grid=grid, self_dims=self_dims, scheme="eighth")
        finally:
            dist.destroy_process_group()
    ndim = sum(dom.decomp)
    assert len(dom.zones) == 2 ** ndim and dom.zone_pairs[1] == dd.eighth_shell_zone_pairs(ndim)[1]
    nb.init_const(ip, dom.ntype, dom.nbfp)
    nb.forget_exclusions()
    npad = nb.put_on_grid_zones(dom.box_search, dom.zones, dom.x_all, dom.type_all, dom.q_all, pbc_shift_dim=dom.pbc_shift_dim)
    assert npad == dom.npad
    nb.make_pairlist(dom.rlist, dom.excl_index, dom.excl_a, dom.min_ci_balanced, iloc=abi.LOCAL, zone_pairs=dom.zone_pairs[0])
    nb.make_pairlist(dom.rlist, None, None, dom.min_ci_balanced, iloc=abi.NONLOCAL, zone_pairs=dom.zone_pairs[1])
    _assert_same_list(nb.download_pairlist(abi.LOCAL), dom.pl_local)
    _assert_same_list(nb.download_pairlist(abi.NONLOCAL), dom.pl_nonlocal)
    if ndim > 1:
        # i-clusters of imported zones are in the list
        assert dom.pl_nonlocal.sci["sci"].max() * 64 >= dom.npad_local
    # tables the device search does not take
    for iloc, zp in ((abi.LOCAL, ((0, 0, 2),)), (abi.NONLOCAL, ((1, 1, 2),)), (abi.NONLOCAL, ((0, 0, 2),)), (abi.NONLOCAL, ((0, 1, 99),)),
                     (abi.NONLOCAL, ((1, 2, 2), (0, 1, 2))), (abi.NONLOCAL, ((0, 1, 2), (0, 1, 2)))):   # i-zones not ascending / listed twice
        with pytest.raises(g.gpu.HipError):
            nb.make_pairlist(dom.rlist, None, None, dom.min_ci_balanced, iloc=iloc, zone_pairs=zp)


def test_device_search_at_config4_size(nb):
    """1,533,168 atoms (BASELINE config 4): the device grid and list against the host builder's, whole arrays"""
    s = g.system.water_box(13, 13, 14)
    ip = g.system.reaction_field(1.0)
    mcb = nb.min_ci_balanced()
    p = g.workload.prepare(s, ip, mcb)
    nb.forget_exclusions()
    assert _device_search(nb, s, ip, mcb) == p.natoms
    grid = nb.download_grid()
    assert np.array_equal(grid["a"], p.host.atom_order()) and np.array_equal(grid["xq"].view(np.uint32), p.xq.view(np.uint32))
    _assert_same_list(nb.download_pairlist(), p.pl)
    # one force evaluation on it: Newton's third law over the whole box
    f, _, _ = _forces_resident(nb, 0)
    assert np.abs(f.astype(np.float64).sum(axis=0)).max() < 1e-3 * np.abs(f).max()


def test_progressive_balance_switch(nb):
    """both builders with progressive balancing off (every entry split at the average pair budget, the reference's
    plain split_sci_entry) still agree, and the switch changes the split"""
    s = g.system.water_box(2, 2, 2)
    ip = g.system.reaction_field(0.9)
    lists = []
    for on in (False, True):
        h = g.host.NbnxnHost(s.box)
        h.set_progressive_balance(on)
        h.put_on_grid(0, (0, 0, 0), s.box, 0, s.x)
        xq, _ = h.set_atomdata(s.ntype + 1, s.type, s.q)
        h.copy_x_to_xq(s.x, xq)
        pl = h.make_pairlist(abi.LOCAL, ip.rlist, xq, s.excl_index, s.excl_a, 600)
        nb.set_progressive_balance(on)
        nb.forget_exclusions()
        _device_search(nb, s, ip, 600)
        _assert_same_list(nb.download_pairlist(), pl)
        lists.append(pl)
    nb.set_progressive_balance(True)
    assert lists[0].ncj4 == lists[1].ncj4 and lists[0].nci_tot == lists[1].nci_tot      # the same pairs ...
    len0 = np.sort(lists[0].sci["cj4_ind_end"] - lists[0].sci["cj4_ind_start"])
    len1 = np.sort(lists[1].sci["cj4_ind_end"] - lists[1].sci["cj4_ind_start"])
    assert len1.max() > len0.max() and np.median(len1[:len(len1) // 8]) <= np.median(len0[:len(len0) // 8])   # ... cut differently


def test_list_scratch_overflow_takes_the_two_walk_route():
    """The list is normally made in one walk per entry into a scratch range of 512 j-clusters; an entry that outgrows
    its range sends the whole list through the count + fill walks instead.  With the range cut to 8 every entry
    overflows: the list must still be the host builder's."""
    import os
    s = g.system.water_box(2, 2, 2, model="spce")
    ip = g.system.ewald(0.9, rlist=1.0)
    p = g.workload.prepare(s, ip, 400)
    os.environ["GMX_HIP_NB_LIST_TMP_CAP"] = "8"
    try:
        h = g.gpu.NbnxnHip(0)           # the search state (and its range size) is made per handle, on first use
        _device_search(h, s, ip, 400)
    finally:
        del os.environ["GMX_HIP_NB_LIST_TMP_CAP"]
    _assert_same_list(h.download_pairlist(), p.pl)
    o = O.kernel_gpu_list(p.xq, p.type, p.ntype, p.nbfp, p.shift_vec, p.pl, ip)
    f, elj, eel = _forces_resident(h, abi.FORCE_VIRIAL)
    assert relrms(f, o["f"]) < FORCE_RTOL
    h.close()
