"""The multi-GPU form of the path on CPU: world_size 1 to 8 over gloo (z slabs, and bricks in two and three
dimensions), with the oracle in place of the HIP kernels.  What is under test is everything bench.py --gpus N adds
around the kernel: the cut, the halo zones and their grids, local / non-local lists, exclusions across
the cuts, the per-step coordinate and force messages (one per forward neighbour), the periodic crossings with their
shift forces and the energy / shift-force reduction (gromacs-cosmo_amd/dd.py).  Reference: the same global
system as ONE domain through the same oracle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, model, min_ci_balanced, ret, grid=None, self_dims=(2,), scheme="half", balance=0):
    import gromacs_cosmo_amd as g
    from gromacs_cosmo_amd import abi, dd
    import oracle as O
    from util import virial_from
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        ip = g.system.reaction_field(0.9) if model in ("spc", "chains_xy") else g.system.ewald(0.9)
        if model == "chains_xy":
            # a cell skewed in the xy plane (box[YY][XX] != 0), box[ZZ] along z: decomposable along z
            glob = g.system.lj_chain_mixture(2744, ntype=5, seed=7, skew=(6, 0, 0))
        elif model == "film":
            gr = grid or (1, 1, world)
            glob = g.system.water_film(2 * gr[0], 2 * gr[1], 3 * gr[2], liquid_fraction=0.4, vapour_keep=6, model="spce")
        else:
            gr = grid or (1, 1, world)
            glob = g.system.water_box(2 * gr[0], 2 * gr[1], 2 * gr[2], model=model)
        if balance:
            dom, history = dd.balance_domain(glob, ip, rank, world, torch.device("cpu"), min_ci_balanced, grid=grid, self_dims=self_dims,
                                             scheme=scheme, iterations=balance)
        else:
            dom = dd.build_domain(glob, ip, rank, world, torch.device("cpu"), min_ci_balanced, grid=grid, self_dims=self_dims, scheme=scheme)
        xq_t = torch.from_numpy(dom.xq)                 # aliases dom.xq
        f_t = torch.zeros((dom.npad, 3), dtype=torch.float64)
        halo = dd.StagedHalo(lambda d: dd.HaloExchange(d, xq_t, f_t), dom) if scheme == "staged" else dd.HaloExchange(dom, xq_t, f_t)
        # the imported coordinates must really arrive by message: wreck them first
        want = dom.xq[dom.halo_slots].copy()
        dom.xq[dom.halo_slots, :3] = 1e6
        halo.send_x()
        assert np.array_equal(dom.xq[dom.halo_slots], want)

        o_l = O.kernel_gpu_list(dom.xq, dom.type, dom.ntype, dom.nbfp, dom.shift_vec, dom.pl_local, ip)
        o_n = O.kernel_gpu_list(dom.xq, dom.type, dom.ntype, dom.nbfp, dom.shift_vec, dom.pl_nonlocal, ip)
        f_t += torch.from_numpy(np.asarray(o_l["f"], dtype=np.float64) + np.asarray(o_n["f"], dtype=np.float64))
        # non-local entries only touch home i-atoms and halo j-atoms
        assert dom.pl_nonlocal.nsci > 0 and (scheme != "half" or dom.pl_nonlocal.sci["sci"].max() * 64 < dom.npad_local)
        halo.return_f()
        out = torch.zeros((abi.OUT_SLOTS, abi.OUT_STRIDE), dtype=torch.float64)
        out[0, 0] = o_l["Vvdw"] + o_n["Vvdw"]
        out[0, 1] = o_l["Vc"] + o_n["Vc"]
        out[0, 2:2 + 135] = torch.from_numpy((np.asarray(o_l["fshift"]) + np.asarray(o_n["fshift"])).reshape(-1))
        e = dd.fold_and_reduce(out.view(-1), halo).numpy()

        fnb = f_t.numpy().astype(np.float32)
        f_home = np.zeros((dom.nhome + dom.nhalo, 3), dtype=np.float32)
        dom.host.add_f(fnb, f_home)
        # virial pieces of this rank: home atoms only (the imported ones gave their forces back)
        x_home = glob.x[dom.home_ids].astype(np.float64)
        xf = x_home.T @ f_home[:dom.nhome].astype(np.float64)
        xf_t = torch.from_numpy(xf)
        dist.all_reduce(xf_t)

        # single-domain reference, computed by every rank for its own atoms
        prep = g.workload.prepare(glob, ip, 0)
        o = O.kernel_gpu_list(prep.xq, prep.type, prep.ntype, prep.nbfp, prep.shift_vec, prep.pl, ip)
        f_ref = np.zeros((glob.natoms, 3), dtype=np.float32)
        prep.host.add_f(np.asarray(o["f"], dtype=np.float32), f_ref)
        err = float(np.sqrt(((f_home[:dom.nhome] - f_ref[dom.home_ids]) ** 2).sum() / (f_ref[dom.home_ids] ** 2).sum()))
        vir_ref = virial_from(np.concatenate([glob.x, np.zeros((glob.natoms, 1), np.float32)], 1), f_ref,
                              prep.shift_vec, np.asarray(o["fshift"]))
        vir_dd = -0.5 * (xf_t.numpy() + dom.shift_vec.astype(np.float64).T @ e[2:2 + 135].reshape(45, 3))
        ret[rank] = dict(err=err, e_lj=e[0], e_el=e[1], e_lj_ref=o["Vvdw"], e_el_ref=o["Vc"],
                         vir_err=float(np.abs(vir_dd - vir_ref).max() / np.abs(vir_ref).max()),
                         nhome=dom.nhome, nhalo=dom.nhalo, natoms=glob.natoms,
                         nsci=(dom.pl_local.nsci, dom.pl_nonlocal.nsci), nmsg=len(dom.msgs), nzones=len(dom.zones),
                         peers=sorted({m.recv_from for m in dom.msgs}),
                         nonhome_i=int((dom.pl_nonlocal.sci["sci"] * 64 >= dom.npad_local).sum()),
                         history=dom.balance_history, thickness=(dom.hi - dom.lo).tolist(),
                         load=dom.pl_local.nci_tot + dom.pl_nonlocal.nci_tot)
    finally:
        dist.destroy_process_group()


# world 1: one rank that is its own neighbour (periodic in z through a halo of its own atoms): the form in which a
# one-GPU box can run the whole decomposed step over RCCL (tests/test_gpu_z_dd.py)
@pytest.mark.parametrize("world,model,mcb", [(2, "spc", 0), (2, "spce", 200), (3, "spc", 0), (1, "spce", 0), (2, "chains_xy", 0)])
def test_slab_decomposition_matches_single_domain(world, model, mcb):
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_worker, args=(world, _free_port(), model, mcb, ret), nprocs=world, join=True)
        res = [ret[r] for r in range(world)]
    assert sum(r["nhome"] for r in res) == res[0]["natoms"]
    for r in res:
        assert r["nhalo"] > 0 and r["nsci"][1] > 0
        # both sides are the double oracle on float32 coordinates; they differ by the float32 rounding
        # of z + box_z for atoms imported across the periodic wrap (~1e-7 of a distance)
        assert r["err"] < 2e-6, r
        assert abs(r["e_lj"] - r["e_lj_ref"]) < 1e-6 * abs(r["e_lj_ref"]), r
        assert abs(r["e_el"] - r["e_el_ref"]) < 1e-6 * abs(r["e_el_ref"]), r
        assert r["vir_err"] < 1e-5, r


# Bricks in two and three dimensions: half shell with direct peers (4 messages per rank in 2-D, 13 in 3-D).  With two
# ranks along a dimension the neighbour above is also the neighbour below (two messages between the same pair, with
# different shifts); (3, 2, 1) has distinct neighbours along x.  One rank that is its own neighbour in all three
# dimensions carries the 13-message step alone.
@pytest.mark.parametrize("world,grid,self_dims,model,mcb", [
    (4, (2, 2, 1), (), "spce", 0), (4, (1, 2, 2), (), "spc", 150), (2, (2, 1, 1), (), "spc", 0),
    (6, (3, 2, 1), (), "spc", 0), (8, (2, 2, 2), (), "spce", 0), (1, (1, 1, 1), (0, 1, 2), "spce", 0), (1, (1, 1, 1), (0, 2), "spc", 0)])
def test_brick_decomposition_matches_single_domain(world, grid, self_dims, model, mcb):
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_worker, args=(world, _free_port(), model, mcb, ret, grid, self_dims), nprocs=world, join=True)
        res = [ret[r] for r in range(world)]
    assert sum(r["nhome"] for r in res) == res[0]["natoms"]
    ncut = sum(1 for d in range(3) if grid[d] > 1 or (world == 1 and d in self_dims))
    for r in res:
        assert r["nmsg"] == {1: 1, 2: 4, 3: 13}[ncut] and r["nzones"] == r["nmsg"] + 1
        assert r["nhalo"] > 0 and r["nsci"][1] > 0
        assert r["err"] < 2e-6, r
        assert abs(r["e_lj"] - r["e_lj_ref"]) < 1e-6 * abs(r["e_lj_ref"]), r
        assert abs(r["e_el"] - r["e_el_ref"]) < 1e-6 * abs(r["e_el_ref"]), r
        assert r["vir_err"] < 1e-5, r


# The reference's eighth shell (dd_zo / dd_zp3, domdec.c:432-449) with the same direct messages: 1 / 3 / 7 imported zones,
# and pairs BETWEEN imported zones in the non-local list, whose i-clusters are then not all home clusters.
@pytest.mark.parametrize("world,grid,self_dims,model,mcb", [
    (4, (2, 2, 1), (), "spce", 0), (4, (1, 2, 2), (), "spc", 150), (8, (2, 2, 2), (), "spce", 0), (6, (3, 2, 1), (), "spc", 0),
    (1, (1, 1, 1), (0, 1, 2), "spce", 0), (2, (1, 1, 2), (), "spc", 0)])
@pytest.mark.parametrize("scheme", ["eighth", "staged"])
def test_eighth_shell_decomposition_matches_single_domain(world, grid, self_dims, model, mcb, scheme):
    """scheme "staged": the same zones brought by the reference's pulses, one per cut dimension, the later ones forwarding
    what the earlier ones brought (dd_move_x / dd_move_f, domdec.c:667-870); same lists, same result."""
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_worker, args=(world, _free_port(), model, mcb, ret, grid, self_dims, scheme), nprocs=world, join=True)
        res = [ret[r] for r in range(world)]
    assert sum(r["nhome"] for r in res) == res[0]["natoms"]
    ncut = sum(1 for d in range(3) if grid[d] > 1 or (world == 1 and d in self_dims))
    for r in res:
        assert r["nmsg"] == ({1: 1, 2: 3, 3: 7}[ncut] if scheme == "eighth" else ncut) and r["nzones"] == 2 ** ncut
        assert (r["nonhome_i"] > 0) == (ncut > 1)        # zone x zone pairs exist from two cut dimensions on
        assert r["err"] < 2e-6, r
        assert abs(r["e_lj"] - r["e_lj_ref"]) < 1e-6 * abs(r["e_lj_ref"]), r
        assert abs(r["e_el"] - r["e_el_ref"]) < 1e-6 * abs(r["e_el_ref"]), r
        assert r["vir_err"] < 1e-5, r


# Load balancing by the reference's rule (set_dd_cell_sizes_dlb_root, domdec.c:3429-3498) on the ranks' cluster-pair
# counts: a liquid film under thin vapour, cut into equal slabs / bricks, is far out of balance; the moved planes
# bring the heaviest rank down, and the decomposed step still equals the single domain.
@pytest.mark.parametrize("world,grid,scheme", [(4, (1, 1, 4), "half"), (4, (1, 2, 2), "eighth")])
def test_load_balanced_cuts(world, grid, scheme):
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_worker, args=(world, _free_port(), "film", 0, ret, grid, (), scheme, 24), nprocs=world, join=True)
        res = [ret[r] for r in range(world)]
    assert sum(r["nhome"] for r in res) == res[0]["natoms"]
    h = res[0]["history"]
    assert all(r["history"] == h for r in res)
    loads = np.array([r["load"] for r in res], dtype=np.float64)
    assert h[0] > 0.8 and loads.max() / loads.mean() - 1 < 0.1, (h, loads)      # from 80-130 % over the mean to under 10 %
    assert len({round(r["thickness"][2], 4) for r in res}) > 1      # the planes moved
    for r in res:
        assert min(r["thickness"]) > 0.9
        # relative to the forces of the rank's own atoms, which are small where the rank holds vapour (4e-6 there
        # with equal slabs, too: the float32 rounding of z + box_z of the imported atoms)
        assert r["err"] < 1e-5, r
        assert abs(r["e_lj"] - r["e_lj_ref"]) < 1e-6 * abs(r["e_lj_ref"]), r
        assert abs(r["e_el"] - r["e_el_ref"]) < 1e-6 * abs(r["e_el_ref"]), r
        assert r["vir_err"] < 1e-5, r


def test_dlb_rule():
    """dlb_update_cuts against the reference's arithmetic worked by hand (domdec.c:3469-3497): relax 0.5, the whole
    row scaled when the largest change exceeds 10 %, the minimum cell size enforced"""
    from gromacs_cosmo_amd import dd
    cuts = dd.uniform_cuts((1, 1, 4))
    # loads 4 ranks along z: mean 100; imbalances +1, 0, -0.5, -0.5 -> changes -0.5, 0, .25, .25, largest 0.5 > 0.1 -> sc = 0.1
    new = dd.dlb_update_cuts(cuts, (1, 1, 4), [200, 100, 50, 50], (0, 0, 0))
    size = 0.25 * (1 - 0.1 * np.array([1.0, 0.0, -0.5, -0.5]))
    assert np.allclose(np.diff(new[2]), size / size.sum(), rtol=0, atol=1e-12)
    assert np.array_equal(new[0], cuts[0]) and np.array_equal(new[1], cuts[1])
    # small imbalance: the plain underrelaxed step
    new = dd.dlb_update_cuts(cuts, (1, 1, 4), [104, 100, 98, 98], (0, 0, 0))
    size = 0.25 * (1 - 0.5 * np.array([0.04, 0.0, -0.02, -0.02]))
    assert np.allclose(np.diff(new[2]), size / size.sum(), rtol=0, atol=1e-12)
    # the limit: a cell already at the minimum stays there, the others share the rest
    new = dd.dlb_update_cuts((cuts[0], cuts[1], np.array([0.0, 0.2, 0.5, 0.75, 1.0])), (1, 1, 4), [200, 100, 50, 50], (0, 0, 0.2))
    d = np.diff(new[2])
    assert abs(d[0] - 0.2) < 1e-12 and abs(d.sum() - 1) < 1e-12 and (d[1:] > 0.2).all()
    # rows of a 2 x 2 grid in y and z: row loads are sums over the other dimension
    new = dd.dlb_update_cuts(dd.uniform_cuts((1, 2, 2)), (1, 2, 2), [30, 10, 30, 10], (0, 0, 0))     # ranks (y0,z0) (y1,z0) (y0,z1) (y1,z1)
    assert np.diff(new[1])[0] < 0.5 and np.allclose(np.diff(new[2]), 0.5)
    with pytest.raises(ValueError):
        dd.dlb_update_cuts(cuts, (1, 1, 4), [1, 1, 1, 1], (0, 0, 0.3))


def test_decomposition_across_a_skewed_box_vector_is_refused():
    import gromacs_cosmo_amd as g
    from gromacs_cosmo_amd import dd
    glob = g.system.lj_chain_mixture(2744, ntype=5, seed=7, skew=(6, -5, 4))
    with pytest.raises(ValueError):
        dd.build_domain(glob, g.system.reaction_field(0.9), 0, 2, torch.device("cpu"))
