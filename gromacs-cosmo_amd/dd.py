"""Spatial domain decomposition of the Verlet nonbonded path, one process per GPU: z slabs by default, bricks in two
or three dimensions on request.

The reference decomposes space (mdlib/domdec.c) and computes non-local pairs on the GPU in a
second stream (sim_util.c:1012-1212: local launch, halo x, non-local launch, copy-backs,
halo f).  The path's multi-GPU form here has the same data flow with a simpler zone scheme:

  * the box is cut into grid = (nx, ny, nz) bricks of equal size (default (1, 1, world): z slabs); a dimension that
    is not cut stays periodic inside the search, a cut one gets no periodic shift there (nbnxn_search.c:4906-4927);
  * half shell with direct peers: a rank imports, from each of its "forward" neighbours (offsets (dx, dy, dz) in
    {-1, 0, 1}^3 over the cut dimensions with (dz, dy, dx) > (0, 0, 0): 1, 4 or 13 of them), the atoms within rlist
    of the face, edge or corner it shares with that neighbour.  Each import is one grid zone and one message straight
    from the rank that owns the atoms.  Home x halo pairs are the NON-LOCAL list, home x home pairs the LOCAL list;
    the i-clusters of both are home clusters, so no pair is ever computed between two imported zones.  Every pair of
    atoms in different bricks is seen by exactly one of the two ranks: the one for which the other is forward.
    The reference's eighth shell (dd_zo / dd_zp3, domdec.c:433-449) imports from 7 zones in staged pulses that
    forward corner data through intermediate ranks (domdec.c:667-758) and computes zone x zone pairs; on a node
    whose 8 GPUs are all linked to each other every neighbour is one hop away, the faces carry nearly all of the
    volume in both schemes (13x13x14 tiles on 2x2x2 ranks: 28 % of the home atoms here, 26 % there), and one grouped
    RCCL call moves all messages at once.  Measured on one rank that is its own neighbour: a tie with two cut dimensions,
    the eighth shell 21 % faster with three (7 messages and zones instead of 13): see scheme="eighth" below.  Bricks must be thicker than rlist so that only neighbours interact;
  * scheme="staged": the same zones and lists brought by the reference's pulses, one per cut dimension, later ones forwarding what
    earlier ones brought (_staged_pulses, StagedHalo, nbnxn_hip_dd_set_phases); measured 14-16 % slower than the direct messages;
  * scheme="eighth" is the reference's own zone scheme with the same direct messages: the 1 / 3 / 7 zones of dd_zo, and the
    zone pairs of dd_zp1 / dd_zp2 / dd_zp3 in the lists, so the non-local list also holds pairs BETWEEN imported zones and its
    i-clusters are not all home clusters (what stock mdrun's search hands the kernel under domain decomposition).  The host
    builder (nbnxn_host_make_pairlist_zones) and the device search (nbnxn_hip_make_pairlist_zones) make the same lists;
  * per step: halo coordinates travel to the ranks that import them (plus the box vectors of the periodic
    boundaries they cross), halo forces travel back and are added to the owner's forces -- those that crossed a
    boundary also to the shift force of that crossing, the rule of dd_move_f (domdec.c:837-851); {e_lj, e_el,
    fshift} are summed over ranks.

Three ways to move the halo, one data flow:
  * `RcclHalo` (what bench.py --gpus N uses): the HIP module issues pack kernel, grouped RCCL sends / receives and
    unpack kernel itself on its non-local stream (nbnxn_hip_dd_*, its own communicator);
  * `DeviceHalo`: the module's pack / unpack kernels, the messages through torch.distributed (backend "nccl" = RCCL
    over xGMI on the GPU box, "gloo" in rehearsals on one GPU);
  * `HaloExchange`: torch indexing operations and torch.distributed, on CPU and GPU tensors alike -- what
    tests/test_dd_cpu.py binds to the oracle.
`SlabRun` binds them to the HIP module (device buffers wrapped in place, local / non-local HIP streams).  One rank is
a valid decomposition too: it is its own neighbour, periodic through a halo of its own atoms in the dimensions
named by `self_dims` (z by default; all three give the 13-message step on a one-GPU box).
"""
import os
import sys
from dataclasses import dataclass, field

import numpy as np
import torch
import torch.distributed as dist

from . import abi
from . import system as sysmod
from .host import NbnxnHost, PairList


@dataclass
class Message:
    """One halo message per step and direction: my home atoms `send_sel` go to `send_to`, `n_recv` atoms arrive from
    `recv_from` (its message of the same offset) and form one grid zone here."""
    offset: tuple              # (dx, dy, dz): where the rank I import from sits relative to me
    send_to: int
    recv_from: int
    n_send: int
    n_recv: int
    shift: np.ndarray          # [3], added to the coordinates on the way (periodic boundaries crossed)
    shift_index: int           # the shift force the returned forces count towards (-1: none)


def shift_index(tx, ty, tz):
    """XYZ2IS (types/ishift.h:45-55)"""
    return 5 * (3 * (tz + 1) + ty + 1) + tx + 2


def rank_of(idx, grid):
    return (int(idx[2]) * grid[1] + int(idx[1])) * grid[0] + int(idx[0])


def coords_of(rank, grid):
    return (rank % grid[0], (rank // grid[0]) % grid[1], rank // (grid[0] * grid[1]))


def forward_offsets(decomp, scheme="half"):
    """The zones a rank imports, as offsets (dx, dy, dz) of the rank that owns them.
    "half": the half shell over the cut dimensions, offsets in {-1, 0, 1}^3 with (dz, dy, dx) > (0, 0, 0).
    "eighth": the reference's eighth shell, the zones 1 .. 2^ndim - 1 of dd_zo (domdec.c:432-434) over the cut dimensions
    in x, y, z order; with it the lists also hold pairs between imported zones (eighth_shell_zone_pairs)."""
    if scheme == "eighth":
        dims = [d for d in range(3) if decomp[d]]
        offs = []
        for zo in DD_ZO[1:2 ** len(dims)]:
            o = [0, 0, 0]
            for k, d in enumerate(dims):
                o[d] = zo[k]
            offs.append(tuple(o))
        return offs
    r = [(-1, 0, 1) if d else (0,) for d in decomp]
    return [(dx, dy, dz) for dz in r[2] for dy in r[1] for dx in r[0] if (dz, dy, dx) > (0, 0, 0)]


# The DD zone order and the zone pairs of the reference (mdlib/domdec.c:432-449): {i-zone, first j-zone, end j-zone}.
# i-zones with a non-zero component see only j-zones with that component 0, so every pair of bricks meets on one rank.
DD_ZO = ((0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 1, 1), (0, 0, 1), (1, 0, 1), (1, 1, 1))
DD_ZP = {1: ((0, 0, 2),), 2: ((0, 0, 4), (1, 3, 4)), 3: ((0, 0, 8), (1, 3, 6), (2, 5, 6), (3, 5, 7))}


def eighth_shell_zone_pairs(ndim):
    """(local, non-local) zone pairs of the eighth shell: home x home is the local list, the rest of dd_zp the non-local."""
    nonlocal_ = []
    for iz, j0, j1 in DD_ZP[ndim]:
        if iz == 0:
            nonlocal_.append((0, 1, j1))
        else:
            nonlocal_.append((iz, j0, j1))
    return ((0, 0, 1),), tuple(nonlocal_)


@dataclass
class Domain:
    rank: int
    world: int
    grid: tuple                # ranks along x, y, z
    decomp: tuple              # which dimensions are cut (a single rank cuts the ones it is its own neighbour in)
    box: np.ndarray            # global box (its diagonal)
    box_search: np.ndarray     # the box as the search takes it: [3], or the [3, 3] matrix of a cell skewed in xy
    lo: np.ndarray             # corners of the home brick
    hi: np.ndarray
    nhome: int                 # home atoms (caller order: home first, halo after)
    nhalo: int
    home_ids: np.ndarray       # global atom ids of the home atoms
    halo_ids: np.ndarray       # global atom ids of the imported atoms, in arrival order (message by message)
    host: NbnxnHost
    xq: np.ndarray             # [npad, 4] grid order, home grid first
    type: np.ndarray           # [npad]
    ntype: int
    nbfp: np.ndarray
    shift_vec: np.ndarray
    pl_local: PairList
    pl_nonlocal: PairList
    npad: int
    npad_local: int
    msgs: list                 # [Message], in the order of forward_offsets(decomp)
    send_slots: np.ndarray     # grid slots of the home atoms that go out, message by message
    halo_slots: np.ndarray     # grid slots the imported atoms occupy here, message by message
    # what the device search needs to redo the gridding and both lists (nbnxn_hip_put_on_grid_zones)
    zones: list = None         # [(corner0, corner1, a0, n)], home zone first
    x_all: np.ndarray = None   # caller order: home atoms, then the imported ones
    type_all: np.ndarray = None
    q_all: np.ndarray = None
    excl_index: np.ndarray = None
    excl_a: np.ndarray = None
    rlist: float = 0.0
    min_ci_balanced: int = 0
    scheme: str = "half"     # "half": half shell, home i-clusters only; "eighth": the reference's zones and zone pairs
    zone_pairs: tuple = (None, None)   # (local, non-local) rows {i-zone, first j-zone, end j-zone}; None = home i-clusters only
    balance_history: list = None       # balance_domain: max/mean - 1 of the ranks' cluster-pair counts, per iteration

    @property
    def pair_slots(self):
        return (self.pl_local.nci_tot + self.pl_nonlocal.nci_tot) * 64

    @property
    def pbc_shift_dim(self):
        return tuple(0 if d else 1 for d in self.decomp)


def _staged(t):
    """gloo moves host memory only: device tensors are staged through the host for it (the
    one-GPU rehearsal of tests/test_gpu_z_dd.py); RCCL takes them as they are."""
    return t is not None and t.is_cuda and dist.get_backend() == "gloo"


def _exchange(send, dst, recv, src):
    """Ring step: everybody sends `send` (rows along dim 0) to `dst` and receives `recv` from `src` at once.

    Issued as ONE all_to_all_single whose split sizes are zero except towards dst / from src: on RCCL
    that is a single grouped ncclSend/ncclRecv pair put on the stream from C++ (measured on MI355X,
    CPU cost per call: 14.5 us, against 67 us for batch_isend_irecv of the same two messages).  It is a
    collective call: every rank of the group makes it, also one with nothing to send or receive."""
    if _staged(send) or _staged(recv):
        hs = send.cpu() if send is not None else None
        hr = torch.empty(recv.shape, dtype=recv.dtype) if recv is not None else None
        _exchange(hs, dst, hr, src)
        if recv is not None:
            recv.copy_(hr)
        return
    if send is None and recv is None:
        raise ValueError("an exchange needs at least one buffer (for its dtype and device)")
    like = send if send is not None else recv
    if send is None:
        send = like.new_empty((0,) + tuple(like.shape[1:]))
    if recv is None:
        recv = like.new_empty((0,) + tuple(like.shape[1:]))
    world = dist.get_world_size()
    ins, outs = [0] * world, [0] * world
    ins[dst] = send.shape[0]
    outs[src] = recv.shape[0]
    dist.all_to_all_single(recv, send, output_split_sizes=outs, input_split_sizes=ins)


def _exchange_split(send, in_splits, recv, out_splits):
    """One all_to_all_single with per-rank row counts: all messages of a step and direction at once."""
    if _staged(send) or _staged(recv):
        hs = send.cpu()
        hr = torch.empty(recv.shape, dtype=recv.dtype)
        _exchange_split(hs, in_splits, hr, out_splits)
        recv.copy_(hr)
        return
    dist.all_to_all_single(recv, send, output_split_sizes=out_splits, input_split_sizes=in_splits)


def uniform_cuts(grid):
    """cell boundaries as fractions of the box, per dimension: equal bricks"""
    return tuple(np.linspace(0.0, 1.0, int(grid[d]) + 1) for d in range(3))


DD_CELL_MARGIN = 1.0001     # domdec.c:452
DLB_RELAX = 0.5             # set_dd_cell_sizes_dlb_root, domdec.c:3440
DLB_CHANGE_LIMIT = 0.10     # comm->dlb_scale_lim = 10 %, domdec.c:6654 (GMX_DLB_MAX_BOX_SCALING)


def dlb_update_cuts(cuts, grid, rank_loads, min_frac, relax=DLB_RELAX, change_limit=DLB_CHANGE_LIMIT):
    """One step of the reference's dynamic load balancing (set_dd_cell_sizes_dlb_root, domdec.c:3467-3498, and the
    limits of dd_cell_sizes_dlb_root_enforce_limits, :3221-3300) on a rectilinear grid: along every cut dimension the
    load of a row of cells is the sum over the ranks in it; a cell's thickness changes by -relax * (its relative
    imbalance), all cells of the row scaled down together when the largest change would exceed change_limit; no cell
    gets thinner than min_frac[d] of the box, the others share what is left.  rank_loads: one number per rank (here
    cluster pairs of its two lists, what the kernel's time is proportional to; the reference takes measured force
    time, with a GPU the time waited for it, sim_util.c:1514-1518).  The reference staggers the boundaries of the
    later dimensions per row of the earlier ones; this grid keeps them straight, so every face is still shared by
    exactly two bricks and the direct messages stay as they are."""
    loads = np.asarray(rank_loads, dtype=np.float64).reshape(int(grid[2]), int(grid[1]), int(grid[0]))   # rank_of order
    new = []
    for d in range(3):
        n = int(grid[d])
        old = np.asarray(cuts[d], dtype=np.float64)
        if n == 1:
            new.append(old.copy())
            continue
        row = loads.sum(axis=tuple(a for a in range(3) if a != 2 - d))
        aver = row.sum() / n
        imbalance = (row - aver) / (aver if aver > 0 else 1.0)
        change_max = np.abs(relax * imbalance).max()
        sc = relax * (change_limit / change_max if change_max > change_limit else 1.0)
        size = (old[1:] - old[:-1]) * (1.0 - sc * imbalance)
        minf = float(min_frac[d])
        if n * minf > 1.0:
            raise ValueError(f"{n} cells of at least {minf:.4f} of the box do not fit along {'xyz'[d]}")
        fixed = np.zeros(n, dtype=bool)
        while True:
            free = ~fixed
            scale = (1.0 - fixed.sum() * minf) / size[free].sum()
            out = np.where(fixed, minf, size * scale)
            below = free & (out < minf)
            if not below.any():
                break
            fixed |= below
        b = np.concatenate([[0.0], np.cumsum(out)])
        b[-1] = 1.0
        new.append(b)
    return tuple(new)


def balance_domain(glob, ip, rank, world, device, min_ci_balanced=0, grid=None, self_dims=(2,), scheme="half", iterations=24,
                   tolerance=0.02):
    """Static use of the reference's load-balancing rule: build the domains, gather every rank's cluster-pair count,
    move the cut planes (dlb_update_cuts), repeat; keeps the cut with the smallest maximum.  Every rank computes the
    same planes from the same gathered numbers.  Returns (Domain, history of max/mean - 1 per iteration)."""
    grid = (1, 1, world) if grid is None else tuple(int(v) for v in grid)
    box = np.asarray(glob.box, dtype=np.float64)
    box = np.diag(box) if box.shape == (3, 3) else box
    min_frac = [float(ip.rlist) * DD_CELL_MARGIN * 1.001 / float(box[d]) for d in range(3)]
    cuts, best, history = uniform_cuts(grid), None, []
    for it in range(iterations + 1):
        dom = build_domain(glob, ip, rank, world, device, min_ci_balanced, grid=grid, self_dims=self_dims, scheme=scheme, cuts=cuts)
        mine = torch.tensor([dom.pl_local.nci_tot + dom.pl_nonlocal.nci_tot], dtype=torch.int64)
        if dist.get_backend() != "gloo":
            mine = mine.to(device)
        parts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        loads = np.array([int(p.item()) for p in parts], dtype=np.float64)
        history.append(float(loads.max() / loads.mean() - 1.0))
        if best is None or loads.max() < best[0]:
            best = (loads.max(), dom)
        if it == iterations or history[-1] < tolerance or world == 1:
            break
        cuts = dlb_update_cuts(cuts, grid, loads, min_frac)
    dom = best[1]
    dom.balance_history = history
    return dom, history


def _staged_pulses(decomp, grid, me, rank, box, lo, rl, home_ids, x_home, device):
    """The reference's communication set-up for one pulse per cut dimension (setup_dd_communication, domdec.c:8100-8500,
    with cells thicker than the cut-off; dd_move_x, :667-758): along x, then y, then z, a rank sends the neighbour below
    everything it HOLDS within rlist of its lower face -- home atoms and what the earlier pulses brought -- so the
    pulse along y forwards zone (1,0,0) as the receiver's zone (1,1,0), and the pulse along z forwards three zones.
    The atoms of a pulse travel grouped by the zone they become, in dd_zo order, so the imported atoms end up in zone
    order 1 .. 2^ndim - 1 and the grids and lists are those of the eighth shell with direct messages."""
    dims = [d for d in range(3) if decomp[d]]
    zo = [tuple(v[:len(dims)]) for v in DD_ZO[:2 ** len(dims)]]
    zone_index = {v: k for k, v in enumerate(zo)}
    ids_held, x_held = home_ids.astype(np.int64), x_home.astype(np.float32)
    zone_held = np.zeros(len(home_ids), dtype=np.int64)
    zone_n = {0: len(home_ids)}
    msgs, sels = [], []
    for k, d in enumerate(dims):
        near = np.nonzero((x_held[:, d] - np.float32(lo[d])) < rl)[0]
        becomes = np.array([zone_index[tuple(1 if j == k else zo[z][j] for j in range(len(dims)))] for z in range(len(zo))])[zone_held[near]]
        sel = near[np.argsort(becomes, kind="stable")]
        t = [0, 0, 0]
        t[d] = 1 if me[d] == 0 else 0                          # the receiver sits at the top of the box: + box
        shift = np.array([t[j] * float(box[j]) for j in range(3)], dtype=np.float32)
        o = tuple(1 if j == d else 0 for j in range(3))
        send_to = rank_of([(me[j] - o[j]) % grid[j] for j in range(3)], grid)
        recv_from = rank_of([(me[j] + o[j]) % grid[j] for j in range(3)], grid)
        cnt_send = torch.from_numpy(np.bincount(becomes, minlength=8).astype(np.int64)).to(device)
        cnt_recv = torch.zeros(8, dtype=torch.int64, device=device)
        _exchange(cnt_send, send_to, cnt_recv, recv_from)
        cnt = cnt_recv.cpu().numpy()
        nr = int(cnt.sum())
        ids_recv = torch.zeros(nr, dtype=torch.int64, device=device)
        x_recv = torch.zeros((nr, 3), dtype=torch.float32, device=device)
        _exchange(torch.from_numpy(ids_held[sel]).to(device), send_to, ids_recv, recv_from)
        _exchange(torch.from_numpy(x_held[sel] + shift).to(device), send_to, x_recv, recv_from)
        msgs.append(Message(offset=o, send_to=send_to, recv_from=recv_from, n_send=len(sel), n_recv=nr, shift=shift,
                            shift_index=shift_index(*t) if any(t) else -1))
        sels.append(sel)
        ids_held = np.concatenate([ids_held, ids_recv.cpu().numpy()])
        x_held = np.concatenate([x_held, x_recv.cpu().numpy()])
        zone_held = np.concatenate([zone_held, np.repeat(np.arange(8), cnt)])
        for z in range(8):
            if cnt[z]:
                zone_n[z] = int(cnt[z])
    nhome = len(home_ids)
    zone_parts, a = [], nhome
    for z in range(1, len(zo)):
        off = [0, 0, 0]
        for j, d in enumerate(dims):
            off[d] = zo[z][j]
        n = zone_n.get(z, 0)
        zone_parts.append((tuple(off), n, x_held[a:a + n]))
        a += n
    assert a == len(ids_held) and (np.diff(zone_held) >= 0).all()
    return msgs, sels, [ids_held[nhome:]], [x_held[nhome:]], zone_parts


def build_domain(glob, ip, rank, world, device, min_ci_balanced=0, grid=None, self_dims=(2,), scheme="half", cuts=None):
    """Cuts rank `rank`'s brick out of the global system `glob` (every rank holds the synthetic
    global topology; coordinates of imported atoms still arrive by message, as they would in a
    run), grids home and halo zones, and builds both lists.  grid = ranks along (x, y, z), default z slabs.
    cuts: per dimension the cell boundaries as fractions of the box (uniform_cuts / dlb_update_cuts); None = equal bricks."""
    if world < 1:
        raise ValueError("domain decomposition needs at least one rank")
    grid = (1, 1, world) if grid is None else tuple(int(v) for v in grid)
    if len(grid) != 3 or min(grid) < 1 or grid[0] * grid[1] * grid[2] != world:
        raise ValueError(f"a grid of {grid} ranks does not hold {world} ranks")
    decomp = tuple(grid[d] > 1 or (world == 1 and d in self_dims) for d in range(3))
    # A triclinic cell may be cut along a dimension whose box vector lies along its axis and into which no other box
    # vector reaches (the halo shift is then a plain translation along that axis): along z, box[YY][XX] may be skewed,
    # the search handles it through its x/y shifts.
    box_search = np.asarray(glob.box, dtype=np.float32)
    if box_search.shape == (3, 3):
        m = box_search
        bad = {2: m[2, 0] != 0 or m[2, 1] != 0, 1: m[1, 0] != 0 or m[2, 1] != 0, 0: m[1, 0] != 0 or m[2, 0] != 0}
        for d in range(3):
            if decomp[d] and bad[d]:
                raise ValueError(f"a decomposition along {'xyz'[d]} needs box[{'XYZ'[d]}{'XYZ'[d]}] along its axis and no other box vector reaching into it")
        box = np.ascontiguousarray(np.diag(box_search))
    else:
        box = box_search
    me = coords_of(rank, grid)
    if cuts is None:
        ext = np.array([float(box[d]) / grid[d] for d in range(3)])
        cell3 = [np.clip(np.floor(glob.x[:, d].astype(np.float64) / ext[d]).astype(np.int64), 0, grid[d] - 1) for d in range(3)]
        lo = np.array([me[d] * ext[d] for d in range(3)])
        hi = lo + ext
        thinnest = ext
    else:
        # cut planes moved by the load balancing: the same planes for every row of cells (a rectilinear grid)
        bnd = []
        for d in range(3):
            b = np.asarray(cuts[d], dtype=np.float64)
            if b.shape != (grid[d] + 1,) or b[0] != 0.0 or b[-1] != 1.0 or not (np.diff(b) > 0).all():
                raise ValueError(f"cuts along {'xyz'[d]} must be {grid[d] + 1} increasing fractions from 0 to 1")
            bnd.append(b * float(box[d]))
        cell3 = [np.clip(np.searchsorted(bnd[d], glob.x[:, d].astype(np.float64), side="right") - 1, 0, grid[d] - 1) for d in range(3)]
        lo = np.array([bnd[d][me[d]] for d in range(3)])
        hi = np.array([bnd[d][me[d] + 1] for d in range(3)])
        thinnest = np.array([np.diff(bnd[d]).min() for d in range(3)])
    for d in range(3):
        # one rank is its own neighbour: periodic through a halo of its own atoms, which needs room for both
        if decomp[d] and thinnest[d] <= ip.rlist * (2 if grid[d] == 1 else 1):
            raise ValueError(f"brick thickness {thinnest[d]:.3f} nm along {'xyz'[d]} must exceed {'2 x ' if grid[d] == 1 else ''}rlist {ip.rlist:.3f} nm")
    owner = (cell3[2] * grid[1] + cell3[1]) * grid[0] + cell3[0]
    home_ids = np.nonzero(owner == rank)[0]
    nhome = len(home_ids)
    x_home = glob.x[home_ids]
    rl = np.float32(ip.rlist)

    msgs, sels, halo_id_parts, halo_x_parts = [], [], [], []
    if scheme not in ("half", "eighth", "staged"):
        raise ValueError(f"unknown decomposition scheme {scheme!r}")
    zone_parts = None
    if scheme == "staged":
        msgs, sels, halo_id_parts, halo_x_parts, zone_parts = _staged_pulses(decomp, grid, me, rank, box, lo, rl, home_ids, x_home, device)
    for o in (forward_offsets(decomp, scheme) if scheme != "staged" else ()):
        # what the rank at me - o imports from me: my atoms within rlist of the face / edge / corner we share
        sel = np.ones(nhome, dtype=bool)
        t = [0, 0, 0]
        for d in range(3):
            if o[d] == 1:
                sel &= (x_home[:, d] - np.float32(lo[d])) < rl
                t[d] = 1 if me[d] == 0 else 0                    # it sits at the top of the box: + box
            elif o[d] == -1:
                sel &= (np.float32(hi[d]) - x_home[:, d]) <= rl
                t[d] = -1 if me[d] == grid[d] - 1 else 0         # it sits at the bottom of the box: - box
        sel = np.nonzero(sel)[0]
        shift = np.array([t[d] * float(box[d]) for d in range(3)], dtype=np.float32)
        send_to = rank_of([(me[d] - o[d]) % grid[d] for d in range(3)], grid)
        recv_from = rank_of([(me[d] + o[d]) % grid[d] for d in range(3)], grid)
        n_send = torch.tensor([len(sel)], dtype=torch.int64, device=device)
        n_recv = torch.zeros(1, dtype=torch.int64, device=device)
        _exchange(n_send, send_to, n_recv, recv_from)
        nr = int(n_recv.item())
        ids_send = torch.from_numpy(home_ids[sel].astype(np.int64)).to(device)
        x_send = torch.from_numpy(x_home[sel] + shift).to(device)
        ids_recv = torch.zeros(nr, dtype=torch.int64, device=device)
        x_recv = torch.zeros((nr, 3), dtype=torch.float32, device=device)
        _exchange(ids_send, send_to, ids_recv, recv_from)
        _exchange(x_send, send_to, x_recv, recv_from)
        msgs.append(Message(offset=o, send_to=send_to, recv_from=recv_from, n_send=len(sel), n_recv=nr, shift=shift,
                            shift_index=shift_index(*t) if any(t) else -1))
        sels.append(sel)
        halo_id_parts.append(ids_recv.cpu().numpy())
        halo_x_parts.append(x_recv.cpu().numpy())
    halo_ids = np.concatenate(halo_id_parts) if halo_id_parts else np.zeros(0, np.int64)
    x_halo = np.concatenate(halo_x_parts) if halo_x_parts else np.zeros((0, 3), np.float32)
    nhalo = len(halo_ids)

    # grids: zone 0 = the home brick, then one zone per message that brought atoms: the rlist-thick layer beyond the
    # face (edge, corner) it came across
    pbc_shift_dim = tuple(0 if d else 1 for d in decomp)
    host = NbnxnHost(box_search, pbc_shift_dim=pbc_shift_dim)
    c0 = [float(lo[d]) if decomp[d] else 0.0 for d in range(3)]
    c1 = [float(hi[d]) if decomp[d] else float(box[d]) for d in range(3)]
    zones = [(tuple(c0), tuple(c1), 0, nhome)]
    host.put_on_grid(0, zones[0][0], zones[0][1], 0, x_home)
    a0 = nhome
    if zone_parts is None:
        zone_parts = [(m.offset, m.n_recv, xh) for m, xh in zip(msgs, halo_x_parts)]      # one zone per message
    for off, nz, xh in zone_parts:
        if nz == 0:
            if scheme != "half":
                raise ValueError(f"zone {off} of the eighth shell is empty: the zone pairs need every zone on the grid")
            continue
        z0 = [c1[d] if off[d] == 1 else (c0[d] - float(rl) if off[d] == -1 else c0[d]) for d in range(3)]
        z1 = [c1[d] + float(rl) if off[d] == 1 else (c0[d] if off[d] == -1 else c1[d]) for d in range(3)]
        zones.append((tuple(z0), tuple(z1), a0, nz))
        host.put_on_grid(len(zones) - 1, zones[-1][0], zones[-1][1], a0, xh)
        a0 += nz
    if len(zones) > abi.MAX_ZONES:
        raise ValueError(f"{len(zones)} grid zones, the search takes {abi.MAX_ZONES}")
    ids_all = np.concatenate([home_ids, halo_ids])
    x_all = np.concatenate([x_home, x_halo]).astype(np.float32)
    ntype = glob.ntype + 1
    xq, typ = host.set_atomdata(ntype, glob.type[ids_all], glob.q[ids_all])
    host.copy_x_to_xq(x_all, xq)

    # topology exclusions in local numbering; partners that are not on this rank drop out.  A partner maps to every
    # copy of it that this rank holds: its home atom and its halo images (several when a rank is the neighbour on more
    # than one side; an exclusion between copies that are out of range of each other is never looked at).
    nloc = len(ids_all)
    by_id = np.argsort(ids_all, kind="stable")
    ids_sorted = ids_all[by_id]
    cnt = (glob.excl_index[1:] - glob.excl_index[:-1])[ids_all]
    starts = glob.excl_index[:-1][ids_all]
    flat = np.repeat(starts - np.concatenate([[0], np.cumsum(cnt)[:-1]]), cnt) + np.arange(int(cnt.sum()))
    row = np.repeat(np.arange(nloc), cnt)
    gp = glob.excl_a[flat]
    first = np.searchsorted(ids_sorted, gp, side="left")
    ncopy = np.searchsorted(ids_sorted, gp, side="right") - first
    rows = np.repeat(row, ncopy)
    within = np.arange(int(ncopy.sum())) - np.repeat(np.concatenate([[0], np.cumsum(ncopy)[:-1]]), ncopy)
    partners = by_id[np.repeat(first, ncopy) + within]
    order = np.lexsort((partners, rows))
    rows, partners = rows[order], partners[order]
    excl_index = np.zeros(nloc + 1, dtype=np.int32)
    np.add.at(excl_index, rows + 1, 1)
    excl_index = np.cumsum(excl_index).astype(np.int32)
    excl_a = partners.astype(np.int32)

    # half shell: home x home and home x every imported zone; eighth shell: the reference's zone pairs, with pairs
    # between imported zones in the non-local list (its i-clusters are then not all home clusters)
    zp_local, zp_nonlocal = (None, None) if scheme == "half" else eighth_shell_zone_pairs(sum(decomp))
    pl_local = host.make_pairlist(abi.LOCAL, ip.rlist, xq, excl_index, excl_a, min_ci_balanced, zone_pairs=zp_local)
    pl_nonlocal = host.make_pairlist(abi.NONLOCAL, ip.rlist, xq, excl_index, excl_a, min_ci_balanced, zone_pairs=zp_nonlocal)
    cell = host.cell_index()
    send_slots = np.concatenate([cell[sel] for sel in sels]).astype(np.int64) if sels else np.zeros(0, np.int64)
    return Domain(rank=rank, world=world, grid=grid, decomp=decomp, box=box, box_search=box_search, lo=lo, hi=hi, nhome=nhome, nhalo=nhalo,
                  home_ids=home_ids, halo_ids=halo_ids, host=host, xq=xq, type=typ, ntype=ntype, nbfp=glob.nbfp(),
                  shift_vec=host.shift_vec(), pl_local=pl_local, pl_nonlocal=pl_nonlocal,
                  npad=len(xq), npad_local=host.natoms_padded_local, msgs=msgs,
                  send_slots=send_slots, halo_slots=cell[nhome:nhome + nhalo].astype(np.int64),
                  zones=zones, x_all=x_all, type_all=glob.type[ids_all].astype(np.int32), q_all=glob.q[ids_all].astype(np.float32),
                  excl_index=excl_index, excl_a=excl_a, rlist=float(ip.rlist), min_ci_balanced=int(min_ci_balanced), scheme=scheme,
                  zone_pairs=(zp_local, zp_nonlocal))


class _Layout:
    """Row orders of the per-step buffers.  The slot lists of a Domain are message by message; one all_to_all_single
    wants the rows it sends ordered by destination rank and delivers the rows it receives ordered by source rank.
    Messages between the same two ranks keep their message order on both sides (both count them by the same offset)."""

    def __init__(self, dom):
        ms = dom.msgs
        s_start = np.concatenate([[0], np.cumsum([m.n_send for m in ms])]).astype(np.int64)
        r_start = np.concatenate([[0], np.cumsum([m.n_recv for m in ms])]).astype(np.int64)
        self.out_order = sorted(range(len(ms)), key=lambda k: (ms[k].send_to, k))      # coordinates out / forces in
        self.in_order = sorted(range(len(ms)), key=lambda k: (ms[k].recv_from, k))     # coordinates in / forces out
        cat = lambda parts: np.concatenate(parts).astype(np.int64) if parts else np.zeros(0, np.int64)
        self.send_slots = cat([dom.send_slots[s_start[k]:s_start[k + 1]] for k in self.out_order])
        self.halo_slots = cat([dom.halo_slots[r_start[k]:r_start[k + 1]] for k in self.in_order])
        self.send_rows = np.zeros(dom.world, dtype=np.int64)     # rows per rank: coordinates out = forces in
        self.recv_rows = np.zeros(dom.world, dtype=np.int64)     # coordinates in = forces out
        self.out_seg, off = [], 0                                 # (message, first row, rows) in the send-side buffers
        for k in self.out_order:
            self.send_rows[ms[k].send_to] += ms[k].n_send
            self.out_seg.append((k, off, ms[k].n_send))
            off += ms[k].n_send
        for k in self.in_order:
            self.recv_rows[ms[k].recv_from] += ms[k].n_recv
        self.send_rows, self.recv_rows = [int(v) for v in self.send_rows], [int(v) for v in self.recv_rows]
        self.n_send, self.n_recv = int(s_start[-1]), int(r_start[-1])


class HaloExchange:
    """The per-step messages, on tensors that alias the engine's coordinate and force arrays:
    xq_t [npad, 4] and f_t [npad, 3] (grid order).  All messages of a direction go out in one all_to_all_single."""

    def __init__(self, dom, xq_t, f_t):
        dev = xq_t.device
        self.dom, self.xq_t, self.f_t = dom, xq_t, f_t
        self.lay = lay = _Layout(dom)
        self.send_slots = torch.from_numpy(lay.send_slots).to(dev)
        self.halo_slots = torch.from_numpy(lay.halo_slots).to(dev)
        shift_rows = np.zeros((lay.n_send, 4), dtype=np.float32)
        for k, off, n in lay.out_seg:
            shift_rows[off:off + n, :3] = dom.msgs[k].shift
        self.shift_rows = torch.from_numpy(shift_rows).to(dev)
        self.x_out = torch.zeros((lay.n_send, 4), dtype=torch.float32, device=dev)
        self.x_in = torch.zeros((lay.n_recv, 4), dtype=torch.float32, device=dev)
        self.f_out = torch.zeros((lay.n_recv, 3), dtype=f_t.dtype, device=dev)
        self.f_in = torch.zeros((lay.n_send, 3), dtype=f_t.dtype, device=dev)
        # forces that come back across a periodic boundary count towards the shift force of that crossing, as
        # dd_move_f does for shifted pulses (mdlib/domdec.c:837-851): the virial needs  shift vector (x) their sum
        self.wrap_seg = [(dom.msgs[k].shift_index, off, n) for k, off, n in lay.out_seg if dom.msgs[k].shift_index >= 0]
        self.fshift_wrap = torch.zeros((abi.SHIFTS, 3), dtype=torch.float64, device=dev)

    def send_x(self):
        """home {x,y,z,q} rows go to the ranks that import them; mine arrive from the forward neighbours."""
        torch.index_select(self.xq_t, 0, self.send_slots, out=self.x_out)
        self.x_out += self.shift_rows
        _exchange_split(self.x_out, self.lay.send_rows, self.x_in, self.lay.recv_rows)
        self.xq_t.index_copy_(0, self.halo_slots, self.x_in)

    def return_f(self):
        """forces on imported atoms go back to their owner, who adds them."""
        torch.index_select(self.f_t, 0, self.halo_slots, out=self.f_out)
        _exchange_split(self.f_out, self.lay.recv_rows, self.f_in, self.lay.send_rows)
        self.f_t.index_add_(0, self.send_slots, self.f_in)
        self.fshift_wrap.zero_()
        for si, off, n in self.wrap_seg:
            self.fshift_wrap[si] += self.f_in[off:off + n].to(torch.float64).sum(0)

    def add_wrap_fshift(self, out_t):
        """the same term, added straight into accumulation slot 0 of the engine's output buffer"""
        if self.wrap_seg:
            out_t[2:2 + 3 * abi.SHIFTS] += self.fshift_wrap.reshape(-1).to(out_t.dtype)


class _PulseView:
    """What the exchange classes read of a Domain, for ONE of its messages"""

    def __init__(self, dom, k):
        s0, r0 = sum(m.n_send for m in dom.msgs[:k]), sum(m.n_recv for m in dom.msgs[:k])
        self.msgs, self.world, self.rank = [dom.msgs[k]], dom.world, dom.rank
        self.send_slots = dom.send_slots[s0:s0 + dom.msgs[k].n_send]
        self.halo_slots = dom.halo_slots[r0:r0 + dom.msgs[k].n_recv]


class StagedHalo:
    """The reference's staged pulses on top of either exchange class (dd_move_x, domdec.c:667-758; dd_move_f, :784-870):
    one exchange per pulse, coordinates in pulse order -- a later pulse sends slots an earlier one filled --, forces in
    reverse order, so that what comes back for a forwarded atom is in its slot before that slot's own pulse returns it.
    make(view) builds the one-message exchange of a pulse (HaloExchange or DeviceHalo)."""

    def __init__(self, make, dom):
        self.parts = [make(_PulseView(dom, k)) for k in range(len(dom.msgs))]
        self.halo_slots = torch.cat([p.halo_slots.to(torch.int64) for p in self.parts])
        self.wrap_seg = [s for p in self.parts for s in getattr(p, "wrap_seg", [])]

    def send_x(self):
        for p in self.parts:
            p.send_x()

    def return_f(self):
        for p in reversed(self.parts):
            p.return_f()

    @property
    def fshift_wrap(self):
        return sum(p.fshift_wrap for p in self.parts)

    def add_wrap_fshift(self, out_t):
        for p in self.parts:
            p.add_wrap_fshift(out_t)


class DeviceHalo:
    """The same messages with the buffers packed and unpacked by the HIP module's halo kernels
    (nbnxn_hip_halo_*) on its non-local stream, and the shift-force term of a periodic crossing added by the
    kernel that adds the returned forces (one pack and one unpack launch per message on the sending side, one
    launch for all on the importing side).  The exchange itself goes out on torch's CURRENT stream, which the
    caller sets to the module's non-local stream (SlabRun.step)."""

    def __init__(self, nb, dom, dev):
        import ctypes as C
        self.nb, self.dom = nb, dom
        self.lay = lay = _Layout(dom)
        self.send_slots = torch.from_numpy(lay.send_slots.astype(np.int32)).to(dev)
        self.halo_slots = torch.from_numpy(lay.halo_slots.astype(np.int32)).to(dev)
        self.n_send, self.n_halo = lay.n_send, lay.n_recv
        self.x_out = torch.zeros((self.n_send, 4), dtype=torch.float32, device=dev)
        self.x_in = torch.zeros((self.n_halo, 4), dtype=torch.float32, device=dev)
        self.f_out = torch.zeros((self.n_halo, 3), dtype=torch.float32, device=dev)
        self.f_in = torch.zeros((self.n_send, 3), dtype=torch.float32, device=dev)
        self._p = [t.data_ptr() for t in (self.send_slots, self.halo_slots, self.x_out, self.x_in, self.f_out, self.f_in)]
        # (rows, first row, shift or None, shift index) of every message with something to send
        self.seg = [(n, off, (C.c_float * 3)(*[float(v) for v in dom.msgs[k].shift]) if dom.msgs[k].shift_index >= 0 else None,
                     dom.msgs[k].shift_index) for k, off, n in lay.out_seg if n > 0]

    def send_x(self):
        p = self._p
        for n, off, shift, _ in self.seg:
            self.nb.halo_pack_x(p[0] + 4 * off, n, shift, p[2] + 16 * off, abi.NONLOCAL)
        _exchange_split(self.x_out, self.lay.send_rows, self.x_in, self.lay.recv_rows)
        self.nb.halo_unpack_x(p[1], self.n_halo, p[3], abi.NONLOCAL)

    def return_f(self):
        p = self._p
        self.nb.halo_pack_f(p[1], self.n_halo, p[4], abi.NONLOCAL)
        _exchange_split(self.f_out, self.lay.recv_rows, self.f_in, self.lay.send_rows)
        for n, off, _, si in self.seg:
            self.nb.halo_unpack_f(p[0] + 4 * off, n, p[5] + 12 * off, si, abi.NONLOCAL)


class RcclHalo:
    """The same messages issued by the HIP module itself (nbnxn_hip_dd_*): one pack kernel, one grouped set of
    ncclSend / ncclRecv between devices, one unpack kernel, all on the module's non-local stream from ONE call
    through the C ABI, however many messages there are; the slot reduction is the module's ncclAllReduce.  No torch
    stream is involved, so the step needs no set_stream and no hand-over between torch's collective stream and the
    module's.  The communicator is the module's own: rank 0 makes the RCCL id, torch.distributed only carries its
    128 bytes."""

    def __init__(self, nb, dom, dev):
        self.nb, self.dom = nb, dom
        cuda_group = dist.get_backend() != "gloo"
        idt = torch.zeros(abi.DD_UNIQUE_ID_BYTES, dtype=torch.uint8, device=dev if cuda_group else "cpu")
        if dom.rank == 0:
            idt.copy_(torch.frombuffer(bytearray(nb.dd_get_unique_id()), dtype=torch.uint8))
        dist.broadcast(idt, 0)
        nb.dd_init(bytes(idt.cpu().numpy().tobytes()), dom.rank, dom.world)
        self.halo_slots = torch.from_numpy(dom.halo_slots.astype(np.int64)).to(dev)
        nb.dd_set_messages([(m.send_to, m.recv_from, m.n_send, m.n_recv, m.shift_index, m.shift) for m in dom.msgs],
                           dom.send_slots, dom.halo_slots)
        if dom.scheme == "staged":
            nb.dd_set_phases(list(range(1, len(dom.msgs) + 1)))      # every pulse a phase of its own (nbnxn_hip_dd_set_phases)

    def send_x(self):
        self.nb.dd_move_x()

    def return_f(self):
        self.nb.dd_move_f()


def reduce_slots(out_t):
    """Per-step form: the raw accumulation slots summed over ranks in place, ONE collective and no
    arithmetic kernels; fold_slots() turns them into {e_lj, e_el, fshift} when somebody asks."""
    if _staged(out_t):
        h = out_t.cpu()
        dist.all_reduce(h)
        out_t.copy_(h)
    else:
        dist.all_reduce(out_t)


def fold_slots(out_t):
    return out_t.view(-1, abi.OUT_STRIDE).to(torch.float64).sum(0)


def fold_and_reduce(out_t, halo=None):
    """{e_lj, e_el, fshift[45][3]} summed over the accumulation slots in double, then over ranks."""
    e = out_t.view(-1, abi.OUT_STRIDE).to(torch.float64).sum(0)
    if halo is not None and halo.wrap_seg:
        e[2:2 + 3 * abi.SHIFTS] += halo.fshift_wrap.reshape(-1)
    if _staged(e):
        h = e.cpu()
        dist.all_reduce(h)
        e.copy_(h)
    else:
        dist.all_reduce(e)
    return e


class _DevArray:
    """A raw device pointer as a __cuda_array_interface__ object, so torch aliases the HIP module's buffers."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f4", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class SlabRun:
    """The multi-GPU workload of bench.py: water_box(tx * nx, ty * ny, tz * nz) (weak scaling) or water_box(tx, ty, tz)
    (strong scaling, BASELINE configs[3]) cut into grid = (nx, ny, nz) bricks of equal size, default `world` z slabs."""

    def __init__(self, tiles, model, ip, rank, world, local_rank, flags, min_ci_balanced=None, device_search=True,
                 device_halo=True, rccl_direct=False, strong=False, grid=None, self_dims=(2,), scheme="half", balance=0, film=None):
        """balance: iterations of the reference's load-balancing rule on the ranks' cluster-pair counts before the run
        (balance_domain; 0 = equal bricks).  film = (liquid_fraction, vapour_keep): system.water_film instead of the full
        box, the workload that needs it.  device_halo: halo buffers packed by the module's kernels (else torch indexing operations), messages
        through torch.distributed.  rccl_direct: everything, messages included, issued by the module on its own
        RCCL communicator (needs one GPU per rank)."""
        from . import gpu
        self.flags = flags
        self.dev = torch.device("cuda", local_rank)
        # weak scaling: `tiles` per rank, stacked along z; strong scaling: `tiles` is the whole box, cut into `world` slabs
        grid = (1, 1, world) if grid is None else tuple(int(v) for v in grid)
        whole = [tiles[d] * (1 if strong else grid[d]) for d in range(3)]
        glob = sysmod.water_box(*whole, model=model) if film is None else sysmod.water_film(*whole, film[0], film[1], model=model)
        self.glob = glob
        self.natoms_global = glob.natoms
        self.nb = nb = gpu.NbnxnHip(local_rank, local_and_nonlocal=True)
        mcb = nb.min_ci_balanced() if min_ci_balanced is None else min_ci_balanced
        verbose = os.environ.get("NBNXN_DD_VERBOSE") == "1"
        import time as _time
        _t0 = _time.perf_counter()

        def say(what):
            if verbose:
                print(f"[dd rank {rank}] {_time.perf_counter() - _t0:6.1f} s  {what}", file=sys.stderr, flush=True)
        if balance > 0 and world > 1:
            self.dom = dom = balance_domain(glob, ip, rank, world, self.dev, mcb, grid=grid, self_dims=self_dims, scheme=scheme,
                                            iterations=balance)[0]
        else:
            self.dom = dom = build_domain(glob, ip, rank, world, self.dev, mcb, grid=grid, self_dims=self_dims, scheme=scheme)
        say(f"domain built: {dom.nhome} home + {dom.nhalo} halo atoms, lists {dom.pl_local.nsci} / {dom.pl_nonlocal.nsci} entries")
        nb.init_const(ip, dom.ntype, dom.nbfp)
        if device_search:
            # grids and both lists made on the device; the host-side builder's results, which the
            # halo bookkeeping above came from, double as the check that both see the same grid
            npad = nb.put_on_grid_zones(dom.box_search, dom.zones, dom.x_all, dom.type_all, dom.q_all, pbc_shift_dim=dom.pbc_shift_dim)
            assert npad == dom.npad
            nb.make_pairlist(dom.rlist, dom.excl_index, dom.excl_a, dom.min_ci_balanced, iloc=abi.LOCAL, zone_pairs=dom.zone_pairs[0])
            if len(dom.zones) > 1:
                nb.make_pairlist(dom.rlist, None, None, dom.min_ci_balanced, iloc=abi.NONLOCAL, zone_pairs=dom.zone_pairs[1])
            for iloc, pl in ((abi.LOCAL, dom.pl_local), (abi.NONLOCAL, dom.pl_nonlocal))[:len(dom.zones)]:
                got = nb.download_pairlist(iloc)
                assert (got.nsci, got.ncj4, got.nexcl, got.nci_tot) == (pl.nsci, pl.ncj4, pl.nexcl, pl.nci_tot)
                assert np.array_equal(got.sci, pl.sci)
        else:
            nb.init_atomdata(dom.npad, dom.npad_local, dom.ntype, dom.type)
            nb.set_atom_order(dom.nhome + dom.nhalo, dom.host.atom_order(), dom.host.cell_index())   # for copy_x_to_xq / add_f
            nb.upload_shiftvec(dom.shift_vec)
            nb.init_pairlist(dom.pl_local, abi.LOCAL)
            nb.init_pairlist(dom.pl_nonlocal, abi.NONLOCAL)
            assert nb.verify_pairlist(dom.pl_local, abi.LOCAL) == 0
            assert nb.verify_pairlist(dom.pl_nonlocal, abi.NONLOCAL) == 0
        say("lists on the device")
        # first pass the reference way: host xq in, forces out, both lists pruned, streams drained
        self.f_first, self.e_first = nb.compute_both(dom.xq, flags)
        say("first pass done")
        d_xq, _, _ = nb.device_buffers()
        self.xq_t = torch.as_tensor(_DevArray(d_xq, (dom.npad, 4)), device=self.dev)
        # the clear is folded into the kernel launches: two output buffers alternate, so the
        # tensors that alias {f, energy slots} are looked up by address after every clear_outputs
        nb.set_clear_in_kernel(True)
        self._views = {}
        self.f_t, self.out_t = self._current_views()
        if rccl_direct:
            # every rank checks on its own that the module can reach RCCL (library found, symbols bound, an id made),
            # and the ranks agree before any of them enters the collective set-up: a rank without it would leave the
            # others waiting.  Without it the messages go through torch.distributed.
            ok = 1
            try:
                nb.dd_get_unique_id()
            except gpu.HipError as err:
                ok = 0
                print(f"rank {rank}: the module's RCCL path is not available ({err}); using torch.distributed", file=sys.stderr, flush=True)
            t = torch.tensor([ok], dtype=torch.int32, device=self.dev if dist.get_backend() != "gloo" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            rccl_direct = bool(int(t.item()))
        self.device_halo, self.rccl_direct = device_halo, rccl_direct
        self._reduction_owed = False
        self.exchange_beside_local = nb.reserved_cus() > 0 and os.environ.get("NBNXN_DD_X_FIRST") != "1"
        self.one_call = os.environ.get("NBNXN_DD_STEP_CALLS") != "many"
        if rccl_direct:
            # The module's communicator and its first exchange, checked: the exchange must bring exactly the coordinates
            # the set-up brought through torch.distributed.  A rank whose RCCL call returns an error, or whose halo comes
            # out different, takes every rank back to torch.distributed (the ranks agree first).
            # What the fall-back can and cannot recover from: an error RETURNED by a call (library not found, a refused
            # argument) and a wrong halo are local facts a rank can report, and the ranks then agree.  A peer that never
            # posts its side of the exchange is not: RCCL's kernel then never ends, nothing on this stream can follow,
            # and pretending to fall back would only move the hang.  So the wait for the first exchange has a deadline,
            # and a rank that runs into it ends the run with an error.
            ok = 1
            want = slots = None
            try:
                self.halo = RcclHalo(nb, dom, self.dev)
                slots = self.halo.halo_slots
                want = self.xq_t[slots].clone()
                self.xq_t[slots, :3] = 1e6
                torch.cuda.synchronize(self.dev)          # torch's stream wrote; the module's non-local stream exchanges
                self.halo.send_x()
                done = torch.cuda.Event()
                done.record(torch.cuda.ExternalStream(nb.stream(abi.NONLOCAL), device=self.dev))
                t_end = _time.perf_counter() + float(os.environ.get("NBNXN_DD_FIRST_EXCHANGE_TIMEOUT_S", "60"))
                while not done.query():
                    if _time.perf_counter() > t_end:
                        raise SystemExit(f"rank {rank}: the module's first halo exchange over RCCL did not complete (a peer never posted "
                                         "its side, or RCCL cannot reach it): no fall-back is possible from here")
                    _time.sleep(0.001)
                torch.cuda.synchronize(self.dev)
                if not torch.equal(self.xq_t[slots], want):
                    ok = 0
                    print(f"rank {rank}: the module's first halo exchange over RCCL brought other coordinates than the set-up; "
                          "using torch.distributed", file=sys.stderr, flush=True)
            except gpu.HipError as err:
                ok = 0
                print(f"rank {rank}: the module's RCCL exchange failed ({err}); using torch.distributed", file=sys.stderr, flush=True)
            if not ok and want is not None:
                torch.cuda.synchronize(self.dev)
                self.xq_t[slots] = want                   # whatever went wrong after the halo rows were wrecked: put them back
                torch.cuda.synchronize(self.dev)
            t = torch.tensor([ok], dtype=torch.int32, device=self.dev if dist.get_backend() != "gloo" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            if not int(t.item()):
                try:
                    nb.dd_finish()
                except gpu.HipError:
                    pass
                self.rccl_direct = rccl_direct = False
        if not rccl_direct:
            make = (lambda d: DeviceHalo(nb, d, self.dev)) if device_halo else (lambda d: HaloExchange(d, self.xq_t, self.f_t))
            self.halo = StagedHalo(make, dom) if dom.scheme == "staged" else make(dom)
        self.s_loc = torch.cuda.ExternalStream(nb.stream(abi.LOCAL), device=self.dev)
        self.s_nl = torch.cuda.ExternalStream(nb.stream(abi.NONLOCAL), device=self.dev)
        self.s_default = torch.cuda.default_stream(self.dev)

    def _current_views(self):
        _, d_f, _ = self.nb.device_buffers()
        v = self._views.get(d_f)
        if v is None:
            d_out, nslots, stride = self.nb.output_slots()
            assert (nslots, stride) == (abi.OUT_SLOTS, abi.OUT_STRIDE)
            v = (torch.as_tensor(_DevArray(d_f, (self.dom.npad, 3)), device=self.dev),
                 torch.as_tensor(_DevArray(d_out, (abi.OUT_SLOTS * abi.OUT_STRIDE,)), device=self.dev))
            self._views[d_f] = v
        return v

    @property
    def pair_slots_local(self):
        return self.dom.pair_slots

    def describe(self):
        d = self.dom
        route = {"halo_route": "module-rccl" if self.rccl_direct else ("torch+module-kernels" if self.device_halo else "torch"),
                 "rccl_ranks": self.nb.dd_comm_count() if self.rccl_direct else None}
        return {**route, "dd_grid": list(d.grid), "halo_messages": len(d.msgs), "halo_zones": len(d.zones) - 1,
                "atoms_rank0": d.nhome, "halo_atoms_rank0": d.nhalo, "nsci_local": d.pl_local.nsci,
                "nsci_nonlocal": d.pl_nonlocal.nsci, "cluster_pairs_rank0": d.pl_local.nci_tot + d.pl_nonlocal.nci_tot,
                **({"load_imbalance_per_balancing_step": [round(v, 3) for v in d.balance_history]} if d.balance_history else {})}

    def step(self, ev=None):
        nb, fl = self.nb, self.flags
        if self.rccl_direct:
            return self._step_rccl_direct(ev)
        if not self.device_halo:
            nb.clear_outputs(fl)                          # a switch of output buffers (or a fill the first time)
            return self._step_torch_halo(ev)
        # the same order of collectives as _step_rccl_direct, for the same reason (torch's RCCL stream also runs them in
        # issue order): this step's coordinate exchange, then the last step's slot reduction, then the kernels.
        # The collectives follow torch's CURRENT stream; set_stream costs a third of the context manager.
        torch.cuda.set_stream(self.s_nl)
        self.halo.send_x()                                # pack, exchange, unpack
        self.flush()                                      # local stream: the last step's slot reduction, if owed
        nb.clear_outputs(fl)
        if ev:
            ev[0].record(self.s_loc)
        nb.launch_kernel_resident(fl, abi.LOCAL)          # local stream; also zeroes the idle output buffer
        if ev:
            ev[1].record(self.s_loc)
        nb.launch_kernel_resident(fl, abi.NONLOCAL)
        torch.cuda.set_stream(self.s_nl)
        self.halo.return_f()                              # atomic adds: no need to wait for the local kernel
        nb.stream_wait(abi.LOCAL, abi.NONLOCAL)           # ONE cross-stream wait per step
        torch.cuda.set_stream(self.s_default)
        self._reduction_owed = bool(fl & abi.FORCE_VIRIAL)

    def _step_rccl_direct(self, ev=None):
        """The same step with the module issuing its own RCCL calls, in an order chosen for what RCCL's kernels need.

        RCCL runs one communicator's operations in the order they were issued, and its send / receive kernel cannot
        start once the local kernel holds every wave slot.  So the coordinate exchange of THIS step
        is issued before the slot reduction of the LAST one: it runs on an idle chip right after the last step's forces
        came back, the reduction follows it on the local stream, and the local kernel follows the reduction.  Issued the
        other way round the exchange would become runnable at the same moment as the local kernel and could lose the
        chip to it for the whole kernel.  The reduction is off the path to the next coordinates, as in an MD step, and
        flush() issues the one still owed before anybody reads the energies."""
        nb, fl = self.nb, self.flags
        beside = self.exchange_beside_local               # the local stream leaves CUs free: RCCL's kernel finds room
        if self.one_call:
            # the same sequence from one call through the C ABI (nbnxn_hip_dd_step): the host's issue time limits the step
            h0 = h1 = None
            if ev:
                for e in ev:
                    if not e.cuda_event:
                        e.record(self.s_loc)              # torch makes the event on its first record
                h0, h1 = ev[0].cuda_event, ev[1].cuda_event
            nb.dd_step(fl, self._reduction_owed, beside, h0, h1)
            self._reduction_owed = bool(fl & abi.FORCE_VIRIAL)
            return
        if not beside:
            nb.dd_move_x()                                # non-local stream, behind the last step's returned forces
        self.flush()                                      # local stream: the last step's slot reduction, if owed
        nb.clear_outputs(fl)                              # a switch of output buffers (or a fill the first time)
        if ev:
            ev[0].record(self.s_loc)
        nb.launch_kernel_resident(fl, abi.LOCAL)          # also zeroes the idle output buffer (the one just reduced)
        if ev:
            ev[1].record(self.s_loc)
        if beside:
            nb.dd_move_x()                                # non-local stream, beside the local kernel
        nb.launch_kernel_resident(fl, abi.NONLOCAL)       # beside the local kernel, once the halo has arrived
        nb.dd_move_f()
        nb.stream_wait(abi.LOCAL, abi.NONLOCAL)           # ONE cross-stream wait: what follows on the local stream sees all of it
        self._reduction_owed = bool(fl & abi.FORCE_VIRIAL)

    def flush(self):
        """issues what a step left for the next one to issue: the slot reduction of the last step"""
        if self._reduction_owed:
            if self.rccl_direct:
                self.nb.dd_sum_outputs(abi.LOCAL)
            else:
                torch.cuda.set_stream(self.s_loc)
                reduce_slots(self._current_views()[1])
                torch.cuda.set_stream(self.s_default)
            self._reduction_owed = False

    def _step_torch_halo(self, ev=None):
        """the same step with the halo buffers packed by torch indexing operations (HaloExchange)"""
        nb, fl = self.nb, self.flags
        self.f_t, self.out_t = self._current_views()
        self.halo.f_t = self.f_t
        if ev:
            ev[0].record(self.s_loc)
        nb.launch_kernel_resident(fl, abi.LOCAL)
        if ev:
            ev[1].record(self.s_loc)
        with torch.cuda.stream(self.s_nl):
            self.halo.send_x()
        nb.launch_kernel_resident(fl, abi.NONLOCAL)       # non-local stream, after the halo arrived
        with torch.cuda.stream(self.s_nl):
            self.s_nl.wait_stream(self.s_loc)             # owner-side adds and the reduction see the local kernel done
            self.halo.return_f()
            if fl & abi.FORCE_VIRIAL:
                self.halo.add_wrap_fshift(self.out_t)
                reduce_slots(self.out_t)
        self.s_loc.wait_stream(self.s_nl)

    def read_energies(self):
        """{e_lj, e_el, fshift[45][3]} of the last step, summed over ranks (valid until the next step)."""
        self.flush()
        torch.cuda.synchronize(self.dev)
        return fold_slots(self._current_views()[1])

    def forces_home(self):
        """home forces in caller order of this rank's home atoms (after a step)."""
        torch.cuda.synchronize(self.dev)
        fnb = self._current_views()[0].cpu().numpy()
        f = np.zeros((self.dom.nhome + self.dom.nhalo, 3), dtype=np.float32)
        self.dom.host.add_f(fnb, f)
        return f[:self.dom.nhome]
