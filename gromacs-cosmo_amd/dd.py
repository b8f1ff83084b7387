"""1-D slab domain decomposition of the Verlet nonbonded path, one process per GPU.

The reference decomposes space (mdlib/domdec.c) and computes non-local pairs on the GPU in a
second stream (sim_util.c:1012-1212: local launch, halo x, non-local launch, copy-backs,
halo f).  The path's multi-GPU form here is the smallest scheme with the same data flow:

  * z slabs of equal thickness, rank r owns z in [r Lz, (r+1) Lz); x and y stay periodic
    inside the search, z is decomposed (no periodic shift along z: nbnxn_search.c:4906-4927);
  * half shell: rank r imports the atoms of rank r+1 that lie within rlist above its upper
    face (zone 1 of its grid) and computes home x halo pairs in the NON-LOCAL list; home x home
    pairs are the LOCAL list.  Slabs must be thicker than rlist so that only neighbours interact;
  * per step: halo coordinates travel DOWN (r -> r-1), halo forces travel UP (r -> r+1) and are
    added to the owner's forces; {e_lj, e_el, fshift} are summed over ranks.

Three ways to move the halo, one data flow:
  * `RcclHalo` (what bench.py --gpus N uses): the HIP module issues pack kernel, grouped RCCL send / receive and
    unpack kernel itself on its non-local stream (nbnxn_hip_dd_*, its own communicator);
  * `DeviceHalo`: the module's pack / unpack kernels, the messages through torch.distributed (backend "nccl" = RCCL
    over xGMI on the GPU box, "gloo" in rehearsals on one GPU);
  * `HaloExchange`: torch indexing operations and torch.distributed, on CPU and GPU tensors alike -- what
    tests/test_dd_cpu.py binds to the oracle.
`SlabRun` binds them to the HIP module (device buffers wrapped in place, local / non-local HIP streams).  One rank is
a valid decomposition too: it is its own neighbour, periodic in z through a halo of its own atoms.
"""
import os
import sys
from dataclasses import dataclass

import numpy as np
import torch
import torch.distributed as dist

from . import abi
from . import system as sysmod
from .host import NbnxnHost, PairList


@dataclass
class Domain:
    rank: int
    world: int
    box: np.ndarray            # global box (its diagonal)
    box_search: np.ndarray     # the box as the search takes it: [3], or the [3, 3] matrix of a cell skewed in xy
    lz: float                  # slab thickness
    nhome: int                 # home atoms (caller order: home first, halo after)
    nhalo: int
    home_ids: np.ndarray       # global atom ids of the home atoms
    halo_ids: np.ndarray       # global atom ids of the imported atoms, in arrival order
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
    send_slots: np.ndarray     # grid slots of the home atoms the lower neighbour imports
    send_zshift: float         # added to their z on the way (periodic wrap at the bottom rank)
    halo_slots: np.ndarray     # grid slots the imported atoms occupy here
    # what the device search needs to redo the gridding and both lists (nbnxn_hip_put_on_grid_zones)
    zones: list = None         # [(corner0, corner1, a0, n)], home zone first
    x_all: np.ndarray = None   # caller order: home atoms, then the imported ones
    type_all: np.ndarray = None
    q_all: np.ndarray = None
    excl_index: np.ndarray = None
    excl_a: np.ndarray = None
    rlist: float = 0.0
    min_ci_balanced: int = 0

    @property
    def pair_slots(self):
        return (self.pl_local.nci_tot + self.pl_nonlocal.nci_tot) * 64

    @property
    def up(self):
        return (self.rank + 1) % self.world

    @property
    def down(self):
        return (self.rank - 1) % self.world


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


def build_domain(glob, ip, rank, world, device, min_ci_balanced=0):
    """Cuts rank `rank`'s slab out of the global system `glob` (every rank holds the synthetic
    global topology; coordinates of imported atoms still arrive by message, as they would in a
    run), grids home and halo zones, and builds both lists."""
    if world < 1:
        raise ValueError("domain decomposition needs at least one rank")
    # a triclinic cell may be decomposed along z as long as box[ZZ] lies along z (the halo shift is then the plain
    # z translation these messages apply); box[YY][XX] may be skewed, the search handles it through its x/y shifts
    box_search = np.asarray(glob.box, dtype=np.float32)
    if box_search.shape == (3, 3):
        if box_search[2, 0] != 0 or box_search[2, 1] != 0:
            raise ValueError("slab decomposition along z needs box[ZZ] along z")
        box = np.ascontiguousarray(np.diag(box_search))
    else:
        box = box_search
    lz = float(box[2]) / world
    # one rank is its own neighbour: periodic in z through a halo of its own atoms, which needs room for both
    if lz <= ip.rlist * (2 if world == 1 else 1):
        raise ValueError(f"slab thickness {lz:.3f} nm must exceed {'2 x ' if world == 1 else ''}rlist {ip.rlist:.3f} nm")
    z = glob.x[:, 2].astype(np.float64)
    owner = np.clip(np.floor(z / lz).astype(np.int64), 0, world - 1)
    home_ids = np.nonzero(owner == rank)[0]
    nhome = len(home_ids)
    x_home = glob.x[home_ids]

    # what the lower neighbour imports from me: my atoms within rlist of my lower face
    zrel = x_home[:, 2] - np.float32(rank * lz)
    down_sel = np.nonzero(zrel < ip.rlist)[0]
    send_zshift = float(box[2]) if rank == 0 else 0.0
    up, down = (rank + 1) % world, (rank - 1) % world

    n_send = torch.tensor([len(down_sel)], dtype=torch.int64, device=device)
    n_recv = torch.zeros(1, dtype=torch.int64, device=device)
    _exchange(n_send, down, n_recv, up)
    nhalo = int(n_recv.item())
    ids_send = torch.from_numpy(home_ids[down_sel].astype(np.int64)).to(device)
    xs = x_home[down_sel].copy()
    xs[:, 2] += np.float32(send_zshift)
    x_send = torch.from_numpy(xs).to(device)
    ids_recv = torch.zeros(nhalo, dtype=torch.int64, device=device)
    x_recv = torch.zeros((nhalo, 3), dtype=torch.float32, device=device)
    _exchange(ids_send, down, ids_recv, up)
    _exchange(x_send, down, x_recv, up)
    halo_ids = ids_recv.cpu().numpy()
    x_halo = x_recv.cpu().numpy()

    # grids: zone 0 = home slab, zone 1 = the rlist-thick layer above it
    host = NbnxnHost(box_search, pbc_shift_dim=(1, 1, 0))
    z0 = rank * lz
    zones = [((0.0, 0.0, z0), (float(box[0]), float(box[1]), z0 + lz), 0, nhome)]
    host.put_on_grid(0, zones[0][0], zones[0][1], 0, x_home)
    if nhalo > 0:
        zones.append(((0.0, 0.0, z0 + lz), (float(box[0]), float(box[1]), z0 + lz + ip.rlist), nhome, nhalo))
        host.put_on_grid(1, zones[1][0], zones[1][1], nhome, x_halo)
    ids_all = np.concatenate([home_ids, halo_ids])
    x_all = np.concatenate([x_home, x_halo]).astype(np.float32)
    ntype = glob.ntype + 1
    xq, typ = host.set_atomdata(ntype, glob.type[ids_all], glob.q[ids_all])
    host.copy_x_to_xq(x_all, xq)

    # topology exclusions in local numbering; partners that are not on this rank drop out.  A partner maps to every
    # copy of it that this rank holds: its home atom and / or its halo image (both only when one rank is its own
    # neighbour; an exclusion between copies that are out of range of each other is never looked at).
    nloc = len(ids_all)
    g2l_home = np.full(glob.natoms, -1, dtype=np.int64)
    g2l_home[home_ids] = np.arange(nhome)
    g2l_halo = np.full(glob.natoms, -1, dtype=np.int64)
    g2l_halo[halo_ids] = nhome + np.arange(nhalo)
    cnt = (glob.excl_index[1:] - glob.excl_index[:-1])[ids_all]
    starts = glob.excl_index[:-1][ids_all]
    flat = np.repeat(starts - np.concatenate([[0], np.cumsum(cnt)[:-1]]), cnt) + np.arange(int(cnt.sum()))
    row = np.repeat(np.arange(nloc), cnt)
    gp = glob.excl_a[flat]
    rows = np.concatenate([row, row])
    partners = np.concatenate([g2l_home[gp], g2l_halo[gp]])
    keep = partners >= 0
    rows, partners = rows[keep], partners[keep]
    order = np.lexsort((partners, rows))
    rows, partners = rows[order], partners[order]
    excl_index = np.zeros(nloc + 1, dtype=np.int32)
    np.add.at(excl_index, rows + 1, 1)
    excl_index = np.cumsum(excl_index).astype(np.int32)
    excl_a = partners.astype(np.int32)

    pl_local = host.make_pairlist(abi.LOCAL, ip.rlist, xq, excl_index, excl_a, min_ci_balanced)
    pl_nonlocal = host.make_pairlist(abi.NONLOCAL, ip.rlist, xq, excl_index, excl_a, min_ci_balanced)
    cell = host.cell_index()
    return Domain(rank=rank, world=world, box=box, box_search=box_search, lz=lz, nhome=nhome, nhalo=nhalo, home_ids=home_ids,
                  halo_ids=halo_ids, host=host, xq=xq, type=typ, ntype=ntype, nbfp=glob.nbfp(),
                  shift_vec=host.shift_vec(), pl_local=pl_local, pl_nonlocal=pl_nonlocal,
                  npad=len(xq), npad_local=host.natoms_padded_local,
                  send_slots=cell[down_sel].astype(np.int64), send_zshift=send_zshift,
                  halo_slots=cell[nhome:nhome + nhalo].astype(np.int64),
                  zones=zones, x_all=x_all, type_all=glob.type[ids_all].astype(np.int32), q_all=glob.q[ids_all].astype(np.float32),
                  excl_index=excl_index, excl_a=excl_a, rlist=float(ip.rlist), min_ci_balanced=int(min_ci_balanced))


class HaloExchange:
    """The per-step messages, on tensors that alias the engine's coordinate and force arrays:
    xq_t [npad, 4] and f_t [npad, 3] (grid order)."""

    def __init__(self, dom, xq_t, f_t):
        dev = xq_t.device
        self.dom, self.xq_t, self.f_t = dom, xq_t, f_t
        self.send_slots = torch.from_numpy(dom.send_slots).to(dev)
        self.halo_slots = torch.from_numpy(dom.halo_slots).to(dev)
        self.zshift = torch.tensor([0.0, 0.0, dom.send_zshift, 0.0], dtype=torch.float32, device=dev)
        self.x_out = torch.zeros((len(dom.send_slots), 4), dtype=torch.float32, device=dev)
        self.x_in = torch.zeros((dom.nhalo, 4), dtype=torch.float32, device=dev)
        self.f_out = torch.zeros((dom.nhalo, 3), dtype=f_t.dtype, device=dev)
        self.f_in = torch.zeros((len(dom.send_slots), 3), dtype=f_t.dtype, device=dev)
        # forces that come back across the periodic wrap count towards the shift force of
        # (0, 0, +1), as dd_move_f does for shifted pulses (mdlib/domdec.c:1046-1068): the
        # virial needs  box_z (x) sum of those forces
        self.wrap_shift = 5 * (3 * (1 + 1) + 0 + 1) + 0 + 2 if dom.send_zshift != 0.0 else -1
        self.fshift_wrap = torch.zeros(3, dtype=torch.float64, device=dev)

    def send_x(self):
        """home {x,y,z,q} rows the lower neighbour imports go down; mine arrive from above."""
        torch.index_select(self.xq_t, 0, self.send_slots, out=self.x_out)
        if self.wrap_shift >= 0:
            self.x_out += self.zshift
        _exchange(self.x_out, self.dom.down, self.x_in, self.dom.up)
        self.xq_t.index_copy_(0, self.halo_slots, self.x_in)

    def return_f(self):
        """forces on imported atoms go back up to their owner, who adds them."""
        torch.index_select(self.f_t, 0, self.halo_slots, out=self.f_out)
        _exchange(self.f_out, self.dom.up, self.f_in, self.dom.down)
        self.f_t.index_add_(0, self.send_slots, self.f_in)
        if self.wrap_shift >= 0:
            self.fshift_wrap = self.f_in.to(torch.float64).sum(0)

    def add_wrap_fshift(self, out_t):
        """the same term, added straight into accumulation slot 0 of the engine's output buffer"""
        if self.wrap_shift >= 0:
            out_t[2 + 3 * self.wrap_shift:5 + 3 * self.wrap_shift] += self.fshift_wrap.to(out_t.dtype)


class DeviceHalo:
    """The same messages with the buffers packed and unpacked by the HIP module's halo kernels
    (nbnxn_hip_halo_*) on its non-local stream: four kernel launches through the C ABI per step instead
    of nine torch operations, and the shift-force term of the periodic wrap is added by the kernel
    that adds the returned forces.  The exchange itself goes out on torch's CURRENT stream, which the
    caller sets to the module's non-local stream (SlabRun.step)."""

    def __init__(self, nb, dom, dev):
        import ctypes as C
        self.nb, self.dom = nb, dom
        self.send_slots = torch.from_numpy(dom.send_slots.astype(np.int32)).to(dev)
        self.halo_slots = torch.from_numpy(dom.halo_slots.astype(np.int32)).to(dev)
        self.n_send, self.n_halo = len(dom.send_slots), int(dom.nhalo)
        self.x_out = torch.zeros((self.n_send, 4), dtype=torch.float32, device=dev)
        self.x_in = torch.zeros((self.n_halo, 4), dtype=torch.float32, device=dev)
        self.f_out = torch.zeros((self.n_halo, 3), dtype=torch.float32, device=dev)
        self.f_in = torch.zeros((self.n_send, 3), dtype=torch.float32, device=dev)
        wrap = dom.send_zshift != 0.0
        self.shift = (C.c_float * 3)(0.0, 0.0, float(dom.send_zshift)) if wrap else None
        self.wrap_shift = 5 * (3 * (1 + 1) + 0 + 1) + 0 + 2 if wrap else -1     # shift (0, 0, +1), as in HaloExchange
        self._p = [t.data_ptr() for t in (self.send_slots, self.halo_slots, self.x_out, self.x_in, self.f_out, self.f_in)]

    def send_x(self):
        p = self._p
        self.nb.halo_pack_x(p[0], self.n_send, self.shift, p[2], abi.NONLOCAL)
        _exchange(self.x_out, self.dom.down, self.x_in, self.dom.up)
        self.nb.halo_unpack_x(p[1], self.n_halo, p[3], abi.NONLOCAL)

    def return_f(self):
        p = self._p
        self.nb.halo_pack_f(p[1], self.n_halo, p[4], abi.NONLOCAL)
        _exchange(self.f_out, self.dom.up, self.f_in, self.dom.down)
        self.nb.halo_unpack_f(p[0], self.n_send, p[5], self.wrap_shift, abi.NONLOCAL)


class RcclHalo:
    """The same messages issued by the HIP module itself (nbnxn_hip_dd_*): pack kernel, one grouped
    ncclSend / ncclRecv between devices, unpack kernel, all on the module's non-local stream from ONE call
    through the C ABI; the slot reduction is the module's ncclAllReduce.  No torch stream is involved, so the
    step needs no set_stream and no hand-over between torch's collective stream and the module's.  The
    communicator is the module's own: rank 0 makes the RCCL id, torch.distributed only carries its 128 bytes."""

    def __init__(self, nb, dom, dev):
        self.nb, self.dom = nb, dom
        cuda_group = dist.get_backend() != "gloo"
        idt = torch.zeros(abi.DD_UNIQUE_ID_BYTES, dtype=torch.uint8, device=dev if cuda_group else "cpu")
        if dom.rank == 0:
            idt.copy_(torch.frombuffer(bytearray(nb.dd_get_unique_id()), dtype=torch.uint8))
        dist.broadcast(idt, 0)
        nb.dd_init(bytes(idt.cpu().numpy().tobytes()), dom.rank, dom.world)
        wrap = dom.send_zshift != 0.0
        self.wrap_shift = 5 * (3 * (1 + 1) + 0 + 1) + 0 + 2 if wrap else -1     # shift (0, 0, +1), as in HaloExchange
        self.halo_slots = torch.from_numpy(dom.halo_slots.astype(np.int64)).to(dev)
        nb.dd_set_pulse(dom.down, dom.up, dom.send_slots, dom.halo_slots, (0.0, 0.0, float(dom.send_zshift)) if wrap else None,
                        self.wrap_shift)

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
    if halo is not None and halo.wrap_shift >= 0:
        e[2 + 3 * halo.wrap_shift:5 + 3 * halo.wrap_shift] += halo.fshift_wrap
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
    """The multi-GPU workload of bench.py: water_box(tx, ty, tz * world) (weak scaling) or water_box(tx, ty, tz)
    (strong scaling, BASELINE configs[3]) cut into `world` z slabs of equal thickness."""

    def __init__(self, tiles, model, ip, rank, world, local_rank, flags, min_ci_balanced=None, device_search=True,
                 device_halo=True, rccl_direct=False, strong=False):
        """device_halo: halo buffers packed by the module's kernels (else torch indexing operations), messages
        through torch.distributed.  rccl_direct: everything, messages included, issued by the module on its own
        RCCL communicator (needs one GPU per rank)."""
        from . import gpu
        self.flags = flags
        self.dev = torch.device("cuda", local_rank)
        # weak scaling: `tiles` per rank, stacked along z; strong scaling: `tiles` is the whole box, cut into `world` slabs
        glob = sysmod.water_box(tiles[0], tiles[1], tiles[2] * (1 if strong else world), model=model)
        self.natoms_global = glob.natoms
        self.nb = nb = gpu.NbnxnHip(local_rank, local_and_nonlocal=True)
        mcb = nb.min_ci_balanced() if min_ci_balanced is None else min_ci_balanced
        self.dom = dom = build_domain(glob, ip, rank, world, self.dev, mcb)
        nb.init_const(ip, dom.ntype, dom.nbfp)
        if device_search:
            # grids and both lists made on the device; the host-side builder's results, which the
            # halo bookkeeping above came from, double as the check that both see the same grid
            npad = nb.put_on_grid_zones(dom.box_search, dom.zones, dom.x_all, dom.type_all, dom.q_all, pbc_shift_dim=(1, 1, 0))
            assert npad == dom.npad
            nb.make_pairlist(dom.rlist, dom.excl_index, dom.excl_a, dom.min_ci_balanced, iloc=abi.LOCAL)
            if len(dom.zones) > 1:
                nb.make_pairlist(dom.rlist, None, None, dom.min_ci_balanced, iloc=abi.NONLOCAL)
            for iloc, pl in ((abi.LOCAL, dom.pl_local), (abi.NONLOCAL, dom.pl_nonlocal))[:len(dom.zones)]:
                got = nb.download_pairlist(iloc)
                assert (got.nsci, got.ncj4, got.nexcl, got.nci_tot) == (pl.nsci, pl.ncj4, pl.nexcl, pl.nci_tot)
                assert np.array_equal(got.sci, pl.sci)
        else:
            nb.init_atomdata(dom.npad, dom.npad_local, dom.ntype, dom.type)
            nb.upload_shiftvec(dom.shift_vec)
            nb.init_pairlist(dom.pl_local, abi.LOCAL)
            nb.init_pairlist(dom.pl_nonlocal, abi.NONLOCAL)
            assert nb.verify_pairlist(dom.pl_local, abi.LOCAL) == 0
            assert nb.verify_pairlist(dom.pl_nonlocal, abi.NONLOCAL) == 0
        # first pass the reference way: host xq in, forces out, both lists pruned, streams drained
        self.f_first, self.e_first = nb.compute_both(dom.xq, flags)
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
        if rccl_direct:
            self.halo = RcclHalo(nb, dom, self.dev)
        else:
            self.halo = DeviceHalo(nb, dom, self.dev) if device_halo else HaloExchange(dom, self.xq_t, self.f_t)
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
        return {"atoms_rank0": d.nhome, "halo_atoms_rank0": d.nhalo, "nsci_local": d.pl_local.nsci,
                "nsci_nonlocal": d.pl_nonlocal.nsci, "cluster_pairs_rank0": d.pl_local.nci_tot + d.pl_nonlocal.nci_tot}

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
