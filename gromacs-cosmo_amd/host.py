"""Host-side producers of the GPU list format: Python mirror over libnbnxn_host.so
(include/nbnxn_host.h).  Mirrors the reference call sequence of do_force_cutsVERLET
(mdlib/sim_util.c:907-1044): put_on_grid -> atomdata_set -> make_pairlist -> copy_x.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_lib = None


def _load():
    global _lib
    if _lib is not None:
        return _lib
    path = os.path.join(abi.lib_dir(), "libnbnxn_host.so")
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = C.CDLL(path)
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    fp = C.POINTER(C.c_float)
    ip = C.POINTER(C.c_int)
    lib.nbnxn_host_last_error.restype = C.c_char_p
    lib.nbnxn_host_create.restype = vp
    lib.nbnxn_host_destroy.argtypes = [vp]
    lib.nbnxn_host_set_box.argtypes = [vp, fp, ip]
    lib.nbnxn_host_set_box_triclinic.argtypes = [vp, fp, ip]
    lib.nbnxn_host_put_on_grid.argtypes = [vp, ci, fp, fp, ci, ci, vp, cf]
    lib.nbnxn_host_natoms_padded.argtypes = [vp]
    lib.nbnxn_host_natoms_padded_local.argtypes = [vp]
    lib.nbnxn_host_ncolumns.argtypes = [vp, ci, ip, ip]
    lib.nbnxn_host_nsupercells.argtypes = [vp]
    lib.nbnxn_host_atom_order.argtypes = [vp]
    lib.nbnxn_host_atom_order.restype = ip
    lib.nbnxn_host_cell_index.argtypes = [vp]
    lib.nbnxn_host_cell_index.restype = ip
    lib.nbnxn_host_set_atomdata.argtypes = [vp, ci, vp, vp, vp, vp]
    lib.nbnxn_host_copy_x_to_xq.argtypes = [vp, vp, vp]
    lib.nbnxn_host_add_f.argtypes = [vp, vp, ci, ci, vp]
    lib.nbnxn_host_make_pairlist.argtypes = [vp, ci, cf, vp, vp, ci, vp]
    lib.nbnxn_host_make_pairlist_zones.argtypes = [vp, ci, ci, vp, cf, vp, vp, ci, vp]
    lib.nbnxn_host_pairlist.argtypes = [vp, ci]
    lib.nbnxn_host_pairlist.restype = C.POINTER(abi.Pairlist)
    lib.nbnxn_host_shift_vec.argtypes = [vp, vp]
    lib.nbnxn_host_set_progressive_balance.argtypes = [vp, ci]
    _lib = lib
    return lib


class HostError(RuntimeError):
    pass


class PairList:
    """Owned numpy copies of one GPU-format list (sci / cj4 / excl)."""

    def __init__(self, sci, cj4, excl, nci_tot, rlist):
        self.sci, self.cj4, self.excl = sci, cj4, excl
        self.nci_tot, self.rlist = int(nci_tot), float(rlist)

    @property
    def nsci(self):
        return len(self.sci)

    @property
    def ncj4(self):
        return len(self.cj4)

    @property
    def nexcl(self):
        return len(self.excl)

    def stats(self):
        return dict(nsci=self.nsci, ncj4=self.ncj4, nexcl=self.nexcl, nci_tot=self.nci_tot,
                    pair_slots=self.nci_tot * 64)


class NbnxnHost:
    def __init__(self, box, pbc_shift_dim=(1, 1, 1)):
        self.lib = _load()
        self.h = self.lib.nbnxn_host_create()
        # box: [3] rectangular, or the [3, 3] GROMACS matrix (rows are box vectors) of a triclinic cell
        box = np.asarray(box, dtype=np.float32)
        p = (C.c_int * 3)(*[int(v) for v in pbc_shift_dim])
        if box.shape == (3, 3):
            self.box_matrix = box.copy()
            self.box = np.ascontiguousarray(np.diag(box))
            self._chk(self.lib.nbnxn_host_set_box_triclinic(self.h, (C.c_float * 9)(*box.reshape(-1)), p))
        else:
            self.box = box
            self.box_matrix = np.diag(box).astype(np.float32)
            self._chk(self.lib.nbnxn_host_set_box(self.h, (C.c_float * 3)(*box), p))
        self.natoms = 0

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.nbnxn_host_destroy(self.h)
            self.h = None

    def _chk(self, rc):
        if rc != 0:
            raise HostError(self.lib.nbnxn_host_last_error().decode())

    def put_on_grid(self, zone, corner0, corner1, a0, x, atom_density=0.0):
        x = np.ascontiguousarray(x, dtype=np.float32)
        c0 = (C.c_float * 3)(*[float(v) for v in corner0])
        c1 = (C.c_float * 3)(*[float(v) for v in corner1])
        self._x_keep = x
        self._chk(self.lib.nbnxn_host_put_on_grid(self.h, zone, c0, c1, a0, len(x), abi.ptr(x), atom_density))
        self.natoms = a0 + len(x)

    @property
    def natoms_padded(self):
        return self.lib.nbnxn_host_natoms_padded(self.h)

    @property
    def natoms_padded_local(self):
        return self.lib.nbnxn_host_natoms_padded_local(self.h)

    def ncolumns(self, zone=0):
        a, b = C.c_int(), C.c_int()
        self._chk(self.lib.nbnxn_host_ncolumns(self.h, zone, C.byref(a), C.byref(b)))
        return a.value, b.value

    def atom_order(self):
        p = self.lib.nbnxn_host_atom_order(self.h)
        return np.ctypeslib.as_array(p, shape=(self.natoms_padded,)).copy()

    def cell_index(self):
        p = self.lib.nbnxn_host_cell_index(self.h)
        return np.ctypeslib.as_array(p, shape=(self.natoms,)).copy()

    def set_atomdata(self, ntype_with_dummy, type_, q):
        type_ = np.ascontiguousarray(type_, dtype=np.int32)
        q = np.ascontiguousarray(q, dtype=np.float32)
        assert len(type_) == self.natoms and len(q) == self.natoms
        xq = np.zeros((self.natoms_padded, 4), dtype=np.float32)
        tout = np.zeros(self.natoms_padded, dtype=np.int32)
        self._chk(self.lib.nbnxn_host_set_atomdata(self.h, ntype_with_dummy, abi.ptr(type_), abi.ptr(q),
                                                    abi.ptr(xq), abi.ptr(tout)))
        return xq, tout

    def copy_x_to_xq(self, x, xq):
        x = np.ascontiguousarray(x, dtype=np.float32)
        assert x.shape == (self.natoms, 3) and xq.shape == (self.natoms_padded, 4)
        self._chk(self.lib.nbnxn_host_copy_x_to_xq(self.h, abi.ptr(x), abi.ptr(xq)))

    def add_f(self, fnb, f, a0=0, n=None):
        n = self.natoms - a0 if n is None else n
        assert fnb.dtype == np.float32 and f.dtype == np.float32
        self._chk(self.lib.nbnxn_host_add_f(self.h, abi.ptr(fnb), a0, n, abi.ptr(f)))

    def make_pairlist(self, iloc, rlist, xq, excl_index=None, excl_a=None, min_ci_balanced=0, zone_pairs=None):
        """zone_pairs: [(i-zone, first j-zone, end j-zone)], the reference's dd_zp tables (domdec.c:433-449); None = the
        home zone against itself (LOCAL) or against every halo zone (NONLOCAL)."""
        if excl_index is not None:
            excl_index = np.ascontiguousarray(excl_index, dtype=np.int32)
            excl_a = np.ascontiguousarray(excl_a, dtype=np.int32)
        if zone_pairs is None:
            self._chk(self.lib.nbnxn_host_make_pairlist(self.h, iloc, rlist, abi.ptr(excl_index), abi.ptr(excl_a),
                                                         min_ci_balanced, abi.ptr(xq)))
        else:
            zp = np.ascontiguousarray(zone_pairs, dtype=np.int32).reshape(-1, 3)
            self._chk(self.lib.nbnxn_host_make_pairlist_zones(self.h, iloc, len(zp), abi.ptr(zp), rlist, abi.ptr(excl_index),
                                                               abi.ptr(excl_a), min_ci_balanced, abi.ptr(xq)))
        v = self.lib.nbnxn_host_pairlist(self.h, iloc).contents

        def grab(addr, n, dt):
            if n == 0:
                return np.zeros(0, dtype=dt)
            buf = (C.c_char * (n * dt.itemsize)).from_address(addr)
            return np.frombuffer(buf, dtype=dt).copy()

        return PairList(grab(v.sci, v.nsci, abi.sci_dtype), grab(v.cj4, v.ncj4, abi.cj4_dtype),
                        grab(v.excl, v.nexcl, abi.excl_dtype), v.nci_tot, v.rlist)

    def pairlist_struct(self, iloc):
        """The nbnxn_pairlist_t-compatible view owned by the C++ side (valid until the next make_pairlist)."""
        return self.lib.nbnxn_host_pairlist(self.h, iloc)

    def set_progressive_balance(self, on):
        self._chk(self.lib.nbnxn_host_set_progressive_balance(self.h, int(on)))

    def shift_vec(self):
        sv = np.zeros((abi.SHIFTS, 3), dtype=np.float32)
        self._chk(self.lib.nbnxn_host_shift_vec(self.h, abi.ptr(sv)))
        return sv
