"""Puts a System through the host producers in the order do_force_cutsVERLET does
(mdlib/sim_util.c:907-1044): grid -> atom data -> pair list, ready for the device calls."""
from dataclasses import dataclass

import numpy as np

from . import abi
from .host import NbnxnHost, PairList


@dataclass
class Prepared:
    host: NbnxnHost
    xq: np.ndarray          # [natoms_padded, 4] grid order
    type: np.ndarray        # [natoms_padded]
    ntype: int              # including the dummy type
    nbfp: np.ndarray        # [ntype, ntype, 2]
    shift_vec: np.ndarray   # [45, 3]
    pl: PairList
    natoms: int             # padded

    @property
    def pair_slots(self):
        """The reference's counting convention: cluster pairs x 64 (nbnxn_search.c:5770-5777)."""
        return self.pl.nci_tot * 64


def prepare(system, ip, min_ci_balanced=0):
    h = NbnxnHost(system.box)
    h.put_on_grid(0, (0.0, 0.0, 0.0), system.box_diag, 0, system.x)
    ntype = system.ntype + 1
    xq, typ = h.set_atomdata(ntype, system.type, system.q)
    h.copy_x_to_xq(system.x, xq)
    pl = h.make_pairlist(abi.LOCAL, ip.rlist, xq, system.excl_index, system.excl_a, min_ci_balanced)
    return Prepared(host=h, xq=xq, type=typ, ntype=ntype, nbfp=system.nbfp(), shift_vec=h.shift_vec(),
                    pl=pl, natoms=len(xq))


def upload(nb, prep, ip):
    """init_const -> init_atomdata -> upload_shiftvec -> init_pairlist (forcerec.c:2108, sim_util.c:1007-1034)."""
    nb.init_const(ip, prep.ntype, prep.nbfp)
    nb.init_atomdata(prep.natoms, prep.natoms, prep.ntype, prep.type)
    nb.upload_shiftvec(prep.shift_vec)
    nb.init_pairlist(prep.pl, abi.LOCAL)
