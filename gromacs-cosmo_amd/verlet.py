"""The callers either side of the kernel, in the order do_force_cutsVERLET runs them
(mdlib/sim_util.c:907-1530), with everything resident on the device:

  search steps (every nstlist):  nbnxn_put_on_grid + nbnxn_atomdata_set + nbnxn_cuda_init_atomdata
                                 (:967-1007)            -> nbnxn_hip_put_on_grid
                                 nbnxn_make_pairlist + nbnxn_cuda_init_pairlist (:1021-1034)
                                                        -> nbnxn_hip_make_pairlist
  other steps:                   nbnxn_atomdata_copy_x_to_nbat_x (:1044) -> nbnxn_hip_copy_x_to_xq
  every step:                    nbnxn_cuda_clear_outputs (:1486), do_nb_verlet (:622),
                                 nbnxn_atomdata_add_nbat_f_to_f (:1502)  -> nbnxn_hip_add_f

x and f are caller-order device arrays (torch tensors); nothing but the energies crosses PCIe.
The list is built with rlist >= rcoulomb and reused for nstlist steps, pruned by the first kernel
that runs on it, as in the reference.  With rlist_inner set, the list is instead re-pruned to that
inner radius by the prune-only kernel (nbnxn_hip_launch_kernel_pruneonly): all of it on search
steps, and on the other steps either all of it every nstprune steps (prune_parts = 1) or one of
prune_parts slices every step (rolling)."""
import numpy as np

from . import abi


class VerletLoop:
    def __init__(self, nb, system, ip, nstlist=10, min_ci_balanced=None, rlist_inner=None, nstprune=0, prune_parts=1):
        self.nb, self.s, self.ip, self.nstlist = nb, system, ip, int(nstlist)
        self.rlist_inner, self.nstprune, self.prune_parts = rlist_inner, int(nstprune), int(prune_parts)
        assert rlist_inner is None or (ip.rcoulomb <= rlist_inner <= ip.rlist and (nstprune > 0 or prune_parts > 1))
        self.mcb = nb.min_ci_balanced() if min_ci_balanced is None else min_ci_balanced
        nb.init_const(ip, system.ntype + 1, system.nbfp())
        self.box = np.ascontiguousarray(system.box, dtype=np.float32)
        self.step = 0
        self.nsearch = 0

    def search(self, x):
        """x: torch float32 [n, 3] on the device, atoms inside the box (the caller wraps them on search steps,
        as put_atoms_in_box does before nbnxn_put_on_grid)."""
        first = self.nsearch == 0
        self.nb.put_on_grid(self.box, x, self.s.type if first else None, self.s.q if first else None)
        self.nb.make_pairlist(self.ip.rlist, self.s.excl_index if first else None, self.s.excl_a if first else None, self.mcb)
        self.nsearch += 1

    def force(self, x, f, flags=0):
        """One step: f (torch float32 [n, 3], device) gets the nonbonded forces ADDED.  Returns None, or
        (e_lj, e_el, fshift) when flags has GMX_FORCE_VIRIAL (that read-back synchronises)."""
        nb = self.nb
        assert x.is_cuda and f.is_cuda and x.dtype == f.dtype and x.is_contiguous() and f.is_contiguous()
        if self.step % self.nstlist == 0:
            self.search(x)
            if self.rlist_inner is not None:
                nb.prune_only(self.rlist_inner, self.prune_parts)          # a new list: all of it
        else:
            nb.copy_x_to_xq(x.data_ptr())
            if self.rlist_inner is not None:
                if self.prune_parts > 1:
                    nb.prune_only(self.rlist_inner, self.prune_parts)      # rolling: one slice per step
                elif self.step % self.nstprune == 0:
                    nb.prune_only(self.rlist_inner, 1)
        nb.clear_outputs(flags)
        nb.launch_kernel_resident(flags)
        nb.add_f(f.data_ptr())
        self.step += 1
        if flags & abi.FORCE_VIRIAL:
            return nb.read_outputs()
        return None
