"""Python mirror of the C ABI in include/nbnxn_hip.h (libnbnxn_hip.so), with the same call
names and argument meaning as the reference's nbnxn_cuda_* module
(legacyheaders/nbnxn_cuda_data_mgmt.h:59-139, mdlib/nbnxn_cuda/nbnxn_cuda.h:64-86).

There is no CPU fallback: if the HIP library is missing or no gfx950 device is present,
every entry point raises.
"""
import ctypes as C
import os
import sys

import numpy as np

from . import abi

_lib = None


class HipError(RuntimeError):
    pass


def _share_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64 (rpath $ORIGIN).  Two HIP
    runtimes in one process do not both see the GPU, and which one gets loaded depends on whether
    torch or this library came first.  When a torch with a bundled runtime is installed, load THAT
    copy first (without importing torch), so that libnbnxn_hip.so and a later `import torch` share it."""
    if "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        rt = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(rt):
            C.CDLL(rt, mode=C.RTLD_GLOBAL)
    except (OSError, ImportError, ValueError):
        pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("NBNXN_HIP_LIB") or os.path.join(abi.lib_dir(), "libnbnxn_hip.so")   # override: kernel A/B builds
    if not os.path.exists(path):
        raise HipError(f"{path} is missing: the HIP extension was not built (python -c 'import __graft_entry__ as g; g.build()')")
    _share_torch_hip_runtime()
    lib = C.CDLL(path)
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    lib.nbnxn_hip_last_error.restype = C.c_char_p
    lib.nbnxn_hip_set_abort_on_error.argtypes = [ci]
    lib.nbnxn_hip_detect_gpus.argtypes = [C.POINTER(ci), C.c_char_p, ci]
    lib.nbnxn_hip_get_gpu_device_info_string.argtypes = [C.c_char_p, ci, ci]
    lib.nbnxn_hip_sizeof_dev_info.restype = C.c_size_t
    lib.nbnxn_hip_detect_gpus_info.argtypes = [C.POINTER(abi.GpuDevInfo), ci, C.POINTER(ci), C.POINTER(ci), C.c_char_p, ci]
    lib.nbnxn_hip_pick_compatible_gpus.argtypes = [C.POINTER(abi.GpuDevInfo), ci, C.POINTER(ci), ci, C.POINTER(ci)]
    lib.nbnxn_hip_check_selected_gpus.argtypes = [C.POINTER(ci), C.POINTER(abi.GpuDevInfo), ci, C.POINTER(ci), ci]
    lib.nbnxn_hip_get_gpu_device_id.argtypes = [C.POINTER(abi.GpuDevInfo), ci, C.POINTER(ci), ci, ci]
    lib.nbnxn_hip_do_quick_memtest.argtypes = [ci]
    lib.nbnxn_hip_do_full_memtest.argtypes = [ci]
    lib.nbnxn_hip_do_timed_memtest.argtypes = [ci, ci]
    lib.nbnxn_hip_pmalloc.argtypes = [C.POINTER(vp), C.c_size_t]
    lib.nbnxn_hip_pfree.argtypes = [vp]
    lib.nbnxn_hip_init.argtypes = [vp, C.POINTER(vp), ci, ci]
    lib.nbnxn_hip_init_const.argtypes = [vp, C.POINTER(abi.InteractionConst), C.POINTER(abi.VerletGroup)]
    lib.nbnxn_hip_init_const_flat.argtypes = [vp, ci] + [cf] * 10 + [ci, vp]
    lib.nbnxn_hip_init_pairlist.argtypes = [vp, C.POINTER(abi.Pairlist), ci]
    lib.nbnxn_hip_init_pairlist_flat.argtypes = [vp, ci, cf, ci, vp, ci, vp, ci, vp, ci]
    lib.nbnxn_hip_init_atomdata.argtypes = [vp, C.POINTER(abi.Atomdata)]
    lib.nbnxn_hip_init_atomdata_flat.argtypes = [vp, ci, ci, ci, vp]
    lib.nbnxn_hip_pme_loadbal_update_param.argtypes = [vp, C.POINTER(abi.InteractionConst)]
    lib.nbnxn_hip_upload_shiftvec.argtypes = [vp, C.POINTER(abi.Atomdata)]
    lib.nbnxn_hip_upload_shiftvec_flat.argtypes = [vp, vp]
    lib.nbnxn_hip_clear_outputs.argtypes = [vp, ci]
    lib.nbnxn_hip_free.argtypes = [vp]
    lib.nbnxn_hip_get_timings.argtypes = [vp]
    lib.nbnxn_hip_get_timings.restype = C.POINTER(abi.WallclockGpu)
    lib.nbnxn_hip_reset_timings.argtypes = [vp]
    lib.nbnxn_hip_min_ci_balanced.argtypes = [vp]
    lib.nbnxn_hip_is_kernel_ewald_analytical.argtypes = [vp]
    lib.nbnxn_hip_launch_kernel.argtypes = [vp, C.POINTER(abi.Atomdata), ci, ci]
    lib.nbnxn_hip_launch_kernel_flat.argtypes = [vp, vp, ci, ci]
    lib.nbnxn_hip_launch_kernel_resident.argtypes = [vp, ci, ci]
    lib.nbnxn_hip_launch_kernel_noalloc.argtypes = [vp, ci, ci, ci, vp, vp, vp]
    lib.nbnxn_hip_launch_cpyback.argtypes = [vp, C.POINTER(abi.Atomdata), ci, ci]
    lib.nbnxn_hip_launch_cpyback_flat.argtypes = [vp, vp, ci, ci]
    lib.nbnxn_hip_wait_gpu.argtypes = [vp, C.POINTER(abi.Atomdata), ci, ci, vp, vp, vp]
    lib.nbnxn_hip_wait_gpu_flat.argtypes = [vp, ci, ci, vp, vp, vp]
    lib.nbnxn_hip_get_device_buffers.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    lib.nbnxn_hip_get_stream.argtypes = [vp, ci]
    lib.nbnxn_hip_get_stream.restype = vp
    lib.nbnxn_hip_read_outputs.argtypes = [vp, vp, vp, vp]
    lib.nbnxn_hip_verify_pairlist_flat.argtypes = [vp, ci, ci, vp, ci, vp, ci, vp]
    lib.nbnxn_hip_download_cj4.argtypes = [vp, ci, ci, vp]
    lib.nbnxn_hip_last_kernel_ms.argtypes = [vp, ci]
    lib.nbnxn_hip_last_kernel_ms.restype = cf
    lib.nbnxn_hip_set_prune.argtypes = [vp, ci, ci]
    lib.nbnxn_hip_set_vdw_modifier_flat.argtypes = [vp, ci] + [cf] * 8
    lib.nbnxn_hip_set_ljpme_flat.argtypes = [vp, ci, cf, cf, vp]
    lib.nbnxn_hip_download_coulomb_tab.argtypes = [vp, vp, ci, C.POINTER(ci), C.POINTER(cf)]
    lib.nbnxn_hip_launch_kernel_pruneonly.argtypes = [vp, ci, cf, ci]
    lib.nbnxn_hip_put_on_grid.argtypes = [vp, vp, vp, ci, vp, ci, vp, vp, cf]
    lib.nbnxn_hip_put_on_grid_zones.argtypes = [vp, vp, vp, ci, C.POINTER(abi.GridZone), ci, vp, ci, vp, vp]
    lib.nbnxn_hip_make_pairlist.argtypes = [vp, ci, cf, vp, vp, ci]
    lib.nbnxn_hip_make_pairlist_zones.argtypes = [vp, ci, ci, vp, cf, vp, vp, ci]
    lib.nbnxn_hip_forget_exclusions.argtypes = [vp]
    lib.nbnxn_hip_set_progressive_balance.argtypes = [vp, ci]
    lib.nbnxn_hip_download_grid.argtypes = [vp] + [vp] * 8
    lib.nbnxn_hip_download_pairlist.argtypes = [vp, ci, vp, vp, vp, vp]
    lib.nbnxn_hip_set_atom_order.argtypes = [vp, ci, vp, vp]
    lib.nbnxn_hip_upload_xq_flat.argtypes = [vp, vp, ci]
    lib.nbnxn_hip_copy_x_to_xq.argtypes = [vp, vp, ci, ci]
    lib.nbnxn_hip_add_f.argtypes = [vp, vp, ci, ci, ci, ci]
    lib.nbnxn_hip_halo_pack_x.argtypes = [vp, vp, ci, vp, vp, ci]
    lib.nbnxn_hip_halo_unpack_x.argtypes = [vp, vp, ci, vp, ci]
    lib.nbnxn_hip_halo_pack_f.argtypes = [vp, vp, ci, vp, ci]
    lib.nbnxn_hip_halo_unpack_f.argtypes = [vp, vp, ci, vp, ci, ci]
    lib.nbnxn_hip_stream_wait.argtypes = [vp, ci, ci]
    lib.nbnxn_hip_reserved_cus.argtypes = [vp]
    lib.nbnxn_hip_dd_get_unique_id.argtypes = [C.c_char_p]
    lib.nbnxn_hip_dd_init.argtypes = [vp, C.c_char_p, ci, ci]
    lib.nbnxn_hip_dd_comm_count.argtypes = [vp, C.POINTER(C.c_int)]
    lib.nbnxn_hip_dd_set_pulse.argtypes = [vp, ci, ci, vp, ci, vp, ci, vp, ci]
    lib.nbnxn_hip_dd_set_messages.argtypes = [vp, ci, vp, vp, vp]
    lib.nbnxn_hip_dd_set_phases.argtypes = [vp, ci, vp]
    lib.nbnxn_hip_dd_move_x.argtypes = [vp]
    lib.nbnxn_hip_dd_step.argtypes = [vp, ci, ci, ci, vp, vp]
    lib.nbnxn_hip_dd_move_f.argtypes = [vp]
    lib.nbnxn_hip_dd_sum_outputs.argtypes = [vp, ci]
    lib.nbnxn_hip_dd_finish.argtypes = [vp]
    lib.nbnxn_hip_get_output_slots.argtypes = [vp, C.POINTER(vp), C.POINTER(ci), C.POINTER(ci)]
    lib.nbnxn_hip_set_timing.argtypes = [vp, ci]
    lib.nbnxn_hip_set_clear_in_kernel.argtypes = [vp, ci]
    lib.nbnxn_hip_set_copy_kernels.argtypes = [vp, ci]
    lib.nbnxn_hip_set_kernel_variant.argtypes = [vp, ci]
    # the host runtime raises instead of aborting the interpreter (gmx_fatal would abort)
    lib.nbnxn_hip_set_abort_on_error(0)
    _lib = lib
    return lib


def detect_gpus():
    lib = load()
    n = C.c_int(0)
    buf = C.create_string_buffer(256)
    rc = lib.nbnxn_hip_detect_gpus(C.byref(n), buf, 256)
    if rc != 0:
        raise HipError("no usable GPU: " + buf.value.decode())
    return n.value


def detect_gpus_info():
    """detect_cuda_gpus with the per-device checks: list of abi.GpuDevInfo, number of compatible devices."""
    lib = load()
    n, nc = C.c_int(0), C.c_int(0)
    buf = C.create_string_buffer(256)
    if lib.nbnxn_hip_detect_gpus_info(None, 0, C.byref(n), C.byref(nc), buf, 256) != 0:
        raise HipError("GPU detection failed: " + buf.value.decode())
    info = (abi.GpuDevInfo * max(n.value, 1))()
    if lib.nbnxn_hip_detect_gpus_info(info, n.value, C.byref(n), C.byref(nc), buf, 256) != 0:
        raise HipError("GPU detection failed: " + buf.value.decode())
    return info, n.value, nc.value


class NbnxnHip:
    """One nbnxn_hip handle (the reference's nbnxn_cuda_ptr_t)."""

    def __init__(self, device_id=0, local_and_nonlocal=False):
        self.lib = load()
        self.h = C.c_void_p()
        self._chk(self.lib.nbnxn_hip_init(None, C.byref(self.h), device_id, int(local_and_nonlocal)))
        self.natoms = 0
        self._keep = {}

    def _chk(self, rc):
        if rc != 0:
            raise HipError(self.lib.nbnxn_hip_last_error().decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.nbnxn_hip_free(self.h)
            self.h = None

    def __del__(self):
        self.close()

    # ---- data management (nbnxn_cuda_data_mgmt.h) ----
    def init_const(self, ip, ntype, nbfp):
        nbfp = np.ascontiguousarray(nbfp, dtype=np.float32)
        assert nbfp.size == ntype * ntype * 2
        self._chk(self.lib.nbnxn_hip_init_const_flat(self.h, ip.el_type, ip.rcoulomb, ip.rvdw, ip.rlist, ip.epsfac,
                                                     ip.k_rf, ip.c_rf, ip.ewaldcoeff_q, ip.sh_ewald,
                                                     ip.disp_cpot, ip.rep_cpot, ntype, abi.ptr(nbfp)))
        self._chk(self.lib.nbnxn_hip_set_vdw_modifier_flat(self.h, ip.vdw_modifier, ip.rvdw_switch, ip.disp_c2, ip.disp_c3,
                                                           ip.rep_c2, ip.rep_c3, ip.sw_c3, ip.sw_c4, ip.sw_c5))
        if ip.vdwtype == abi.VDW_PME:
            from .system import nbfp_comb
            comb = nbfp_comb(nbfp, ntype, ip.ljpme_comb_rule)
            self._chk(self.lib.nbnxn_hip_set_ljpme_flat(self.h, ip.ljpme_comb_rule, ip.ewaldcoeff_lj, ip.sh_lj_ewald,
                                                        abi.ptr(comb)))

    def prune_only(self, rlist_inner, num_parts=1, iloc=abi.LOCAL):
        """dynamic pruning of the installed list to rlist_inner with the coordinates on the device"""
        self._chk(self.lib.nbnxn_hip_launch_kernel_pruneonly(self.h, iloc, rlist_inner, num_parts))

    # ---- pair search on the device ----
    @staticmethod
    def _box_matrix(box):
        """[3] (rectangular) or [3, 3] (GROMACS matrix, rows are box vectors) -> the 9 floats the C ABI takes"""
        box = np.asarray(box, dtype=np.float32)
        return np.ascontiguousarray(np.diag(box) if box.shape == (3,) else box, dtype=np.float32).reshape(9)

    def put_on_grid(self, box, x, type_, q, pbc_shift_dim=None, atom_density=0.0):
        """nbnxn_put_on_grid + nbnxn_atomdata_set + init_atomdata + copy_x on the device.  x: numpy
        [n,3] or an object with __cuda_array_interface__ (device array).  Returns the padded atom count."""
        box = self._box_matrix(box)
        if type_ is not None:
            type_ = np.ascontiguousarray(type_, dtype=np.int32)
            q = np.ascontiguousarray(q, dtype=np.float32)
        pbc = None if pbc_shift_dim is None else np.ascontiguousarray(pbc_shift_dim, dtype=np.int32)
        if hasattr(x, "__cuda_array_interface__"):
            xp, on_dev, n = C.c_void_p(x.__cuda_array_interface__["data"][0]), 1, x.__cuda_array_interface__["shape"][0]
        else:
            x = np.ascontiguousarray(x, dtype=np.float32)
            xp, on_dev, n = abi.ptr(x), 0, len(x)
        assert type_ is None or (len(type_) == n and len(q) == n)
        self._chk(self.lib.nbnxn_hip_put_on_grid(self.h, abi.ptr(box), abi.ptr(pbc), n, xp, on_dev, abi.ptr(type_), abi.ptr(q),
                                                 atom_density))
        dims = np.zeros(4, dtype=np.int32)
        self._chk(self.lib.nbnxn_hip_download_grid(self.h, abi.ptr(dims), None, None, None, None, None, None, None))
        self.natoms = int(dims[3])
        self.natoms_real = n
        return self.natoms

    def put_on_grid_zones(self, box, zones, x, type_, q, pbc_shift_dim=None):
        """zones: list of (corner0, corner1, a0, n) -- home zone first, then the halo zones; x / type_ / q hold all
        their atoms in that order.  Returns (padded atoms of all zones, padded atoms of the home zone)."""
        box = self._box_matrix(box)
        if type_ is not None:
            type_ = np.ascontiguousarray(type_, dtype=np.int32)
            q = np.ascontiguousarray(q, dtype=np.float32)
        pbc = None if pbc_shift_dim is None else np.ascontiguousarray(pbc_shift_dim, dtype=np.int32)
        zs = (abi.GridZone * len(zones))()
        for k, (c0, c1, a0, n) in enumerate(zones):
            zs[k].corner0[:] = [float(v) for v in c0]
            zs[k].corner1[:] = [float(v) for v in c1]
            zs[k].a0, zs[k].n, zs[k].atom_density = int(a0), int(n), 0.0
        if hasattr(x, "__cuda_array_interface__"):
            xp, on_dev, n = C.c_void_p(x.__cuda_array_interface__["data"][0]), 1, x.__cuda_array_interface__["shape"][0]
        else:
            x = np.ascontiguousarray(x, dtype=np.float32)
            xp, on_dev, n = abi.ptr(x), 0, len(x)
        self._chk(self.lib.nbnxn_hip_put_on_grid_zones(self.h, abi.ptr(box), abi.ptr(pbc), len(zones), zs, n, xp, on_dev,
                                                       abi.ptr(type_), abi.ptr(q)))
        dims = np.zeros(4, dtype=np.int32)
        self._chk(self.lib.nbnxn_hip_download_grid(self.h, abi.ptr(dims), None, None, None, None, None, None, None))
        self.natoms = int(dims[3])
        self.natoms_real = n
        return self.natoms

    def make_pairlist(self, rlist, excl_index=None, excl_a=None, min_ci_balanced=0, iloc=abi.LOCAL, zone_pairs=None):
        """zone_pairs: rows {i-zone, first j-zone, end j-zone} of the reference's dd_zp tables (eighth shell), or None."""
        if excl_index is not None:
            excl_index = np.ascontiguousarray(excl_index, dtype=np.int32)
            excl_a = np.ascontiguousarray(excl_a, dtype=np.int32)
        if zone_pairs is None:
            self._chk(self.lib.nbnxn_hip_make_pairlist(self.h, iloc, rlist, abi.ptr(excl_index), abi.ptr(excl_a), min_ci_balanced))
        else:
            zp = np.ascontiguousarray(zone_pairs, dtype=np.int32).reshape(-1, 3)
            self._chk(self.lib.nbnxn_hip_make_pairlist_zones(self.h, iloc, len(zp), abi.ptr(zp), rlist, abi.ptr(excl_index),
                                                             abi.ptr(excl_a), min_ci_balanced))

    def set_progressive_balance(self, on):
        self._chk(self.lib.nbnxn_hip_set_progressive_balance(self.h, int(on)))

    def forget_exclusions(self):
        self._chk(self.lib.nbnxn_hip_forget_exclusions(self.h))

    def download_grid(self):
        dims = np.zeros(4, dtype=np.int32)
        self._chk(self.lib.nbnxn_hip_download_grid(self.h, abi.ptr(dims), None, None, None, None, None, None, None))
        ncx, ncy, nc, npad = (int(v) for v in dims)
        out = dict(ncx=ncx, ncy=ncy, nc=nc, a=np.zeros(npad, np.int32), cell=np.zeros(self.natoms_real, np.int32),
                   bb=np.zeros((nc * 8, 6), np.float32), bbcz=np.zeros((nc, 2), np.float32), nsubc=np.zeros(nc, np.int32),
                   xq=np.zeros((npad, 4), np.float32), type=np.zeros(npad, np.int32))
        self._chk(self.lib.nbnxn_hip_download_grid(self.h, None, abi.ptr(out["a"]), abi.ptr(out["cell"]), abi.ptr(out["bb"]),
                                                   abi.ptr(out["bbcz"]), abi.ptr(out["nsubc"]), abi.ptr(out["xq"]), abi.ptr(out["type"])))
        return out

    def download_pairlist(self, iloc=abi.LOCAL):
        from .host import PairList
        cnt = np.zeros(4, dtype=np.int32)
        self._chk(self.lib.nbnxn_hip_download_pairlist(self.h, iloc, abi.ptr(cnt), None, None, None))
        sci = np.zeros(int(cnt[0]), dtype=abi.sci_dtype)
        cj4 = np.zeros(int(cnt[1]), dtype=abi.cj4_dtype)
        excl = np.zeros(int(cnt[2]), dtype=abi.excl_dtype)
        self._chk(self.lib.nbnxn_hip_download_pairlist(self.h, iloc, None, abi.ptr(sci), abi.ptr(cj4), abi.ptr(excl)))
        return PairList(sci, cj4, excl, int(cnt[3]), 0.0)

    def coulomb_tab(self):
        """(table [size] float32, scale) of the tabulated Ewald flavour, in the reference's units."""
        n, sc = C.c_int(0), C.c_float(0)
        self._chk(self.lib.nbnxn_hip_download_coulomb_tab(self.h, None, 0, C.byref(n), C.byref(sc)))
        tab = np.zeros(n.value, dtype=np.float32)
        self._chk(self.lib.nbnxn_hip_download_coulomb_tab(self.h, abi.ptr(tab), n.value, C.byref(n), C.byref(sc)))
        return tab, sc.value

    def init_const_struct(self, ic, nbv_group):
        self._chk(self.lib.nbnxn_hip_init_const(self.h, C.byref(ic), C.byref(nbv_group)))

    def init_atomdata(self, natoms, natoms_local, ntype, type_):
        type_ = np.ascontiguousarray(type_, dtype=np.int32)
        assert len(type_) == natoms
        self._chk(self.lib.nbnxn_hip_init_atomdata_flat(self.h, natoms, natoms_local, ntype, abi.ptr(type_)))
        self.natoms = natoms

    def init_atomdata_struct(self, nbat):
        self._chk(self.lib.nbnxn_hip_init_atomdata(self.h, C.byref(nbat)))
        self.natoms = nbat.natoms

    def init_pairlist(self, pl, iloc=abi.LOCAL):
        self._chk(self.lib.nbnxn_hip_init_pairlist_flat(self.h, iloc, pl.rlist, pl.nsci, abi.ptr(pl.sci), pl.ncj4,
                                                        abi.ptr(pl.cj4), pl.nexcl, abi.ptr(pl.excl), pl.nci_tot))

    def init_pairlist_struct(self, pl_struct_ptr, iloc=abi.LOCAL):
        self._chk(self.lib.nbnxn_hip_init_pairlist(self.h, pl_struct_ptr, iloc))

    def verify_pairlist(self, pl, iloc=abi.LOCAL):
        n = self.lib.nbnxn_hip_verify_pairlist_flat(self.h, iloc, pl.nsci, abi.ptr(pl.sci), pl.ncj4, abi.ptr(pl.cj4),
                                                    pl.nexcl, abi.ptr(pl.excl))
        if n < 0:
            raise HipError(self.lib.nbnxn_hip_last_error().decode())
        return n

    def download_cj4(self, pl, iloc=abi.LOCAL):
        out = np.zeros(pl.ncj4, dtype=abi.cj4_dtype)
        self._chk(self.lib.nbnxn_hip_download_cj4(self.h, iloc, pl.ncj4, abi.ptr(out)))
        return out

    def upload_shiftvec(self, shift_vec):
        sv = np.ascontiguousarray(shift_vec, dtype=np.float32)
        assert sv.size == abi.SHIFTS * 3
        self._chk(self.lib.nbnxn_hip_upload_shiftvec_flat(self.h, abi.ptr(sv)))

    def pme_loadbal_update_param(self, ic):
        self._chk(self.lib.nbnxn_hip_pme_loadbal_update_param(self.h, C.byref(ic)))

    def clear_outputs(self, flags):
        self._chk(self.lib.nbnxn_hip_clear_outputs(self.h, flags))

    def min_ci_balanced(self):
        return self.lib.nbnxn_hip_min_ci_balanced(self.h)

    def is_kernel_ewald_analytical(self):
        return bool(self.lib.nbnxn_hip_is_kernel_ewald_analytical(self.h))

    def timings(self):
        p = self.lib.nbnxn_hip_get_timings(self.h)
        return p.contents if p else None

    def reset_timings(self):
        self._chk(self.lib.nbnxn_hip_reset_timings(self.h))

    def set_kernel_variant(self, v):
        self._chk(self.lib.nbnxn_hip_set_kernel_variant(self.h, v))

    # ---- device-side atom-data moves (nbnxn_atomdata.c:1141, :1600) ----
    def set_atom_order(self, natoms_real, a, cell):
        a = np.ascontiguousarray(a, dtype=np.int32)
        cell = np.ascontiguousarray(cell, dtype=np.int32)
        assert len(a) == self.natoms and len(cell) == natoms_real
        self._chk(self.lib.nbnxn_hip_set_atom_order(self.h, natoms_real, abi.ptr(a), abi.ptr(cell)))
        self.natoms_real = natoms_real

    def upload_xq(self, xq, iloc=abi.LOCAL):
        assert xq.dtype == np.float32 and xq.shape == (self.natoms, 4) and xq.flags.c_contiguous
        self._keep["xq"] = xq
        self._chk(self.lib.nbnxn_hip_upload_xq_flat(self.h, abi.ptr(xq), iloc))

    def copy_x_to_xq(self, x, iloc=abi.LOCAL):
        """x: numpy [natoms_real, 3] float32 (host) or an int device address."""
        if isinstance(x, int):
            self._chk(self.lib.nbnxn_hip_copy_x_to_xq(self.h, C.c_void_p(x), 1, iloc))
        else:
            assert x.dtype == np.float32 and x.shape == (self.natoms_real, 3) and x.flags.c_contiguous
            self._keep["x"] = x
            self._chk(self.lib.nbnxn_hip_copy_x_to_xq(self.h, abi.ptr(x), 0, iloc))

    def add_f(self, f, a0=0, n=None, aloc=abi.LOCAL):
        """f: numpy [natoms_real, 3] float32 (host, updated on return) or an int device address."""
        n = self.natoms_real - a0 if n is None else n
        if isinstance(f, int):
            self._chk(self.lib.nbnxn_hip_add_f(self.h, C.c_void_p(f), 1, a0, n, aloc))
        else:
            assert f.dtype == np.float32 and f.shape == (self.natoms_real, 3) and f.flags.c_contiguous
            self._chk(self.lib.nbnxn_hip_add_f(self.h, abi.ptr(f), 0, a0, n, aloc))

    # ---- DD halo moves on the device (device addresses as ints; slots int32) ----
    def halo_pack_x(self, d_slots, n, shift, d_buf, iloc=abi.NONLOCAL):
        sh = shift if shift is None or isinstance(shift, C.Array) else (C.c_float * 3)(*[float(v) for v in shift])
        self._chk(self.lib.nbnxn_hip_halo_pack_x(self.h, d_slots, n, sh, d_buf, iloc))

    def halo_unpack_x(self, d_slots, n, d_buf, iloc=abi.NONLOCAL):
        self._chk(self.lib.nbnxn_hip_halo_unpack_x(self.h, d_slots, n, d_buf, iloc))

    def halo_pack_f(self, d_slots, n, d_buf, iloc=abi.NONLOCAL):
        self._chk(self.lib.nbnxn_hip_halo_pack_f(self.h, d_slots, n, d_buf, iloc))

    def halo_unpack_f(self, d_slots, n, d_buf, shift_index=-1, iloc=abi.NONLOCAL):
        self._chk(self.lib.nbnxn_hip_halo_unpack_f(self.h, d_slots, n, d_buf, shift_index, iloc))

    def reserved_cus(self):
        return int(self.lib.nbnxn_hip_reserved_cus(self.h))

    def stream_wait(self, iloc_waits, iloc_done):
        self._chk(self.lib.nbnxn_hip_stream_wait(self.h, iloc_waits, iloc_done))

    # ---- domain decomposition over RCCL, issued by the library (nbnxn_hip_dd_*) ----
    def dd_get_unique_id(self):
        buf = C.create_string_buffer(abi.DD_UNIQUE_ID_BYTES)
        self._chk(self.lib.nbnxn_hip_dd_get_unique_id(buf))
        return buf.raw

    def dd_init(self, unique_id, rank, nranks):
        assert len(unique_id) == abi.DD_UNIQUE_ID_BYTES
        self._chk(self.lib.nbnxn_hip_dd_init(self.h, unique_id, rank, nranks))

    def dd_comm_count(self):
        """ranks of the module's RCCL communicator, from ncclCommCount"""
        n = C.c_int(0)
        self._chk(self.lib.nbnxn_hip_dd_comm_count(self.h, C.byref(n)))
        return n.value

    def dd_set_pulse(self, send_to, recv_from, send_slots, recv_slots, shift=None, shift_index=-1):
        ss = np.ascontiguousarray(send_slots, dtype=np.int32)
        rs = np.ascontiguousarray(recv_slots, dtype=np.int32)
        sh = None if shift is None else np.ascontiguousarray(shift, dtype=np.float32)
        self._chk(self.lib.nbnxn_hip_dd_set_pulse(self.h, send_to, recv_from, abi.ptr(ss), len(ss), abi.ptr(rs), len(rs), abi.ptr(sh), shift_index))

    def dd_set_messages(self, msgs, send_slots, recv_slots):
        """msgs: list of (send_to, recv_from, n_send, n_recv, shift_index, shift[3]); the slot arrays hold the
        slots of all messages back to back, in message order (nbnxn_hip_dd_set_messages)."""
        ms = (abi.DdMessage * len(msgs))()
        for k, (to, frm, ns, nr, si, sh) in enumerate(msgs):
            ms[k].send_to, ms[k].recv_from, ms[k].n_send, ms[k].n_recv, ms[k].shift_index = int(to), int(frm), int(ns), int(nr), int(si)
            ms[k].shift[:] = [float(v) for v in sh]
        ss = np.ascontiguousarray(send_slots, dtype=np.int32)
        rs = np.ascontiguousarray(recv_slots, dtype=np.int32)
        assert len(ss) == sum(m[2] for m in msgs) and len(rs) == sum(m[3] for m in msgs)
        self._chk(self.lib.nbnxn_hip_dd_set_messages(self.h, len(msgs), ms, abi.ptr(ss), abi.ptr(rs)))

    def dd_set_phases(self, phase_end):
        pe = np.ascontiguousarray(phase_end, dtype=np.int32)
        self._chk(self.lib.nbnxn_hip_dd_set_phases(self.h, len(pe), abi.ptr(pe)))

    def dd_step(self, flags, sum_last_outputs, exchange_beside_local, ev_before_local=None, ev_after_local=None):
        self._chk(self.lib.nbnxn_hip_dd_step(self.h, flags, int(sum_last_outputs), int(exchange_beside_local), ev_before_local, ev_after_local))

    def dd_move_x(self):
        self._chk(self.lib.nbnxn_hip_dd_move_x(self.h))

    def dd_move_f(self):
        self._chk(self.lib.nbnxn_hip_dd_move_f(self.h))

    def dd_sum_outputs(self, iloc=abi.LOCAL):
        self._chk(self.lib.nbnxn_hip_dd_sum_outputs(self.h, iloc))

    def dd_finish(self):
        self._chk(self.lib.nbnxn_hip_dd_finish(self.h))

    def set_copy_kernels(self, on):
        """xq in / f and partial sums out by kernels on mapped host arrays (default) or by the copy engine"""
        self._chk(self.lib.nbnxn_hip_set_copy_kernels(self.h, int(on)))

    def set_clear_in_kernel(self, on):
        self._chk(self.lib.nbnxn_hip_set_clear_in_kernel(self.h, int(on)))

    def set_timing(self, on):
        self._chk(self.lib.nbnxn_hip_set_timing(self.h, int(on)))

    def set_prune(self, iloc, on):
        self._chk(self.lib.nbnxn_hip_set_prune(self.h, iloc, int(on)))

    # ---- per step (nbnxn_cuda.h) ----
    def launch_kernel(self, xq, flags, iloc=abi.LOCAL):
        assert xq.dtype == np.float32 and xq.shape == (self.natoms, 4) and xq.flags.c_contiguous
        self._keep["xq"] = xq
        self._chk(self.lib.nbnxn_hip_launch_kernel_flat(self.h, abi.ptr(xq), flags, iloc))

    def launch_kernel_resident(self, flags, iloc=abi.LOCAL):
        self._chk(self.lib.nbnxn_hip_launch_kernel_resident(self.h, flags, iloc))

    def launch_kernel_noalloc(self, flags, natoms, d_xq, d_f, d_e_fshift=None, iloc=abi.LOCAL):
        """the entry point that allocates nothing: device addresses (ints) of the caller's xq [natoms, 4],
        f [natoms, 3] (added into) and, with FORCE_VIRIAL, e_fshift [137] (assigned)"""
        self._chk(self.lib.nbnxn_hip_launch_kernel_noalloc(self.h, flags, iloc, natoms, C.c_void_p(d_xq), C.c_void_p(d_f),
                                                           C.c_void_p(d_e_fshift) if d_e_fshift else None))

    def launch_cpyback(self, f, flags, aloc=abi.LOCAL):
        assert f.dtype == np.float32 and f.shape == (self.natoms, 3) and f.flags.c_contiguous
        self._keep["f"] = f
        self._chk(self.lib.nbnxn_hip_launch_cpyback_flat(self.h, abi.ptr(f), flags, aloc))

    def wait_gpu(self, flags, aloc=abi.LOCAL, e_lj=None, e_el=None, fshift=None):
        self._chk(self.lib.nbnxn_hip_wait_gpu_flat(self.h, flags, aloc, abi.ptr(e_lj), abi.ptr(e_el), abi.ptr(fshift)))

    def read_outputs(self):
        e = np.zeros(2, dtype=np.float32)
        fs = np.zeros((abi.SHIFTS, 3), dtype=np.float32)
        self._chk(self.lib.nbnxn_hip_read_outputs(self.h, abi.ptr(e[0:1]), abi.ptr(e[1:2]), abi.ptr(fs)))
        return float(e[0]), float(e[1]), fs

    def device_buffers(self):
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._chk(self.lib.nbnxn_hip_get_device_buffers(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def output_slots(self):
        a, n, st = C.c_void_p(), C.c_int(), C.c_int()
        self._chk(self.lib.nbnxn_hip_get_output_slots(self.h, C.byref(a), C.byref(n), C.byref(st)))
        return a.value, n.value, st.value

    def stream(self, iloc=abi.LOCAL):
        return self.lib.nbnxn_hip_get_stream(self.h, iloc)

    def last_kernel_ms(self, iloc=abi.LOCAL):
        return float(self.lib.nbnxn_hip_last_kernel_ms(self.h, iloc))

    # ---- convenience: one full step as do_force_cutsVERLET issues it (sim_util.c:1012-1486) ----
    def compute(self, xq, flags=0, iloc=abi.LOCAL):
        """clear_outputs -> launch_kernel -> launch_cpyback -> wait_gpu.  Returns f, (e_lj, e_el, fshift)."""
        f = np.zeros((self.natoms, 3), dtype=np.float32)
        e_lj = np.zeros(1, dtype=np.float32)
        e_el = np.zeros(1, dtype=np.float32)
        fshift = np.zeros((abi.SHIFTS, 3), dtype=np.float32)
        self.clear_outputs(flags | abi.FORCE_VIRIAL)
        self.launch_kernel(xq, flags, iloc)
        self.launch_cpyback(f, flags, iloc)
        self.wait_gpu(flags, iloc, e_lj, e_el, fshift)
        return f, (float(e_lj[0]), float(e_el[0]), fshift)

    def compute_both(self, xq, flags=0):
        """One step with local and non-local work in the reference's order (sim_util.c:1012-1212,
        1440-1486): both launches, non-local copy-back and wait first, then local."""
        f = np.zeros((self.natoms, 3), dtype=np.float32)
        e_lj = np.zeros(1, dtype=np.float32)
        e_el = np.zeros(1, dtype=np.float32)
        fshift = np.zeros((abi.SHIFTS, 3), dtype=np.float32)
        self.clear_outputs(flags | abi.FORCE_VIRIAL)
        self.launch_kernel(xq, flags, abi.LOCAL)
        self.launch_kernel(xq, flags, abi.NONLOCAL)
        self.launch_cpyback(f, flags, abi.NONLOCAL)
        self.launch_cpyback(f, flags, abi.LOCAL)
        self.wait_gpu(flags, abi.NONLOCAL)
        self.wait_gpu(flags, abi.LOCAL, e_lj, e_el, fshift)
        return f, (float(e_lj[0]), float(e_el[0]), fshift)
