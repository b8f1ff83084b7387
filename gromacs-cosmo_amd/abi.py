"""ctypes / numpy mirrors of the C ABI in include/nbnxn_hip.h and include/nbnxn_host.h.

The struct layouts follow the reference's (legacyheaders/types/nbnxn_pairlist.h:102-271,
types/interaction_const.h:49-130, types/nb_verlet.h:91-96); oracle/abi_check.c verifies the
C side against the reference headers, tests/test_abi.py verifies this file against the C side.
"""
import ctypes as C
import os

import numpy as np

CLUSTER_SIZE = 8
NCL_PER_SUPERCL = 8
JGROUP_SIZE = 4
SHIFTS = 45
CENTRAL = 22

FORCE_VIRIAL = 1 << 8
FORCE_ENERGY = 1 << 9

LOCAL, NONLOCAL = 0, 1
# accumulation slots of the device output buffer (csrc/nbnxn_kernel.h)
OUT_SLOTS, OUT_STRIDE = 64, 144
DD_UNIQUE_ID_BYTES = 128   # NBX_DD_UNIQUE_ID_BYTES
MAX_ZONES = 14             # NBX_MAX_ZONES
DD_MAX_MESSAGES = 13       # NBX_DD_MAX_MESSAGES
EL_RF, EL_EWALD_ANA, EL_EWALD_TAB = 0, 1, 2
EEL_CUT, EEL_RF, EEL_GRF, EEL_PME, EEL_EWALD = 0, 1, 2, 3, 4
VDW_CUT, VDW_PME = 0, 5
LJCR_GEOM, LJCR_LB, LJCR_NONE = 0, 1, 2
MOD_POTSHIFT = 1
MOD_NONE, MOD_POTSWITCH, MOD_EXACTCUTOFF, MOD_FORCESWITCH = 2, 3, 4, 5

# numpy record layouts of the list elements
sci_dtype = np.dtype([("sci", "<i4"), ("shift", "<i4"), ("cj4_ind_start", "<i4"), ("cj4_ind_end", "<i4")])
cj4_dtype = np.dtype([("cj", "<i4", (4,)), ("imei", [("imask", "<u4"), ("excl_ind", "<i4")], (2,))])
excl_dtype = np.dtype([("pair", "<u4", (32,))])
assert sci_dtype.itemsize == 16 and cj4_dtype.itemsize == 32 and excl_dtype.itemsize == 128


class CacheProtect(C.Structure):
    _fields_ = [("dummy", C.c_int * 16)]


class Pairlist(C.Structure):
    """nbnxn_pairlist_t"""
    _fields_ = [
        ("cp0", CacheProtect),
        ("alloc", C.c_void_p), ("free", C.c_void_p),
        ("bSimple", C.c_int),
        ("na_ci", C.c_int), ("na_cj", C.c_int), ("na_sc", C.c_int),
        ("rlist", C.c_float),
        ("nci", C.c_int), ("ci", C.c_void_p), ("ci_nalloc", C.c_int),
        ("nsci", C.c_int), ("sci", C.c_void_p), ("sci_nalloc", C.c_int),
        ("ncj", C.c_int), ("cj", C.c_void_p), ("cj_nalloc", C.c_int),
        ("ncj4", C.c_int), ("cj4", C.c_void_p), ("cj4_nalloc", C.c_int),
        ("nexcl", C.c_int), ("excl", C.c_void_p), ("excl_nalloc", C.c_int),
        ("nci_tot", C.c_int),
        ("work", C.c_void_p),
        ("cp1", CacheProtect),
    ]


class PairlistSet(C.Structure):
    _fields_ = [
        ("nnbl", C.c_int), ("nbl", C.POINTER(C.POINTER(Pairlist))),
        ("bCombined", C.c_int), ("bSimple", C.c_int),
        ("natpair_ljq", C.c_int), ("natpair_lj", C.c_int), ("natpair_q", C.c_int),
        ("nbl_fep", C.c_void_p),
    ]


class AtomdataOutput(C.Structure):
    _fields_ = [
        ("f", C.c_void_p), ("fshift", C.c_void_p),
        ("nV", C.c_int), ("Vvdw", C.c_void_p), ("Vc", C.c_void_p),
        ("nVS", C.c_int), ("VSvdw", C.c_void_p), ("VSc", C.c_void_p),
    ]


class BufferFlags(C.Structure):
    _fields_ = [("nflag", C.c_int), ("flag", C.c_void_p), ("flag_nalloc", C.c_int)]


class Atomdata(C.Structure):
    """nbnxn_atomdata_t"""
    _fields_ = [
        ("alloc", C.c_void_p), ("free", C.c_void_p),
        ("ntype", C.c_int), ("nbfp", C.c_void_p),
        ("comb_rule", C.c_int), ("nbfp_comb", C.c_void_p), ("nbfp_s4", C.c_void_p),
        ("natoms", C.c_int), ("natoms_local", C.c_int),
        ("type", C.c_void_p), ("lj_comb", C.c_void_p),
        ("XFormat", C.c_int), ("FFormat", C.c_int),
        ("q", C.c_void_p),
        ("na_c", C.c_int), ("nenergrp", C.c_int), ("neg_2log", C.c_int),
        ("energrp", C.c_void_p),
        ("bDynamicBox", C.c_int),
        ("shift_vec", C.c_void_p),
        ("xstride", C.c_int), ("fstride", C.c_int),
        ("x", C.c_void_p),
        ("simd_4xn_diagonal_j_minus_i", C.c_void_p), ("simd_2xnn_diagonal_j_minus_i", C.c_void_p),
        ("simd_exclusion_filter1", C.c_void_p), ("simd_exclusion_filter2", C.c_void_p),
        ("simd_interaction_array", C.c_void_p),
        ("nout", C.c_int), ("out", C.POINTER(AtomdataOutput)),
        ("nalloc", C.c_int),
        ("bUseBufferFlags", C.c_int),
        ("buffer_flags", BufferFlags),
        ("bUseTreeReduce", C.c_int),
        ("syncStep", C.c_void_p),
    ]


class ShiftConsts(C.Structure):
    _fields_ = [("c2", C.c_float), ("c3", C.c_float), ("cpot", C.c_float)]


class SwitchConsts(C.Structure):
    _fields_ = [("c3", C.c_float), ("c4", C.c_float), ("c5", C.c_float)]


class InteractionConst(C.Structure):
    """interaction_const_t"""
    _fields_ = [
        ("vdwtype", C.c_int), ("vdw_modifier", C.c_int),
        ("rvdw", C.c_float), ("rvdw_switch", C.c_float),
        ("dispersion_shift", ShiftConsts), ("repulsion_shift", ShiftConsts),
        ("vdw_switch", SwitchConsts),
        ("sh_invrc6", C.c_float),
        ("eeltype", C.c_int), ("coulomb_modifier", C.c_int),
        ("rcoulomb", C.c_float), ("rlist", C.c_float), ("rlistlong", C.c_float),
        ("ewaldcoeff_q", C.c_float), ("ewaldcoeff_lj", C.c_float),
        ("ljpme_comb_rule", C.c_int),
        ("sh_ewald", C.c_float), ("sh_lj_ewald", C.c_float),
        ("epsilon_r", C.c_float), ("epsfac", C.c_float),
        ("epsilon_rf", C.c_float), ("k_rf", C.c_float), ("c_rf", C.c_float),
        ("tabq_scale", C.c_float), ("tabq_size", C.c_int),
        ("tabq_coul_F", C.c_void_p), ("tabq_coul_V", C.c_void_p), ("tabq_coul_FDV0", C.c_void_p),
        ("tabq_vdw_F", C.c_void_p), ("tabq_vdw_V", C.c_void_p), ("tabq_vdw_FDV0", C.c_void_p),
    ]


class VerletGroup(C.Structure):
    """nonbonded_verlet_group_t"""
    _fields_ = [
        ("nbl_lists", PairlistSet), ("nbat", C.POINTER(Atomdata)),
        ("kernel_type", C.c_int), ("ewald_excl", C.c_int),
    ]


class KTime(C.Structure):
    _fields_ = [("t", C.c_double), ("c", C.c_int)]


class WallclockGpu(C.Structure):
    _fields_ = [
        ("ktime", (KTime * 2) * 2),
        ("nb_h2d_t", C.c_double), ("nb_d2h_t", C.c_double), ("nb_c", C.c_int),
        ("pl_h2d_t", C.c_double), ("pl_h2d_c", C.c_int),
    ]


def lib_dir():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib")


def ptr(a):
    """void* of a numpy array (or None)."""
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)


class GpuDevInfo(C.Structure):
    """nbx_gpu_dev_info_t (include/nbnxn_hip.h): the plain-type stand-in for cuda_dev_info_t."""
    _fields_ = [("id", C.c_int), ("stat", C.c_int), ("name", C.c_char * 256), ("arch", C.c_char * 64),
                ("compute_units", C.c_int), ("clock_mhz", C.c_int), ("total_mem", C.c_size_t)]


GPU_COMPATIBLE, GPU_NONEXISTENT, GPU_INCOMPATIBLE, GPU_INSANE = 0, 1, 2, 3


class DdMessage(C.Structure):
    """nbx_dd_message_t (include/nbnxn_hip.h)"""
    _fields_ = [("send_to", C.c_int), ("recv_from", C.c_int), ("n_send", C.c_int), ("n_recv", C.c_int),
                ("shift_index", C.c_int), ("shift", C.c_float * 3)]


class GridZone(C.Structure):
    """nbx_grid_zone_t (include/nbnxn_hip.h)"""
    _fields_ = [("corner0", C.c_float * 3), ("corner1", C.c_float * 3), ("a0", C.c_int), ("n", C.c_int),
                ("atom_density", C.c_float)]
