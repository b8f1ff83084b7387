"""CPU: the C-ABI libraries load and export every symbol include/*.h declares; the ctypes
mirrors have the C struct sizes (which oracle/abi_check.c ties to the reference headers)."""
import ctypes as C
import os
import re

import pytest

from gromacs_cosmo_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"\w+)\s*\(", text)))


def test_hip_library_exports_every_declared_symbol():
    names = declared("nbnxn_hip.h", "nbnxn_hip_")
    assert len(names) >= 35
    lib = C.CDLL(os.path.join(abi.lib_dir(), "libnbnxn_hip.so"))
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_host_library_exports_every_declared_symbol():
    names = declared("nbnxn_host.h", "nbnxn_host_")
    assert len(names) >= 15
    lib = C.CDLL(os.path.join(abi.lib_dir(), "libnbnxn_host.so"))
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_ctypes_mirror_sizes_match_reference_structs():
    # sizes printed by oracle/_ref/abi_check (built against the reference headers)
    assert C.sizeof(abi.Pairlist) == 256
    assert C.sizeof(abi.PairlistSet) == 48
    assert C.sizeof(abi.AtomdataOutput) == 64
    assert C.sizeof(abi.Atomdata) == 256
    assert C.sizeof(abi.InteractionConst) == 176
    assert C.sizeof(abi.VerletGroup) == 64
    assert C.sizeof(abi.WallclockGpu) == 104
    assert abi.sci_dtype.itemsize == 16 and abi.cj4_dtype.itemsize == 32 and abi.excl_dtype.itemsize == 128


def test_abi_check_against_reference_headers():
    exe = os.path.join(ROOT, "oracle", "_ref", "abi_check")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref not built (no reference tree here)")
    import subprocess
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert "ABI check passed" in out.stdout
