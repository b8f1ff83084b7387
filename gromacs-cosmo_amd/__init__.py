"""gromacs-cosmo_amd: MI355X-native Verlet cluster-pair nonbonded path (GROMACS 5.0.4 nbnxn).

Only what the hot path needs: the HIP kernels and C ABI (csrc/, include/), the host-side
list producers, and Python mirrors used by tests, bench.py and the multi-GPU driver.
"""
from . import abi, gpu, host, system, verlet, workload  # noqa: F401
