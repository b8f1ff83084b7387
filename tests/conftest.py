import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if not os.path.exists(os.path.join(ROOT, "gromacs_cosmo_amd")):
    # the importable alias of the package directory (its name has a hyphen)
    os.symlink("gromacs-cosmo_amd", os.path.join(ROOT, "gromacs_cosmo_amd"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
