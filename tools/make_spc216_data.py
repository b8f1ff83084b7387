"""Converts the reference's equilibrated SPC water box (share/top/spc216.gro, a data file
the reference's own mdrun tests use: src/programs/mdrun/tests/) into the small .npz the
synthetic-workload generator tiles.  Data only: coordinates in nm, box in nm.
Run once in the build container: python tools/make_spc216_data.py
"""
import sys
import numpy as np

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/share/top/spc216.gro"
lines = open(src).read().splitlines()
n = int(lines[1])
xyz = np.array([[float(l[20:28]), float(l[28:36]), float(l[36:44])] for l in lines[2:2 + n]], dtype=np.float64)
box = np.array([float(t) for t in lines[2 + n].split()[:3]], dtype=np.float64)
names = [l[10:15].strip() for l in lines[2:2 + n]]
assert names[:3] == ["OW", "HW1", "HW2"] and n == 648
np.savez("gromacs-cosmo_amd/data/spc216.npz", x=xyz.astype(np.float32), box=box.astype(np.float32))
print("wrote", n, "atoms, box", box)
