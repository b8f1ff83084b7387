"""Counts the cluster pairs left in the 97k lists after the kernel's prune step."""
import sys; sys.path.insert(0, ".")
import numpy as np
import gromacs_cosmo_amd as g
nb = g.gpu.NbnxnHip(0)
for name, ip, model in (("rf", g.system.reaction_field(1.0), "spc"), ("ewald", g.system.ewald(1.0), "spce")):
    s = g.system.water_box(5, 5, 6, model=model)
    p = g.workload.prepare(s, ip, nb.min_ci_balanced())
    g.workload.upload(nb, p, ip)
    nb.compute(p.xq, 0)
    cj4 = nb.download_cj4(p.pl)
    im = np.ascontiguousarray(cj4["imei"]["imask"][:, 0]).astype(np.uint32)
    bits = np.unpackbits(im.view(np.uint8)).sum()
    nz = sum(int(((im >> (8 * k)) & 0xff != 0).sum()) for k in range(4))
    print(name, "nsci", p.pl.nsci, "ncj4", p.pl.ncj4, "cluster pairs before", p.pl.nci_tot, "after prune", int(bits),
          "j-clusters with any pair", nz, flush=True)
