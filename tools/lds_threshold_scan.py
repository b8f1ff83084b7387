"""Where does the per-entry LJ table in LDS stop paying?  The 100 k-atom chain mixture (BASELINE configs[4] in small)
with 2 ... 16 atom types, Ewald V+F (and LJ-PME V+F), once with the LDS table and once with the global table
(GMX_HIP_NB_LDS_TABLE_MAX_TYPES=0, a child process each: the switch is read once per process).
usage: python tools/lds_threshold_scan.py [natoms]"""
import json
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(natoms, ntypes):
    import time
    import torch
    import gromacs_cosmo_amd as g
    from gromacs_cosmo_amd import abi
    nb = g.gpu.NbnxnHip(0)
    nb.set_timing(False)
    nb.set_clear_in_kernel(True)
    out = {}
    for nt in ntypes:
        s = g.system.lj_chain_mixture(natoms, ntype=nt)
        for name, ip in (("ewald", g.system.ewald(1.0)), ("ljpme", g.system.with_ljpme(g.system.ewald(1.0), abi.LJCR_GEOM))):
            p = g.workload.prepare(s, ip, nb.min_ci_balanced())
            g.workload.upload(nb, p, ip)
            fl = abi.FORCE_VIRIAL
            nb.compute(p.xq, fl)
            for _ in range(20):
                nb.clear_outputs(fl); nb.launch_kernel_resident(fl)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                nb.clear_outputs(fl); nb.launch_kernel_resident(fl)
            torch.cuda.synchronize()
            out[f"{nt}:{name}"] = (time.perf_counter() - t0) / 200 * 1e6
    nb.close()
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), [int(v) for v in sys.argv[3].split(",")])
        sys.exit(0)
    natoms = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    ntypes = "2,4,6,8,10,12,14,15"      # + the filler type = ntype of the table
    res = {}
    for label, env in (("lds", {}), ("global", {"GMX_HIP_NB_LDS_TABLE_MAX_TYPES": "0"})):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(natoms), ntypes], env=dict(os.environ, **env),
                           capture_output=True, text=True, timeout=900)
        if r.returncode != 0:
            raise SystemExit(r.stderr[-2000:])
        res[label] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    print(f"{natoms}-atom chain mixture, kernel us per step (wall, back-to-back launches): LDS table / global table")
    for k in res["lds"]:
        nt, name = k.split(":")
        nt1 = int(nt) + 1
        lds = 4100 + nt1 * (1168 if name == "ljpme" else 544)
        print(f"  {nt1:2d} types in the table, {name:5s}: {res['lds'][k]:7.1f} / {res['global'][k]:7.1f}   ratio {res['lds'][k] / res['global'][k]:.3f}"
              f"   (LDS table: {lds} B per workgroup -> {min(32, 163840 // (-(-lds // 1280) * 1280))} workgroups per CU by LDS)")
