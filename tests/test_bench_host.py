"""CPU: the host-side parts of bench.py that decide what the line may quote -- the hardware-counted flops are tied to the
kernel source they were counted on, the HBM traffic figure to a committed profile, and the parity block's figures against
the reference's own kernels to the committed 97 k-atom fixtures."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import gromacs_cosmo_amd as g  # noqa: E402
import oracle as O  # noqa: E402


def test_counted_flops_belong_to_the_kernel_source_in_the_tree():
    """roofline.achieved_counted is quoted only while nbnxn_kernel.hip is the source the counters were taken on; a kernel
    change without a new `bash tools/profile_round.sh` pass would silently drop it from the line, so it fails here first"""
    sha = hashlib.sha256(open(os.path.join(ROOT, "gromacs-cosmo_amd", "csrc", "nbnxn_kernel.hip"), "rb").read()).hexdigest()
    for workload, ener, nrnb in (("ewald", True, 107), ("rf", False, 38)):
        cf = bench.counted_flops(workload, ener)
        assert cf is not None, f"no profiles/r*_pmc_flops.json for kernel source {sha[:12]}: run tools/profile_round.sh on the GPU box"
        c = cf["counters_per_launch"]
        want = 64.0 * (2 * c["SQ_INSTS_VALU_FMA_F32"] + c["SQ_INSTS_VALU_MUL_F32"] + c["SQ_INSTS_VALU_ADD_F32"] + c["SQ_INSTS_VALU_TRANS_F32"])
        assert cf["flop_per_launch"] == want
        # the hardware issues fewer FP32 operations than nrnb prices (pruned slots, masked lanes still issue), never more
        assert 0.4 * nrnb * 66325760 < want < nrnb * 66325760
        assert 0.3 < cf["lanes_on_share"] < 1.0
    assert bench.counted_flops("ewald", False) is None and bench.counted_flops("rf", True) is None     # not profiled


def test_traffic_profile_is_committed():
    """bench.py quotes the newest profiles/r*_hbm_traffic.json: it must belong to the same profiling pass as the counted
    flops of the kernel source in the tree (tools/profile_round.sh + tools/collect_profiles.py write both)"""
    import glob
    path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")))[-1]
    d = json.load(open(path))
    assert d["hbm_bytes_per_launch"] == 1000 * (2 * d["FETCH_SIZE_KB_per_launch"] + d["WRITE_SIZE_KB_per_launch"])
    cf = bench.counted_flops("ewald", True)
    assert cf is not None and cf["file"].split("_pmc_flops")[0] == os.path.basename(path).split("_hbm_traffic")[0]


def test_parity_block_quotes_the_reference_kernels_fixture():
    """oracle_check on the bench workload, fed the oracle's own forces in place of the GPU's: the block names the fixture
    and holds the figures the line would carry (forces of the two reference kernels within 1e-5 of the double oracle, the
    one-float energy sums of nbnxn_kernel_ref within 2e-4, those of nbnxn_kernel_gpu_ref within 2e-5)"""
    s = g.system.water_box(5, 5, 6, model="spce")
    ip = g.system.ewald(1.0)
    p = g.workload.prepare(s, ip, 10240)
    o = O.kernel_gpu_list(p.xq, p.type, p.ntype, p.nbfp, p.shift_vec, p.pl, ip)
    chk = bench.oracle_check(p, ip, np.asarray(o["f"], dtype=np.float32), (o["Vvdw"], o["Vc"]), golden=True)
    assert chk["force_rel_rms_vs_oracle"] < 1e-7 and chk["pair_slots"] == p.pair_slots
    r = chk["vs_reference_kernels"]
    assert r["fixture"] == "tests/golden/spce97k_ewald10.npz"
    assert r["force_sample_rel_rms_vs_nbnxn_kernel_ref"] < 1e-5 and r["force_sample_rel_rms_vs_nbnxn_kernel_gpu_ref"] < 1e-5
    assert r["e_el_rel_vs_nbnxn_kernel_ref"] < 2e-4 and r["e_lj_rel_vs_nbnxn_kernel_ref"] < 2e-4
    assert r["e_el_rel_vs_nbnxn_kernel_gpu_ref"] < 2e-5 and r["e_lj_rel_vs_nbnxn_kernel_gpu_ref"] < 2e-5
    assert "vs_reference_kernels" not in bench.oracle_check(p, ip, np.asarray(o["f"], dtype=np.float32), (o["Vvdw"], o["Vc"]))
