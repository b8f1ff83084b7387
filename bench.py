#!/usr/bin/env python3
"""bench.py -- pair-interactions/s and nonbonded ns/day of the Verlet cluster-pair path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload ewald|rf] [--no-cpu]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N = 1: BASELINE.json configs[2], the configuration the metric is quoted on: SPC/E water,
97,200 atoms (spc216 tiled 5x5x6), analytical Ewald real space + LJ, rc = rlist = 1.0 nm,
energies + virial every step.  (--workload rf gives configs[1]: SPC, reaction field.)
N > 1: weak scaling by default: N slabs of that box stacked along z (97,200 x N atoms), one rank per
GPU, 1-D spatial domain decomposition with RCCL halo exchange (gromacs-cosmo_amd/dd.py).
--strong: the box given by --tiles is ONE workload cut into N z slabs (BASELINE.json configs[3]:
`--tiles 13,13,14 --strong` is the 1,533,168-atom box on 1 / 2 / 4 / 8 GPUs; "scaling": "strong").
`python bench.py --gpus N` starts its own N ranks (child processes, before this process touches the
GPU) when it is not already running under torch.distributed.run.

One step = clear outputs (folded into the previous kernel launch: ping-pong output buffers,
nbnxn_hip_set_clear_in_kernel) + nonbonded kernel on coordinates and pair list
already resident in HBM; the list was pruned by the first warm-up step, as it is once per
list lifetime in the reference.  pair interactions = cluster pairs in the uploaded list x 64,
the reference's counting convention (nbnxn_search.c:5770-5777), NOT the pruned count.
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
if not os.path.exists(os.path.join(ROOT, "gromacs_cosmo_amd")):
    os.symlink("gromacs-cosmo_amd", os.path.join(ROOT, "gromacs_cosmo_amd"))

# flops per pair slot of the reference's cost model (gmxlib/nrnb.c:101-106)
FLOPS = {("rf", False): 38, ("rf", True): 54, ("ewald", False): 66, ("ewald", True): 107}
FP32_VECTOR_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md, chip-level parameters
DT_FS = 2.0


def oracle_check(prep, ip, f_gpu, e_gpu, golden=False):
    """Checker leg (runs with the CPU baseline, outside every timed region): the double-precision
    oracle over the same list gives the parity of THIS run's forces and energies and the
    reference's three pair counts (nbnxn_kernel_gpu_ref.c:362-375)."""
    import oracle as O
    o = O.kernel_gpu_list(prep.xq, prep.type, prep.ntype, prep.nbfp, prep.shift_vec, prep.pl, ip)
    fo = np.asarray(o["f"], dtype=np.float64)
    err = float(np.sqrt(((f_gpu.astype(np.float64) - fo) ** 2).sum() / (fo ** 2).sum()))
    out = {"force_rel_rms_vs_oracle": err,
           "e_lj_rel": abs(e_gpu[0] - o["Vvdw"]) / abs(o["Vvdw"]), "e_el_rel": abs(e_gpu[1] - o["Vc"]) / abs(o["Vc"]),
           "pair_slots": int(o["counts"][0]), "pair_slots_after_prune": int(o["counts"][1]),
           "pairs_in_cutoff_non_filler": int(o["counts"][2])}
    # against the committed outputs of the REFERENCE's own kernels for this very system (tests/golden/make_golden.py ran
    # nbnxn_kernel_ref and nbnxn_kernel_gpu_ref, compiled where they lie, on the same grid-ordered atoms): a sample of the
    # forces and both energies.  nbnxn_kernel_ref sums a list's energy in one float and is itself 8e-5 from the double
    # oracle at this size, so the deviation from north_star's 1e-5 on energies stays visible in the line.
    fixture = os.path.join(ROOT, "tests", "golden", "spce97k_ewald10.npz" if ip.el_type != 0 else "spc97k_rf10.npz")
    if golden and os.path.exists(fixture):
        d = np.load(fixture)
        if int(d["natoms_padded"]) == len(f_gpu):
            st = int(d["stride"])
            rr = lambda a, b: float(np.sqrt(((a.astype(np.float64) - b.astype(np.float64)) ** 2).sum() / (b.astype(np.float64) ** 2).sum()))
            out["vs_reference_kernels"] = {
                "fixture": "tests/golden/" + os.path.basename(fixture),
                "force_sample_rel_rms_vs_nbnxn_kernel_ref": rr(f_gpu[::st], d["ref_f_sample"]),
                "force_sample_rel_rms_vs_nbnxn_kernel_gpu_ref": rr(f_gpu[::st], d["gpuref_f_sample"]),
                "e_el_rel_vs_nbnxn_kernel_ref": abs(e_gpu[1] - float(d["ref_Vc"])) / abs(float(d["ref_Vc"])),
                "e_lj_rel_vs_nbnxn_kernel_ref": abs(e_gpu[0] - float(d["ref_Vvdw"])) / abs(float(d["ref_Vvdw"])),
                "e_el_rel_vs_nbnxn_kernel_gpu_ref": abs(e_gpu[1] - float(d["gpuref_Vc"])) / abs(float(d["gpuref_Vc"])),
                "e_lj_rel_vs_nbnxn_kernel_gpu_ref": abs(e_gpu[0] - float(d["gpuref_Vvdw"])) / abs(float(d["gpuref_Vvdw"]))}
    return out


def large_system_leg(g, abi, ip, flags, model, flops_pair, peak_tflops, tiles=(13, 13, 14), nsteps=40):
    """Secondary figure, outside the timed region of `value`: the same kernel on BASELINE config 4's
    1.53 M atoms on this one GPU.  At 97 k atoms the list has to be split into ~15 k short entries to fill
    256 CUs (nbnxn_hip_min_ci_balanced), and every entry pays its own prologue and i-force reduction; at
    this size the entries are whole super-clusters, which shows what the pair loop itself reaches."""
    import time
    import torch
    nb = g.gpu.NbnxnHip(0)
    nb.set_timing(False)
    nb.set_clear_in_kernel(True)
    system = g.system.water_box(*tiles, model=model)
    prep = g.workload.prepare(system, ip, nb.min_ci_balanced())
    g.workload.upload(nb, prep, ip)
    nb.compute(prep.xq, flags)                                   # prune step, coordinates now resident
    for _ in range(5):
        nb.clear_outputs(flags); nb.launch_kernel_resident(flags)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(nsteps):
        nb.clear_outputs(flags); nb.launch_kernel_resident(flags)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / nsteps
    nb.close()
    return {"atoms": system.natoms, "pair_slots_per_step": prep.pair_slots, "nsci": prep.pl.nsci, "ms_per_step": ms,
            "pairs_per_s": prep.pair_slots / (ms * 1e-3),
            "roofline_frac": prep.pair_slots * flops_pair / (ms * 1e-3) / 1e12 / peak_tflops,
            "note": "wall time per step (not kernel events), same flavour and flop count as the headline workload"}


def counted_flops(workload, ener):
    """Hardware-counted FP32 operations per launch of the timed kernel (rocprofv3 --pmc with the per-type VALU counters,
    tools/profile_round.sh -> tools/pmc_flops.py): quoted only while nbnxn_kernel.hip is byte for byte the source the
    counters were taken on (its SHA-256 is stored with them) and the workload is one of the two that were profiled."""
    import glob
    import hashlib
    key = {("ewald", True): "ewald_vf", ("rf", False): "rf_f"}.get((workload, ener))
    if key is None:
        return None
    sha = hashlib.sha256(open(os.path.join(ROOT, "gromacs-cosmo_amd", "csrc", "nbnxn_kernel.hip"), "rb").read()).hexdigest()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_flops.json")), reverse=True):
        d = json.load(open(path))
        if d.get("kernel_source_sha256") == sha and key in d:
            return dict(d[key], file=os.path.basename(path), counting_rule=d["counting_rule"])
    return None


def mixture_leg(g, abi, peak_tflops, sizes=(100000, 3000000)):
    """Secondary figure, outside the timed region of `value`: BASELINE configs[4] in kind -- the 8-type chain mixture with
    15-30 exclusions per atom (system.lj_chain_mixture; 9 types in the table with the filler type), analytical Ewald,
    energies + virial every step, at ~100 k and 3 M atoms on this one GPU, and once with LJ-PME.  The per-entry LJ table
    in LDS grows with the number of types (544 B per type, 1168 B with the LJ-PME grid column), so this is where its
    cost in resident workgroups would show."""
    import torch
    out = {}
    nb = g.gpu.NbnxnHip(0)
    nb.set_timing(False)
    nb.set_clear_in_kernel(True)
    fl = abi.FORCE_VIRIAL
    for natoms in sizes:
        s = g.system.lj_chain_mixture(natoms)
        for name, ip in (("ewald", g.system.ewald(1.0)),) + ((("ewald_ljpme", g.system.with_ljpme(g.system.ewald(1.0), abi.LJCR_GEOM)),) if natoms <= 200000 else ()):
            p = g.workload.prepare(s, ip, nb.min_ci_balanced())
            g.workload.upload(nb, p, ip)
            nb.compute(p.xq, fl)                      # prune step, coordinates resident
            n = 200 if natoms <= 200000 else 30
            for _ in range(10):
                nb.clear_outputs(fl); nb.launch_kernel_resident(fl)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                nb.clear_outputs(fl); nb.launch_kernel_resident(fl)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3 / n
            # the library's rule (lds_table_max_types, nbnxn_kernel.hip): plain LJ up to 16 types, LJ-PME up to 6
            ljpme = name.endswith("ljpme")
            cap = int(os.environ.get("GMX_HIP_NB_LDS_TABLE_MAX_TYPES", 6 if ljpme else 16))
            in_lds = p.ntype <= min(cap, 16)
            lds = 4100 + (p.ntype * (1168 if ljpme else 544) if in_lds else 0)
            out[f"{name}_{natoms}"] = {"atoms": s.natoms, "ntype": int(p.ntype), "pair_slots_per_step": p.pair_slots, "nsci": p.pl.nsci, "nexcl": p.pl.nexcl,
                                       "ms_per_step": ms, "pairs_per_s": p.pair_slots / (ms * 1e-3),
                                       "roofline_frac": p.pair_slots * FLOPS[("ewald", True)] / (ms * 1e-3) / 1e12 / peak_tflops,
                                       "lj_table": "lds" if in_lds else "global", "lds_bytes_per_workgroup": lds}
    nb.close()
    out["note"] = ("wall time per step of back-to-back launches; roofline_frac by nrnb's 107 flops per pair slot as for the headline (LJ-PME costs more "
                   "than that per pair, so its fraction understates it); 9 atom types: the per-entry LJ table is in LDS for plain LJ and in global memory for LJ-PME "
                   "(the measured switch points: tools/lds_threshold_scan.py, profiles/r3_b_lds_threshold.txt)")
    return out


def strong_leg(g, abi, dd, dist, torch, ip, flags, model, rank, world, local_rank, rccl_direct, flops_pair, tiles=(13, 13, 14), nsteps=60):
    """Secondary figure of an N > 1 run, outside the timed region of `value`: BASELINE configs[3], ONE 1,533,168-atom box cut
    into `world` z slabs (strong scaling), so that one driver command gives the weak-scaling line and this point."""
    run = dd.SlabRun(tiles, model, ip, rank, world, local_rank, flags, rccl_direct=rccl_direct, strong=True)
    run.nb.set_timing(False)
    for _ in range(10):
        run.step()
    run.flush()
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(nsteps):
        run.step()
    run.flush()
    torch.cuda.synchronize()
    dist.barrier()
    wall = time.perf_counter() - t0
    t = torch.tensor([wall, float(run.pair_slots_local)], dtype=torch.float64, device="cuda" if dist.get_backend() != "gloo" else "cpu")
    tmax, tsum = t.clone(), t.clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
    d = run.describe()
    natoms = run.natoms_global
    torch.cuda.synchronize()
    dist.barrier()
    run.nb.close()
    ms = 1e3 * float(tmax[0]) / nsteps
    return {"atoms": natoms, "scaling": "strong", "n_gpus": world, "ms_per_step": ms, "pairs_per_s": float(tsum[1]) / (ms * 1e-3),
            "ns_per_day": 86400.0 * DT_FS * 1e-6 / (ms * 1e-3), "roofline_frac_of_n_gpus": float(tsum[1]) * flops_pair / (ms * 1e-3) / 1e12 / (FP32_VECTOR_PEAK_TFLOPS * world),
            "halo_route": d["halo_route"], "rccl_ranks": d["rccl_ranks"], "halo_atoms_rank0": d["halo_atoms_rank0"], "steps": nsteps,
            "note": "python bench.py --gpus N --tiles 13,13,14 --strong gives the same as a headline line"}


def decomposed_step_leg(timeout_s=150):
    """Secondary figure, outside the timed region of `value`: the domain-decomposed step (what --gpus N runs on every
    rank) on ONE rank that is its own neighbour, so that the halo messages are real RCCL sends / receives on this one
    GPU.  Run as a child process with a time limit (it brings up a process group and an RCCL communicator, which the
    headline measurement must not depend on); any failure is reported instead of a number."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--dd-self", "--no-cpu", "--steps", "200", "--warmup", "20"],
                           env=env, capture_output=True, text=True, timeout=timeout_s)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not line:
            return {"error": f"exit code {r.returncode}", "stderr_tail": r.stderr[-300:]}
        d = json.loads(line[-1])
        return {"ms_per_step": d["ms_per_step"], "pairs_per_s": d["value"], "parallelism": d["config"]["parallelism"],
                "halo_atoms": d["config"].get("halo_atoms_rank0"), "local_kernel_ms": d["roofline"]["kernel_ms"],
                "note": "python bench.py --dd-self: local + non-local kernels, halo x / f over RCCL (to itself), slot all-reduce, per step"}
    except subprocess.TimeoutExpired:
        return {"error": f"no result within {timeout_s} s"}
    except (OSError, ValueError, KeyError) as err:
        return {"error": repr(err)}


def pcie_step_leg(g, abi, prep, ip, flags, nsteps=200):
    """Secondary figure, outside the timed region of `value`: the step through the calls sim_util.c makes, with the
    reference's calling convention: host xq in, f and energies out EVERY step (clear_outputs -> launch_kernel with host
    coordinates -> launch_cpyback -> wait_gpu), host arrays pinned as the reference's nbat->alloc hook makes them
    (nbnxn_hip_pmalloc = pmalloc, forcerec.c:2247-2262).  This is the PCIe-inclusive rate; it is never `value`."""
    import ctypes as C
    import time
    nb = g.gpu.NbnxnHip(0)
    nb.set_timing(False)
    nb.set_clear_in_kernel(True)      # as include/nbnxn_hip_gmx.h's nbnxn_cuda_init sets it: clear_outputs only flips buffers
    g.workload.upload(nb, prep, ip)
    held = []

    def pinned(shape):
        n = int(np.prod(shape)) * 4
        ptr = C.c_void_p()
        nb.lib.nbnxn_hip_pmalloc(C.byref(ptr), n)
        if not ptr.value:
            raise RuntimeError("nbnxn_hip_pmalloc failed")
        held.append(ptr)
        return np.frombuffer((C.c_char * n).from_address(ptr.value), dtype=np.float32).reshape(shape)

    xq = pinned(prep.xq.shape)
    xq[:] = prep.xq
    f, e_lj, e_el, fsh = pinned((prep.natoms, 3)), pinned((1,)), pinned((1,)), pinned((45, 3))

    def step():
        nb.clear_outputs(flags)
        nb.launch_kernel(xq, flags)
        nb.launch_cpyback(f, flags)
        nb.wait_gpu(flags, abi.LOCAL, e_lj, e_el, fsh)
    for _ in range(20):
        step()
    t0 = time.perf_counter()
    for _ in range(nsteps):
        step()
    ms = (time.perf_counter() - t0) * 1e3 / nsteps
    del xq, f, e_lj, e_el, fsh
    for ptr in held:
        nb.lib.nbnxn_hip_pfree(ptr)
    nb.close()
    return {"ms_per_step": ms, "pairs_per_s": prep.pair_slots / (ms * 1e-3), "steps": nsteps,
            "h2d_bytes_per_step": int(prep.xq.nbytes), "d2h_bytes_per_step": int(prep.natoms * 12 + (2 + 45 * 3) * 4),
            "note": "clear_outputs -> launch_kernel(host xq) -> launch_cpyback -> wait_gpu every step, pinned host arrays, "
                    "one blocking wait per step: the reference's calling convention (sim_util.c:622, 1147, 1385); xq in and "
                    "f + partial-sum slots out move by copy kernels on the mapped host arrays (NBNXN_HIP_COPY_KERNELS=0: copy engine)"}


def verlet_loop_leg(g, abi, system, ip, flags, nstlist=10, nsteps=100):
    """Secondary figure, outside the timed region of `value`: the whole device-resident step of
    gromacs-cosmo_amd/verlet.py (search on the device every nstlist steps, x gather, kernel, f scatter
    into a caller-order device array), atoms at rest, list radius = cut-off as in the headline config."""
    import time
    import torch
    nb = g.gpu.NbnxnHip(0)
    nb.set_timing(False)
    nb.set_clear_in_kernel(True)
    loop = g.verlet.VerletLoop(nb, system, ip, nstlist=nstlist)
    xd = torch.from_numpy(system.x).cuda()
    fd = torch.zeros_like(xd)
    for _ in range(nstlist + 2):
        loop.force(xd, fd, flags & ~abi.FORCE_VIRIAL)        # warm-up incl. one search
    loop.step = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(nsteps):
        loop.force(xd, fd, flags & ~abi.FORCE_VIRIAL)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / nsteps
    t0 = time.perf_counter()
    for _ in range(5):
        loop.search(xd)
    torch.cuda.synchronize()
    search_ms = (time.perf_counter() - t0) * 1e3 / 5
    nb.close()
    return {"ms_per_step": ms, "nstlist": nstlist, "steps": nsteps, "search_ms": search_ms,
            "note": "forces only; device search + x gather + kernel + f scatter per step, caller-order x and f resident in HBM"}


def cpu_threads():
    """Threads for the CPU yardstick: the box's CPU share (OMP_NUM_THREADS is set to it on the GPU box, where
    os.cpu_count() shows the whole machine), at most the cores this process may run on and the reference's own limit of
    32 lists (NBNXN_BUFFERFLAG_MAX_THREADS, nbnxn_pairlist.h:209)."""
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        share = int(os.environ.get("OMP_NUM_THREADS", "0"))
    except ValueError:
        share = 0
    return max(1, min(share if share > 0 else avail, avail, 32))


def _time_calls(fn, seconds, max_calls=100000):
    fn()
    n, t0 = 0, time.perf_counter()
    while True:
        fn()
        n += 1
        dt = time.perf_counter() - t0
        if dt > seconds or n >= max_calls:
            return n, dt


def cpu_baseline(prep, ip, ener, seconds=8.0):
    """The CPU yardstick north_star names: the reference's SIMD 4xN OpenMP kernel (oracle/_ref/libnbnxn_ref_simd.so,
    the reference's simd_4xn units compiled for AVX2_256 from where they lie, oracle/ref_simd_harness.c) on the host
    cores, same atoms and the same atom pairs as the GPU list (as a 4x8 list), same flavour as the timed GPU kernel
    (analytical Ewald or RF; energies as the GPU step has them), plus its force-only rate.  Bounded sample: repeats
    for ~`seconds` per figure.  Falls back to the reference's plain-C nbnxn_kernel_ref (oracle/_ref/libnbnxn_ref.so),
    then to the repo's scalar float port, when the library above it was not built."""
    import oracle as O
    threads = cpu_threads()
    host = {"threads": threads, "host_cores": os.cpu_count(), "cores_available": len(os.sched_getaffinity(0)),
            "survey_simd_4xn_8_threads_pair_per_s": 4.2e9}
    el = "analytical Ewald" if ip.el_type else "RF"
    if O.have_ref_simd():
        rk = O.RefKernelSimd4xN(prep.xq, prep.type, prep.ntype, prep.nbfp, prep.shift_vec, prep.pl, ip, nlist=threads)
        pairs = rk.ncj * 32      # its own list: 4x8 cluster pairs x 32 = the GPU list's cluster pairs x 64
        n, dt = _time_calls(lambda: rk.run(calc_ener=ener, nthreads=threads, outputs=False), seconds)
        nf, dtf = (n, dt) if not ener else _time_calls(lambda: rk.run(calc_ener=False, nthreads=threads, outputs=False), seconds / 2)
        rk.close()
        return dict({"value": pairs * n / dt, "unit": "pair/s", "cores": threads, "kind": "reference",
                     "kernel": f"nbnxn_kernel_simd_4xn (AVX2_256 4x8, {el}, {'energies' if ener else 'forces only'}, OpenMP)",
                     "sample": f"{n} calls of the reference's SIMD 4xN kernel ({el}, {'V+F' if ener else 'F'}) on the same {len(prep.xq)} atoms, "
                               f"{rk.ncj} 4x8 cluster pairs per call over {threads} lists / threads, {dt:.1f} s",
                     "ms_per_step": 1e3 * dt / n, "force_only_value": pairs * nf / dtf, "force_only_ms_per_step": 1e3 * dtf / nf}, **host)
    if O.have_ref():
        rk = O.RefKernelRef(prep.xq, prep.type, prep.ntype, prep.nbfp, prep.shift_vec, prep.pl, ip, nlist=threads)
        n, dt = _time_calls(lambda: rk.run(calc_ener=False, nthreads=threads, outputs=False), seconds, 200)
        pairs = rk.ncj * 16      # its own list: 4x4 cluster pairs x 16
        return dict({"value": pairs * n / dt, "unit": "pair/s", "cores": threads, "kind": "reference",
                     "kernel": f"nbnxn_kernel_ref (plain-C 4x4, {'tabulated Ewald' if ip.el_type else 'RF'}, forces only, OpenMP)",
                     "sample": f"{n} force-only calls of nbnxn_kernel_ref (plain-C 4x4, {'tabulated Ewald' if ip.el_type else 'RF'}) "
                               f"on the same {len(prep.xq)} atoms, {rk.ncj} 4x4 cluster pairs per call, {dt:.1f} s",
                     "ms_per_step": 1e3 * dt / n}, **host)
    t0 = time.perf_counter()
    O.kernel_gpu_list_f32(prep.xq, prep.type, prep.ntype, prep.nbfp, prep.shift_vec, prep.pl, ip)
    dt = time.perf_counter() - t0
    return dict({"value": prep.pair_slots / dt, "unit": "pair/s", "cores": 1, "kind": "port", "kernel": "scalar float port of the GPU-layout kernel",
                 "sample": f"1 force-only call of the scalar float port over the GPU list, {dt:.1f} s", "ms_per_step": 1e3 * dt}, **dict(host, threads=1))


def self_launch(nranks, timeout_s=1500):
    """`python bench.py --gpus N` as typed: one child per rank with the environment torch.distributed.run would give
    it (rendezvous on 127.0.0.1, a free port), every child's stderr passed through, rank 0's stdout returned as ours."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(nranks):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), LOCAL_RANK=str(r),
                   WORLD_SIZE=str(nranks), LOCAL_WORLD_SIZE=str(nranks))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    rc = 0
    try:
        out, _ = procs[0].communicate(timeout=timeout_s)
        sys.stdout.write(out)
        sys.stdout.flush()
        for p in procs:
            rc = max(rc, abs(p.wait(timeout=120)))
    except subprocess.TimeoutExpired:
        rc = 124
    finally:
        for p in procs:                      # exactly the children started here
            if p.poll() is None:
                p.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="ewald", choices=["ewald", "rf"])
    ap.add_argument("--force-only", action="store_true", help="F kernel instead of energies+virial every step")
    ap.add_argument("--tiles", default="5,5,6")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-split", action="store_true", help="do not split list entries for load balance")
    ap.add_argument("--variant", type=int, default=-1, help="kernel variant for A/B runs (-1: library default)")
    ap.add_argument("--event-every", type=int, default=10,
                    help="bracket the kernel of every Nth timed step with HIP events (an event pair costs device time itself)")
    ap.add_argument("--dd-self", action="store_true",
                    help="measurement aid: the domain-decomposed step on ONE rank that is its own neighbour (periodic in z through "
                         "a halo of its own atoms), so the messages are real RCCL sends / receives on a one-GPU box")
    ap.add_argument("--mcb", type=int, default=-1, help="list split target (-1: nbnxn_hip_min_ci_balanced)")
    ap.add_argument("--precondition", type=float, default=0.5,
                    help="seconds of untimed steps before the warm-up steps, so that the timed region runs at the clock the GPU sustains")
    ap.add_argument("--dd-grid", default="", help="ranks along x,y,z (default 1,1,N: z slabs); e.g. 2,2,2 for --gpus 8")
    ap.add_argument("--dd-scheme", default="auto", choices=["auto", "half", "eighth", "staged"],
                    help="half shell with home i-clusters only, or the reference's eighth shell with pairs between imported zones "
                         "(staged: its zones brought by the reference's pulses, one per dimension, instead of direct messages); "
                         "auto: the eighth shell when two or three dimensions are cut (3 / 7 messages instead of 4 / 13; measured 5 % / 23 % faster), "
                         "with one cut dimension the two are the same thing")
    ap.add_argument("--dd-self-dims", default="2", help="with --dd-self: the dimensions the one rank is its own neighbour in (0,1,2 = the 13-message step)")
    ap.add_argument("--dd-film", default="", help="liquid_fraction,vapour_keep: a liquid film under thin vapour instead of the full box "
                                                  "(a density step along z, the workload load balancing is for); not the headline workload")
    ap.add_argument("--dd-balance", type=int, default=0,
                    help="iterations of the reference's load-balancing rule (domdec.c:3429) on the ranks' cluster-pair counts before the run")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: --tiles is the whole box, cut into --gpus slabs (default: --tiles per GPU)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: be one.  N fresh child processes, started before this process has imported torch or
        # touched HIP (a process that holds the GPU must never be replaced or forked into ranks); rank 0's JSON line
        # is this command's output.
        raise SystemExit(self_launch(args.gpus))

    print(f"[bench] importing torch (the first import on a fresh box can take a minute or two)", file=sys.stderr, flush=True)
    import torch
    import torch.distributed as dist
    import gromacs_cosmo_amd as g
    from gromacs_cosmo_amd import abi

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (the HIP path has no CPU fallback)")
    # rehearsal of the N > 1 path on a one-GPU box: BENCH_REHEARSAL=1 puts every rank on cuda:0 and
    # uses gloo (RCCL needs one GPU per rank); its numbers mean nothing, it only proves the plumbing
    rehearsal = os.environ.get("BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    decomposed = world > 1 or args.dd_self
    if decomposed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    t_start = time.perf_counter()

    def progress(what):
        # one line per phase on stderr: a long run shows where it is (and a harness that watches for output sees it alive)
        print(f"[bench rank {rank}] {time.perf_counter() - t_start:7.1f} s  {what}", file=sys.stderr, flush=True)

    tiles = tuple(int(t) for t in args.tiles.split(","))
    ener = not args.force_only
    flags = abi.FORCE_VIRIAL if ener else 0
    model = "spce" if args.workload == "ewald" else "spc"
    ip = g.system.ewald(1.0) if args.workload == "ewald" else g.system.reaction_field(1.0)
    flops_pair = FLOPS[(args.workload, ener)]

    if not decomposed:
        system = g.system.water_box(*tiles, model=model)
        nb = g.gpu.NbnxnHip(local_rank)
        nb.set_timing(False)     # the module's own per-launch event pair costs ~7 us of device time a step
        nb.set_clear_in_kernel(True)   # each launch zeroes the next step's output buffer: no fill dispatch
        if args.variant >= 0:
            nb.set_kernel_variant(args.variant)
        mcb = 0 if args.no_split else (args.mcb if args.mcb >= 0 else nb.min_ci_balanced())
        prep = g.workload.prepare(system, ip, mcb)
        g.workload.upload(nb, prep, ip)
        assert nb.verify_pairlist(prep.pl) == 0
        natoms = system.natoms
        pair_slots = prep.pair_slots
        roof_slots = prep.pair_slots
        # coordinates resident in HBM; this first call also prunes the fresh list
        f0, e0 = nb.compute(prep.xq, abi.FORCE_VIRIAL)

        def step(ev=None):
            nb.clear_outputs(flags)
            if ev:
                ev[0].record(stream)
            nb.launch_kernel_resident(flags)
            if ev:
                ev[1].record(stream)
        stream = torch.cuda.ExternalStream(nb.stream(abi.LOCAL))
        sync = lambda: torch.cuda.synchronize()
        parallelism = "1 GPU"
        cfg_extra = {"nsci": prep.pl.nsci, "ncj4": prep.pl.ncj4, "cluster_pairs": prep.pl.nci_tot}
    else:
        from gromacs_cosmo_amd import dd
        # the halo messages and the slot reduction are issued by the HIP module on its own RCCL communicator
        # (nbnxn_hip_dd_*); BENCH_DD_TORCH=1 (and the one-GPU rehearsal, where RCCL cannot run two ranks on one
        # device) sends them through torch.distributed instead
        rccl_direct = not rehearsal and os.environ.get("BENCH_DD_TORCH") != "1"
        grid = tuple(int(v) for v in args.dd_grid.split(",")) if args.dd_grid else (1, 1, world)
        self_dims = tuple(int(v) for v in args.dd_self_dims.split(",")) if args.dd_self else ()
        ncut = sum(1 for d in range(3) if grid[d] > 1 or d in self_dims)
        if args.dd_scheme == "auto":
            args.dd_scheme = "eighth" if ncut >= 2 else "half"
        run = dd.SlabRun(tiles, model, ip, rank, world, local_rank, flags, rccl_direct=rccl_direct, strong=args.strong, grid=grid,
                         self_dims=self_dims, scheme=args.dd_scheme, balance=args.dd_balance,
                         film=(lambda v: (float(v[0]), int(v[1])))(args.dd_film.split(",")) if args.dd_film else None)
        if rccl_direct and not run.rccl_direct:
            # the module's own RCCL path was asked for and every rank fell back to torch.distributed: that is not the
            # configuration this line is supposed to measure (BENCH_DD_TORCH=1 asks for the torch route on purpose)
            print(f"[bench rank {rank}] the module's RCCL path is not in use (see the ranks' messages above): refusing to report "
                  "a number for another route; BENCH_DD_TORCH=1 measures the torch.distributed route", file=sys.stderr, flush=True)
            run.nb.close()
            dist.destroy_process_group()
            raise SystemExit(3)
        run.nb.set_timing(False)     # as for one GPU: the bench times the local kernel with its own events
        natoms = run.natoms_global
        pair_slots = run.pair_slots_local    # summed over ranks below
        roof_slots = run.dom.pl_local.nci_tot * 64   # the timed kernel is this rank's LOCAL-list launch
        step = run.step          # step(ev) records ev around the local kernel
        stream = torch.cuda.ExternalStream(run.nb.stream(abi.LOCAL))
        sync = lambda: torch.cuda.synchronize()
        shape = "z slabs" if grid == (1, 1, world) and self_dims in ((), (2,)) and args.dd_scheme == "half" else f"{grid[0]}x{grid[1]}x{grid[2]} bricks, {args.dd_scheme} shell, {len(run.dom.msgs)} direct peers' messages"
        parallelism = f"dd{world}{' (one rank, its own neighbour)' if world == 1 else ''} ({shape}, RCCL halo x/f " + ("issued by the HIP module)" if rccl_direct else "through torch.distributed)")
        cfg_extra = run.describe()
        if args.mcb >= 0 or args.variant >= 0 or args.no_split:
            raise SystemExit("--mcb/--variant/--no-split are single-GPU A/B switches")

    progress(f"set up: {natoms} atoms, {parallelism}")
    finish = run.flush if decomposed else (lambda: None)   # what the last step left for the next one to issue
    # Preconditioning, untimed and before the W warm-up steps: the GPU's clock governor takes ~100 ms of load to reach
    # the frequency it then sustains (measured on the box: the same kernel runs 81 us in the first 20 ms of load and
    # 72-73 us from ~150 ms on), and a molecular-dynamics run is millions of back-to-back steps.  So the step runs
    # for --precondition seconds first; with K = 20 the timed region would otherwise sit inside the ramp.
    t_pre = time.perf_counter()
    n_pre = 0
    def keep_preconditioning():
        go = args.precondition > 0 and time.perf_counter() - t_pre < args.precondition
        if decomposed and world > 1:
            # every rank must run the same number of steps (each step is a set of matched messages): rank 0's clock decides
            t = torch.tensor([1 if go else 0], dtype=torch.int32, device="cpu" if rehearsal else "cuda")
            dist.broadcast(t, 0)
            go = bool(int(t.item()))
        return go
    while keep_preconditioning():
        for _ in range(100):
            step()
        finish()
        sync()
        n_pre += 100
    if decomposed:
        dist.barrier()
    progress(f"preconditioned: {n_pre} steps")
    for _ in range(args.warmup):
        step()
    finish()
    sync()
    if decomposed:
        dist.barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(ev[k] if k % args.event_every == 0 else None)
    t_issued = time.perf_counter()      # the host has issued every step; the device may still be working
    finish()
    sync()
    if decomposed:
        dist.barrier()
    t1 = time.perf_counter()
    wall = t1 - t0
    progress(f"timed {args.steps} steps: {1e3 * wall / args.steps:.4f} ms per step")
    kernel_ms_raw = float(np.mean([a.elapsed_time(b) for a, b in ev[::args.event_every]]))   # the nonbonded kernel alone, HIP events on its stream
    # an event pair with nothing between its records reads ~4 us on this stack: calibrate and subtract,
    # so that the figure is comparable with rocprofv3's dispatch begin/end duration of the same kernel
    cal = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(50)]
    for a, b in cal:
        a.record(stream)
        b.record(stream)
    sync()
    event_pair_ms = float(np.median([a.elapsed_time(b) for a, b in cal]))
    kernel_ms = max(kernel_ms_raw - event_pair_ms, 1e-6)

    if decomposed:
        t = torch.tensor([wall, float(pair_slots), kernel_ms], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        wall = float(tmax[0])
        pair_slots = float(tsum[1])
        # kernel_ms stays rank 0's own: it is paired with rank 0's local-list pair count in the roofline

    # HBM bytes per launch of the timed kernel: PMC passes cannot run inside this process, so the
    # figure measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on this very command is kept
    # under profiles/ and quoted only when the workload is the one it was measured on
    traffic = None
    # (the newest pass of tools/profile_round.sh + tools/collect_profiles.py)
    tpath = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")))[-1]
    tname = os.path.basename(tpath)
    if not decomposed and args.workload == "ewald" and ener and tiles == (5, 5, 6) and args.mcb < 0 and not args.no_split \
            and os.path.exists(tpath):
        traffic = json.load(open(tpath))["hbm_bytes_per_launch"]

    if rank == 0:
        ms_per_step = 1e3 * wall / args.steps
        value = pair_slots * args.steps / wall
        achieved_tflops = roof_slots * flops_pair / (kernel_ms * 1e-3) / 1e12
        out = {
            "metric": "pair interactions/s (cluster pairs x 64 per step, nonbonded Verlet kernel)",
            "value": value, "unit": "pair/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU over gloo, numbers meaningless)" if rehearsal else ""),
            "ns_per_day": 86400.0 * DT_FS * 1e-6 / (ms_per_step * 1e-3),
            "ns_per_day_note": f"nonbonded only, dt = {DT_FS} fs",
            "host_issue_ms_per_step": 1e3 * (t_issued - t0) / args.steps,
            "config": dict({"workload": f"{model.upper()} water {natoms} atoms ({'x'.join(map(str, tiles))} spc216 tiles"
                                        f"{' per GPU' if world > 1 and not args.strong else ''}{', one box cut into ' + (str(world) + ' z slabs' if not args.dd_grid else args.dd_grid.replace(',', 'x') + ' bricks') if world > 1 and args.strong else ''}), "
                                        f"{'analytical Ewald real-space' if args.workload == 'ewald' else 'reaction-field'} + LJ, "
                                        f"rc=rlist=1.0 nm, {'energies+virial every step' if ener else 'forces only'}",
                            "parallelism": parallelism, "pair_slots_per_step": pair_slots,
                            "preconditioning": f"{n_pre} untimed steps ({args.precondition} s) before the {args.warmup} warm-up steps: GPU clock at its sustained level",
                            "flops_per_pair": flops_pair}, **cfg_extra),
            "roofline": {"bound": "fp32-vector", "achieved": achieved_tflops, "peak": FP32_VECTOR_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved_tflops / FP32_VECTOR_PEAK_TFLOPS, "traffic": traffic,
                         "traffic_note": f"HBM bytes per launch from profiles/{tname} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)" if traffic else None,
                         "kernel_ms": kernel_ms, "kernel_ms_events_raw": kernel_ms_raw, "event_pair_ms": event_pair_ms,
                         "note": "algorithmic flops = pair slots x nrnb.c cost (gmxlib/nrnb.c:101-106); duration = mean HIP-event time around the nonbonded kernel launch inside the timed region minus the latency of an empty event pair"},
        }
        cf = counted_flops(args.workload, ener) if not decomposed and tiles == (5, 5, 6) and args.mcb < 0 and not args.no_split else None
        if cf:
            out["roofline"]["achieved_counted"] = cf["flop_per_launch"] / (kernel_ms * 1e-3) / 1e12
            out["roofline"]["frac_counted"] = out["roofline"]["achieved_counted"] / FP32_VECTOR_PEAK_TFLOPS
            out["roofline"]["counted"] = {"flop_per_launch": cf["flop_per_launch"], "counters_per_launch": cf["counters_per_launch"],
                                          "lanes_on_share": cf["lanes_on_share"], "rule": cf["counting_rule"], "file": "profiles/" + cf["file"]}
        if decomposed:
            out["halo_route"] = cfg_extra.get("halo_route")
            out["rccl_ranks"] = cfg_extra.get("rccl_ranks")
        if not args.no_cpu and not decomposed and natoms > 400000:
            # a box of BASELINE configs[3] size: the CPU leg alone (bounded); the full-list oracle pass and the
            # secondary figures belong to the default workload
            out["cpu_baseline"] = cpu_baseline(prep, ip, ener)
        elif not args.no_cpu and not decomposed:
            cb = cpu_baseline(prep, ip, ener)
            out["cpu_baseline"] = cb
            chk = oracle_check(prep, ip, f0, e0, golden=(tiles == (5, 5, 6)))
            out["parity"] = chk
            # the same kernel time priced over the slots the pruned list still walks (the launch that is timed runs on
            # the pruned list; `frac` above counts every uploaded slot, the reference's convention)
            out["roofline"]["frac_post_prune_slots"] = chk["pair_slots_after_prune"] * flops_pair / (kernel_ms * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TFLOPS
            out["roofline"]["frac_pairs_in_cutoff"] = chk["pairs_in_cutoff_non_filler"] * flops_pair / (kernel_ms * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TFLOPS
            out["useful_pairs_per_s"] = chk["pairs_in_cutoff_non_filler"] * args.steps / wall
            out["pcie_step"] = pcie_step_leg(g, abi, prep, ip, flags)
            out["verlet_loop"] = verlet_loop_leg(g, abi, system, ip, flags)
            out["large_system"] = large_system_leg(g, abi, ip, flags, model, flops_pair, FP32_VECTOR_PEAK_TFLOPS)
            out["decomposed_step_one_rank"] = decomposed_step_leg()
            if args.workload == "ewald" and ener and tiles == (5, 5, 6):
                out["large_mixture"] = mixture_leg(g, abi, FP32_VECTOR_PEAK_TFLOPS)
    strong = None
    if decomposed:
        sync()
        dist.barrier()
        run.nb.close()               # also destroys the module's RCCL communicator, on every rank
        if world > 1 and not args.strong and not args.dd_self and not args.dd_grid and not args.dd_film and tiles == (5, 5, 6) \
                and os.environ.get("BENCH_NO_STRONG_LEG") != "1":
            progress("secondary leg: BASELINE configs[3], one box cut into slabs (strong scaling)")
            try:
                strong = strong_leg(g, abi, dd, dist, torch, ip, flags, model, rank, world, local_rank, rccl_direct, flops_pair,
                                    tiles=(13, 13, 14) if not rehearsal else (2, 2, 4))
            except (RuntimeError, AssertionError, MemoryError) as err:
                strong = {"error": repr(err)[:300]}
        dist.destroy_process_group()
    if rank == 0:
        if strong is not None:
            out["strong_1p5M"] = strong
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
