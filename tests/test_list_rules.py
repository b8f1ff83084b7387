"""The grid and the list of a hand-made toy system, EXACTLY, against what the reference's rules give for it.

`nbnxn_search.c` cannot be compiled in this image (it needs cmake-generated headers), so the list the
reference would make for the 97,200-atom box cannot be produced here, and its published statistics (1,570 super-cells,
1,042,421 cluster pairs) are those of mdrun's start configuration -- genconf's box after grompp and the constraints of
the first step --, not of the raw spc216 tiling this repo builds: the super-cell count depends only on how many atoms
fall into each x/y column, i.e. on the coordinates.  What CAN be pinned without those coordinates is every rule by
which the reference turns coordinates into a grid and a list.  This file restates those rules in plain Python, each
with the lines it comes from, applies them to a 150-atom system small enough to check by eye, and demands the builder's
arrays entry for entry:

  * number of columns              set_grid_size_xy            nbnxn_search.c:433-486
  * column of an atom              calc_column_indices         :1493-1590
  * order inside a column          sort_columns_supersub       :1374-1490  (z; then y per 32, BACKWARDS in every odd
                                   + sort_atoms                :610-739     z-half; then x per 16, backwards in every
                                                                            odd y-half; ties by atom index, reversed
                                                                            in a backward sort; unused slots last)
  * filler coordinates             nbnxn_atomdata.c:256-295 with fill_cell's sx / sy / sz (nbnxn_search.c:1474-1479)
  * j-cells visited, in which order  nbnxn_make_pairlist_part  :5160-5330  (same grid, central shift: cj >= ci)
  * j-clusters, interaction masks  make_cluster_list_supersub  :2994-3122  (own cell: i-clusters 0..cjo only)
  * one cj4 range per entry        new_sci_entry / close_ci_entry_supersub :3936-4060
  * diagonal exclusion words       set_self_and_newton_excls_supersub :2759-2782, numbering :2715-2728
  * order of the entries           sort_sci                    :5517-5580

The device search is held to the host builder entry for entry elsewhere (tests/test_gpu_search.py), so what is pinned
here holds for both."""
import numpy as np

import gromacs_cosmo_amd as g
from gromacs_cosmo_amd import abi
from gromacs_cosmo_amd.host import NbnxnHost

NA_C, NA_SC, FAR = 8, 64, 107.0


def toy():
    """150 atoms in a 4 x 2 x 2 nm cell without periodicity: 85 land in the column x < 2, 65 in the other.  Coordinates
    are multiples of 1/64 nm (exact in float), with deliberate ties in z, y and x so that the tie rules show."""
    rng = np.random.default_rng(12)
    n = 150
    x = np.empty((n, 3), dtype=np.float32)
    left = np.zeros(n, dtype=bool)
    left[rng.permutation(n)[:85]] = True
    x[:, 0] = np.where(left, rng.integers(2, 126, n), rng.integers(130, 254, n)) / 64.0
    x[:, 1] = rng.integers(2, 126, n) / 64.0
    x[:, 2] = rng.integers(2, 126, n) / 64.0
    x[7, 2] = x[3, 2]; x[40, 2] = x[3, 2]          # equal z: the order falls back on the atom index
    x[11, 1] = x[5, 1]; x[90, 0] = x[60, 0]
    return x, np.array([4.0, 2.0, 2.0], dtype=np.float32)


def expected_grid(x, box, tie=1):
    """tie=1: the reference's rule (equal coordinates in atom-index order, the whole order reversed in a backward sort);
    tie=-1 is the opposite rule, used only to show that the toy system can tell the two apart"""
    n = len(x)
    # set_grid_size_xy: ~cubic sub-cells of na_c atoms, columns of 2 x 2 of them, rounded DOWN (:466-475)
    dens = np.float32(n) / (box[0] * box[1] * box[2])
    tlen = np.float32(NA_C / dens) ** np.float32(1.0 / 3.0)
    ncx, ncy = max(1, int(box[0] / (tlen * 2))), max(1, int(box[1] / (tlen * 2)))
    sx, sy = box[0] / ncx, box[1] / ncy
    # calc_column_indices (:1545-1570): truncation, clamped to the grid
    col = [min(max(int(xi[0] / sx), 0), ncx - 1) * ncy + min(max(int(xi[1] / sy), 0), ncy - 1) for xi in x]
    a, first_cell = [], []
    for c in range(ncx * ncy):
        cx, cy = divmod(c, ncy)
        first_cell.append(len(a) // NA_SC)
        atoms = sorted([i for i in range(n) if col[i] == c], key=lambda i: (x[i, 2], tie * i))     # z, ties by index
        ncz = -(-len(atoms) // NA_SC)
        out = []
        for sub_z in range(2 * ncz):
            half = atoms[32 * sub_z:32 * sub_z + 32]
            half = sorted(half, key=lambda i: (x[i, 1], tie * i), reverse=bool(sub_z & 1))            # :1448
            for sub_y in range(2):
                quarter = half[16 * sub_y:16 * sub_y + 16]
                quarter = sorted(quarter, key=lambda i: (x[i, 0], tie * i), reverse=bool(sub_y & 1))  # :1462, cz * 2 even
                out += quarter
        a += out + [-1] * (ncz * NA_SC - len(out))                                                 # :1485-1488
    first_cell.append(len(a) // NA_SC)
    return (ncx, ncy), np.array(a), first_cell


def expected_fillers(a, ncy, first_cell):
    """slot -> (x, y, z) of a filler: -107 (1 + sx), -107 (1 + sy), -107 (1 + sz + i), i the place in its cluster"""
    out = {}
    for s, at in enumerate(a):
        if at >= 0:
            continue
        sc = s // NA_SC
        c = max(k for k in range(len(first_cell) - 1) if first_cell[k] <= sc)
        cx, cy = divmod(c, ncy)
        cl = (s % NA_SC) // NA_C
        sub_z = 2 * (sc - first_cell[c]) + cl // 4
        out[s] = (-FAR * (1 + NA_C * (2 * cx + (cl & 1))), -FAR * (1 + NA_C * (2 * cy + ((cl >> 1) & 1))), -FAR * (1 + NA_C * sub_z + s % NA_C))
    return out


def expected_list(a):
    """every cluster pair in range, no periodic images, no splitting: entries, groups, masks, diagonal exclusions"""
    nc = len(a) // NA_SC
    nsubc = [min(8, -(-int((a[NA_SC * c:NA_SC * (c + 1)] >= 0).sum()) // NA_C)) for c in range(nc)]
    sci, cj4, excl = [], [], [np.full(32, 0xffffffff, dtype=np.uint32)]
    for ci in range(nc):
        start = len(cj4)
        slot = 0
        for cj in range(ci, nc):                               # :5268-5281, cells ascend with the column index
            for cjo in range(nsubc[cj]):
                if slot % 4 == 0:
                    cj4.append({"cj": [0, 0, 0, 0], "imask": 0, "ei": [0, 0]})
                grp, jm = cj4[-1], slot % 4
                grp["cj"][jm] = cj * 8 + cjo
                ci1 = cjo + 1 if cj == ci else nsubc[ci]       # :3008-3015
                for i in range(ci1):
                    grp["imask"] |= 1 << (jm * 8 + i)
                if cj == ci:                                   # :2759-2782: j <= i of the diagonal cluster pair
                    for w in range(2):
                        if grp["ei"][w] == 0:
                            grp["ei"][w] = len(excl)
                            excl.append(np.full(32, 0xffffffff, dtype=np.uint32))
                    for ej in range(8):
                        for ei in range(ej, 8):
                            excl[grp["ei"][ej >> 2]][(ej & 3) * 8 + ei] &= np.uint32(~(1 << (jm * 8 + cjo)) & 0xffffffff)
                slot += 1
        if len(cj4) > start:
            sci.append((ci, 22, start, len(cj4)))              # CENTRAL = 22, types/ishift.h:45-55
    # sort_sci: counting sort on min(m, groups), larger first, order kept among equals
    m = (2 * len(cj4)) // len(sci)
    if len(cj4) > len(sci):
        sci = [e for k in range(m, -1, -1) for e in sci if min(m, e[3] - e[2]) == k]
    return sci, cj4, excl, sum(bin(grp["imask"]).count("1") for grp in cj4)


def test_toy_grid_follows_the_reference_rules():
    x, box = toy()
    (ncx, ncy), a, first_cell = expected_grid(x, box)
    assert (ncx, ncy) == (2, 1) and first_cell == [0, 2, 4]        # 85 and 65 atoms: two super-cells each
    h = NbnxnHost(box, pbc_shift_dim=(0, 0, 0))
    h.put_on_grid(0, (0.0, 0.0, 0.0), box, 0, x)
    assert h.ncolumns() == (2, 1)
    assert np.array_equal(h.atom_order(), a)
    cell = np.full(len(x), -1)
    cell[a[a >= 0]] = np.nonzero(a >= 0)[0]
    assert np.array_equal(h.cell_index(), cell)
    # the toy system has equal coordinates inside sort ranges: the opposite tie rule gives another order
    assert not np.array_equal(expected_grid(x, box, tie=-1)[1], a)
    # atom data in grid order: coordinates of the atoms, the reference's far-away fillers with zero charge and the filler type
    xq, typ = h.set_atomdata(3, np.ones(len(x), dtype=np.int32), np.full(len(x), 0.5, dtype=np.float32))
    h.copy_x_to_xq(x, xq)
    fill = expected_fillers(a, ncy, first_cell)
    assert len(fill) == 4 * NA_SC - len(x)
    for s, at in enumerate(a):
        if at >= 0:
            assert np.array_equal(xq[s, :3], x[at]) and xq[s, 3] == 0.5 and typ[s] == 1
        else:
            assert tuple(float(v) for v in xq[s, :3]) == fill[s] and xq[s, 3] == 0.0 and typ[s] == 2


def test_toy_list_follows_the_reference_rules():
    x, box = toy()
    _, a, _ = expected_grid(x, box)
    h = NbnxnHost(box, pbc_shift_dim=(0, 0, 0))
    h.put_on_grid(0, (0.0, 0.0, 0.0), box, 0, x)
    xq, _ = h.set_atomdata(3, np.ones(len(x), dtype=np.int32), np.zeros(len(x), dtype=np.float32))
    h.copy_x_to_xq(x, xq)
    pl = h.make_pairlist(abi.LOCAL, 10.0, xq)                       # rlist beyond the cell: every cluster pair is in range
    sci, cj4, excl, nci_tot = expected_list(a)
    assert (pl.nsci, pl.ncj4, pl.nexcl, pl.nci_tot) == (len(sci), len(cj4), len(excl), nci_tot)
    assert [tuple(int(e[k]) for k in ("sci", "shift", "cj4_ind_start", "cj4_ind_end")) for e in pl.sci] == sci
    for k, grp in enumerate(cj4):
        got = pl.cj4[k]
        assert int(got["imei"]["imask"][0]) == grp["imask"] and int(got["imei"]["imask"][1]) == grp["imask"], k
        assert [int(v) for v in got["imei"]["excl_ind"]] == grp["ei"], k
        for jm in range(4):
            if (grp["imask"] >> (8 * jm)) & 0xff:                   # slots without a j-cluster are undefined (:2460-2472)
                assert int(got["cj"][jm]) == grp["cj"][jm], (k, jm)
    for k, e in enumerate(excl):
        assert np.array_equal(pl.excl[k]["pair"], e), k
    # four entries: each cell against itself and the cells after it; the first has most groups, and sort_sci keeps that order
    assert [e[0] for e in sci] == [0, 1, 2, 3]
