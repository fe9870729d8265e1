"""Synthetic adaptive grids in the DART-Ray array layout.

The layout is the contract of the reference's grid builders
(`create_adap_grid_trustI.f90:157-395`, child order `:335-347`) as read by
`read_main_grid` (`io_routines.f90:297`) and consumed by the ray tracer through
`smooth_grid_routines.f90:42-58,74-80` (`cchild`, `lvl`, `cindex`, `ccoord`, `csize`):

* cell 0 is the root; the children of a cell are contiguous, `cchild[i]` is the index of the
  first child or -1 for a leaf; the root's children start at index 1
  (`ccindd_to_cc`, `smooth_grid_routines.f90:820-852`, starts its descent at l=1);
* the root splits `base[0]**3` ways, deeper levels `base[1]**3`; child order is x fastest,
  `k = (iz*b + iy)*b + ix`;
* `cindex` ORs `(k+1) * basediv[0] * basediv[1]**(lvl-2)` per level (`basediv` = next power of
  two >= base**3 + 1, `set_base`, `smooth_grid_routines.f90:622-649`);
* the builders keep face neighbours within one level of each other (`check_level_jump`,
  `smooth_grid_routines.f90:774-787`); `balance=True` reproduces that.

This module is our own generator (the reference's builders are out of scope, SURVEY.md section 8f
rank 4): breadth-first by level, vectorised with numpy so that 10^6-cell trees build in seconds.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, Optional

import numpy as np

# Single-precision literals in the reference (smooth_grid_routines.f90: cspeed, parsec).
CSPEED = float(np.float32(2.99792458e8))  # m/s
PARSEC = float(np.float32(3.08567758e16))  # m
CS_PC = CSPEED / PARSEC  # light speed in pc/s, `cs` for units_csize='pc' (rt_routines.f90:5192)

# The reference builds pi from single-precision asin(1.) (smooth_grid_routines.f90:311-317).
HALFPI_F32 = float(np.float32(np.arcsin(np.float32(1.0))))
PI_REF = 2.0 * HALFPI_F32


def basediv(base: int) -> int:
    """`set_base` (smooth_grid_routines.f90:632-643): smallest 2**j with base**3 < 2**j."""
    j = 0
    while not (base ** 3 < 2 ** j):
        j += 1
    return 2 ** j


def level_sizes(modelsize: float, base, max_lvl: int) -> np.ndarray:
    """`make_csize_arr` (io_routines.f90:669-696): csize_arr(0:max_lvl)."""
    binv = (1.0 / float(base[0]), 1.0 / float(base[1]))

    def ipow(x, n):  # integer power by binary exponentiation, as a Fortran runtime does
        r = 1.0
        while n > 0:
            if n & 1:
                r *= x
            x *= x
            n >>= 1
        return r

    out = np.empty(max_lvl + 1)
    for i in range(max_lvl + 1):
        out[i] = modelsize * (binv[0] * ipow(binv[1], i - 1)) if i > 0 else modelsize
    return out


@dataclass
class Tree:
    modelsize: float
    base: tuple
    max_lvl: int
    cchild: np.ndarray  # int32 [N]
    lvl: np.ndarray  # int32 [N]
    cindex: np.ndarray  # int64 [N]
    ccoord: np.ndarray  # f64 [N,3]  (Fortran (3,N))
    csize: np.ndarray  # f64 [N]
    icoord: np.ndarray  # int64 [N,3] integer coordinates at the cell's own level resolution
    extra: Dict[str, np.ndarray] = field(default_factory=dict)

    @property
    def ncell(self) -> int:
        return int(self.cchild.shape[0])

    @property
    def leaf(self) -> np.ndarray:
        return self.cchild == -1


def _res(base, level: int) -> int:
    return 1 if level == 0 else base[0] * base[1] ** (level - 1)


def _key(c: np.ndarray, res: int) -> np.ndarray:
    return (c[:, 2] * res + c[:, 1]) * res + c[:, 0]


def build_tree(modelsize: float, base, max_lvl: int,
               refine: Callable[[int, np.ndarray, float], np.ndarray],
               balance: bool = True) -> Tree:
    """Build a tree. `refine(level, centres[n,3], size)` -> bool[n] says which cells of `level`
    (1 <= level < max_lvl) split further. The root always splits."""
    base = (int(base[0]), int(base[1]))
    sizes = level_sizes(modelsize, base, max_lvl)

    # Pass 1: sets of refined cells per level, as integer coordinates.
    refined = {0: np.zeros((1, 3), dtype=np.int64)}
    cur = _children_coords(refined[0], base[0])
    for l in range(1, max_lvl):
        res = _res(base, l)
        ctr = (cur + 0.5) * sizes[l] - 0.5 * modelsize
        m = np.asarray(refine(l, ctr, sizes[l]), dtype=bool)
        refined[l] = cur[m]
        if refined[l].shape[0] == 0:
            for k in range(l + 1, max_lvl):
                refined[k] = np.zeros((0, 3), dtype=np.int64)
            break
        cur = _children_coords(refined[l], base[1])
    for l in range(1, max_lvl):
        refined.setdefault(l, np.zeros((0, 3), dtype=np.int64))

    # Pass 2 (fine -> coarse): 2:1 balance across faces. A refined cell q at level l+1 (its children
    # live at l+2) forces the level-l parent of each of its six face neighbours to be refined.
    if balance:
        for l in range(max_lvl - 2, 0, -1):
            q = refined[l + 1]
            if q.shape[0] == 0:
                continue
            res1 = _res(base, l + 1)
            need = []
            for ax in range(3):
                for s in (-1, 1):
                    n = q.copy()
                    n[:, ax] += s
                    ok = (n[:, ax] >= 0) & (n[:, ax] < res1)
                    need.append(n[ok] // base[1])
            # ancestors of refined cells must be refined too
            need.append(q // base[1])
            need = np.concatenate(need, axis=0)
            both = np.concatenate([refined[l], need], axis=0)
            res = _res(base, l)
            _, idx = np.unique(_key(both, res), return_index=True)
            refined[l] = both[np.sort(idx)]
        # level-1 ancestors exist by construction (root always splits)

    # Pass 3: linearise breadth-first. Level-l cells = children of refined level-(l-1) cells in the
    # order of their parents; refined level-l cells must then be looked up among them.
    lv_coords = [np.zeros((1, 3), dtype=np.int64)]
    lv_parent = [np.full(1, -1, dtype=np.int64)]
    lv_cchild = []
    offset = [0]
    par_refined_idx = np.array([0], dtype=np.int64)  # indices (within level) of refined cells
    for l in range(0, max_lvl):
        b = base[0] if l == 0 else base[1]
        coords = lv_coords[l]
        nref = par_refined_idx.shape[0]
        cc = np.full(coords.shape[0], -1, dtype=np.int64)
        next_off = offset[l] + coords.shape[0]
        cc[par_refined_idx] = next_off + np.arange(nref, dtype=np.int64) * b ** 3
        lv_cchild.append(cc)
        offset.append(next_off)
        kids = _children_coords(coords[par_refined_idx], b)
        lv_coords.append(kids)
        lv_parent.append(np.repeat(par_refined_idx + offset[l], b ** 3))
        if l + 1 < max_lvl and refined[l + 1].shape[0] > 0:
            res = _res(base, l + 1)
            kkeys = _key(kids, res)
            order = np.argsort(kkeys, kind="stable")
            rkeys = _key(refined[l + 1], res)
            pos = np.searchsorted(kkeys[order], rkeys)
            ok = (pos < kkeys.shape[0])
            pos = np.where(ok, pos, 0)
            hit = ok & (kkeys[order][pos] == rkeys)
            par_refined_idx = np.sort(order[pos[hit]])
        else:
            par_refined_idx = np.zeros(0, dtype=np.int64)
        if kids.shape[0] == 0:
            break
    nlev = len(lv_coords)
    while nlev > 0 and lv_coords[nlev - 1].shape[0] == 0:
        nlev -= 1
    lv_cchild.append(np.full(lv_coords[nlev - 1].shape[0], -1, dtype=np.int64)) \
        if len(lv_cchild) < nlev else None
    used_max_lvl = nlev - 1

    N = sum(lv_coords[l].shape[0] for l in range(nlev))
    cchild = np.empty(N, dtype=np.int32)
    lvl = np.empty(N, dtype=np.int32)
    cindex = np.zeros(N, dtype=np.int64)
    ccoord = np.zeros((N, 3))
    csize = np.empty(N)
    icoord = np.zeros((N, 3), dtype=np.int64)
    bd = (basediv(base[0]), basediv(base[1]))
    pos = 0
    for l in range(nlev):
        n = lv_coords[l].shape[0]
        sl = slice(pos, pos + n)
        cchild[sl] = lv_cchild[l][:n] if l < len(lv_cchild) else -1
        lvl[sl] = l
        csize[sl] = sizes[l]
        icoord[sl] = lv_coords[l]
        if l > 0:
            b = base[0] if l == 1 else base[1]
            par = lv_parent[l]
            loc = lv_coords[l] % b
            k = (loc[:, 2] * b + loc[:, 1]) * b + loc[:, 0]
            mult = 1 if l == 1 else bd[0] * bd[1] ** (l - 2)
            cindex[sl] = cindex[par] | ((k + 1) * mult)
            # create_adap_grid_trustI.f90:360-371: centre relative to the parent's centre
            if b % 2 == 1:
                off = (loc - b // 2).astype(np.float64)
            else:
                off = (loc - b // 2 + 0.5).astype(np.float64)
            ccoord[sl] = ccoord[par] + sizes[l] * off
        pos += n
    return Tree(modelsize=float(modelsize), base=base, max_lvl=int(used_max_lvl),
                cchild=cchild, lvl=lvl, cindex=cindex, ccoord=ccoord, csize=csize, icoord=icoord)


def _children_coords(parents: np.ndarray, b: int) -> np.ndarray:
    """Children of `parents` (integer coords), parent-major, x fastest within a parent."""
    if parents.shape[0] == 0:
        return np.zeros((0, 3), dtype=np.int64)
    iz, iy, ix = np.meshgrid(np.arange(b), np.arange(b), np.arange(b), indexing="ij")
    loc = np.stack([ix.ravel(), iy.ravel(), iz.ravel()], axis=1).astype(np.int64)  # x fastest
    return (parents[:, None, :] * b + loc[None, :, :]).reshape(-1, 3)


# ----------------------------------------------------------------------------------------------
# Opacity / phase-function set-up (`set_npix_arr`, rt_routines.f90:5235-5315)
# ----------------------------------------------------------------------------------------------

def kp_for_g(g: float, kp_sca_max: int) -> int:
    """HEALPix k for the scattering-source sphere at one wavelength; -1 = isotropic.
    Restates rt_routines.f90:5254-5273 including its single-precision constants."""
    g = abs(float(g))
    g_th = 2 * float(np.float32(1e-3))
    if g < g_th:
        return -1
    pmax = (1 - g ** 2) / (1 + g ** 2 - 2 * g) ** 1.5
    pmin = (1 - g ** 2) / (1 + g ** 2 + 2 * g) ** 1.5
    twothird = float(np.float32(2.0) / np.float32(3.0))
    cos_hm = ((2 * (1 - g ** 2) / (pmax + pmin)) ** twothird - (1 + g ** 2)) / (-2 * g)
    fwhm = 2 * np.arccos(cos_hm)
    delta_min = fwhm / 5
    l2 = float(np.log10(np.float32(2.0)))
    k_min = int(1 / (2 * l2) * np.log10(4 * PI_REF / (12 * delta_min ** 2)))
    return min(k_min, kp_sca_max)


# ----------------------------------------------------------------------------------------------
# Problem definitions ("cases") shared by the oracle driver, the CPU restatement and the GPU path
# ----------------------------------------------------------------------------------------------

@dataclass
class Case:
    """Everything `rt_loop` reads from module state (SURVEY.md section 8b "State in")."""
    tree: Tree
    dens: np.ndarray  # f64 [N,L]   dens_arr(L,N) : extinction coefficient per pc
    dens_stars: np.ndarray  # f64 [N,L]   dens_stars_arr(L,N) : volume emissivity
    ksca_norm: np.ndarray  # f64 [L]     albedo
    gsca: np.ndarray  # f64 [L]
    p_src_pos: np.ndarray  # f64 [nsrc,3]
    p_src_lum: np.ndarray  # f64 [nsrc,L]
    dir_out: np.ndarray  # f64 [ndir,2] (theta, phi)
    kp_sca_max: int = 1
    rad_lim: float = 0.1
    accuracy: float = 0.1
    conv_en_lim: float = 0.01
    bm_par: int = 10
    bm_par_sca: int = 5
    bm_par_max: int = 200
    wall_on: tuple = (False, False, False)
    wall_coord: np.ndarray = field(default_factory=lambda: np.zeros((3, 2)))  # model units
    max_sca_iterations: int = -1  # <0: unlimited
    dist_obs: float = 10e3
    name: str = "case"

    @property
    def lnum(self) -> int:
        return int(self.dens.shape[1])

    def write(self, path: str) -> None:
        """Little-endian stream file read by oracle/ref_driver.f90 and the C restatement."""
        t = self.tree
        N, L = t.ncell, self.lnum
        nsrc, ndir = self.p_src_pos.shape[0], self.dir_out.shape[0]
        hdr = np.array([0x44525431, N, t.max_lvl, t.base[0], t.base[1], L, nsrc, ndir,
                        self.kp_sca_max, self.bm_par, self.bm_par_sca, self.bm_par_max,
                        int(self.wall_on[0]), int(self.wall_on[1]), int(self.wall_on[2]),
                        self.max_sca_iterations], dtype=np.int32)
        dbl = np.array([t.modelsize, self.rad_lim, self.accuracy, self.conv_en_lim, self.dist_obs,
                        CS_PC] + list(np.asarray(self.wall_coord, dtype=np.float64).ravel()),
                       dtype=np.float64)
        with open(path, "wb") as f:
            hdr.tofile(f)
            dbl.tofile(f)
            t.cchild.astype(np.int32).tofile(f)
            t.lvl.astype(np.int32).tofile(f)
            t.cindex.astype(np.int64).tofile(f)
            np.ascontiguousarray(t.ccoord, dtype=np.float64).tofile(f)  # (3,N) Fortran order
            t.csize.astype(np.float64).tofile(f)
            np.ascontiguousarray(self.dens, dtype=np.float64).tofile(f)  # (L,N) Fortran order
            np.ascontiguousarray(self.dens_stars, dtype=np.float64).tofile(f)
            np.ascontiguousarray(self.ksca_norm, dtype=np.float64).tofile(f)
            np.ascontiguousarray(self.gsca, dtype=np.float64).tofile(f)
            np.ascontiguousarray(self.p_src_pos, dtype=np.float64).tofile(f)  # (3,nsrc)
            np.ascontiguousarray(self.p_src_lum, dtype=np.float64).tofile(f)  # (L,nsrc)
            np.ascontiguousarray(self.dir_out.T, dtype=np.float64).tofile(f)  # (ndir,2) Fortran


def slab_case(tau_z: float = 0.01, base=(10, 2), max_lvl: int = 2, nlambda: int = 1,
              albedo=0.6, g=0.55, lum: float = 1.0, ndir: int = 2, refine_layers: int = 1,
              emissivity: float = 0.0, name: str = "slab", walls: bool = False,
              modelsize: float = 10.0, dens_scale=None, **kw) -> Case:
    """TRUST-I-like slab (user_routines_trustI.f90:15-19, 167-218): uniform dust in
    x,y in [-5,5], z in [-5,-2]; one point source at (0,0,4); the cells whose top face touches
    z=-2 are refined level after level (`create_adap_grid_trustI.f90:88-89`)."""
    z1 = -2.0 * modelsize / 10.0
    z0 = -5.0 * modelsize / 10.0

    def refine(level, ctr, size):
        return np.abs(ctr[:, 2] + size / 2.0 - z1) / abs(z1) < 1e-5

    tree = build_tree(modelsize, base, max_lvl, refine, balance=True)
    N = tree.ncell
    inside = (tree.ccoord[:, 2] >= z0) & (tree.ccoord[:, 2] <= z1)
    dens_ref = np.where(inside, tau_z / (z1 - z0), 0.0)
    dens_ref[~tree.leaf] = np.where(inside[~tree.leaf], tau_z / (z1 - z0), 0.0)
    L = nlambda
    if dens_scale is None:
        dens_scale = np.array([1.0 / (1.0 + 0.35 * i) for i in range(L)])
    alb = np.broadcast_to(np.asarray(albedo, dtype=np.float64), (L,)).copy()
    gg = np.broadcast_to(np.asarray(g, dtype=np.float64), (L,)).copy()
    dens = dens_ref[:, None] * np.asarray(dens_scale)[None, :]
    dens_stars = np.zeros((N, L))
    if emissivity > 0:
        dens_stars[inside & tree.leaf] = emissivity
    theta = np.linspace(0.0, np.pi, ndir) if ndir > 1 else np.array([0.5])
    dir_out = np.stack([theta, np.zeros_like(theta)], axis=1) if ndir > 0 else np.zeros((0, 2))
    p_lum = np.array([[lum * (1.0 + 0.1 * i) for i in range(L)]])
    wall_coord = np.zeros((3, 2))
    wall_on = (False, False, False)
    if walls:  # z_wall_coord = 0, 0.3 in units of modelsize (set_walls, rt_routines.f90:2620-2621)
        wall_on = (False, False, True)
        wall_coord[2] = np.array([0.0, 0.3]) * modelsize - modelsize / 2.0
    return Case(tree=tree, dens=dens, dens_stars=dens_stars, ksca_norm=alb, gsca=gg,
                p_src_pos=np.array([[0.0, 0.0, 4.0 * modelsize / 10.0]]), p_src_lum=p_lum,
                dir_out=dir_out, wall_on=wall_on, wall_coord=wall_coord, name=name, **kw)


def read_case(path: str, name: str = "case") -> Case:
    """Inverse of `Case.write`."""
    with open(path, "rb") as f:
        hdr = np.fromfile(f, np.int32, 16)
        dbl = np.fromfile(f, np.float64, 12)
        if hdr[0] != 0x44525431:
            raise ValueError("not a DART-Ray case file")
        N, max_lvl, b0, b1, L, nsrc, ndir = (int(x) for x in hdr[1:8])
        cchild = np.fromfile(f, np.int32, N)
        lvl = np.fromfile(f, np.int32, N)
        cindex = np.fromfile(f, np.int64, N)
        ccoord = np.fromfile(f, np.float64, 3 * N).reshape(N, 3)
        csize = np.fromfile(f, np.float64, N)
        dens = np.fromfile(f, np.float64, N * L).reshape(N, L)
        dens_stars = np.fromfile(f, np.float64, N * L).reshape(N, L)
        ksca = np.fromfile(f, np.float64, L)
        gsca = np.fromfile(f, np.float64, L)
        ppos = np.fromfile(f, np.float64, 3 * nsrc).reshape(nsrc, 3)
        plum = np.fromfile(f, np.float64, nsrc * L).reshape(nsrc, L)
        dirs = np.fromfile(f, np.float64, 2 * ndir).reshape(2, ndir).T.copy()
    res = np.array([_res((b0, b1), int(l)) for l in range(max_lvl + 1)])
    icoord = np.floor((ccoord + 0.5 * dbl[0]) / csize[:, None]).astype(np.int64)
    tree = Tree(modelsize=float(dbl[0]), base=(b0, b1), max_lvl=max_lvl, cchild=cchild, lvl=lvl,
                cindex=cindex, ccoord=ccoord, csize=csize, icoord=icoord)
    return Case(tree=tree, dens=dens, dens_stars=dens_stars, ksca_norm=ksca, gsca=gsca,
                p_src_pos=ppos, p_src_lum=plum, dir_out=dirs, kp_sca_max=int(hdr[8]),
                bm_par=int(hdr[9]), bm_par_sca=int(hdr[10]), bm_par_max=int(hdr[11]),
                wall_on=(bool(hdr[12]), bool(hdr[13]), bool(hdr[14])),
                wall_coord=dbl[6:12].reshape(3, 2).copy(), max_sca_iterations=int(hdr[15]),
                rad_lim=float(dbl[1]), accuracy=float(dbl[2]), conv_en_lim=float(dbl[3]),
                dist_obs=float(dbl[4]), name=name)


def blob_case(base=(4, 2), max_lvl: int = 3, nlambda: int = 2, tau: float = 1.0,
              albedo=(0.6, 0.5, 0.4), g=(0.55, 0.0, 0.3), ndir: int = 2, balance: bool = True,
              emissivity: float = 1.0, p_src=((0.31, -0.23, 0.13),), lum: float = 50.0,
              modelsize: float = 8.0, blob_r: float = 2.0, blob_c=(0.5, 0.4, -0.3),
              walls: bool = False, name: str = "blob", rad_lim: float = 0.1037, **kw) -> Case:
    """A dense Gaussian blob, refined towards its centre, with diffuse stellar emissivity and point
    sources: exercises every branch of `deposit` (rt_routines.f90:2426-2537), including the
    go_low / 4-sibling merge path (ray_list.f90:210-296) when rays leave the refined core.

    Source positions and rad_lim are deliberately "un-round": with a source at z=0.1, a cell face at
    z=0.5, a HEALPix ring at cos(theta)=0.5 and rad_lim*modelsize=0.8 the precalc test
    `psel > rad_lim*modelsize` (rt_routines.f90:2430) lands exactly on its threshold and its outcome
    depends on the last bit of the math library's cos(acos(0.5)), i.e. differs between any two libms."""
    blob_c = np.asarray(blob_c, dtype=np.float64)

    def refine(level, ctr, size):
        d = np.sqrt(((ctr - blob_c) ** 2).sum(axis=1))
        return d < blob_r * (0.5 ** (level - 1)) + 0.5 * size

    tree = build_tree(modelsize, base, max_lvl, refine, balance=balance)
    N, L = tree.ncell, nlambda
    d2 = ((tree.ccoord - blob_c) ** 2).sum(axis=1)
    prof = np.exp(-d2 / (2 * (0.6 * blob_r) ** 2))
    dens_ref = tau / blob_r * prof
    dens_ref[prof < 1e-3] = 0.0
    scale = np.array([1.0 / (1.0 + 0.5 * i) for i in range(L)])
    dens = dens_ref[:, None] * scale[None, :]
    dens_stars = np.zeros((N, L))
    em = emissivity * np.exp(-d2 / (2 * (1.2 * blob_r) ** 2))
    em[em < 0.05 * emissivity] = 0.0
    dens_stars[:, :] = (em * tree.leaf)[:, None] * np.array([1.0 + 0.2 * i for i in range(L)])[None, :]
    alb = np.array([albedo[i % len(albedo)] for i in range(L)], dtype=np.float64)
    gg = np.array([g[i % len(g)] for i in range(L)], dtype=np.float64)
    ppos = np.asarray(p_src, dtype=np.float64).reshape(-1, 3)
    plum = np.array([[lum * (1.0 + 0.3 * i) / (1 + k) for i in range(L)] for k in range(ppos.shape[0])]).reshape(-1, L)
    theta = np.array([0.4 + 0.9 * j for j in range(ndir)])
    phi = np.array([0.3 + 1.7 * j for j in range(ndir)])
    dir_out = np.stack([theta, phi], axis=1) if ndir > 0 else np.zeros((0, 2))
    wall_coord = np.zeros((3, 2))
    wall_on = (False, False, False)
    if walls:
        wall_on = (True, False, True)
        wall_coord[0] = np.array([0.1, 0.9]) * modelsize - modelsize / 2.0
        wall_coord[2] = np.array([0.2, 0.8]) * modelsize - modelsize / 2.0
    return Case(tree=tree, dens=dens, dens_stars=dens_stars, ksca_norm=alb, gsca=gg, p_src_pos=ppos,
                p_src_lum=plum, dir_out=dir_out, wall_on=wall_on, wall_coord=wall_coord, name=name,
                rad_lim=rad_lim, **kw)


# TRUST-I benchmark inputs at the 32 wavelengths <= 80 um of lambda_list_trustI_basic.dat
# (max_lambda_stars = 80, TRUSTI_GRIDS/input_trustI_tau0.01_LD.in:26). Columns: lambda [um],
# kext(lambda)/kext(1 um), albedo, g (linear interpolation of DUST_OPACITY/TRUST/
# ZDA_BARE_GR_S_Effective.dat as read_av_opacities does, sed_routines.f90:1870-1898), and the
# stellar luminosity [erg/s/Hz] (log-log interpolation of TRUSTI_GRIDS/BB_T10000_L100000.dat as
# read_planck_table_trustI does, user_routines_trustI.f90:147-156). Data values, not code.
TRUSTI_TABLE = np.array([
    [0.1, 1.48628301e+01, 2.52981121e-01, 6.01118286e-01, 4.76832615e+20],
    [0.1232847, 1.00709124e+01, 3.25539366e-01, 6.40600022e-01, 3.85294469e+21],
    [0.1519911, 7.70059606e+00, 3.63591507e-01, 6.29863804e-01, 1.86367634e+22],
    [0.1873817, 7.94580506e+00, 4.10966812e-01, 5.73687408e-01, 5.94661528e+22],
    [0.231013, 8.88322940e+00, 4.13310438e-01, 5.12659389e-01, 1.35522835e+23],
    [0.2848036, 5.86727588e+00, 5.06974607e-01, 5.01633103e-01, 2.35545137e+23],
    [0.3511192, 4.68497404e+00, 5.20719047e-01, 4.97661707e-01, 3.29767403e+23],
    [0.4328761, 3.72095632e+00, 5.34140383e-01, 4.74774658e-01, 3.89268628e+23],
    [0.5336699, 2.84719102e+00, 5.44275949e-01, 4.37074093e-01, 4.02310769e+23],
    [0.6579332, 2.09286437e+00, 5.38148445e-01, 3.87315011e-01, 3.75280092e+23],
    [0.8111308, 1.47695426e+00, 5.10677160e-01, 3.26241452e-01, 3.23622778e+23],
    [1, 1.00000000e+00, 4.62392434e-01, 2.56744818e-01, 2.62814005e+23],
    [1.232847, 6.52910690e-01, 3.95526564e-01, 1.79934337e-01, 2.03844874e+23],
    [1.519911, 4.18986620e-01, 3.19163076e-01, 1.11819171e-01, 1.52623360e+23],
    [1.873817, 2.66698116e-01, 2.39483173e-01, 6.73208815e-02, 1.11199553e+23],
    [2.31013, 1.70983358e-01, 1.64225864e-01, 4.17986483e-02, 7.93224327e+22],
    [2.848036, 1.12893789e-01, 1.04609117e-01, 2.80629336e-02, 5.56555751e+22],
    [3.511192, 7.73526699e-02, 6.35725160e-02, 2.03539864e-02, 3.85464952e+22],
    [4.328761, 5.46361755e-02, 3.74490654e-02, 1.50211806e-02, 2.64245569e+22],
    [5.336699, 4.02422391e-02, 2.09754844e-02, 1.05937462e-02, 1.79681859e+22],
    [6.579332, 3.22588818e-02, 1.04977963e-02, 7.13060833e-03, 1.21392724e+22],
    [8.111308, 6.85582275e-02, 1.85445941e-03, 4.22375202e-03, 8.15904440e+21],
    [10, 1.69830399e-01, 6.59885530e-04, 2.58846264e-03, 5.46122723e+21],
    [12.32847, 7.17302125e-02, 6.67741106e-04, 1.84036727e-03, 3.64336360e+21],
    [15.19911, 4.84080702e-02, 3.31415484e-04, 2.68951938e-04, 2.42411830e+21],
    [18.73817, 6.81248935e-02, 1.33411280e-04, -2.46638927e-04, 1.60943094e+21],
    [23.1013, 4.91642129e-02, 9.13372963e-05, -5.98686433e-04, 1.06669851e+21],
    [28.48036, 4.19282936e-02, 5.03081880e-05, -1.04008515e-03, 7.05992378e+20],
    [35.11192, 3.34789598e-02, 3.09689191e-05, -1.27201698e-03, 4.66735618e+20],
    [43.28761, 1.97148319e-02, 2.35483794e-05, -8.91400727e-04, 3.08277586e+20],
    [53.36699, 1.12982455e-02, 1.75951571e-05, 1.91485524e-04, 2.03467388e+20],
    [65.79332, 6.75419853e-03, 1.25726332e-05, 1.06116246e-03, 1.34210697e+20],
])


def trusti_case(tau_z: float = 1.0, max_lvl: int = 4, wavelengths=None, base=(10, 2), name: str = "trustI",
                top_layers: int = 1, **kw) -> Case:
    """The TRUST-I slab benchmark as the reference sets it up (BASELINE configs 1 and 2):
    TRUSTI_GRIDS/input_grid_trustI_tau0.01_LD.in (modelsize 10 pc, base 10,2, refinement of the
    cells touching the slab's top face z = -2), input_trustI_tau0.01_LD.in (kp_sca_max=1,
    rad_lim=0.1, accuracy=0.1, conv_en_lim=0.01, bm_par=10, bm_par_sca=5, bm_par_max=200, z wall
    0..0.3, 7 observer directions theta = 0..pi), point source at (0,0,4) with the T=10^4 K black body.
    `wavelengths`: indices into TRUSTI_TABLE (default: the single 1-um entry for tau_z=0.01-style
    runs is index 11; pass range(32) for the full list). `top_layers` > 1 also refines the cells
    below the top layer (used to reach ~10^5 cells)."""
    if wavelengths is None:
        wavelengths = [11]
    tab = TRUSTI_TABLE[list(wavelengths)]
    modelsize = 10.0
    z0, z1 = -5.0, -2.0

    def refine(level, ctr, size):
        top = np.abs(ctr[:, 2] + size / 2.0 - z1) / abs(z1) < 1e-5
        if top_layers > 1:
            depth = (z1 - (ctr[:, 2] + size / 2.0)) / size  # 0 for the top layer, 1 for the next ...
            top = top | ((depth > -0.5) & (depth < top_layers - 0.5) & (ctr[:, 2] > z0) & (ctr[:, 2] < z1))
        return top

    tree = build_tree(modelsize, base, max_lvl, refine, balance=True)
    N = tree.ncell
    inside = (tree.ccoord[:, 2] >= z0) & (tree.ccoord[:, 2] <= z1)
    dens_ref = np.where(inside, tau_z / (z1 - z0), 0.0)
    dens = dens_ref[:, None] * tab[None, :, 1]
    L = tab.shape[0]
    theta = np.array([0.0, 0.523599, 1.04720, 1.57080, 2.09440, 2.61799, 3.14159])  # dir_out_trustI.dat
    dir_out = np.stack([theta, np.zeros_like(theta)], axis=1)
    wall_coord = np.zeros((3, 2))
    wall_coord[2] = np.array([0.0, 0.3]) * modelsize - modelsize / 2.0
    return Case(tree=tree, dens=dens, dens_stars=np.zeros((N, L)), ksca_norm=tab[:, 2].copy(), gsca=tab[:, 3].copy(),
                p_src_pos=np.array([[0.0, 0.0, 4.0]]), p_src_lum=tab[None, :, 4].copy(), dir_out=dir_out,
                kp_sca_max=1, rad_lim=0.1, accuracy=0.1, conv_en_lim=0.01, bm_par=10, bm_par_sca=5, bm_par_max=200,
                wall_on=(False, False, True), wall_coord=wall_coord, dist_obs=10e3, name=name, **kw)


# ----------------------------------------------------------------------------------------------
# BASELINE configs 3-5: synthetic counterparts in the reference layout.
# The reference's own builders and density profiles (create_adap_grid_*.f90, user_routines_*.f90)
# are out of scope; these generators take the set-up from the shipped input files and fill the
# grids with analytic profiles, so that every number can be regenerated from this file alone.
# ----------------------------------------------------------------------------------------------

def _observer_dirs(ndir: int) -> np.ndarray:
    theta = np.linspace(0.0, PI_REF, ndir) if ndir > 1 else np.array([0.7])
    phi = np.array([0.37 * j for j in range(ndir)])
    return np.stack([theta, phi], axis=1) if ndir > 0 else np.zeros((0, 2))


def shell_case(max_lvl: int = 3, base=(10, 3), nlambda: int = 4, modelsize: float = 6.0, tau_pc: float = 0.48,
               semi_axes=(1.09, 2.18, 2.18), half_width: float = 0.10, dtau_max: float = 0.048, ndir: int = 3,
               lum: float = 4.3e30, name: str = "magtar_shell", **kw) -> Case:
    """BASELINE config 3, the MAGTAR dust shell (MAGTAR_GRIDS/input_grid_shell.in): base 10,3, max_lvl 3,
    an ellipsoidal shell of semi-axes (1.09, 2.18, 2.18) pc and normalised half-width 0.10 with
    extinction 0.48 pc^-1 at the reference wavelength, refined where a cell's optical depth exceeds
    0.048, one point source at the origin (param_point_sources.dat). `nlambda` wavelengths share the
    TRUST dust column ratios of TRUSTI_TABLE around 1 um (the MAGTAR opacity file is not restated)."""
    ax = np.asarray(semi_axes, dtype=np.float64)

    def shell_r(ctr):
        return np.sqrt(((ctr / ax) ** 2).sum(axis=1))  # 1 on the shell's mid surface

    def refine(level, ctr, size):
        near = np.abs(shell_r(ctr) - 1.0) < half_width + 0.87 * size / ax.min()
        return near & (tau_pc * size > dtau_max)

    tree = build_tree(modelsize, base, max_lvl, refine, balance=True)
    N = tree.ncell
    idx = [8, 11, 14, 17, 5, 20, 2, 23][:nlambda] if nlambda <= 8 else list(range(nlambda))
    tab = TRUSTI_TABLE[idx]
    inside = np.abs(shell_r(tree.ccoord) - 1.0) < half_width
    dens = np.where(inside, tau_pc, 0.0)[:, None] * tab[None, :, 1]
    L = tab.shape[0]
    plum = (lum * tab[:, 4] / tab[:, 4].max())[None, :]
    return Case(tree=tree, dens=dens, dens_stars=np.zeros((N, L)), ksca_norm=tab[:, 2].copy(), gsca=tab[:, 3].copy(),
                p_src_pos=np.array([[0.013, -0.021, 0.017]]), p_src_lum=plum, dir_out=_observer_dirs(ndir),
                kp_sca_max=1, rad_lim=0.1031, accuracy=0.1, conv_en_lim=0.01, bm_par=10, bm_par_sca=5, bm_par_max=200,
                name=name, **kw)


def galaxy_case(max_lvl: int = 3, base=(3, 3), nlambda: int = 3, modelsize: float = 28000.0, hr: float = 4000.0,
                hz_dust: float = 200.0, hz_stars: float = 350.0, tau_face: float = 1.5, bulge_r: float = 1200.0,
                dtau_max: float = 0.05, ndir: int = 2, walls: bool = True, name: str = "galaxy", **kw) -> Case:
    """BASELINE config 4, GALAXY example 1 (GALAXY_GRIDS/input_grid_galaxy_example1.in): modelsize 28 kpc,
    27-ary tree (base 3,3), every emitting leaf cell is a source (no point sources), bm_par 5,
    accuracy 0.3, z walls 0.3-0.7 (input_galaxy_example1.in). Dust and stars follow double-exponential
    discs (scale length hr, scale heights hz_*) plus a Gaussian stellar bulge: an analytic stand-in for
    user_routines_galaxy.f90. Refinement: cells whose central optical depth exceeds dtau_max."""
    kdust0 = tau_face / (2.0 * hz_dust)  # face-on optical depth through the centre = tau_face

    def kdust(c):
        R = np.sqrt(c[:, 0] ** 2 + c[:, 1] ** 2)
        return kdust0 * np.exp(-R / hr - np.abs(c[:, 2]) / hz_dust)

    def refine(level, ctr, size):
        # the value at the point of the cell nearest to the mid-plane / axis decides (profiles peak there)
        near = ctr.copy()
        near[:, 2] = np.sign(ctr[:, 2]) * np.maximum(np.abs(ctr[:, 2]) - size / 2.0, 0.0)
        Rn = np.maximum(np.sqrt(ctr[:, 0] ** 2 + ctr[:, 1] ** 2) - 0.71 * size, 0.0)
        k = kdust0 * np.exp(-Rn / hr - np.abs(near[:, 2]) / hz_dust)
        return k * size > dtau_max

    tree = build_tree(modelsize, base, max_lvl, refine, balance=True)
    N = tree.ncell
    if nlambda <= 6:
        lam_scale = np.array([1.6, 1.0, 0.55, 0.3, 0.18, 0.1][:nlambda])
        star_spec = np.array([0.4, 1.0, 1.3, 1.1, 0.8, 0.5][:nlambda])
        alb = np.array([0.62, 0.58, 0.5, 0.4, 0.3, 0.2][:nlambda])
        gg = np.array([0.55, 0.5, 0.42, 0.3, 0.2, 0.1][:nlambda])
    else:
        # many wavelengths (BASELINE config 4 names 24; the shipped list has 16 below 5 um): the TRUST dust columns from
        # 0.15 um on, extinction relative to the 0.53-um entry, a 10^4-K stellar spectrum shape
        tab = TRUSTI_TABLE[2:2 + nlambda]
        lam_scale = tab[:, 1] / TRUSTI_TABLE[8, 1]
        star_spec = tab[:, 4] / tab[:, 4].max()
        alb, gg = tab[:, 2].copy(), tab[:, 3].copy()
    L = lam_scale.shape[0]
    kd = kdust(tree.ccoord)
    kd[kd < 1e-4 * kdust0] = 0.0
    dens = kd[:, None] * lam_scale[None, :]
    c = tree.ccoord
    R = np.sqrt(c[:, 0] ** 2 + c[:, 1] ** 2)
    stars = np.exp(-R / hr - np.abs(c[:, 2]) / hz_stars) + 4.0 * np.exp(-(c ** 2).sum(axis=1) / (2 * bulge_r ** 2))
    stars[stars < 2e-3] = 0.0
    dens_stars = (stars * tree.leaf)[:, None] * star_spec[None, :] * 1e20
    wall_coord = np.zeros((3, 2))
    wall_on = (False, False, False)
    if walls:
        wall_on = (False, False, True)
        wall_coord[2] = np.array([0.3, 0.7]) * modelsize - modelsize / 2.0
    return Case(tree=tree, dens=dens, dens_stars=dens_stars, ksca_norm=alb,
                gsca=gg, p_src_pos=np.zeros((0, 3)), p_src_lum=np.zeros((0, L)),
                dir_out=_observer_dirs(ndir), kp_sca_max=1, rad_lim=0.1017, accuracy=0.3, conv_en_lim=0.01, bm_par=5,
                bm_par_sca=5, bm_par_max=200, wall_on=wall_on, wall_coord=wall_coord, dist_obs=10e6, name=name, **kw)


def nbody_case(max_lvl: int = 3, base=(3, 3), nlambda: int = 2, modelsize: float = 40.0, npart: int = 20000,
               max_part: int = 24, seed: int = 708, ndir: int = 2, name: str = "nbody", **kw) -> Case:
    """BASELINE config 5, N-body/SPH example 1: the snapshot is not shipped (.gitignore:17-20), so the tree is
    refined around pseudo-particles drawn from a fixed-seed (PCG64, 708) Plummer sphere + exponential disc:
    a cell splits while it holds more than `max_part` particles (the criterion of
    create_adap_grid_Nbody_SPH.f90 restated). Gas particles give the dust density, star particles the
    emissivity; every emitting leaf is a source. The tree is irregular and unbalanced levels are possible
    before the 2:1 pass, which is what makes it a different test from the analytic grids."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n1 = npart // 2
    # Plummer sphere, scale a
    a = 0.06 * modelsize
    u = rng.random(n1)
    r = a / np.sqrt(u ** (-2.0 / 3.0) - 1.0)
    ct = 2 * rng.random(n1) - 1
    ph = 2 * np.pi * rng.random(n1)
    st = np.sqrt(1 - ct * ct)
    p1 = np.stack([r * st * np.cos(ph), r * st * np.sin(ph), r * ct], axis=1)
    n2 = npart - n1
    R = rng.exponential(0.12 * modelsize, n2)
    ph2 = 2 * np.pi * rng.random(n2)
    z = rng.laplace(0.0, 0.012 * modelsize, n2)
    p2 = np.stack([R * np.cos(ph2), R * np.sin(ph2), z], axis=1)
    part = np.concatenate([p1, p2], axis=0) + np.array([0.011, -0.007, 0.003]) * modelsize
    part = part[(np.abs(part) < 0.4999 * modelsize).all(axis=1)]
    is_gas = rng.random(part.shape[0]) < 0.5

    def count(level, ctr, size, sel=None):
        p = part if sel is None else part[sel]
        res = int(round(modelsize / size))
        ip = np.clip(np.floor((p + modelsize / 2.0) / size).astype(np.int64), 0, res - 1)
        keys = (ip[:, 2] * res + ip[:, 1]) * res + ip[:, 0]
        ic = np.clip(np.floor((ctr + modelsize / 2.0) / size).astype(np.int64), 0, res - 1)
        ckeys = (ic[:, 2] * res + ic[:, 1]) * res + ic[:, 0]
        uk, cnt = np.unique(keys, return_counts=True)
        pos = np.searchsorted(uk, ckeys)
        pos = np.clip(pos, 0, max(uk.shape[0] - 1, 0))
        hit = uk[pos] == ckeys if uk.shape[0] else np.zeros(ckeys.shape[0], dtype=bool)
        return np.where(hit, cnt[pos] if uk.shape[0] else 0, 0)

    def refine(level, ctr, size):
        return count(level, ctr, size) > max_part

    tree = build_tree(modelsize, base, max_lvl, refine, balance=True)
    N = tree.ncell
    L = nlambda
    gas = np.zeros(N)
    stars = np.zeros(N)
    for l in range(1, tree.max_lvl + 1):
        m = (tree.lvl == l) & tree.leaf
        if not m.any():
            continue
        size = float(tree.csize[m][0])
        gas[m] = count(l, tree.ccoord[m], size, is_gas) / size ** 3
        stars[m] = count(l, tree.ccoord[m], size, ~is_gas) / size ** 3
    # normalise: optical depth ~1 through the densest cell at the first wavelength
    kmax = gas.max() if gas.max() > 0 else 1.0
    smallest = float(tree.csize.min())
    if L <= 4:
        kscale, sspec = np.array([1.0, 0.45, 0.2, 0.1][:L]), np.array([1.0, 1.4, 1.1, 0.7][:L])
        alb, gg = np.array([0.6, 0.5, 0.4, 0.3][:L]), np.array([0.5, 0.35, 0.2, 0.0][:L])
    else:  # many wavelengths (BASELINE config 5 names 28): TRUST dust columns from 0.15 um on, relative to 0.53 um
        tab = TRUSTI_TABLE[2:2 + L]
        kscale, sspec = tab[:, 1] / TRUSTI_TABLE[8, 1], tab[:, 4] / tab[:, 4].max()
        alb, gg = tab[:, 2].copy(), tab[:, 3].copy()
    dens = (gas / kmax * (1.0 / smallest))[:, None] * kscale[None, :]
    dens_stars = stars[:, None] * sspec[None, :]
    return Case(tree=tree, dens=dens, dens_stars=dens_stars, ksca_norm=alb,
                gsca=gg, p_src_pos=np.zeros((0, 3)), p_src_lum=np.zeros((0, L)),
                dir_out=_observer_dirs(ndir), kp_sca_max=1, rad_lim=0.1023, accuracy=0.3, conv_en_lim=0.01, bm_par=5,
                bm_par_sca=5, bm_par_max=200, dist_obs=10e3, name=name, **kw)
