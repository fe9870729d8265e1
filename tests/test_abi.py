"""CPU-side checks of the boundary: the C-ABI library loads without a GPU and exports every entry
point include/dartray_rt.h declares; compute entry points fail loudly when no device is present
(there is no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "dartray_rt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(drt_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pkg):
    lib = ctypes.CDLL(pkg.host.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/dartray_rt.h but not exported"
    assert set(pkg.host.EXPORTS) <= set(names)


def test_no_cpu_fallback(pkg):
    """Without a gfx950 device the product refuses to run; it never routes through the oracle."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    case = pkg.grids.blob_case(base=(4, 2), max_lvl=2, nlambda=1, blob_r=1.5)
    with pytest.raises(pkg.DartRayError):
        pkg.DartRayGPU(case)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "dart-ray_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".f90", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle_py" not in text and "liboracle" not in text and "dartray_oracle.h" not in text, f


def test_grid_layout_invariants(pkg):
    """Layout contract of smooth_grid_routines.f90:42-58: children contiguous, x fastest; ccindd_to_cc
    (smooth_grid_routines.f90:820) finds every leaf again from its cindex code (SURVEY 8c iv)."""
    import numpy as np
    g = pkg.grids
    for case in (g.slab_case(max_lvl=3), g.blob_case(base=(3, 3), max_lvl=3, balance=False)):
        t = case.tree
        b = t.base
        bd = (g.basediv(b[0]), g.basediv(b[1]))
        leaves = np.flatnonzero(t.leaf)
        for i in leaves[:: max(1, len(leaves) // 300)]:
            l = 1
            for lev in range(1, t.lvl[i] + 1):
                bb = b[0] if lev == 1 else b[1]
                div = 1 if lev == 1 else bd[0] * bd[1] ** (lev - 2)
                mask = (bd[0] if lev == 1 else bd[1]) - 1
                k = ((int(t.cindex[i]) // div) & mask) - 1
                assert 0 <= k < bb ** 3
                l = l + k
                if lev < t.lvl[i]:
                    assert t.cchild[l] != -1
                    l = int(t.cchild[l])
            assert l == i
        # 2:1 balance across faces when requested
    t = g.slab_case(max_lvl=4).tree
    assert t.ncell == 21801 and int(t.leaf.sum()) == 19200


def test_config_builders_keep_the_reference_layout():
    """Generators of BASELINE configs 3-5 (dart-ray_amd/grids.py): children contiguous and x-fastest, cindex
    decodes to the cell, face neighbours within one level, every emitting cell a leaf."""
    import numpy as np
    from conftest import load_pkg
    g = load_pkg().grids
    for case in (g.shell_case(max_lvl=2, nlambda=2), g.galaxy_case(max_lvl=3), g.nbody_case(max_lvl=3)):
        t = case.tree
        N = t.ncell
        assert t.cchild[0] == 1 and t.lvl[0] == 0
        par = np.flatnonzero(t.cchild >= 0)
        for i in par[:: max(1, par.size // 200)]:
            b = t.base[0] if t.lvl[i] == 0 else t.base[1]
            kids = np.arange(t.cchild[i], t.cchild[i] + b ** 3)
            assert kids[-1] < N and (t.lvl[kids] == t.lvl[i] + 1).all()
            # child k = (iz*b + iy)*b + ix sits at parent corner + (ix+.5, iy+.5, iz+.5) * child size
            k = np.arange(b ** 3)
            off = np.stack([k % b, (k // b) % b, k // (b * b)], axis=1)
            want = t.ccoord[i] - 0.5 * t.csize[i] + (off + 0.5) * t.csize[kids][:, None]
            assert np.allclose(t.ccoord[kids], want, rtol=0, atol=1e-9 * t.modelsize)
        assert ((case.dens_stars.sum(axis=1) > 0) <= t.leaf).all()
        assert np.isfinite(case.dens).all() and (case.dens >= 0).all()


def test_arena_layout_regions_do_not_overlap(pkg):
    """carve_arena's arithmetic (drt_arena_carve, no GPU needed): for every split of an arena among waves the regions
    of one wave's slice - four SoA ray queues, sibling hash, status words, HG normalisations [LP][64], intensities
    [LP][64] - follow each other without overlap, end inside the wave's stride, and all strides fit the arena. A
    queue overflow must therefore surface as DRT_E_CAPACITY, never as a write into a neighbour's slice."""
    rng = np.random.default_rng(7)
    cases = [(16 << 30, 2048, 32), (128 << 30, 4096, 32), (1 << 30, 1024, 1), (64 << 20, 256, 4), (9 << 20, 256, 2)]
    cases += [(int(rng.integers(1 << 20, 1 << 34)), int(rng.integers(1, 1025)) * 4, int(2 ** rng.integers(0, 7))) for _ in range(200)]
    seen_ok = 0
    for arena, waves, lp in cases:
        r = pkg.host.arena_carve(arena, waves, lp)
        if r is None:
            continue
        seen_ok += 1
        cap, hcap, stride, off = r
        assert cap >= 64 and cap % 64 == 0 and hcap >= 2 * cap and (hcap & (hcap - 1)) == 0
        assert stride * waves <= arena and stride % 256 == 0
        lb = 12 * cap + 16 * cap + 8 * cap * lp          # one SoA queue: ipix, meta, cc | prev, dpl | lum[LP]
        sizes = [lb, lb, lb, lb, 4 * hcap, 4 * (4 * hcap + cap), 4 * cap, 512 * lp, 512 * lp]
        for k in range(9):
            assert off[k] + sizes[k] <= off[k + 1], (arena, waves, lp, k)
        assert off[7] % 16 == 0 and off[9] <= stride
    assert seen_ok > 50
    assert pkg.host.arena_carve(1 << 20, 4096, 32) is None  # too small: an error, not a layout


def test_deal_by_cost_is_a_balanced_partition_in_chunks(pkg):
    """The deal of cell sources among ranks (build_src_list, drt_deal_by_cost; the C library's host code, no GPU): every
    source has exactly one owner, runs of ordinary sources stay together in chunks of at most `chunk`, a source above
    1/64 of a rank's share is dealt alone, the heaviest rank carries at most the mean plus one chunk, the result is a pure
    function of the costs (two calls agree), and equal costs give every rank the same number of chunks to within one."""
    host = pkg.host
    rng = np.random.default_rng(5)
    n, npr, chunk = 20000, 8, 64
    cost = rng.lognormal(0.0, 1.0, n)
    cost[:200] *= 300.0            # first-level cells: a few sources carry a large part of the work
    own = host.deal_by_cost(cost, npr, chunk)
    assert own.shape == (n,) and own.min() >= 0 and own.max() == npr - 1
    assert np.array_equal(own, host.deal_by_cost(cost, npr, chunk))
    load = np.array([cost[own == r].sum() for r in range(npr)])
    big = cost.sum() / npr / 64.0
    # chunks: maximal runs of one owner never exceed... (two neighbouring chunks may go to the same rank, so look at the
    # boundaries instead) every change of owner happens at a chunk boundary of the construction
    k, bounds = 0, {0}
    while k < n:
        hi = k
        while hi < n and hi - k < chunk:
            if cost[hi] > big:
                if hi == k:
                    hi += 1
                break
            hi += 1
        bounds.add(hi); k = hi
    changes = set(np.flatnonzero(np.diff(own) != 0) + 1)
    assert changes <= bounds
    largest_chunk = max(cost[a:b].sum() for a, b in zip(sorted(bounds)[:-1], sorted(bounds)[1:]))
    assert load.max() <= load.mean() + largest_chunk
    assert load.max() / load.mean() < 1.02
    eq = host.deal_by_cost(np.ones(64 * 33), npr, 64)
    counts = np.bincount(eq, minlength=npr) // 64
    assert counts.max() - counts.min() <= 1 and counts.sum() == 33
    assert np.array_equal(host.deal_by_cost(cost, 1, chunk), np.zeros(n, dtype=np.int32))


def test_one_hip_runtime_in_the_process(pkg):
    """The torch wheel bundles its own libamdhip64.so / libhsa-runtime64.so; libdartray_rt.so is linked against /opt/rocm's.
    Both carry the SONAME libamdhip64.so.7, but libc10_hip.so asks for the FILE name libamdhip64.so: when the library is
    mapped first, torch's copy is mapped beside it and the second runtime finds no GPU ("No HIP GPUs are available" from
    torch's lazy initialisation - round 2's open item). host.lib() therefore imports torch before it loads the library;
    whichever of the two a caller touches first, the process must hold ONE HIP and ONE HSA runtime. Run in a child
    process: this one has usually imported torch already."""
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import __graft_entry__ as ge\n"
        "pkg = ge._load_pkg()\n"
        "pkg.host.lib()\n"          # the product's loader FIRST, torch only afterwards (device_tensor() / attach_process_group())
        "import torch\n"
        "hip, hsa = set(), set()\n"
        "for line in open('/proc/self/maps'):\n"
        "    p = line.split()[-1]\n"
        "    if 'libamdhip64' in p: hip.add(p)\n"
        "    if 'libhsa-runtime64' in p: hsa.add(p)\n"
        "print(len(hip), len(hsa), sorted(hip), sorted(hsa))\n"
    ) % (ROOT,)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    n_hip, n_hsa = out.stdout.split()[:2]
    assert (n_hip, n_hsa) == ("1", "1"), out.stdout


def test_plan_tasks_rules(pkg):
    """drt_plan_tasks (plan_tasks in dartray_rt.hip, pure): how a cell-source launch is cut into tasks. Every source gets exactly
    one class; a light source's task (G sectors) and a heavy source's tasks (Gi sectors, a power-of-two fraction of G) stay under
    the limit unless Gi is already 1; the explicit heavy tasks cover each heavy source's 48 sectors once; deep = one sector alone
    exceeds the limit; a launch planned from the prior is cut four times finer; the limit is never coarser than the queues of
    the stage's last launch allow; no source is deep in a short, measured launch; and where the sources' own queue lengths are known
    they bound the sectors per task."""
    rng = np.random.default_rng(5)
    n, waves, G = 5000, 3072, 16
    cost = np.exp(rng.normal(0.0, 1.6, n)) * 1e7            # wave cycles, spread over three decades
    cost[:5] *= 3e3                                          # a few sources far beyond the rest
    cls, ntask, limit = pkg.host.plan_tasks(cost, waves, G, heavy_div=8.0, deep_min=0.0, measured=True)
    per_wave = cost.sum() / waves
    assert limit == pytest.approx(per_wave / 8.0, rel=1e-12)
    assert set(np.unique(cls)) <= {-1, 0, 1, 2, 4, 8, 16} and (cls != -2).all()
    light, deep, heavy = cls == 0, cls == -1, cls > 0
    assert light.any() and deep.any() and heavy.any()
    assert (cost[light] * G / 48.0 <= limit).all()
    assert (cost[deep] / 48.0 > limit).all() and (cost[~deep] / 48.0 <= limit).all()
    gi = cls[heavy]
    assert ((cost[heavy] * gi / 48.0 <= limit) | (gi == 1)).all()          # cut until it fits, or down to one sector
    assert (cost[heavy] * np.minimum(2 * gi, G) / 48.0 > limit)[gi < G].all()  # and no finer than needed
    assert ntask == int((48 // gi).sum())
    # the prior: four times finer
    _, _, lim_prior = pkg.host.plan_tasks(cost, waves, G, heavy_div=8.0, deep_min=0.0, measured=False)
    assert lim_prior == pytest.approx(limit / 4.0, rel=1e-12)
    # the queue rule: last launch cut at 1/32 of a wave's share with 8 sectors filled 52 % of the queues -> at most 1/32 * 0.8 / 0.52 at 8
    # sectors, half of that at 16
    _, _, lim_q8 = pkg.host.plan_tasks(cost, waves, 8, heavy_div=8.0, deep_min=0.0, measured=True, f_last=1 / 32, q_fill=0.52, G_last=8)
    assert lim_q8 == pytest.approx(per_wave / 32 * 0.8 / 0.52, rel=1e-12)
    _, _, lim_q16 = pkg.host.plan_tasks(cost, waves, 16, heavy_div=8.0, deep_min=0.0, measured=True, f_last=1 / 32, q_fill=0.52, G_last=8)
    assert lim_q16 == pytest.approx(lim_q8 / 2, rel=1e-12)
    _, _, lim_room = pkg.host.plan_tasks(cost, waves, 8, heavy_div=8.0, deep_min=0.0, measured=True, f_last=1 / 32, q_fill=0.05, G_last=8)
    assert lim_room == pytest.approx(limit, rel=1e-12)                       # plenty of room: the rule does not bind
    # a short measured launch has no deep sources (their sectors are cut one per task instead); an unmeasured one may
    cls_s, _, _ = pkg.host.plan_tasks(cost, waves, G, heavy_div=8.0, deep_min=10 * per_wave, measured=True)
    assert not (cls_s == -1).any() and (cls_s[deep] == 1).all()
    cls_u, _, _ = pkg.host.plan_tasks(cost, waves, G, heavy_div=8.0, deep_min=10 * per_wave, measured=False)
    assert (cls_u == -1).any()
    # batches (deposit log): every launch is 1/nbatch of the work
    _, _, lim_b = pkg.host.plan_tasks(cost, waves, G, nbatch=6, heavy_div=8.0, deep_min=0.0, measured=True)
    assert lim_b == pytest.approx(limit / 6.0, rel=1e-12)
    # the sources' own queues (q48: the queue a 48-sector task of the source would hold; cap rays fit a queue): a source takes the
    # sectors per task that keep its queue under 0.8 cap - whatever its cost - and is deep where one sector does not; the rule on
    # the last launch's one queue is then not applied; sources whose queue is not known are cut by cost alone
    cap = 4000.0
    q48 = np.exp(rng.normal(np.log(6000.0), 1.0, n)); q48[::7] = 0.0; q48[5:10] = 1e6
    cls_q, ntask_q, lim_own = pkg.host.plan_tasks(cost, waves, G, heavy_div=8.0, deep_min=10 * per_wave, measured=True, f_last=1 / 32, q_fill=0.52, G_last=8,
                                                  q48=q48, cap=cap)
    assert lim_own == pytest.approx(limit, rel=1e-12)
    known = q48 > 0
    assert (cls_q[5:10] == -1).all()                                         # one sector would overflow: sub-tree tasks, however short the launch
    deep_q = cls_q == -1
    assert (q48[deep_q] / 48.0 > 0.8 * cap).all()
    g_eff = np.where(cls_q == 0, G, cls_q)
    ok = ~deep_q
    assert (q48[ok] * g_eff[ok] / 48.0 <= 0.8 * cap).all()
    assert ((cost[ok] * g_eff[ok] / 48.0 <= limit) | (g_eff[ok] == 1)).all()
    finer = ok & (g_eff < G)                                                 # and no finer than one of the two rules asks
    g2 = np.minimum(2 * g_eff, G)
    assert ((q48 * g2 / 48.0 > 0.8 * cap) | (cost * g2 / 48.0 > limit))[finer].all()
    assert np.array_equal(cls_q[~known], cls_s[~known])
    assert ntask_q == int((48 // cls_q[cls_q > 0]).sum())
    # heavy_div = 0 switches the cutting off
    cls0, n0, _ = pkg.host.plan_tasks(cost, waves, G, heavy_div=0.0)
    assert (cls0 == 0).all() and n0 == 0


def test_fit_queue_prior(pkg):
    """drt_fit_queue_prior (fit_queue_prior in dartray_rt.hip, pure): the line through (log luminosity, log queue length) of the sources
    an undone launch measured; pairs without a measurement are ignored; the slope stays within [0, 1.5]; nothing is fitted from fewer
    than 64 pairs or from luminosities that are all alike."""
    rng = np.random.default_rng(11)
    n = 4000
    lum = np.exp(rng.normal(70.0, 2.0, n))
    q = 3.0 * lum ** 0.7 * np.exp(rng.normal(0.0, 0.3, n))
    q[::3] = 0.0                                              # not measured
    a, b = pkg.host.fit_queue_prior(lum, q)
    assert b == pytest.approx(0.7, abs=0.02) and a == pytest.approx(np.log(3.0), abs=1.5)
    ok = q > 0
    pred = np.exp(a + b * np.log(lum[ok]))
    assert np.median(q[ok] / pred) == pytest.approx(1.0, abs=0.05)
    assert (q[ok] <= 2.0 * pred).mean() > 0.97               # what the launch plans with: twice the line
    _, b_hi = pkg.host.fit_queue_prior(lum, lum ** 3.0)
    assert b_hi == 1.5
    _, b_lo = pkg.host.fit_queue_prior(lum, lum ** -1.0)
    assert b_lo == 0.0
    assert pkg.host.fit_queue_prior(lum[:63], q[:63] + 1.0) is None
    assert pkg.host.fit_queue_prior(np.full(200, 5.0), np.full(200, 7.0)) is None
