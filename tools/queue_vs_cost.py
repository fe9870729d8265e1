#!/usr/bin/env python3
"""Developer tool: how well do a source's measured wave time and its luminosity predict its longest ray queue?
For the first cell-source stage with work of a configuration: log-log correlation and slope of q48 (queue at 48 sectors per task,
second half of DRT_A_SRC_COST) against the measured cost and against the luminosity, and the spread of q48 / fit.
usage: python tools/queue_vs_cost.py shell|galaxy|nbody|trusti [max_lvl] [nlambda]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge._load_pkg()
g = pkg.grids
which = sys.argv[1]
max_lvl = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nl = int(sys.argv[3]) if len(sys.argv) > 3 else 3
kw = dict(max_sca_iterations=1)
case = {"shell": lambda: g.shell_case(max_lvl=max_lvl, nlambda=nl, **kw), "galaxy": lambda: g.galaxy_case(max_lvl=max_lvl, nlambda=nl, **kw),
        "nbody": lambda: g.nbody_case(max_lvl=max_lvl, nlambda=nl, npart=400000, **kw),
        "trusti": lambda: g.trusti_case(max_lvl=max_lvl, wavelengths=list(range(32)) if nl == 32 else list(np.linspace(0, 21, nl).astype(int)), **kw)}[which]()
gpu = pkg.DartRayGPU(case, arena_bytes=int(os.environ.get("ARENA_GB", "200")) << 30)
N = case.tree.ncell


def report(name):
    both = gpu.device_tensor("src_cost").cpu().numpy()
    cost, q = both[:N], both[N:]
    lum = gpu.get_array("lumcell").reshape(N, -1).sum(axis=1)
    ok = (cost > 0) & (q > 0) & (lum > 0)
    if ok.sum() < 10:
        print(f"{name}: {ok.sum()} sources with a measured queue"); return
    lq, lc, ll = np.log(q[ok]), np.log(cost[ok]), np.log(lum[ok])
    for tag, x in (("cost", lc), ("luminosity", ll)):
        A = np.vstack([x, np.ones_like(x)]).T
        (b, a), *_ = np.linalg.lstsq(A, lq, rcond=None)
        res = lq - (b * x + a)
        print(f"{name}: log q48 against log {tag}: {ok.sum()} sources, correlation {np.corrcoef(x, lq)[0, 1]:.3f}, slope {b:.2f}, q48 / fit: "
              f"90 % below x{np.exp(np.quantile(res, 0.9)):.2f}, 99 % below x{np.exp(np.quantile(res, 0.99)):.2f}, max x{np.exp(res.max()):.2f}")
    top = np.argsort(-q[ok])[:10]
    print(f"   longest queues (q48): {np.round(q[ok][top]).astype(int).tolist()}; their cost rank of {ok.sum()}: {[int((cost[ok] > cost[ok][t]).sum()) for t in top]}")


gpu.rt_prepare("start")
gpu.rt_loop("precalc_cell"); report("precalc_cell")
gpu.rt_loop("precalc_src"); gpu.rt_mpi("precalc_src")
gpu.rt_prepare("prep_part2")
gpu.rt_loop("dir_cell"); report("dir_cell")
gpu.rt_loop("dir_src"); gpu.rt_mpi("dir_src")
if not gpu.scatt_prepare():
    gpu.rt_loop("scatt"); report("scatt 1")
