#!/usr/bin/env python3
"""Developer tool: how the wave time of one scattering step is spread over its sources (DRT_A_SRC_COST), and how long
the longest tasks are next to the kernel. usage: python tools/cost_profile.py [max_lvl] [nlambda] [tau] [top_layers]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge._load_pkg()
max_lvl = int(sys.argv[1]) if len(sys.argv) > 1 else 5
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 1
tau = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
top = int(sys.argv[4]) if len(sys.argv) > 4 else 1
wl = list(range(32)) if nl == 32 else [11]
case = pkg.grids.trusti_case(tau_z=tau, max_lvl=max_lvl, wavelengths=wl, top_layers=top)
g = pkg.DartRayGPU(case, arena_bytes=int(os.environ.get("ARENA_GB", "200")) << 30)
g.rt_prepare("start")
g.rt_loop("precalc_cell"); g.rt_loop("precalc_src"); g.rt_mpi("precalc_src")
g.rt_prepare("prep_part2")
g.rt_loop("dir_cell"); g.rt_loop("dir_src"); g.rt_mpi("dir_src")
assert not g.scatt_prepare()
for rep in range(3):
    g.reset_counters(); g.rt_loop("scatt"); c = g.counters()
    cost = g.get_array("src_cost")
    nz = cost[cost > 0]
    tot = cost.sum()
    srt = np.sort(nz)[::-1]
    print(f"pass {rep}: N={case.tree.ncell} L={case.lnum} kernel {c['kernel_ms']:.1f} ms, {c['intersections']:.3e} intersections, {int(c['sources'])} sources, "
          f"sum cost {tot:.3e} ticks; per ms of kernel {tot / c['kernel_ms']:.3e}", flush=True)
    cum = np.cumsum(srt) / tot
    for k in (1, 10, 100, 1000, 10000):
        if k <= srt.size:
            print(f"   top {k:6d} sources: {cum[k - 1] * 100:6.2f} % of the wave time; the {k}-th costs {srt[k - 1]:.3e} ticks = {srt[k - 1] / tot * 100:.4f} %")
    lv = case.tree.lvl
    idx = np.argsort(cost)[::-1][:8]
    print("   heaviest cells (index, level, cost):", [(int(i), int(lv[i]), f"{cost[i]:.2e}") for i in idx])
