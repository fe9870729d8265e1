#!/usr/bin/env python3
"""Developer tool: per-source wave time and luminosity of the first scattering iterations of the bench grid, dumped for offline analysis."""
import os, sys
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge._load_pkg()
for nl in (1, 32):
    wl = list(range(32)) if nl == 32 else [11]
    case = pkg.grids.trusti_case(tau_z=1.0, max_lvl=5, wavelengths=wl)
    g = pkg.DartRayGPU(case, arena_bytes=200 << 30)
    g.rt_prepare("start")
    g.rt_loop("precalc_cell"); g.rt_loop("precalc_src"); g.rt_mpi("precalc_src")
    g.rt_prepare("prep_part2")
    g.rt_loop("dir_cell"); g.rt_loop("dir_src"); g.rt_mpi("dir_src")
    out = {}
    it = 0
    while not g.scatt_prepare() and it < 3:
        lum = g.get_array("lumcell")
        g.rt_loop("scatt")
        out[f"cost{it}"] = g.get_array("src_cost"); out[f"lum{it}"] = lum
        g.rt_mpi("scatt")
        it += 1
    out["lvl"] = case.tree.lvl; out["ccoord"] = case.tree.ccoord; out["csize"] = case.tree.csize
    np.savez_compressed(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", f"costdump{nl}.npz"), **out)
    g.close()
