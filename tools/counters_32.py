#!/usr/bin/env python3
"""Developer tool: all counters of one scattering step of the bench workload."""
import sys; import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge._load_pkg()
case = pkg.grids.trusti_case(tau_z=1.0, max_lvl=5, wavelengths=list(range(32)))
g = pkg.DartRayGPU(case, arena_bytes=200 << 30)
g.rt_prepare("start"); g.rt_loop("precalc_cell"); g.rt_loop("precalc_src"); g.rt_mpi("precalc_src")
g.rt_prepare("prep_part2"); g.rt_loop("dir_cell"); g.rt_loop("dir_src"); g.rt_mpi("dir_src")
g.scatt_prepare(); g.rt_loop("scatt"); g.reset_counters(); g.rt_loop("scatt")
c = g.counters(); print({k: (float(v) if k == "kernel_ms" else int(v)) for k, v in c.items()}); print("kp", list(g.kp)); print("npix", list(g.npix))
import numpy as np
sc = [float(np.abs(g.scaspe(il)).sum()) for il in range(32)]
print("scaspe sums per wavelength (relative):", [round(x / max(sc), 4) for x in sc])
