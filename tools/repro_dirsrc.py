"""Developer tool: TRUST-I (21 801 cells, 32 wavelengths) through the direct-light stage only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge._load_pkg()
case = pkg.grids.trusti_case(tau_z=1.0, max_lvl=4, wavelengths=list(range(32)))
g = pkg.DartRayGPU(case, arena_bytes=64 << 30)
g.rt_prepare("start")
g.rt_loop("precalc_cell"); g.rt_loop("precalc_src"); g.rt_mpi("precalc_src")
g.rt_prepare("prep_part2")
print("dir_src", flush=True); g.rt_loop("dir_src")
print("ok", g.counters()["intersections"], flush=True)
