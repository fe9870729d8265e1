import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
pkg = ge._load_pkg()
case = pkg.grids.read_case(os.path.join(ROOT, "tests", "golden", "cube65.case"), "cube65") if os.path.exists(os.path.join(ROOT, "tests", "golden", "cube65.case")) else pkg.grids.blob_case(base=(4, 2), max_lvl=2, nlambda=2)
g = pkg.DartRayGPU(case)
g.rt_prepare("start")
g.rt_loop("precalc_cell"); g.rt_loop("precalc_src"); g.rt_mpi("precalc_src")
g.rt_prepare("prep_part2")
print("dir_cell", flush=True); g.rt_loop("dir_cell")
print("dir_src", flush=True); g.rt_loop("dir_src")
print("ok", flush=True)
