#!/usr/bin/env python3
"""Developer tool: wall time of every step of a complete run (dartray_main's stage order) on one GPU, host work included.
usage: python tools/full_run_profile.py [max_lvl] [nlambda] [max_sca_iterations]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge._load_pkg()
max_lvl = int(sys.argv[1]) if len(sys.argv) > 1 else 5
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 32
max_it = int(sys.argv[3]) if len(sys.argv) > 3 else -1
wl = list(range(32)) if nl == 32 else [11]
case = pkg.grids.trusti_case(tau_z=1.0, max_lvl=max_lvl, wavelengths=wl, max_sca_iterations=max_it)
t00 = time.time()
g = pkg.DartRayGPU(case, arena_bytes=int(os.environ.get("ARENA_GB", "200")) << 30)   # ARENA_GB=0: the library's own choice
rows = []


def step(name, fn):
    t = time.time(); r = fn(); dt = time.time() - t
    c = g.counters() if name.startswith(("rt_loop", "iobs")) else None
    rows.append((name, dt, c["kernel_ms"] if c else None))
    return r


rows.append(("create + upload", time.time() - t00, None))
step("rt_prepare(start)", lambda: g.rt_prepare("start"))
step("rt_loop(precalc_cell)", lambda: g.rt_loop("precalc_cell")); step("rt_loop(precalc_src)", lambda: g.rt_loop("precalc_src")); step("rt_mpi", lambda: g.rt_mpi("precalc_src"))
step("rt_prepare(prep_part2)", lambda: g.rt_prepare("prep_part2"))
step("rt_loop(dir_cell)", lambda: g.rt_loop("dir_cell")); step("rt_loop(dir_src)", lambda: g.rt_loop("dir_src")); step("rt_mpi", lambda: g.rt_mpi("dir_src"))
step("iobs(dir_cell)", lambda: g.rt_loop_iobs("i_obs_dir_cell")); step("iobs(dir_src)", lambda: g.rt_loop_iobs("i_obs_dir_src"))
it = 0
while True:
    if step("rt_prepare(scatt)", g.scatt_prepare):
        break
    it += 1
    step(f"rt_loop(scatt {it})", lambda: g.rt_loop("scatt")); step("rt_mpi", lambda: g.rt_mpi("scatt"))
step("rt_prepare(i_obs)", lambda: g.rt_prepare("i_obs")); step("iobs(i_obs)", lambda: g.rt_loop_iobs("i_obs"))
step("sed", g.sed)
tot = sum(r[1] for r in rows)
print(f"N={case.tree.ncell} L={case.lnum}: {it} scattering iterations, {tot:.2f} s in all")
for name, dt, k in rows:
    print(f"  {name:26s} {dt * 1e3:10.1f} ms" + (f"   kernel {k:10.1f} ms" if k is not None else ""))
