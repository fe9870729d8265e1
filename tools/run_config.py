#!/usr/bin/env python3
"""Developer tool: run one of the BASELINE configurations (dart-ray_amd/grids.py) through every stage and print
timings, counters and the energy budget. usage: python tools/run_config.py shell|galaxy|nbody|trusti [max_lvl] [nlambda]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge._load_pkg()
g = pkg.grids
which = sys.argv[1]
max_lvl = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nl = int(sys.argv[3]) if len(sys.argv) > 3 else 3
kw = dict(max_sca_iterations=int(os.environ.get("MAX_IT", "2")))
case = {"shell": lambda: g.shell_case(max_lvl=max_lvl, nlambda=nl, **kw),
        "galaxy": lambda: g.galaxy_case(max_lvl=max_lvl, nlambda=nl, **kw),
        "nbody": lambda: g.nbody_case(max_lvl=max_lvl, nlambda=nl, npart=int(os.environ.get("NPART", "400000")), **kw),
        "trusti": lambda: g.trusti_case(max_lvl=max_lvl, wavelengths=list(np.linspace(0, 21, nl).astype(int)), **kw)}[which]()
t = case.tree
print(f"{case.name}: N={t.ncell} leaves={int(t.leaf.sum())} L={case.lnum} sources: {case.p_src_pos.shape[0]} point + "
      f"{int(((case.dens_stars.sum(axis=1) > 0) & t.leaf).sum())} cells", flush=True)
gpu = pkg.DartRayGPU(case, arena_bytes=int(os.environ.get("ARENA_GB", "64")) << 30)
tot = {"inter": 0, "ms": 0.0}
def on_stage(name):
    pass
t0 = time.time()
# a stage of the large configurations runs for minutes: a sign of life once a minute (a silent GPU job is taken to be hung)
import threading
def _heartbeat():
    while True:
        time.sleep(60)
        print(f"    ... {time.time() - t0:.0f} s", flush=True)
threading.Thread(target=_heartbeat, daemon=True).start()
def stage(name, fn):
    gpu.reset_counters(); ts = time.time(); fn(); c = gpu.counters()
    tot["inter"] += c["intersections"]; tot["ms"] += c["kernel_ms"]
    print(f"  {name:12s} {c['kernel_ms']:10.1f} ms  sources {c['sources']:8d} intersections {c['intersections']:14d} "
          f"{c['intersections'] / max(c['kernel_ms'], 1e-9) * 1e3:.3e}/s  launches {c['launches']} undone {c['retries']}", flush=True)
gpu.rt_prepare("start")
stage("precalc_cell", lambda: gpu.rt_loop("precalc_cell")); stage("precalc_src", lambda: gpu.rt_loop("precalc_src")); gpu.rt_mpi("precalc_src")
gpu.rt_prepare("prep_part2")
stage("dir_cell", lambda: gpu.rt_loop("dir_cell")); stage("dir_src", lambda: gpu.rt_loop("dir_src")); gpu.rt_mpi("dir_src")
it = 0
while not gpu.scatt_prepare():
    it += 1
    stage(f"scatt{it}", lambda: gpu.rt_loop("scatt")); gpu.rt_mpi("scatt")
u = gpu.u_final
vol = (t.csize ** 3)[:, None]
absorbed = (u * case.dens * (1.0 - case.ksca_norm)[None, :] * g.CS_PC * vol).sum(axis=0)
emitted = case.p_src_lum.sum(axis=0) + (case.dens_stars * vol).sum(axis=0)
print(f"  total {tot['ms'] / 1e3:.2f} s kernel, {tot['inter']:.3e} intersections, wall {time.time() - t0:.1f} s; iterations {gpu.iterations}")
print("  absorbed / emitted per wavelength:", np.round(absorbed / np.maximum(emitted, 1e-300), 4), " lum_lost / emitted:", np.round(gpu.lum_lost / np.maximum(emitted, 1e-300), 5))
