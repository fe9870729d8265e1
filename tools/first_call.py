#!/usr/bin/env python3
"""Developer tool: the scattering step of the bench grid (TRUST-I tau = 1, 32 wavelengths, 85 801 cells) called three times with a small
queue arena (ARENA_GB, default 16): what the first, unmeasured call costs (undone attempts included in kernel_ms) and what the later ones settle at.
DRT_NO_QUEUE_PRIOR=1 / DRT_NO_SRC_QUEUES=1 switch the queue rules of the planner off for comparison."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge
pkg = ge._load_pkg()
case = pkg.grids.trusti_case(tau_z=1.0, max_lvl=5, wavelengths=list(range(32)), max_sca_iterations=1)
g = pkg.DartRayGPU(case, arena_bytes=int(os.environ.get("ARENA_GB", "16")) << 30)
g.rt_prepare("start"); g.rt_loop("precalc_cell"); g.rt_loop("precalc_src"); g.rt_mpi("precalc_src"); g.rt_prepare("prep_part2")
g.rt_loop("dir_cell"); g.rt_loop("dir_src"); g.rt_mpi("dir_src")
assert not g.scatt_prepare()
for k in range(3):
    g.reset_counters(); t = time.time(); g.rt_loop("scatt"); c = g.counters()
    print(f"call {k + 1}: wall {time.time() - t:.2f} s kernel {c['kernel_ms']:.0f} ms launches {c['launches']} undone {c['retries']} deep {c['deep_sources']}", flush=True)
