#!/usr/bin/env python3
"""Developer tool: how evenly, and how fast, the two ways of sharing cell sources among ranks split one scattering
iteration of the bench workload. One GPU: the state after the direct-light stage is computed once, then every rank's
share of the scattering step is timed alone (np ranks x 2 partitions). Prints per-rank kernel times; the step time of
an np-GPU job is the slowest rank's. usage: python tools/partition_ab.py [np] [max_lvl] [nlambda]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge._load_pkg()
npr = int(sys.argv[1]) if len(sys.argv) > 1 else 8
max_lvl = int(sys.argv[2]) if len(sys.argv) > 2 else 5
nl = int(sys.argv[3]) if len(sys.argv) > 3 else 32
wl = list(range(32)) if nl == 32 else [11]
case = pkg.grids.trusti_case(tau_z=1.0, max_lvl=max_lvl, wavelengths=wl)
g = pkg.DartRayGPU(case, arena_bytes=int(os.environ.get("ARENA_GB", "128")) << 30)
g.rt_prepare("start")
g.rt_loop("precalc_cell"); g.rt_loop("precalc_src"); g.rt_mpi("precalc_src")
g.rt_prepare("prep_part2")
g.rt_loop("dir_cell"); g.rt_loop("dir_src"); g.rt_mpi("dir_src")
assert not g.scatt_prepare()
g.L_.drt_debug_set_rank.argtypes = [C.c_void_p, C.c_int, C.c_int]
g.rt_loop("scatt")  # warm-up: settles the sectors per task
full = g.counters()
print(f"N={case.tree.ncell} L={case.lnum} one rank: {full['kernel_ms']:.1f} ms, {full['intersections']:.3e} intersections", flush=True)
full_cost = g.get_array("src_cost")  # what rt_mpi would hand every rank: the cost of every source in the previous launch
for part, heavy in (("cyclic64", "0"), ("cyclic64", "16"), ("cost", "16"), ("cost", "32"), ("block", "16"), ("mod", "16")):
    os.environ["DRT_PARTITION"] = part[:6] if part.startswith("cyclic") else part
    os.environ["DRT_HEAVY_DIV"] = heavy
    if part.startswith("cyclic"):
        os.environ["DRT_PARTITION_CHUNK"] = part[6:]
    ms, inter = [], []
    for r in range(npr):
        g.L_.drt_debug_set_rank(g.h, npr, r)
        g.set_array("src_cost", full_cost)
        g.reset_counters(); g.rt_loop("scatt"); c = g.counters()
        ms.append(c["kernel_ms"]); inter.append(c["intersections"])
    print(f"{part:10s} heavy_div={heavy:3s} np={npr}: per-rank ms {[round(x, 1) for x in ms]} max {max(ms):.1f} sum {sum(ms):.1f} "
          f"| intersections max/mean {max(inter) / (sum(inter) / npr):.3f} | speed-up over one rank {full['kernel_ms'] / max(ms):.2f}", flush=True)
g.L_.drt_debug_set_rank(g.h, 1, 0)
os.environ.pop("DRT_PARTITION", None)
for heavy in ("0", "8", "16", "32"):
    os.environ["DRT_HEAVY_DIV"] = heavy
    g.set_array("src_cost", full_cost)
    g.reset_counters(); g.rt_loop("scatt"); c = g.counters()
    print(f"one rank, heavy_div={heavy}: {c['kernel_ms']:.1f} ms", flush=True)
