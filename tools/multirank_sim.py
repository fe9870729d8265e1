#!/usr/bin/env python3
"""Developer tool: an np-rank run of the scattering stage rehearsed on ONE GPU. np library contexts (np_mpi = np, id_mpi =
0..np-1) run every stage on their shares one after the other, rt_mpi runs on aliases of their device buffers with the
all-reduce replaced by an in-process sum (parallel.rt_mpi_local) - the product's multi-rank path, as in
tests/test_gpu_parity.py::test_two_ranks_in_one_process_match_golden. The time of a stage on np GPUs is the slowest
rank's kernel time (the all-reduce is not timed here); a one-rank context runs beside them for the comparison.
usage: python tools/multirank_sim.py [np] [max_lvl] [nlambda] [iterations]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pkg = ge._load_pkg()
from importlib import import_module
parallel = import_module(pkg.__name__ + ".parallel")
npr = int(sys.argv[1]) if len(sys.argv) > 1 else 8
max_lvl = int(sys.argv[2]) if len(sys.argv) > 2 else 5
nl = int(sys.argv[3]) if len(sys.argv) > 3 else 1
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 4
torch.cuda.init()
wl = list(range(32)) if nl == 32 else [11]
case = pkg.grids.trusti_case(tau_z=1.0, max_lvl=max_lvl, wavelengths=wl)
arena = int(os.environ.get("ARENA_GB", "24")) << 30
RTT = pkg.host.RTT
# the one-rank run first, alone on the GPU with the arena bench.py gives a rank (the queues' depth decides how many sectors
# a task can take); the np contexts share the GPU's memory afterwards: ARENA_GB each (default 24)
one = pkg.DartRayGPU(case, arena_bytes=int(os.environ.get("ARENA_ONE_GB", "200")) << 30)
one.rt_prepare("start")
one.rt_loop("precalc_cell"); one.rt_loop("precalc_src"); one.rt_mpi("precalc_src")
one.rt_prepare("prep_part2")
one.rt_loop("dir_cell"); one.rt_loop("dir_src"); one.rt_mpi("dir_src")
t_one = {"dir_src": one.counters()["kernel_ms"]}
for it in range(iters):
    if one.scatt_prepare():
        break
    one.rt_loop("scatt"); t_one[it] = one.counters()["kernel_ms"]; one.rt_mpi("scatt")
u_one = one.u_final
one.close()
ranks = [pkg.DartRayGPU(case, np_mpi=npr, id_mpi=r, arena_bytes=arena) for r in range(npr)]


def mpi(rt):
    torch.cuda.synchronize()
    parallel.rt_mpi_local(RTT[rt], [g.mpi_fields(rt) for g in ranks])
    torch.cuda.synchronize()


def stage(rt):
    ms = []
    for g in ranks:
        g.rt_loop(rt); ms.append(g.counters()["kernel_ms"])
    return ms


for g in ranks:
    g.rt_prepare("start")
for rt in ("precalc_cell", "precalc_src"):
    stage(rt)
mpi("precalc_src")
for g in ranks:
    g.rt_prepare("prep_part2")
for rt in ("dir_cell", "dir_src"):
    ms = stage(rt)
    if rt == "dir_src":
        print(f"dir_src (point source, sectors dealt round the ranks): one rank {t_one['dir_src']:.1f} ms, slowest of {npr} {max(ms):.1f} ms", flush=True)
mpi("dir_src")
tot_np = tot_1 = 0.0
for it in range(iters):
    cn = [g.scatt_prepare() for g in ranks]
    assert all(c == cn[0] for c in cn) and cn[0] == (it not in t_one)
    if cn[0]:
        break
    ms, t1 = stage("scatt"), t_one[it]
    c = ranks[0].counters()
    mpi("scatt")
    tot_np += max(ms); tot_1 += t1
    print(f"scattering iteration {it + 1}: one rank {t1:.1f} ms | {npr} ranks: {[round(x, 1) for x in ms]} max {max(ms):.1f} sum {sum(ms):.1f} "
          f"| speed-up {t1 / max(ms):.2f} | rank 0: {int(c['heavy_tasks'])} heavy tasks, {int(c['deep_sources'])} deep of {int(c['sources'])} sources", flush=True)
d = float(np.abs(ranks[0].u_final - u_one).max() / np.abs(u_one).max())
print(f"{iters} iterations: one rank {tot_1:.1f} ms, {npr} ranks {tot_np:.1f} ms, speed-up {tot_1 / tot_np:.2f}; u_final of rank 0 vs one rank: max rel diff {d:.2e}")
