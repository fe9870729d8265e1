#!/usr/bin/env python3
"""Developer tool: time every stage of a TRUST-I run on the GPU.
usage: python tools/probe.py <max_lvl> <nlambda> [tau_z] [max_iter]"""
import os, sys, time
import ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge._load_pkg()
max_lvl, nl = int(sys.argv[1]), int(sys.argv[2])
tau = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
max_it = int(sys.argv[4]) if len(sys.argv) > 4 else 3
wl = list(range(32)) if nl == 32 else ([11] if nl == 1 else list(np.linspace(0, 21, nl).astype(int)))
t0 = time.time()
case = pkg.grids.trusti_case(tau_z=tau, max_lvl=max_lvl, wavelengths=wl, max_sca_iterations=max_it, top_layers=int(os.environ.get("TOP_LAYERS", "1")))
print(f"case N={case.tree.ncell} leaves={case.tree.leaf.sum()} L={case.lnum} build {time.time()-t0:.2f}s", flush=True)
g = pkg.DartRayGPU(case, arena_bytes=int(os.environ.get("ARENA_GB", "16")) << 30, waves=int(os.environ.get("WAVES", "0")))
def stage(name, fn):
    g.reset_counters(); t = time.time(); fn(); dt = time.time() - t
    c = g.counters()
    bytes_alg = 40 * c["intersections"] + 8 * c["lambda_deposits"] + 16 * c["u_adds"] + 8 * c["u_reads"] + 16 * c["scaspe_slots"]
    ms = c["kernel_ms"]
    import ctypes as C
    if os.environ.get("DRT_PROFILE"):
        raw = (C.c_ulonglong * 32)()
        g.L_.drt_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
        g.L_.drt_debug_counters(g.h, raw, 32)
        tot = max(raw[15], 1)
        names = ["setup", "step", "flush", "park", "next", "commit", "extract"]
        print("    phases % of wave time:", {n: round(100 * raw[8 + i] / tot, 1) for i, n in enumerate(names)},
              "steps", raw[16], "cycles/step", int(tot / max(raw[16], 1)), "sum of wave times / (100 MHz ticks -> ms)", round(tot / 1e5, 1), "busy lanes/step", round(raw[17] / max(raw[16], 1), 1),
              "rays/group", round(raw[19] / max(raw[18], 1), 2))
        print("    wavelength loop % of wave time:", {n: round(100 * raw[22 + i] / tot, 1) for i, n in enumerate(["deposit", "group head", "members", "epilogue"])},
              "lambda iterations/step", round(raw[26] / max(raw[16], 1), 1), "groups/step", round(raw[18] / max(raw[16], 1), 1),
              "member iterations/step by table (kp0, kp1 or memory, isotropic)", [round(raw[27 + i] / max(raw[16], 1), 1) for i in range(3)])
    print(f"{name:14s} wall {dt*1e3:9.1f} ms kernel {ms:9.2f} ms src {c['sources']:7d} inter {c['intersections']:12d} rays {c['rays']:11d} "
          f"sca {c['scaspe_adds']:12d} | {c['intersections']/max(ms,1e-9)*1e3:.3e} inter/s  {bytes_alg/max(ms,1e-9)*1e3/1e9:8.1f} GB/s alg"
          f" | atomics {c['atomic_instructions']:.3e} ({c['scaspe_adds']/max(c['atomic_instructions'],1):.2f} (ray,lambda)/instr, "
          f"{c['atomic_lanes']/max(c['atomic_instructions'],1):.1f} lanes/instr, {8e-6*c['atomic_lanes']/max(ms,1e-9):.1f} GB/s payload)", flush=True)
g.rt_prepare("start")
print("npix", g.npix)
stage("precalc_cell", lambda: g.rt_loop("precalc_cell")); stage("precalc_src", lambda: g.rt_loop("precalc_src")); g.rt_mpi("precalc_src")
g.rt_prepare("prep_part2")
stage("dir_cell", lambda: g.rt_loop("dir_cell")); stage("dir_src", lambda: g.rt_loop("dir_src")); g.rt_mpi("dir_src")
stage("iobs_dir", lambda: (g.rt_loop_iobs("i_obs_dir_cell"), g.rt_loop_iobs("i_obs_dir_src")))
it = 0
if os.environ.get("DRT_SKIP_SCATT_ATOMICS"):
    g.L_.drt_debug_flags.argtypes = [C.c_void_p, C.c_int]
    g.L_.drt_debug_flags(g.h, int(os.environ["DRT_SKIP_SCATT_ATOMICS"]))  # 1: no atomics, 3: no deposit at all (builds with -DDRT_DEBUG_FLAGS)
while not g.scatt_prepare():
    it += 1
    stage(f"scatt{it}", lambda: g.rt_loop("scatt")); g.rt_mpi("scatt")
print("iterations", g.iterations, "lum_lost/L", g.lum_lost / case.p_src_lum[0])
