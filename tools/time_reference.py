#!/usr/bin/env python3
"""Time the REFERENCE's own CPU path on bench-shaped cases (build container only: needs /root/reference, amdflang, MPICH).

north_star asks for the GPU figure "next to the reference CPU path timed on the host cores (core count stated)". The
reference's multi-threaded optimised builds hang in its own thread-termination protocol (SURVEY.md section 8c caveat 2),
so the cores are used the way the reference itself decomposes the work: `mpiexec -n P` single-threaded ranks of
oracle/_ref/ref_driver (the UNMODIFIED reference modules, -O2), sources i = im*np_mpi + id_mpi
(rt_routines.f90:1003-1005), rt_mpi's reductions restated in the driver. Intersections are counted by the C oracle on
the same case (its counters equal the reference's loop trips; the reference has no counter of its own).

usage: python tools/time_reference.py [--ranks 8] [--cases small|all] -> profiles/r02_reference_cpu_timing.json
Never run on the GPU box: the reference cannot travel."""
import argparse, json, os, re, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as ge
import oracle_py as op

ap = argparse.ArgumentParser()
ap.add_argument("--ranks", type=int, default=8)
ap.add_argument("--cases", default="small")
ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_reference_cpu_timing.json"))
args = ap.parse_args()
if not os.path.isdir("/root/reference"):
    raise SystemExit("needs /root/reference (build container)")
exe = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
if not os.path.exists(exe):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
pkg = ge._load_pkg()
g = pkg.grids
cases = [("TRUST-I tau_z=1, 1 wavelength (1 um), 1 801 cells", dict(tau_z=1.0, max_lvl=2, wavelengths=[11])),
         ("TRUST-I tau_z=1, 32 wavelengths, 1 801 cells", dict(tau_z=1.0, max_lvl=2, wavelengths=list(range(32)))),
         ("TRUST-I tau_z=0.01, 1 wavelength, 21 801 cells (BASELINE config 1)", dict(tau_z=0.01, max_lvl=4, wavelengths=[11]))]
if args.cases == "all":
    cases += [("TRUST-I tau_z=1, 1 wavelength, 21 801 cells", dict(tau_z=1.0, max_lvl=4, wavelengths=[11])),
              ("TRUST-I tau_z=1, 32 wavelengths, 21 801 cells", dict(tau_z=1.0, max_lvl=4, wavelengths=list(range(32))))]
ncpu = len(os.sched_getaffinity(0))
cpu = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][:1]
rows = []
for label, kw in cases:
    case = g.trusti_case(max_sca_iterations=1, **kw)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "case.bin")
        case.write(path)
        # intersections per stage: the C oracle (8 threads), stage by stage
        o = op.Oracle(path, nsrc=1, ndir=case.dir_out.shape[0])
        o.set_parallel(1, 0, ncpu)
        cnt = {}
        for rt in ("precalc_cell", "precalc_src"):
            o.rt_loop(rt)
        o.reset_counters(); o.rt_loop("dir_cell"); o.rt_loop("dir_src"); cnt["direct"] = o.counters()["intersections"]; o.rt_mpi("dir_src")
        cn = o.scatt_prepare()
        cnt["scatt1"] = 0
        if not cn:
            o.reset_counters(); t0 = time.time(); o.rt_loop("scatt"); t_port = time.time() - t0
            cnt["scatt1"] = o.counters()["intersections"]
        o.close()
        rec = {"case": label, "cells": int(case.tree.ncell), "wavelengths": int(case.lnum), "intersections_direct": int(cnt["direct"]),
               "intersections_scatt_iteration_1": int(cnt["scatt1"]), "port_threads": ncpu, "port_scatt1_s": t_port if cnt["scatt1"] else None}
        for P in sorted({1, args.ranks}):
            env = dict(os.environ, OMP_NUM_THREADS="1")
            cmd = (["/opt/conda/bin/mpiexec", "-n", str(P)] if P > 1 else []) + [exe, path, os.path.join(tmp, f"out{P}"), "11"]
            t0 = time.time()
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=7200)
            wall = time.time() - t0
            m = re.search(r"timings pre/dir/sca/iobs \[s\]:\s+(\S+)\s+(\S+)\s+(\S+)\s+(\S+)", r.stdout)
            assert r.returncode == 0 and m, r.stdout[-2000:] + r.stderr[-2000:]
            t_pre, t_dir, t_sca = float(m.group(1)), float(m.group(2)), float(m.group(3))
            rec[f"ranks_{P}"] = {"t_precalc_s": t_pre, "t_direct_s": t_dir, "t_scatt_iteration_1_s": t_sca, "wall_s": wall,
                                 "direct_intersections_per_s": cnt["direct"] / max(t_dir, 1e-9),
                                 "scatt_intersections_per_s": cnt["scatt1"] / max(t_sca, 1e-9) if cnt["scatt1"] else None}
            print(f"{label}: {P} rank(s): direct {t_dir:.2f} s ({cnt['direct'] / max(t_dir, 1e-9):.3e}/s), scattering iteration 1 {t_sca:.2f} s "
                  f"({cnt['scatt1'] / max(t_sca, 1e-9):.3e}/s)", flush=True)
        rows.append(rec)
out = {"what": "reference hot-path modules (unmodified, amdflang -O2) timed in the build container; single-threaded MPI ranks, "
               "sources i = im*np_mpi + id_mpi, stage-end reductions as in rt_mpi; intersections counted by the C oracle",
       "host": {"cpu": cpu[0] if cpu else "?", "cores_available": ncpu}, "ranks": args.ranks, "cases": rows}
json.dump(out, open(args.out, "w"), indent=1)
print("->", args.out)
