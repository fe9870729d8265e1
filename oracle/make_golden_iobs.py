#!/usr/bin/env python3
"""Golden fixtures of the i_obs algorithm (dartray_i_obs, dartray_hub.f90:125-168) from the UNMODIFIED reference modules.

ORACLE INFRASTRUCTURE (build container only: needs /root/reference, amdflang, MPICH).

Runs oracle/_ref/ref_driver with stage mask 63 on cases of oracle/make_golden.py: after the whole main run the driver
switches rt_algorithm_ID to rta_i_obs, replaces the observer directions and runs the three observer passes on the
scaspe_tot it holds (ref_driver.f90, "dartray_i_obs"). The main run's outputs must equal the existing fixture bit for bit
(same program, same inputs, one thread); stored are the new directions and the i_obs / SED arrays of the passes, as
tests/golden/<case>_iobsalg.npz.

usage: python oracle/make_golden_iobs.py [case ...]   (default: cube65 blobB)
"""
from __future__ import annotations

import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import make_golden  # noqa: E402
import oracle_py  # noqa: E402


def read_iobs(path):
    with open(path, "rb") as f:
        magic, N, L, ndir, nsrc = (int(x) for x in np.fromfile(f, np.int32, 5))
        assert magic == 0x44524F32
        out = {"dir_alg": np.fromfile(f, np.float64, 2 * ndir).reshape(2, ndir).T.copy()}   # dir_i_out(0:ndir-1, 2)
        for k in ("i_obs_alg_dir", "sed_alg_dir", "i_obs_alg", "sed_alg"):
            n = ndir * (N + nsrc) * L if k.startswith("i_obs") else ndir * L
            a = np.fromfile(f, np.float64, n)
            out[k] = a.reshape(ndir, N + nsrc, L) if k.startswith("i_obs") else a.reshape(ndir, L)
    return out


def make(name, case, outdir):
    exe = os.path.join(HERE, "_ref", "ref_driver")
    with tempfile.TemporaryDirectory() as tmp:
        cpath = os.path.join(tmp, name + ".bin")
        case.write(cpath)
        env = dict(os.environ, OMP_NUM_THREADS="1")
        subprocess.check_call([exe, cpath, os.path.join(tmp, name), "63"], env=env, stdout=subprocess.DEVNULL)
        ref = oracle_py.read_ref_output(os.path.join(tmp, name + ".out"))
        alg = read_iobs(os.path.join(tmp, name + ".iobs"))
    old = np.load(os.path.join(outdir, name + ".npz"))
    for k in ("u_final", "lum_lost", "i_obs", "sed"):
        assert np.array_equal(np.asarray(ref[k]), old[k]), f"{name}: the main run differs from the existing fixture in {k}"
    path = os.path.join(outdir, name + "_iobsalg.npz")
    np.savez_compressed(path, **alg)
    print(f"{name}: {alg['dir_alg'].shape[0]} new directions, i_obs_alg max {alg['i_obs_alg'].max():.3e} "
          f"(direct part {alg['i_obs_alg_dir'].max():.3e}) -> {path} ({os.path.getsize(path) / 1024:.0f} kB)")


if __name__ == "__main__":
    allc = make_golden.cases()
    for name in (sys.argv[1:] or ["cube65", "blobB"]):
        make(name, allc[name], os.path.join(ROOT, "tests", "golden"))
