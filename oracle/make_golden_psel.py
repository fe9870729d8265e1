#!/usr/bin/env python3
"""Golden fixtures of psel_av_arr (print_psel_av: the average path of the rays of every source per direct-light stage and
scattering iteration, rt_routines.f90:1456-1465) from the UNMODIFIED reference modules.

ORACLE INFRASTRUCTURE (build container only). Runs oracle/_ref/ref_driver with stage mask 31 + 64 on cases of
oracle/make_golden.py; the main outputs must equal the existing fixture bit for bit. Stored: psel_av[iteration][source]
(source = cell index, or tot_ncell + point-source index) as tests/golden/<case>_pselav.npz.

usage: python oracle/make_golden_psel.py [case ...]   (default: cube65 blobB)
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


def make(name, case, outdir):
    exe = os.path.join(HERE, "_ref", "ref_driver")
    with tempfile.TemporaryDirectory() as tmp:
        cpath = os.path.join(tmp, name + ".bin")
        case.write(cpath)
        subprocess.check_call([exe, cpath, os.path.join(tmp, name), "95"], env=dict(os.environ, OMP_NUM_THREADS="1"), stdout=subprocess.DEVNULL)
        ref = oracle_py.read_ref_output(os.path.join(tmp, name + ".out"))
        with open(os.path.join(tmp, name + ".psel"), "rb") as f:
            magic, N, nsrc, nit = (int(x) for x in np.fromfile(f, np.int32, 4))
            assert magic == 0x44524F33
            psel = np.fromfile(f, np.float64, (N + nsrc) * nit).reshape(N + nsrc, nit).T.copy()   # psel_av_arr(0, it, sid)
    old = np.load(os.path.join(outdir, name + ".npz"))
    for k in ("u_final", "lum_lost"):
        assert np.array_equal(np.asarray(ref[k]), old[k]), f"{name}: the main run differs from the existing fixture in {k}"
    assert nit == int(old["iterations"])      # index 0: direct light; 1 .. iterations-1: the scattering iterations that ran
    path = os.path.join(outdir, name + "_pselav.npz")
    np.savez_compressed(path, psel_av=psel)
    print(f"{name}: psel_av {psel.shape}, sources with rays per stage {[int((r > 0).sum()) for r in psel]} -> {path}")


if __name__ == "__main__":
    allc = make_golden.cases()
    for name in (sys.argv[1:] or ["cube65", "blobB"]):
        make(name, allc[name], os.path.join(ROOT, "tests", "golden"))
