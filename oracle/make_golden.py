#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the UNMODIFIED reference modules.

ORACLE INFRASTRUCTURE (build container only: needs /root/reference, amdflang, MPICH).

For every case below: build the synthetic problem with dart-ray_amd/grids.py, write it as a case
file, run oracle/_ref/ref_driver (the reference's smooth_grid_routines / healpix_routines /
ray_list / rt_routines compiled in place by `make -C oracle ref`, single-threaded so the result is
deterministic, SURVEY.md section 8c caveat 1), and store inputs + per-stage outputs as one compressed .npz.
Large scaspe arrays are stored as per-cell sums, per-pixel sums and a deterministic sample of
full cells to keep each fixture at a few hundred kB.

usage: python oracle/make_golden.py [case ...]
"""
from __future__ import annotations

import importlib.util
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import oracle_py  # noqa: E402

_spec = importlib.util.spec_from_file_location("dartray_grids", os.path.join(ROOT, "dart-ray_amd", "grids.py"))
grids = importlib.util.module_from_spec(_spec)
sys.modules["dartray_grids"] = grids
_spec.loader.exec_module(grids)

FULL_SCASPE_LIMIT = 20000  # values
NSAMPLE = 24


def cases():
    return {
        # tiny cube, two wavelengths (HG kp=1 and isotropic), point source + diffuse emission
        "cube65": grids.blob_case(base=(4, 2), max_lvl=2, nlambda=2, name="cube65", blob_r=1.5),
        # TRUST-I-like slab of BASELINE.md section 2 ("slab-1801")
        "slab1801": grids.slab_case(max_lvl=2, name="slab1801"),
        # refined blob, balanced tree, 2 wavelengths; go_low / merge path is exercised
        "blobA": grids.blob_case(base=(4, 2), max_lvl=3, nlambda=2, name="blobA"),
        # 27-ary tree (base 3,3), unbalanced, walls on, two point sources, 3 wavelengths
        "blobB": grids.blob_case(base=(3, 3), max_lvl=3, nlambda=3, balance=False, walls=True, name="blobB",
                                 p_src=((0.31, -0.23, 0.13), (-2.03, 1.52, 2.46))),
        # deeper tree, strongly forward scattering, kp_sca_max=2 (192+2 scaspe slots)
        "blobC": grids.blob_case(base=(4, 2), max_lvl=4, nlambda=1, balance=False, name="blobC", tau=3.0,
                                 g=(0.7,), kp_sca_max=2),
        # --- many wavelengths per ray: the instantiations bench.py times (rt_stage_wide<32,4>) ---
        # TRUST-I tau_z=1 with the full 32-wavelength list on the 1 801-cell tree: slot tables [kp 1 x13, kp 0 x10,
        # isotropic x9], 7 observer directions (55 / 19 / 8 scaspe slots)
        "trusti32": grids.trusti_case(tau_z=1.0, max_lvl=2, wavelengths=list(range(32)), name="trusti32",
                                      max_sca_iterations=2),
        # 7 wavelengths (not a power of two), kp 1 / -1 / 0 interleaved so that every 4- and 8-wide wavelength block
        # mixes slot tables; walls on, two point sources, diffuse emission
        "blobD": grids.blob_case(base=(4, 2), max_lvl=3, nlambda=7, walls=True, name="blobD",
                                 albedo=(0.6, 0.5, 0.4, 0.55), g=(0.55, 0.0, 0.1, 0.3, 0.05),
                                 p_src=((0.31, -0.23, 0.13), (-2.03, 1.52, 2.46)), max_sca_iterations=3),
        # 5 wavelengths with kp_sca_max=2: 194-slot tables (more slots than lanes) next to 50-, 14- and 3-slot ones
        "blobE": grids.blob_case(base=(3, 3), max_lvl=2, nlambda=5, name="blobE", tau=2.0, kp_sca_max=2,
                                 albedo=(0.6, 0.5, 0.45), g=(0.7, 0.0, 0.1, 0.3, 0.62), max_sca_iterations=3),
    }


def pack_scaspe(prefix, arrs, out):
    for il, a in enumerate(arrs):
        if a.size <= FULL_SCASPE_LIMIT:
            out[f"{prefix}{il}_full"] = a
        else:
            out[f"{prefix}{il}_cellsum"] = a.sum(axis=1)
            out[f"{prefix}{il}_pixsum"] = a.sum(axis=0)
            nz = np.flatnonzero(np.abs(a).sum(axis=1) > 0)
            if nz.size:
                idx = nz[np.linspace(0, nz.size - 1, min(NSAMPLE, nz.size)).astype(int)]
            else:
                idx = np.zeros(0, dtype=np.int64)
            out[f"{prefix}{il}_sample_idx"] = idx.astype(np.int32)
            out[f"{prefix}{il}_sample"] = a[idx]


def make(name, case, outdir):
    exe = os.path.join(HERE, "_ref", "ref_driver")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", HERE, "ref"])
    with tempfile.TemporaryDirectory() as tmp:
        cpath = os.path.join(tmp, name + ".bin")
        case.write(cpath)
        env = dict(os.environ, OMP_NUM_THREADS="1")
        subprocess.check_call([exe, cpath, os.path.join(tmp, name)], env=env, stdout=subprocess.DEVNULL)
        ref = oracle_py.read_ref_output(os.path.join(tmp, name + ".out"))
        out = {"case_bin": np.fromfile(cpath, dtype=np.uint8)}
    for k in ("npix", "npix_hp", "kp", "en_lim", "u_fest", "u_final_dir", "lum_lost_dir", "u_final", "lum_lost",
              "timings"):
        out[k] = np.asarray(ref[k])
    out["iterations"] = np.int32(ref["iterations"])
    out["src_cell_hosts"] = np.flatnonzero(ref["src_cell"] >= 0).astype(np.int32)
    out["src_cell_ids"] = ref["src_cell"][ref["src_cell"] >= 0].astype(np.int32)
    if ref["nsrc"] > 0:
        out["p_src_pos"] = ref["p_src_pos"]
    if ref["ndir"] > 0:
        for k in ("i_obs_dir", "sed_dir", "i_obs", "sed"):
            out[k] = ref[k]
    pack_scaspe("scaspe_dir", ref["scaspe_dir"], out)
    pack_scaspe("scaspe_tot", ref["scaspe_tot"], out)
    path = os.path.join(outdir, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: N={ref['N']} L={ref['L']} iterations={ref['iterations']} "
          f"ref time {ref['timings'].sum():.1f}s -> {path} ({os.path.getsize(path) / 1024:.0f} kB)")


def main():
    outdir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    allc = cases()
    want = sys.argv[1:] or list(allc)
    for name in want:
        make(name, allc[name], outdir)


if __name__ == "__main__":
    main()
