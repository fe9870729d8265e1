"""The driver shell around the path on CPU: dartray_gpu_main parses the reference's input
files, validates them like check_input (io_routines.f90:2745-3303), reads the grids and scales dens_arr like
scale_dens_arr (rt_routines.f90:5018-5071). `test_run` - the reference's own dry-run flag (rt_routines.f90:492-493) -
stops in front of the RT loops, so none of this needs a GPU."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def _exe(pkg):
    exe = os.path.join(os.path.dirname(pkg.__file__), "fortran", "dartray_gpu_main")
    if not os.path.exists(exe) or not os.path.exists(pkg.io.LIB_PATH):
        pytest.skip("Fortran main program / HDF5 library not built (amdflang or HDF5 missing)")
    return exe


def test_shipped_trustI_input_runs_up_to_the_rt_loops(pkg, tmp_path):
    """The one RT input the reference ships, TRUSTI_GRIDS/input_trustI_tau0.01_LD.in, UNCHANGED, with the reference's own
    data files (wavelength list, observer directions, star position, black-body table, effective opacities) and the
    main grid the reference's create_adap_grid_trustI would write (ours: grids.trusti_case, same file layout). No
    lambda grids, no file_param_src: scale_dens_arr and the TRUST-I source table are what make it run. Build container
    only - the reference's files do not travel."""
    if not os.path.isdir(REF):
        pytest.skip("needs /root/reference (build container)")
    exe = _exe(pkg)
    g = pkg.grids
    os.makedirs(tmp_path / "TRUSTI_GRIDS")
    for f in os.listdir(os.path.join(REF, "TRUSTI_GRIDS")):
        os.symlink(os.path.join(REF, "TRUSTI_GRIDS", f), tmp_path / "TRUSTI_GRIDS" / f)
    os.symlink(os.path.join(REF, "DUST_OPACITY"), tmp_path / "DUST_OPACITY")
    case = g.trusti_case(tau_z=0.01, max_lvl=4, wavelengths=list(range(32)))
    pkg.io.write_main_grid(str(tmp_path / "TRUSTI_GRIDS" / "grid_trustI_tau0.01_l1um_LD.h5"), case.tree, case.dens[:, 11], case.dens_stars[:, 11])
    r = subprocess.run([exe, "TRUSTI_GRIDS/input_trustI_tau0.01_LD.in", "0", "test_run"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    m = re.search(r"tot_ncell (\d+) lnum (\d+) point sources (\d+) directions (\d+)", r.stdout)
    assert m and [int(x) for x in m.groups()] == [21801, 32, 1, 7], r.stdout[-2000:]
    d = [float(x) for x in re.search(r"dens_arr\(first, ref, last wavelength\) max\s+(\S+)\s+(\S+)\s+(\S+)", r.stdout).groups()]
    lum = [float(x) for x in re.search(r"lum_p_src_arr\(first, last wavelength\)\s+(\S+)\s+(\S+)", r.stdout).groups()]
    tab = g.TRUSTI_TABLE
    dref = 0.01 / 3.0
    assert d[1] == pytest.approx(dref, rel=1e-12)
    # the synthetic inputs of the bench (grids.TRUSTI_TABLE) are these very interpolations, tabulated to 9 digits
    assert d[0] == pytest.approx(dref * tab[0, 1], rel=1e-7) and d[2] == pytest.approx(dref * tab[31, 1], rel=1e-7)
    assert lum[0] == pytest.approx(tab[0, 4], rel=1e-7) and lum[1] == pytest.approx(tab[31, 4], rel=1e-7)
    for w in ("no_dust_rt = .FALSE.", "print_maps = .TRUE."):
        assert w in r.stdout


def _inputs(pkg, tmp_path, **kw):
    case = pkg.grids.slab_case(max_lvl=2, nlambda=2, ndir=2)
    inp = pkg.io.write_reference_inputs(case, [0.5, 1.0], str(tmp_path), label="slab", **kw)
    return case, inp


def test_check_input_lists_every_violation(pkg, tmp_path):
    """check_input's ranges (io_routines.f90:3015-3125): kp_sca_max in [0,4], rad_lim in [0,2], accuracy and conv_en_lim
    in [0,1], bm_par* in [0,1000], bm_par_max >= 10*max(bm_par, bm_par_sca), wall fractions ordered within [0,1]. All
    violations are printed before the program stops (error_stop)."""
    exe = _exe(pkg)
    case, inp = _inputs(pkg, tmp_path)
    ok = subprocess.run([exe, inp, "0", "test_run"], capture_output=True, text=True, timeout=120)
    assert ok.returncode == 0, ok.stdout[-2000:] + ok.stderr[-2000:]
    txt = open(inp).read()
    bad = (txt.replace("kp_sca_max = 1", "kp_sca_max = 5").replace("accuracy = 0.1", "accuracy = 1.5")
           .replace("bm_par_max = 200", "bm_par_max = 60").replace("z_wall_on = .FALSE.", "z_wall_on = .TRUE.")
           )
    bad = re.sub(r"z_wall_coord = \S+ \S+", "z_wall_coord = 0.6, 0.3,", bad)
    assert bad != txt and "0.6, 0.3" in bad
    open(inp, "w").write(bad)
    r = subprocess.run([exe, inp, "0", "test_run"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    for msg in ("Invalid kp_sca_max value", "Invalid accuracy value", "Invalid bm_par_max value", "z_wall_coord values not allowed", "STOP(check_input)"):
        assert msg in r.stdout, (msg, r.stdout[-2000:])


def test_missing_files_and_foreign_algorithms_are_refused(pkg, tmp_path):
    exe = _exe(pkg)
    case, inp = _inputs(pkg, tmp_path)
    txt = open(inp).read()
    open(inp, "w").write(txt.replace("rt_algorithm = 'main'", "rt_algorithm = 'dust2D'").replace("file_dir_out = 'dir_out.dat'", "file_dir_out = 'nowhere.dat'"))
    r = subprocess.run([exe, inp, "0", "test_run"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "outside the GPU path" in r.stdout and "nowhere.dat" in r.stdout


def test_i_obs_algorithm_input_is_accepted(pkg, tmp_path):
    """rt_algorithm = 'i_obs' (io_routines.f90:2778-2779): accepted, label_model_out_i_obs defaults to label_model_out
    (:2898-2902) and observer directions are required; test_run stops before the GPU is touched."""
    exe = _exe(pkg)
    case = pkg.grids.slab_case(max_lvl=2, nlambda=2, ndir=2)
    inp = pkg.io.write_reference_inputs(case, [0.5, 1.0], str(tmp_path), label="slab", rt_algorithm="i_obs")
    r = subprocess.run([exe, inp, "0", "test_run"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "test_run" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    txt = open(inp).read()
    open(inp, "w").write(txt.replace("use_dir_out = .TRUE.", "use_dir_out = .FALSE."))
    r = subprocess.run([exe, inp, "0", "test_run"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "needs observer directions" in r.stdout


def test_main_grid_only_needs_emission_free_main_grid(pkg, tmp_path):
    """Without lambda grids there is no spectrum for diffuse stellar emission (the reference's user routines set it):
    refused with a message instead of running with zeros."""
    exe = _exe(pkg)
    case = pkg.grids.blob_case(base=(4, 2), max_lvl=2, nlambda=2)
    inp = pkg.io.write_reference_inputs(case, [0.5, 1.0], str(tmp_path), label="blob")
    txt = open(inp).read()
    open(inp, "w").write(txt.replace("use_lambda_grid = .TRUE.", "use_lambda_grid = .FALSE."))
    r = subprocess.run([exe, inp, "0", "test_run"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "stellar emissivity" in r.stdout


def test_facade_links_against_the_reference_modules(pkg):
    """INTEGRATION.md tells a maintainer to replace `use dartray_state` by the reference's modules. `make linktest` does
    exactly that (-DDRT_REAL_MODULES: `use rt_routines, only: <every name the facade touches>`), compiles the facade
    against the .mod files of the UNMODIFIED reference modules (oracle/_ref/build) and links it with their objects: a
    kind, rank or name mismatch (bm_par, scaspe_arr%a, the rtt_* ids ...) or a clash with rt_routines' own rt_loop would
    fail here. Build container only."""
    import shutil
    if not os.path.isdir(REF) or shutil.which("amdflang") is None:
        pytest.skip("needs /root/reference and amdflang (build container)")
    fdir = os.path.join(os.path.dirname(pkg.__file__), "fortran")
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "build", "rt_routines.mod")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    shutil.rmtree(os.path.join(fdir, "_linktest"), ignore_errors=True)
    r = subprocess.run(["make", "-C", fdir, "linktest"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    out = subprocess.run([os.path.join(fdir, "_linktest", "linktest")], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "facade linked against the reference modules" in out.stdout
