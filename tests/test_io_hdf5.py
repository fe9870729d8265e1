"""HDF5 grid input / field output in the reference's on-disk layout (SURVEY.md section 8f rank 2,
io_routines.f90:184-294, 297-470, 720-794, 1112-1240). CPU only: the library is host code."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import load_pkg

pkg = load_pkg()
hio = pkg.io


@pytest.fixture(scope="module", autouse=True)
def _built():
    if not os.path.exists(hio.LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.dirname(hio.LIB_PATH)])


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(hio.LIB_PATH)
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "dartray_io.h")).read()
    import re
    declared = set(re.findall(r"\b(drt_io_[a-z_]+)\s*\(", hdr))
    assert declared == set(hio.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name


def test_main_grid_round_trip_and_on_disk_contract(tmp_path):
    case = pkg.grids.blob_case(base=(3, 3), max_lvl=3, nlambda=2)
    t = case.tree
    path = str(tmp_path / "grid_main.h5")
    hio.write_main_grid(path, t, case.dens[:, 0], case.dens_stars[:, 0])
    g = hio.read_main_grid(path)
    assert (g["cchild"] == t.cchild).all() and (g["lvl"] == t.lvl).all() and (g["cindex"] == t.cindex).all()
    assert (g["ccoord"] == t.ccoord).all() and (g["csize"] == t.csize).all()
    assert (g["dens"] == case.dens[:, 0]).all() and (g["dens_stars"] == case.dens_stars[:, 0]).all()
    assert tuple(g["base"]) == tuple(t.base)
    h5dump = shutil.which("h5dump") or "/opt/conda/bin/h5dump"
    if not os.path.exists(h5dump):
        pytest.skip("h5dump not available")
    hdr = subprocess.check_output([h5dump, "-H", path], text=True)
    N = t.ncell
    # names, types and C-order extents a file written by print_3d_grid_file has (io_routines.f90:184-243, 701-717)
    want = {"ncell": ("H5T_STD_I32LE", f"( {N} )"), "cchild": ("H5T_STD_I32LE", f"( {N} )"), "lvl": ("H5T_STD_I32LE", f"( {N} )"),
            "cindex": ("H5T_STD_I64LE", f"( {N} )"), "ccoord": ("H5T_IEEE_F64LE", f"( {N}, 3 )"), "csize": ("H5T_IEEE_F64LE", f"( {N} )"),
            "dens": ("H5T_IEEE_F64LE", f"( {N} )"), "dens_stars": ("H5T_IEEE_F64LE", f"( {N} )"), "base": ("H5T_STD_I32LE", "( 2 )")}
    import re
    for name, (typ, ext) in want.items():
        m = re.search(r'DATASET "%s" \{\s*DATATYPE\s+(\S+)\s*DATASPACE\s+SIMPLE \{ (\([^)]*\))' % name, hdr)
        assert m, name
        assert m.group(1) == typ and m.group(2) == ext, (name, m.groups())


def test_lambda_grid_and_size_check(tmp_path):
    d, s = np.linspace(0, 1, 100), np.linspace(2, 3, 100)
    path = str(tmp_path / hio.lambda_grid_filename("m", 0.443))
    assert os.path.basename(path) == "grid_m_l0.443um.h5"
    hio.write_lambda_grid(path, d, s)
    d2, s2 = hio.read_lambda_grid(path, 100)
    assert (d2 == d).all() and (s2 == s).all()
    with pytest.raises(hio.DartRayIOError):  # "dimensions of lambda grid does not coincide" (io_routines.f90:437-444)
        hio.read_lambda_grid(path, 99)
    with pytest.raises(hio.DartRayIOError):
        hio.read_lambda_grid(str(tmp_path / "missing.h5"), 100)


def test_output_arrays_and_file_names(tmp_path):
    # label_wave_arr is written with F9.3 and left-adjusted (io_routines.f90:2301); names from set_filenames :757-790
    assert hio.filename("trustI", 1.0, "ufield") == "grid_trustI_l1.000um_ufield.h5"
    assert hio.filename("trustI", 12.32847, "scaspe_tot") == "grid_trustI_l12.328um_scaspe_tot.h5"
    assert hio.filename("trustI", 0.1519911, "i_obs_part2", "_dust_eff") == "grid_trustI_l0.152um_i_obs_part2_dust_eff.h5"
    assert hio.filename("trustI", -1.0, "sed") == "grid_trustI_sed.h5"
    sc = np.arange(7 * 5, dtype=np.float64).reshape(7, 5)  # [N=7][npix=5] == Fortran scaspe(npix, N)
    p = str(tmp_path / "sc.h5")
    hio.write_array(p, "scaspe_tot", [5, 7], sc)
    hio.add_int(p, "nside_sca", 2)
    assert (hio.read_array(p, "scaspe_tot", [5, 7]) == sc).all()
    assert hio.read_int(p, "nside_sca") == 2
    with pytest.raises(hio.DartRayIOError):
        hio.read_array(p, "scaspe_tot", [7, 5])
    with pytest.raises(hio.DartRayIOError):
        hio.read_array(p, "u_final", [35])


def test_fortran_main_reads_the_references_input_files():
    """The namelists of dart-ray_amd/fortran/dartray_io.f90 are the reference's (io_routines.f90:176-178): its
    shipped input files parse as they are, and the program gets as far as the (not shipped) HDF5 grid.
    Runs only where the reference tree is mounted (never on the GPU box)."""
    ref = "/root/reference"
    exe = os.path.join(os.path.dirname(pkg.__file__), "fortran", "dartray_gpu_main")
    if not os.path.isdir(ref) or not os.path.exists(exe):
        pytest.skip("reference tree or Fortran main program not available")
    for inp, grid in (("TRUSTI_GRIDS/input_trustI_tau0.01_LD.in", "grid_trustI_tau0.01_l1um_LD.h5"),
                      ("MAGTAR_GRIDS/input_shell.in", "grid_shell.h5")):
        out = subprocess.run([exe, inp], cwd=ref, capture_output=True, text=True, timeout=60)
        assert out.returncode != 0
        assert "file not found" in out.stdout + out.stderr and grid in out.stdout + out.stderr   # check_input
