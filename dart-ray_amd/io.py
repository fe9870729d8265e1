"""ctypes binding of dart-ray_amd/io/libdartray_io.so (include/dartray_io.h): the reference's HDF5 grid input
and field output (io_routines.f90 read_main_grid :297, read_lambda_grid :397, print_3d_grid_file :202,
print_big_array :1112, set_filenames :720). Used by tests and tools; the Fortran host binds the same C ABI."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Tuple

import numpy as np

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "io", "libdartray_io.so")
EXPORTS = ["drt_io_last_error", "drt_io_write_main_grid", "drt_io_main_grid_size", "drt_io_read_main_grid",
           "drt_io_write_lambda_grid", "drt_io_read_lambda_grid", "drt_io_write_array", "drt_io_read_array",
           "drt_io_add_int", "drt_io_read_int", "drt_io_filename", "drt_io_lambda_grid_filename", "drt_io_array_dims"]
_lib = None


class DartRayIOError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DartRayIOError(f"{LIB_PATH} not built (make -C dart-ray_amd/io)")
        _lib = C.CDLL(LIB_PATH)
        _lib.drt_io_last_error.restype = C.c_char_p
    return _lib


def _chk(rc):
    if rc != 0:
        raise DartRayIOError(f"dartray_io error {rc}: {lib().drt_io_last_error().decode()}")


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def write_main_grid(path: str, tree, dens_ref: np.ndarray, dens_stars_ref: np.ndarray) -> None:
    """print_3d_grid_file: `tree` is a grids.Tree; dens_ref / dens_stars_ref are the [N] reference-wavelength arrays."""
    N = tree.ncell
    ncell = np.zeros(N, dtype=np.int32)  # per-cell bookkeeping of the builders; the ray tracer only uses its length
    cc = np.ascontiguousarray(tree.ccoord, dtype=np.float64)  # [N][3] row-major == Fortran (3,N)
    base = np.array(tree.base, dtype=np.int32)
    _chk(lib().drt_io_write_main_grid(path.encode(), C.c_int32(N), _p(ncell, C.c_int32), _p(cc, C.c_double),
                                      _p(np.ascontiguousarray(tree.cchild, dtype=np.int32), C.c_int32),
                                      _p(np.ascontiguousarray(tree.cindex, dtype=np.int64), C.c_int64),
                                      _p(np.ascontiguousarray(tree.lvl, dtype=np.int32), C.c_int32),
                                      _p(np.ascontiguousarray(tree.csize, dtype=np.float64), C.c_double),
                                      _p(np.ascontiguousarray(dens_ref, dtype=np.float64), C.c_double),
                                      _p(np.ascontiguousarray(dens_stars_ref, dtype=np.float64), C.c_double),
                                      _p(base, C.c_int32)))


def read_main_grid(path: str) -> Dict[str, np.ndarray]:
    n = C.c_int32(0)
    _chk(lib().drt_io_main_grid_size(path.encode(), C.byref(n)))
    N = n.value
    out = {"ncell": np.zeros(N, np.int32), "ccoord": np.zeros((N, 3)), "cchild": np.zeros(N, np.int32),
           "cindex": np.zeros(N, np.int64), "lvl": np.zeros(N, np.int32), "csize": np.zeros(N), "dens": np.zeros(N),
           "dens_stars": np.zeros(N), "base": np.zeros(2, np.int32)}
    _chk(lib().drt_io_read_main_grid(path.encode(), n, _p(out["ncell"], C.c_int32), _p(out["ccoord"], C.c_double),
                                     _p(out["cchild"], C.c_int32), _p(out["cindex"], C.c_int64), _p(out["lvl"], C.c_int32),
                                     _p(out["csize"], C.c_double), _p(out["dens"], C.c_double), _p(out["dens_stars"], C.c_double),
                                     _p(out["base"], C.c_int32)))
    return out


def write_lambda_grid(path: str, dens: np.ndarray, dens_stars: np.ndarray) -> None:
    d, s = np.ascontiguousarray(dens, dtype=np.float64), np.ascontiguousarray(dens_stars, dtype=np.float64)
    _chk(lib().drt_io_write_lambda_grid(path.encode(), C.c_int32(d.shape[0]), _p(d, C.c_double), _p(s, C.c_double)))


def read_lambda_grid(path: str, N: int) -> Tuple[np.ndarray, np.ndarray]:
    d, s = np.zeros(N), np.zeros(N)
    _chk(lib().drt_io_read_lambda_grid(path.encode(), C.c_int32(N), _p(d, C.c_double), _p(s, C.c_double)))
    return d, s


def write_array(path: str, dsetname: str, fdims, data: np.ndarray) -> None:
    """`fdims` in Fortran order; `data` is the C-contiguous array whose memory equals the Fortran array's
    (e.g. scaspe (npix, N) <- numpy [N][npix])."""
    fd = np.array(list(fdims), dtype=np.int64)
    a = np.ascontiguousarray(data, dtype=np.float64)
    if a.size != int(np.prod(fd)):
        raise DartRayIOError(f"{dsetname}: {a.size} elements for dims {tuple(fd)}")
    _chk(lib().drt_io_write_array(path.encode(), dsetname.encode(), C.c_int(fd.size), _p(fd, C.c_int64), _p(a, C.c_double)))


def read_array(path: str, dsetname: str, fdims) -> np.ndarray:
    fd = np.array(list(fdims), dtype=np.int64)
    a = np.zeros(tuple(int(x) for x in fd[::-1]))
    _chk(lib().drt_io_read_array(path.encode(), dsetname.encode(), C.c_int(fd.size), _p(fd, C.c_int64), _p(a, C.c_double)))
    return a


def add_int(path: str, dsetname: str, value: int) -> None:
    _chk(lib().drt_io_add_int(path.encode(), dsetname.encode(), C.c_int32(value)))


def read_int(path: str, dsetname: str) -> int:
    v = C.c_int32(0)
    _chk(lib().drt_io_read_int(path.encode(), dsetname.encode(), C.byref(v)))
    return v.value


def filename(label: str, lam: float, what: str, chext: str = "") -> str:
    buf = C.create_string_buffer(512)
    _chk(lib().drt_io_filename(buf, C.c_size_t(512), label.encode(), C.c_double(lam), what.encode(), chext.encode()))
    return buf.value.decode()


def lambda_grid_filename(label: str, lam: float) -> str:
    buf = C.create_string_buffer(512)
    _chk(lib().drt_io_lambda_grid_filename(buf, C.c_size_t(512), label.encode(), C.c_double(lam)))
    return buf.value.decode()


def write_outputs(gpu, label: str, lambdas, out_dir: str, stage: str) -> None:
    """make_output (io_routines.f90:1743-1897) for one stage of a DartRayGPU run: 'part1' (u_fest), 'part2'
    (direct light: ufield_part2, scaspe_part2, lum_lost_part2, i_obs_part2, sed_dir) or 'final' (ufield,
    scaspe_tot + nside_sca, lum_lost, i_obs, sed)."""
    N, L, ndir, nsrc = gpu.N, gpu.L, gpu.ndir, gpu.nsrc
    j = lambda name: os.path.join(out_dir, name)
    if stage == "part1":
        u = gpu.u_fest
        for il in range(L):
            write_array(j(filename(label, lambdas[il], "u_fest_part1")), "u_fest", [N], u[:, il])
        return
    part2 = stage == "part2"
    u = gpu.u_final
    io = gpu.i_obs if ndir else None
    for il in range(L):
        write_array(j(filename(label, lambdas[il], "ufield_part2" if part2 else "ufield")), "u_final", [N], u[:, il])
        npix = int(gpu.npix[il])
        if part2:
            write_array(j(filename(label, lambdas[il], "scaspe_part2")), "scaspe", [npix, N], gpu.scaspe(il))
        else:
            p = j(filename(label, lambdas[il], "scaspe_tot"))
            write_array(p, "scaspe_tot", [npix, N], gpu.scaspe_tot(il))
            add_int(p, "nside_sca", 2 ** int(gpu.kp[il]) if gpu.kp[il] >= 0 else 0)   # integer 2**kp_sca_arr: 0 for isotropic (kp = -1), io_routines.f90:1594
        if ndir:
            write_array(j(filename(label, lambdas[il], "i_obs_part2" if part2 else "i_obs")), "i_obs", [N + nsrc, ndir],
                        io[:, :, il])
    write_array(j(filename(label, -1.0, "lum_lost_part2" if part2 else "lum_lost")), "lum_lost", [L], gpu.lum_lost)
    if ndir:
        write_array(j(filename(label, -1.0, "sed_dir" if part2 else "sed")), "sed_arr_dir" if part2 else "sed_arr", [L, ndir], gpu.sed())


def write_reference_inputs(case, lambdas, root: str, label: str = "model", use_lambda_grid: bool = True, iref: int = 0,
                           restore_file_mpi: bool = False, rt_algorithm: str = "main", label_i_obs: str = "",
                           input_name: str = "input.in", dir_out_name: str = "dir_out.dat", only_direct_rt: bool = False,
                           print_psel_av: bool = False) -> str:
    """Write `case` (grids.Case) as the set of files the reference's main programs read (and
    dart-ray_amd/fortran/dartray_gpu_main reads): <root>/GRIDS/ with the HDF5 main grid, one lambda grid per
    wavelength, the wavelength list, dir_out and point-source files, an effective-opacity table in the layout
    of DUST_OPACITY/TRUST/*_Effective.dat and the source spectra table; <root>/RUNS/ for the output; and
    <root>/input.in with the three namelists. Returns the path of input.in.
    use_lambda_grid = False leaves the per-wavelength grids out: the run then scales the main grid's dens with
    kext(lambda)/kext(lambda_ref) from the opacity table (scale_dens_arr, rt_routines.f90:5018), which needs a case whose
    dens columns are multiples of each other and which has no diffuse stellar emission.
    restore_file_mpi: the run may start from the part-1 / part-2 files of an earlier run found in <root>/RUNS
    (check_files, io_routines.f90:893).
    rt_algorithm = 'i_obs' (with label_i_obs, another input_name and dir_out_name in the same root): the input of a
    dartray_i_obs run on the scaspe_tot files a main run left in <root>/RUNS."""
    lambdas = [float(x) for x in lambdas]
    L = case.lnum
    assert len(lambdas) == L and all(b > a for a, b in zip(lambdas, lambdas[1:]))
    gdir, rdir = os.path.join(root, "GRIDS"), os.path.join(root, "RUNS")
    os.makedirs(gdir, exist_ok=True)
    os.makedirs(rdir, exist_ok=True)
    t = case.tree
    write_main_grid(os.path.join(gdir, f"grid_{label}.h5"), t, case.dens[:, iref], case.dens_stars[:, iref])
    if use_lambda_grid:
        for il, lam in enumerate(lambdas):
            write_lambda_grid(os.path.join(gdir, lambda_grid_filename(label, lam)), case.dens[:, il], case.dens_stars[:, il])
    else:
        assert not case.dens_stars.any(), "main-grid-only runs have no way to give diffuse emission a spectrum"
    # extinction coefficient per wavelength relative to the reference one: what scale_dens_arr multiplies dens with
    jmax = int(np.argmax(case.dens[:, iref]))
    kscale = case.dens[jmax, :] / case.dens[jmax, iref]
    assert np.allclose(case.dens, case.dens[:, iref:iref + 1] * kscale[None, :], rtol=1e-12, atol=0.0)
    with open(os.path.join(gdir, "lambda_list.dat"), "w") as f:
        for lam in lambdas:
            f.write(f"{lam!r}\n")
    ndir, nsrc = case.dir_out.shape[0], case.p_src_pos.shape[0]
    if ndir:
        with open(os.path.join(gdir, dir_out_name), "w") as f:
            f.write("theta phi [rad]\n")
            for th, ph in case.dir_out:
                f.write(f"{float(th)!r} {float(ph)!r}\n")
    if nsrc:
        with open(os.path.join(gdir, "pos_point_sources.dat"), "w") as f:
            f.write("x y z\n")
            for p in case.p_src_pos:
                f.write(" ".join(repr(float(x)) for x in p) + "\n")
        with open(os.path.join(gdir, "param_point_sources.dat"), "w") as f:
            f.write("lambda L_nu per source\n")
            rows = [(0.5 * lambdas[0], case.p_src_lum[:, 0])] + [(lam, case.p_src_lum[:, il]) for il, lam in enumerate(lambdas)]
            for lam, lum in rows:
                f.write(f"{lam!r} " + " ".join(repr(float(x)) for x in lum) + "\n")
    with open(os.path.join(gdir, "opacities_Effective.dat"), "w") as f:
        f.write("effective quantities written by dart-ray_amd/io.py\n   lam         Cabs          Csca        Tau\n"
                "   [um]       [cm^2]        [cm^2]      [/N_H]           a           g\n" + "-" * 77 + "\n")
        K = 1e-21
        rows = [(0.5 * lambdas[0], case.ksca_norm[0], case.gsca[0], kscale[0])] + [(lam, case.ksca_norm[il], case.gsca[il], kscale[il]) for il, lam in enumerate(lambdas)]
        for lam, a, g, ks in rows:
            f.write(f"{lam!r} {(1.0 - float(a)) * K * float(ks)!r} {float(a) * K * float(ks)!r} {K * float(ks)!r} {float(a)!r} {float(g)!r}\n")
    wc = np.asarray(case.wall_coord, dtype=np.float64)
    ms = t.modelsize
    frac = [[float(v) for v in row] for row in (wc + ms / 2.0) / ms]  # set_walls maps fractions of modelsize to model coordinates
    fl = lambda b: ".TRUE." if b else ".FALSE."
    inp = os.path.join(root, input_name)
    with open(inp, "w") as f:
        f.write("&dartray_input_strings\n"
                f"label_model_lambda_grid = '{label}',\nlabel_model_out = '{label}',\ngrid_file = 'grid_{label}.h5',\n"
                f"file_dir_out = '{dir_out_name}',\nfile_p_src = 'pos_point_sources.dat'," f"\nfile_param_src = 'param_point_sources.dat',\n"
                f"file_lambda_list = 'lambda_list.dat',\ndir_runs = '{rdir}',\ndir_grid = '{gdir}/',\nrt_algorithm = '{rt_algorithm}',\n" + (f"label_model_out_i_obs = '{label_i_obs}',\n" if label_i_obs else "") + f""
                f"units_lambda = 'um',\ndust_model = 'TRUST',\nfile_av_opacities = '{gdir}/opacities_Effective.dat',\n"
                "units_csize = 'pc',\nunits_luminosity = 'erg/s/Hz',\ndust_heating_type = 'eff'\n/\n")
        f.write("&dartray_input_var\n"
                f"lambda_ref = {lambdas[iref]!r},\nkp_sca_max = {case.kp_sca_max},\nrad_lim = {float(case.rad_lim)!r},\naccuracy = {float(case.accuracy)!r},\n"
                f"conv_en_lim = {float(case.conv_en_lim)!r},\nbm_par = {case.bm_par},\nbm_par_sca = {case.bm_par_sca},\nbm_par_max = {case.bm_par_max},\n"
                f"max_lambda_stars = {lambdas[-1] * 1.001!r},\nmin_lambda_dust = 1.,\ndist_obs = {float(case.dist_obs)!r},\nnpixel_maps = 300.,\n"
                f"x_wall_coord = {frac[0][0]!r}, {frac[0][1]!r},\ny_wall_coord = {frac[1][0]!r}, {frac[1][1]!r},\n"
                f"z_wall_coord = {frac[2][0]!r}, {frac[2][1]!r},\nmax_sca_iterations = {max(case.max_sca_iterations, 0)}\n/\n")
        f.write("&dartray_input_logical\n"
                f"use_lambda_grid = {fl(use_lambda_grid)},\nuse_dir_out = {fl(ndir > 0)},\nuse_p_src = {fl(nsrc > 0)},\nprint_output_part1 = .TRUE.,\n"
                "print_output_part2 = .TRUE.,\nprint_scaspe_part2 = .TRUE.,\nprint_scaspe_tot = .TRUE.,\nprint_sed = .TRUE.,\n"
                f"input_av_opacities = .TRUE.,\nno_communications = .TRUE.,\nno_dust_rt = .TRUE.,\nrestore_file_mpi = {fl(restore_file_mpi)},\nonly_direct_rt = {fl(only_direct_rt)},\nprint_psel_av = {fl(print_psel_av)},\n"
                f"x_wall_on = {fl(case.wall_on[0])},\ny_wall_on = {fl(case.wall_on[1])},\nz_wall_on = {fl(case.wall_on[2])},\n"
                f"limit_scattering_iterations = {fl(case.max_sca_iterations >= 0)}\n/\n")
    return inp
