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
           "drt_io_add_int", "drt_io_read_int", "drt_io_filename", "drt_io_lambda_grid_filename"]
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
            add_int(p, "nside_sca", 2 ** max(int(gpu.kp[il]), 0))
        if ndir:
            write_array(j(filename(label, lambdas[il], "i_obs_part2" if part2 else "i_obs")), "i_obs", [N + nsrc, ndir],
                        io[:, :, il])
    write_array(j(filename(label, -1.0, "lum_lost_part2" if part2 else "lum_lost")), "lum_lost", [L], gpu.lum_lost)
    if ndir:
        write_array(j(filename(label, -1.0, "sed_dir" if part2 else "sed")), "sed_arr_dir" if part2 else "sed_arr", [L, ndir], gpu.sed())
