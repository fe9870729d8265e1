"""ctypes binding of the oracle library and reader of ref_driver output files.

ORACLE INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg (see oracle/dartray_oracle.h).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "liboracle.so")

RTT = dict(precalc_cell=1, precalc_src=2, dir_cell=5, i_obs_dir_cell=6, dir_src=7, i_obs_dir_src=8,
           scatt=10, i_obs=11)


class Counters(C.Structure):
    _fields_ = [("intersections", C.c_int64), ("rays", C.c_int64), ("deposits", C.c_int64),
                ("max_high", C.c_int64), ("max_low", C.c_int64), ("max_ext", C.c_int64),
                ("max_nside", C.c_int)]


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "dartray_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "oracle"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.orc_create_from_file.restype = C.c_void_p
        L.orc_create_from_file.argtypes = [C.c_char_p]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_set_parallel.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        for fn in ("orc_rt_loop", "orc_rt_loop_iobs", "orc_rt_mpi", "orc_run_all"):
            getattr(L, fn).argtypes = [C.c_void_p, C.c_int]
        L.orc_scatt_prepare.argtypes = [C.c_void_p]
        L.orc_scatt_prepare.restype = C.c_int
        L.orc_iobs_prepare.argtypes = [C.c_void_p]
        for fn in ("orc_ncell", "orc_lnum", "orc_iterations"):
            getattr(L, fn).argtypes = [C.c_void_p]
            getattr(L, fn).restype = C.c_int
        for fn in ("orc_npix", "orc_npix_hp", "orc_kp"):
            getattr(L, fn).argtypes = [C.c_void_p, C.c_int]
            getattr(L, fn).restype = C.c_int
        L.orc_en_lim.argtypes = [C.c_void_p]
        L.orc_en_lim.restype = C.c_double
        for fn in ("orc_u_fest", "orc_u_final", "orc_lumcell", "orc_lum_lost", "orc_lum_lost_prev", "orc_i_obs", "orc_p_src_pos"):
            getattr(L, fn).argtypes = [C.c_void_p]
            getattr(L, fn).restype = C.POINTER(C.c_double)
        for fn in ("orc_scaspe", "orc_scaspe_tot", "orc_scaspe_prev"):
            getattr(L, fn).argtypes = [C.c_void_p, C.c_int]
            getattr(L, fn).restype = C.POINTER(C.c_double)
        L.orc_src_cell.argtypes = [C.c_void_p]
        L.orc_src_cell.restype = C.POINTER(C.c_int32)
        L.orc_sed.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.orc_get_counters.argtypes = [C.c_void_p, C.POINTER(Counters)]
        L.orc_set_algorithm.argtypes = [C.c_void_p, C.c_int]
        L.orc_psel_av.argtypes = [C.c_void_p]
        L.orc_psel_av.restype = C.POINTER(C.c_double)
        L.orc_calc_total_luminosity.argtypes = [C.c_void_p]
        L.orc_set_dir_out.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.orc_reset_counters.argtypes = [C.c_void_p]
        L.orc_pix2ang_nest.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_ang2pix_nest.argtypes = [C.c_int, C.c_double, C.c_double]
        L.orc_ang2pix_nest.restype = C.c_int
        _lib = L
    return _lib


class Oracle:
    """One problem instance held by the C restatement."""

    def __init__(self, case_path: str, nsrc: int, ndir: int):
        self.L_ = lib()
        self.h = self.L_.orc_create_from_file(case_path.encode())
        if not self.h:
            raise RuntimeError(f"oracle: cannot read {case_path}")
        self.N = self.L_.orc_ncell(self.h)
        self.L = self.L_.orc_lnum(self.h)
        self.nsrc, self.ndir = nsrc, ndir
        self.npix = [self.L_.orc_npix(self.h, il) for il in range(self.L)]
        self.npix_hp = [self.L_.orc_npix_hp(self.h, il) for il in range(self.L)]
        self.kp = [self.L_.orc_kp(self.h, il) for il in range(self.L)]

    def close(self):
        if self.h:
            self.L_.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_parallel(self, np_=1, id_=0, nthreads=1):
        self.L_.orc_set_parallel(self.h, np_, id_, nthreads)

    def set_algorithm(self, name: str): self.L_.orc_set_algorithm(self.h, {"main": 0, "i_obs": 1}[name])

    def set_dir_out(self, dir_out: np.ndarray):
        """dir_out[ndir][2] = (theta, phi); also zeroes i_obs (create_i_obs)."""
        d = np.ascontiguousarray(np.asarray(dir_out, dtype=np.float64).T)   # theta[ndir] then phi[ndir]
        assert d.shape == (2, self.ndir)
        self.L_.orc_set_dir_out(self.h, d.ctypes.data_as(C.POINTER(C.c_double)))

    @property
    def psel_av(self) -> np.ndarray:
        """psel_av of every source of the last rt_loop ([N + nsrc], rt_routines.f90:1456-1465)."""
        return np.ctypeslib.as_array(self.L_.orc_psel_av(self.h), shape=(self.N + self.nsrc,)).copy()

    def calc_total_luminosity(self): self.L_.orc_calc_total_luminosity(self.h)

    def rt_loop(self, rt): self.L_.orc_rt_loop(self.h, RTT[rt] if isinstance(rt, str) else rt)
    def rt_loop_iobs(self, rt): self.L_.orc_rt_loop_iobs(self.h, RTT[rt] if isinstance(rt, str) else rt)
    def rt_mpi(self, rt): self.L_.orc_rt_mpi(self.h, RTT[rt] if isinstance(rt, str) else rt)
    def scatt_prepare(self) -> bool: return bool(self.L_.orc_scatt_prepare(self.h))
    def iobs_prepare(self): self.L_.orc_iobs_prepare(self.h)
    def run_all(self, stages=31): self.L_.orc_run_all(self.h, stages)

    def _arr(self, ptr, shape):
        return np.ctypeslib.as_array(ptr, shape=shape)

    @property
    def u_fest(self): return self._arr(self.L_.orc_u_fest(self.h), (self.N, self.L))
    @property
    def u_final(self): return self._arr(self.L_.orc_u_final(self.h), (self.N, self.L))
    @property
    def lumcell(self): return self._arr(self.L_.orc_lumcell(self.h), (self.N, self.L))
    @property
    def lum_lost(self): return self._arr(self.L_.orc_lum_lost(self.h), (self.L,))
    @property
    def lum_lost_prev(self): return self._arr(self.L_.orc_lum_lost_prev(self.h), (self.L,))
    @property
    def i_obs(self): return self._arr(self.L_.orc_i_obs(self.h), (max(self.ndir, 1), self.N + self.nsrc, self.L))
    @property
    def src_cell(self): return self._arr(self.L_.orc_src_cell(self.h), (self.N,))
    @property
    def p_src_pos(self): return self._arr(self.L_.orc_p_src_pos(self.h), (max(self.nsrc, 1), 3))
    @property
    def en_lim(self): return self.L_.orc_en_lim(self.h)
    @property
    def iterations(self): return self.L_.orc_iterations(self.h)

    def scaspe(self, il): return self._arr(self.L_.orc_scaspe(self.h, il), (self.N, self.npix[il]))
    def scaspe_tot(self, il): return self._arr(self.L_.orc_scaspe_tot(self.h, il), (self.N, self.npix[il]))
    def scaspe_prev(self, il): return self._arr(self.L_.orc_scaspe_prev(self.h, il), (self.N, self.npix[il]))

    def sed(self):
        out = np.zeros((max(self.ndir, 1), self.L))
        self.L_.orc_sed(self.h, out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def counters(self) -> Dict[str, int]:
        c = Counters()
        self.L_.orc_get_counters(self.h, C.byref(c))
        return {k: int(getattr(c, k)) for k, _ in Counters._fields_}

    def reset_counters(self): self.L_.orc_reset_counters(self.h)


def read_ref_output(path: str, stages: int = 31) -> Dict[str, np.ndarray]:
    """Parse the stream file written by oracle/ref_driver.f90."""
    out: Dict[str, np.ndarray] = {}
    with open(path, "rb") as f:
        magic, N, L, ndir, nsrc = np.fromfile(f, np.int32, 5)
        assert magic == 0x44524F31, "bad ref_driver output"
        N, L, ndir, nsrc = int(N), int(L), int(ndir), int(nsrc)
        out.update(N=N, L=L, ndir=ndir, nsrc=nsrc)
        out["npix"] = np.fromfile(f, np.int32, L)
        out["npix_hp"] = np.fromfile(f, np.int32, L)
        out["kp"] = np.fromfile(f, np.int32, L)
        out["en_lim"] = np.fromfile(f, np.float64, 1)[0]
        out["src_cell"] = np.fromfile(f, np.int32, N)
        if nsrc > 0:
            out["p_src_pos"] = np.fromfile(f, np.float64, 3 * nsrc).reshape(nsrc, 3)
        out["u_fest"] = np.fromfile(f, np.float64, N * L).reshape(N, L)
        out["u_final_dir"] = np.fromfile(f, np.float64, N * L).reshape(N, L)
        out["scaspe_dir"] = [np.fromfile(f, np.float64, N * int(p)).reshape(N, int(p)) for p in out["npix"]]
        out["lum_lost_dir"] = np.fromfile(f, np.float64, L)
        if ndir > 0:
            out["i_obs_dir"] = np.fromfile(f, np.float64, ndir * (N + nsrc) * L).reshape(ndir, N + nsrc, L)
            out["sed_dir"] = np.fromfile(f, np.float64, ndir * L).reshape(ndir, L)
        if stages & 8:
            out["iterations"] = int(np.fromfile(f, np.int32, 1)[0])
            out["u_final"] = np.fromfile(f, np.float64, N * L).reshape(N, L)
            out["scaspe_tot"] = [np.fromfile(f, np.float64, N * int(p)).reshape(N, int(p)) for p in out["npix"]]
            out["lum_lost"] = np.fromfile(f, np.float64, L)
            if ndir > 0:
                out["i_obs"] = np.fromfile(f, np.float64, ndir * (N + nsrc) * L).reshape(ndir, N + nsrc, L)
                out["sed"] = np.fromfile(f, np.float64, ndir * L).reshape(ndir, L)
        out["timings"] = np.fromfile(f, np.float64, 4)
    return out
