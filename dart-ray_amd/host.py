"""ctypes binding of libdartray_rt.so (include/dartray_rt.h) — the thin Python host used by the
tests and bench.py. It mirrors the reference's driver-side names: `rt_prepare`, `rt_loop`,
`rt_loop_iobs`, `rt_mpi` (dartray_hub.f90:12-67) act on a context that holds what the Fortran code
keeps in module variables. There is no CPU fallback: if the library is missing or no gfx950 device
is present, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.environ.get("DRT_LIB", os.path.join(CSRC, "libdartray_rt.so"))

RTT = dict(start=0, precalc_cell=1, precalc_src=2, prep_part2=4, dir_cell=5, i_obs_dir_cell=6, dir_src=7,
           i_obs_dir_src=8, scatt=10, i_obs=11)
A = dict(src_cost=13, psel_av=14, psel_sum=15, psel_cnt=16, u_fest=1, u_final=2, scaspe=3, scaspe_tot=4, scaspe_prev=5, lumcell=6, lum_lost=7, i_obs=8, dens=9,
         dens_stars=10, tot_rad_en=11, lum_lost_prev=12)
UNIQUE_ID_BYTES = 128


class DartRayError(RuntimeError):
    pass


class Params(C.Structure):
    _fields_ = [("kp_sca_max", C.c_int32), ("bm_par", C.c_int32), ("bm_par_sca", C.c_int32), ("bm_par_max", C.c_int32),
                ("rad_lim", C.c_double), ("accuracy", C.c_double), ("conv_en_lim", C.c_double), ("cs", C.c_double),
                ("dist_obs", C.c_double), ("wall_on", C.c_int32 * 3), ("wall_coord", C.c_double * 6),
                ("max_sca_iterations", C.c_int32), ("np_mpi", C.c_int32), ("id_mpi", C.c_int32)]


class Counters(C.Structure):
    _fields_ = [("intersections", C.c_int64), ("rays", C.c_int64), ("deposits", C.c_int64), ("scaspe_adds", C.c_int64),
                ("lambda_deposits", C.c_int64), ("u_adds", C.c_int64), ("u_reads", C.c_int64), ("scaspe_slots", C.c_int64),
                ("sources", C.c_int64), ("kernel_ms", C.c_double), ("launches", C.c_int64),
                ("atomic_instructions", C.c_int64), ("atomic_lanes", C.c_int64),
                ("heavy_tasks", C.c_int64), ("deep_sources", C.c_int64), ("retries", C.c_int64)]


EXPORTS = ["drt_create", "drt_destroy", "drt_last_error", "drt_configure", "drt_set_grid", "drt_set_params",
           "drt_set_wavelengths", "drt_set_point_sources", "drt_set_observers", "drt_rt_prepare", "drt_rt_loop",
           "drt_rt_loop_iobs", "drt_rt_mpi", "drt_get_array", "drt_set_array", "drt_get_npix", "drt_get_src_cell",
           "drt_device_ptr", "drt_en_lim", "drt_iterations", "drt_cnflag", "drt_calc_sed", "drt_get_counters", "drt_reset_counters",
           "drt_comm_unique_id", "drt_comm_init", "drt_comm_destroy", "drt_comm_selftest", "drt_arena_carve", "drt_get_sources", "drt_mark_cost_reduced", "drt_deal_by_cost", "drt_set_algorithm", "drt_set_psel_av", "drt_plan_tasks", "drt_fit_queue_prior"]


def build(force: bool = False) -> str:
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in ("dartray_rt.hip", "rt_kernels.hpp", "rt_wide.hpp", "rt_device.hpp", "Makefile")]
    srcs.append(os.path.join(os.path.dirname(HERE), "include", "dartray_rt.h"))
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", CSRC], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DartRayError(f"{LIB_PATH} is missing: run __graft_entry__.build() (there is no CPU fallback)")
        # ONE HIP runtime per process. The torch wheel bundles its own libamdhip64.so / libhsa-runtime64.so (SONAME
        # libamdhip64.so.7, but NEEDED by libc10_hip.so under the file name libamdhip64.so). Mapped first, that copy also
        # satisfies this library's NEEDED libamdhip64.so.7 and the process holds one runtime. Mapped second - a lazy
        # `import torch` from device_tensor() / attach_process_group() after contexts exist - glibc does not match the file
        # name against /opt/rocm's copy, maps torch's runtime beside it, and the second runtime finds no device ("No HIP GPUs
        # are available", tools/repro_torch_late_init.py). So torch, where it is installed, is imported before the library.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        vp, i32, dp, ip = C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int32)
        L.drt_create.argtypes = [C.POINTER(vp), C.c_int]
        L.drt_destroy.argtypes = [vp]
        L.drt_last_error.argtypes = [vp]
        L.drt_last_error.restype = C.c_char_p
        L.drt_configure.argtypes = [vp, C.c_size_t, C.c_int]
        L.drt_set_grid.argtypes = [vp, i32, i32, ip, ip, ip, C.POINTER(C.c_int64), dp, dp]
        L.drt_set_params.argtypes = [vp, C.POINTER(Params)]
        L.drt_set_wavelengths.argtypes = [vp, i32, dp, dp]
        L.drt_set_point_sources.argtypes = [vp, i32, dp, dp]
        L.drt_set_observers.argtypes = [vp, i32, dp]
        for fn in ("drt_rt_prepare", "drt_rt_loop", "drt_rt_loop_iobs", "drt_rt_mpi"):
            getattr(L, fn).argtypes = [vp, i32]
        L.drt_get_array.argtypes = [vp, i32, i32, dp, C.c_size_t]
        L.drt_set_array.argtypes = [vp, i32, i32, dp, C.c_size_t]
        L.drt_device_ptr.argtypes = [vp, i32, i32, C.POINTER(vp), C.POINTER(C.c_size_t)]
        L.drt_get_npix.argtypes = [vp, i32, ip, ip, ip]
        L.drt_get_src_cell.argtypes = [vp, ip, dp]
        L.drt_en_lim.argtypes = [vp]
        L.drt_en_lim.restype = C.c_double
        L.drt_iterations.argtypes = [vp]
        L.drt_cnflag.argtypes = [vp]
        L.drt_calc_sed.argtypes = [vp, dp]
        L.drt_get_counters.argtypes = [vp, C.POINTER(Counters)]
        L.drt_reset_counters.argtypes = [vp]
        L.drt_comm_unique_id.argtypes = [C.c_char_p]
        L.drt_comm_init.argtypes = [vp, i32, i32, C.c_char_p]
        L.drt_comm_destroy.argtypes = [vp]
        if hasattr(L, "drt_comm_selftest"):
            L.drt_comm_selftest.argtypes = [vp]
        if hasattr(L, "drt_get_sources"):  # absent from libraries of older revisions (A/B runs through DRT_LIB)
            L.drt_get_sources.argtypes = [vp, i32, ip, i32, ip]
            L.drt_arena_carve.argtypes = [C.c_size_t, i32, i32, ip, ip, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        if hasattr(L, "drt_mark_cost_reduced"):
            L.drt_mark_cost_reduced.argtypes = [vp]
        if hasattr(L, "drt_set_algorithm"):
            L.drt_set_algorithm.argtypes = [vp, i32]
        if hasattr(L, "drt_set_psel_av"):
            L.drt_set_psel_av.argtypes = [vp, i32]
        if hasattr(L, "drt_deal_by_cost"):
            L.drt_deal_by_cost.argtypes = [dp, C.c_int64, i32, i32, ip]
        if hasattr(L, "drt_plan_tasks"):
            L.drt_plan_tasks.argtypes = [dp, C.c_int64, i32, i32, i32, C.c_double, C.c_double, i32, C.c_double, C.c_double, i32, dp, C.c_double, ip, dp]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def arena_carve(arena_bytes: int, waves: int, lp: int):
    """drt_arena_carve: (cap, hcap, stride, offsets[10]) or None when the arena is too small. Needs no GPU."""
    cap, hcap, stride = C.c_int32(), C.c_int32(), C.c_uint64()
    off = (C.c_uint64 * 10)()
    rc = lib().drt_arena_carve(arena_bytes, waves, lp, C.byref(cap), C.byref(hcap), C.byref(stride), off)
    return None if rc != 0 else (cap.value, hcap.value, stride.value, list(off))


def plan_tasks(cost, waves: int, G: int, nbatch: int = 1, heavy_div: float = 8.0, deep_min: float = 4.0e6, measured: bool = True,
               f_last: float = 0.0, q_fill: float = 0.0, G_last: int = 0, q48=None, cap: float = 0.0):
    """drt_plan_tasks: (cls, number of explicit heavy tasks, limit). cls[k] = 0 light, -1 deep, else sectors per task. Needs no GPU."""
    cost = np.ascontiguousarray(cost, dtype=np.float64)
    cls = np.empty(cost.size, dtype=np.int32)
    lim = C.c_double()
    if q48 is not None:
        q48 = np.ascontiguousarray(q48, dtype=np.float64)
        assert q48.size == cost.size
    rc = lib().drt_plan_tasks(_dp(cost), cost.size, waves, G, nbatch, heavy_div, deep_min, 1 if measured else 0, f_last, q_fill, G_last,
                              _dp(q48) if q48 is not None else None, cap, cls.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(lim))
    if rc < 0:
        raise ValueError(f"drt_plan_tasks: {rc}")
    return cls, rc, lim.value


def fit_queue_prior(lum, q):
    """drt_fit_queue_prior: (a, b) of log q = a + b log lum, or None when there is nothing to fit. Needs no GPU."""
    lum = np.ascontiguousarray(lum, dtype=np.float64); q = np.ascontiguousarray(q, dtype=np.float64)
    assert lum.size == q.size
    L = lib()
    L.drt_fit_queue_prior.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    a, b = C.c_double(), C.c_double()
    rc = L.drt_fit_queue_prior(_dp(lum), _dp(q), lum.size, C.byref(a), C.byref(b))
    return None if rc != 0 else (a.value, b.value)


def deal_by_cost(cost, np_mpi: int, chunk: int = 64) -> np.ndarray:
    """drt_deal_by_cost: owner rank of each of len(cost) consecutive emitting cells. Needs no GPU."""
    cost = np.ascontiguousarray(cost, dtype=np.float64)
    owner = np.empty(cost.size, dtype=np.int32)
    rc = lib().drt_deal_by_cost(_dp(cost), cost.size, np_mpi, chunk, owner.ctypes.data_as(C.POINTER(C.c_int32)))
    if rc != 0:
        raise ValueError(f"drt_deal_by_cost: {rc}")
    return owner


class DartRayGPU:
    """One RT problem resident on one MI355X. `case` is a grids.Case."""

    def __init__(self, case, device: int = 0, np_mpi: int = 1, id_mpi: int = 0, arena_bytes: int = 0, waves: int = 0):
        self.L_ = lib()
        h = C.c_void_p()
        rc = self.L_.drt_create(C.byref(h), device)
        if rc != 0:
            raise DartRayError(f"drt_create failed ({rc}): a gfx950 device is required, there is no CPU fallback")
        self.h = h
        t = case.tree
        self.N, self.L = t.ncell, case.lnum
        self.nsrc, self.ndir = int(case.p_src_pos.shape[0]), int(case.dir_out.shape[0])
        if arena_bytes or waves:
            self._chk(self.L_.drt_configure(self.h, arena_bytes, waves))
        base = np.asarray(t.base, dtype=np.int32)
        cchild = np.ascontiguousarray(t.cchild, dtype=np.int32)
        lvl = np.ascontiguousarray(t.lvl, dtype=np.int32)
        cindex = np.ascontiguousarray(t.cindex, dtype=np.int64)
        ccoord = np.ascontiguousarray(t.ccoord, dtype=np.float64)
        csize = np.ascontiguousarray(t.csize, dtype=np.float64)
        ip = C.POINTER(C.c_int32)
        self._chk(self.L_.drt_set_grid(self.h, self.N, t.max_lvl, base.ctypes.data_as(ip), cchild.ctypes.data_as(ip),
                                       lvl.ctypes.data_as(ip), cindex.ctypes.data_as(C.POINTER(C.c_int64)),
                                       _dp(ccoord), _dp(csize)))
        from . import grids
        p = Params()
        p.kp_sca_max, p.bm_par, p.bm_par_sca, p.bm_par_max = case.kp_sca_max, case.bm_par, case.bm_par_sca, case.bm_par_max
        p.rad_lim, p.accuracy, p.conv_en_lim, p.cs, p.dist_obs = case.rad_lim, case.accuracy, case.conv_en_lim, grids.CS_PC, case.dist_obs
        for k in range(3):
            p.wall_on[k] = int(case.wall_on[k])
        wc = np.asarray(case.wall_coord, dtype=np.float64).ravel()
        for k in range(6):
            p.wall_coord[k] = float(wc[k])
        p.max_sca_iterations, p.np_mpi, p.id_mpi = case.max_sca_iterations, np_mpi, id_mpi
        self._chk(self.L_.drt_set_params(self.h, C.byref(p)))
        ksca = np.ascontiguousarray(case.ksca_norm, dtype=np.float64)
        gsca = np.ascontiguousarray(case.gsca, dtype=np.float64)
        self._chk(self.L_.drt_set_wavelengths(self.h, self.L, _dp(ksca), _dp(gsca)))
        ppos = np.ascontiguousarray(case.p_src_pos, dtype=np.float64)
        plum = np.ascontiguousarray(case.p_src_lum, dtype=np.float64)
        self._chk(self.L_.drt_set_point_sources(self.h, self.nsrc, _dp(ppos), _dp(plum)))
        dirs = np.ascontiguousarray(case.dir_out.T, dtype=np.float64)  # Fortran (ndir,2)
        self._chk(self.L_.drt_set_observers(self.h, self.ndir, _dp(dirs)))
        self.set_array("dens", np.ascontiguousarray(case.dens, dtype=np.float64))
        self.set_array("dens_stars", np.ascontiguousarray(case.dens_stars, dtype=np.float64))
        self.npix = self.npix_hp = self.kp = None
        self.device = device
        self.np_mpi, self.id_mpi = np_mpi, id_mpi
        self.group = None       # torch.distributed process group for rt_mpi when np_mpi > 1
        self.use_dist = False

    # ---- plumbing
    def _chk(self, rc):
        if rc != 0:
            msg = self.L_.drt_last_error(self.h)
            raise DartRayError(f"dartray_rt error {rc}: {msg.decode() if msg else ''}")

    def close(self):
        if getattr(self, "h", None):
            self.L_.drt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- the reference's driver-side entry points
    def rt_prepare(self, rt):
        self._chk(self.L_.drt_rt_prepare(self.h, RTT[rt] if isinstance(rt, str) else rt))
        if self.npix is None:
            a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
            self.npix, self.npix_hp, self.kp = [], [], []
            for il in range(self.L):
                self._chk(self.L_.drt_get_npix(self.h, il, C.byref(a), C.byref(b), C.byref(c)))
                self.npix.append(a.value); self.npix_hp.append(b.value); self.kp.append(c.value)

    def rt_loop(self, rt): self._chk(self.L_.drt_rt_loop(self.h, RTT[rt] if isinstance(rt, str) else rt))
    def rt_loop_iobs(self, rt): self._chk(self.L_.drt_rt_loop_iobs(self.h, RTT[rt] if isinstance(rt, str) else rt))
    def rt_mpi(self, rt):
        """rt_mpi (io_routines.f90:3667). One rank: the library's own kernels. Several ranks attached with
        attach_process_group(): torch.distributed all-reduces (RCCL) straight on the device buffers."""
        rt = RTT[rt] if isinstance(rt, str) else rt
        if not self.use_dist:
            self._chk(self.L_.drt_rt_mpi(self.h, rt))
            return
        from . import parallel
        parallel.rt_mpi(rt, self.mpi_fields(rt), self.group)

    def mpi_fields(self, rt):
        """The device buffers rt_mpi reduces for this stage, as zero-copy torch tensors."""
        rt = RTT[rt] if isinstance(rt, str) else rt
        fields = {"u_fest": self.device_tensor("u_fest"), "u_final": self.device_tensor("u_final"),
                  "src_cost": self.device_tensor("src_cost"), "src_cost_done": self.mark_cost_reduced}
        if rt != RTT["precalc_src"]:
            if getattr(self, "psel_on", False):   # on every rank or on none: the flag decides, not whether this rank had sources
                fields["psel"] = [self.device_tensor("psel_sum"), self.device_tensor("psel_cnt")]
            fields["lum_lost"] = self.device_tensor("lum_lost")
            fields["lum_lost_prev"] = self.device_tensor("lum_lost_prev")
            fields["scaspe"] = [self.device_tensor("scaspe", il) for il in range(self.L)]
        return fields

    def set_psel_av(self, on: bool = True):
        """print_psel_av: keep the average path of each source's rays per stage (get_array('psel_av'))."""
        self._chk(self.L_.drt_set_psel_av(self.h, 1 if on else 0))
        self.psel_on = bool(on)

    def set_algorithm(self, name: str):
        """'main' or 'i_obs' (drt_set_algorithm)."""
        self._chk(self.L_.drt_set_algorithm(self.h, {"main": 0, "i_obs": 1}[name]))

    def mark_cost_reduced(self):
        """DRT_A_SRC_COST has been summed over the ranks (drt_mark_cost_reduced)."""
        self._chk(self.L_.drt_mark_cost_reduced(self.h))

    def attach_process_group(self, group=None):
        """Use an initialised torch.distributed group (backend nccl = RCCL) for the stage-end reductions."""
        import torch.distributed as dist
        assert dist.is_initialized() and dist.get_world_size(group) == self.np_mpi
        self.group, self.use_dist = group, self.np_mpi > 1

    def device_tensor(self, which: str, il: int = 0):
        """Zero-copy torch view of a device field (drt_device_ptr)."""
        from . import parallel
        p, n = C.c_void_p(), C.c_size_t()
        self._chk(self.L_.drt_device_ptr(self.h, A[which], il, C.byref(p), C.byref(n)))
        return parallel.alias_device(p.value, n.value, self.device)

    def scatt_prepare(self) -> bool:
        """rt_prepare for rtt_scatt; returns cnflag."""
        self.rt_prepare("scatt")
        return bool(self.L_.drt_cnflag(self.h))

    def run_all(self, stages: int = 31, on_stage=None):
        """dartray_main's stage order (dartray_hub.f90:12-67); stage bits as in oracle/ref_driver.f90."""
        self.rt_prepare("start")
        if stages & 1:
            self.rt_loop("precalc_cell"); self.rt_loop("precalc_src"); self.rt_mpi("precalc_src")
            if on_stage: on_stage("precalc")
        self.rt_prepare("prep_part2")
        if stages & 2:
            self.rt_loop("dir_cell"); self.rt_loop("dir_src"); self.rt_mpi("dir_src")
            if on_stage: on_stage("direct")
        if stages & 4:
            self.rt_loop_iobs("i_obs_dir_cell"); self.rt_loop_iobs("i_obs_dir_src")
            if on_stage: on_stage("iobs_direct")
        if stages & 8:
            while not self.scatt_prepare():
                self.rt_loop("scatt"); self.rt_mpi("scatt")
                if on_stage: on_stage("scatt")
            if stages & 16:
                self.rt_prepare("i_obs"); self.rt_loop_iobs("i_obs")
                if on_stage: on_stage("iobs")

    # ---- module state
    def _count(self, which, il):
        NL = self.N * self.L
        return {"u_fest": NL, "u_final": NL, "lumcell": NL, "dens": NL, "dens_stars": NL, "lum_lost": self.L,
                "tot_rad_en": self.L, "lum_lost_prev": self.L, "src_cost": self.N, "psel_av": self.N + self.nsrc, "i_obs": self.ndir * (self.N + self.nsrc) * self.L,
                "scaspe": None, "scaspe_tot": None, "scaspe_prev": None}[which]

    def get_array(self, which: str, il: int = 0) -> np.ndarray:
        n = self._count(which, il)
        if n is None:
            n = self.N * self.npix[il]
        out = np.empty(n, dtype=np.float64)
        self._chk(self.L_.drt_get_array(self.h, A[which], il, _dp(out), n))
        if which in ("u_fest", "u_final", "lumcell", "dens", "dens_stars"):
            return out.reshape(self.N, self.L)
        if which in ("scaspe", "scaspe_tot", "scaspe_prev"):
            return out.reshape(self.N, self.npix[il])
        if which == "i_obs":
            return out.reshape(self.ndir, self.N + self.nsrc, self.L)
        return out

    def set_array(self, which: str, arr: np.ndarray, il: int = 0):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        self._chk(self.L_.drt_set_array(self.h, A[which], il, _dp(arr), arr.size))

    @property
    def u_fest(self): return self.get_array("u_fest")
    @property
    def u_final(self): return self.get_array("u_final")
    @property
    def lum_lost(self): return self.get_array("lum_lost")
    @property
    def i_obs(self): return self.get_array("i_obs")
    @property
    def en_lim(self): return self.L_.drt_en_lim(self.h)
    @property
    def iterations(self): return self.L_.drt_iterations(self.h)

    def scaspe(self, il): return self.get_array("scaspe", il)
    def scaspe_tot(self, il): return self.get_array("scaspe_tot", il)

    def src_cell(self):
        sc = np.empty(self.N, dtype=np.int32)
        pos = np.empty((max(self.nsrc, 1), 3), dtype=np.float64)
        self._chk(self.L_.drt_get_src_cell(self.h, sc.ctypes.data_as(C.POINTER(C.c_int32)), _dp(pos)))
        return sc, pos[: self.nsrc]

    def sed(self) -> np.ndarray:
        out = np.zeros((max(self.ndir, 1), self.L))
        self._chk(self.L_.drt_calc_sed(self.h, _dp(out)))
        return out[: self.ndir]

    def sources(self, rt) -> np.ndarray:
        """Cells this rank traces in rt_loop(rt) (drt_get_sources)."""
        rt = RTT[rt] if isinstance(rt, str) else rt
        n = C.c_int32()
        self._chk(self.L_.drt_get_sources(self.h, rt, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), dtype=np.int32)
        self._chk(self.L_.drt_get_sources(self.h, rt, out.ctypes.data_as(C.POINTER(C.c_int32)), n.value, C.byref(n)))
        return out[: n.value]

    def counters(self) -> Dict[str, float]:
        c = Counters()
        self._chk(self.L_.drt_get_counters(self.h, C.byref(c)))
        return {k: getattr(c, k) for k, _ in Counters._fields_}

    def reset_counters(self): self._chk(self.L_.drt_reset_counters(self.h))

    def comm_selftest(self):
        """drt_comm_selftest: the library's RCCL path on one rank (raises DartRayError when it fails)."""
        self._chk(self.L_.drt_comm_selftest(self.h))

    # ---- multi-GPU
    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(UNIQUE_ID_BYTES)
        if lib().drt_comm_unique_id(buf) != 0:
            raise DartRayError("drt_comm_unique_id failed (RCCL not found)")
        return buf.raw

    def comm_init(self, nranks: int, rank: int, uid: bytes):
        self._chk(self.L_.drt_comm_init(self.h, nranks, rank, uid))
