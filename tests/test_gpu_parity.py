"""Parity of the HIP path (through the C ABI) with the oracle on the same inputs, and with the
golden vectors of the reference itself. Everything here needs a real MI355X."""
import numpy as np
import pytest

import oracle_py as op
from conftest import GOLDEN_CASES, golden_scaspe_check

pytestmark = pytest.mark.gpu

# Tolerances (relative to the array maximum). The GPU path differs from the reference by the order
# of atomic floating-point sums and by <= 2 ulp in device sin/cos/acos/exp; north_star asks for 1e-4
# on SEDs, we hold the fields to 1e-9.
RTOL = 1e-9


def relmax(a, b):
    s = np.abs(b).max()
    return 0.0 if s == 0 else float(np.abs(np.asarray(a) - b).max() / s)


# Element-wise check: u_final and scaspe span many decades between cells near and far from the sources, and a
# tolerance relative to the array maximum would let a faint cell be wrong by a large factor. Every element above
# FLOOR of the maximum must agree to RTOL_ELEM relative to ITSELF (sums of ~1e2..1e5 positive terms in a different
# order: 1e-12 would do; the margin is for libm differences in exp).
RTOL_ELEM = 1e-7
FLOOR = 1e-12


def relelem(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    s = np.abs(b).max()
    if s == 0:
        return float(np.abs(a).max())
    m = np.abs(b) > FLOOR * s
    return float((np.abs(a[m] - b[m]) / np.abs(b[m])).max()) if m.any() else 0.0


@pytest.mark.parametrize("mapping", ["ray_major", "lane_per_wavelength"])
@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_gpu_matches_reference_golden(name, mapping, golden, pkg, monkeypatch):
    """Both lane mappings of the stage kernel: rt_stage_wide (lane = ray, the default for every number
    of wavelengths) and rt_stage_kernel (lane = (ray, wavelength), DRT_RAY_MAJOR=0) against the reference's own output."""
    z, path = golden(name)
    case = pkg.grids.read_case(path, name)
    if mapping == "lane_per_wavelength":
        monkeypatch.setenv("DRT_RAY_MAJOR", "0")
    g = pkg.DartRayGPU(case)
    g.rt_prepare("start")
    assert list(g.npix) == list(z["npix"]) and list(g.kp) == list(z["kp"])
    assert g.en_lim == pytest.approx(float(z["en_lim"]), rel=1e-15)
    sc, pos = g.src_cell()
    hosts = np.flatnonzero(sc >= 0)
    assert list(hosts) == list(z["src_cell_hosts"])
    g.rt_loop("precalc_cell"); g.rt_loop("precalc_src"); g.rt_mpi("precalc_src")
    assert relmax(g.u_fest, z["u_fest"]) <= RTOL and relelem(g.u_fest, z["u_fest"]) <= RTOL_ELEM
    g.rt_prepare("prep_part2")
    g.rt_loop("dir_cell"); g.rt_loop("dir_src"); g.rt_mpi("dir_src")
    assert relmax(g.u_final, z["u_final_dir"]) <= RTOL and relelem(g.u_final, z["u_final_dir"]) <= RTOL_ELEM
    assert relmax(g.lum_lost, z["lum_lost_dir"]) <= 1e-7
    for il in range(g.L):
        golden_scaspe_check(z, "scaspe_dir", il, g.scaspe(il), RTOL, name, RTOL_ELEM)
    if g.ndir:
        g.rt_loop_iobs("i_obs_dir_cell"); g.rt_loop_iobs("i_obs_dir_src")
        assert relmax(g.i_obs, z["i_obs_dir"]) <= RTOL and relelem(g.i_obs, z["i_obs_dir"]) <= RTOL_ELEM
        assert relmax(g.sed(), z["sed_dir"]) <= RTOL and relelem(g.sed(), z["sed_dir"]) <= RTOL_ELEM
    while not g.scatt_prepare():
        g.rt_loop("scatt"); g.rt_mpi("scatt")
    assert g.iterations == int(z["iterations"])
    assert relmax(g.u_final, z["u_final"]) <= RTOL and relelem(g.u_final, z["u_final"]) <= RTOL_ELEM
    assert relmax(g.lum_lost, z["lum_lost"]) <= 1e-7
    for il in range(g.L):
        golden_scaspe_check(z, "scaspe_tot", il, g.scaspe_tot(il), RTOL, name, RTOL_ELEM)
    if g.ndir:
        g.rt_prepare("i_obs"); g.rt_loop_iobs("i_obs")
        assert relmax(g.i_obs, z["i_obs"]) <= RTOL and relelem(g.i_obs, z["i_obs"]) <= RTOL_ELEM
        assert relmax(g.sed(), z["sed"]) <= 1e-9 and relelem(g.sed(), z["sed"]) <= RTOL_ELEM  # north_star: 1e-4
    g.close()


def _arena_for_cap(pkg, waves, lp, want_cap):
    """Smallest arena (in 64 KiB steps) that gives `waves` wavefronts queues of at least want_cap rays."""
    a = 64 << 10
    while True:
        r = pkg.host.arena_carve(a, waves, lp)
        if r is not None and r[0] >= want_cap:
            return a
        a += 64 << 10


@pytest.mark.parametrize("name", ["blobB", "blobD"])
def test_queue_overflow_is_undone_and_retried(name, golden, pkg):
    """An arena that gives 16 waves ray queues of 512 rays (256 waves: 32, below the minimum of 64): the cell-source stages overflow at 16, 8, ... sectors per
    task and the point-source stages at full width. Every overflowing launch must be undone (accumulators, counters, lum_lost,
    scaspe restored from the snapshot) and repeated with deeper queues - fewer sectors per task for cell sources,
    fewer waves for point sources - and the end result must still equal the reference's golden vectors."""
    z, path = golden(name)
    case = pkg.grids.read_case(path, name)
    lp = 1
    while lp < case.lnum:
        lp *= 2
    # room for 64-ray queues of 16 waves, 256 waves asked for: the library must also come down in waves by itself
    g = pkg.DartRayGPU(case, arena_bytes=_arena_for_cap(pkg, 16, lp, 512), waves=256)
    launches, retries = {}, {}
    g.rt_prepare("start")
    for rt in ("precalc_cell", "precalc_src"):
        g.rt_loop(rt); launches[rt] = g.counters()["launches"]; retries[rt] = g.counters()["retries"]
    g.rt_mpi("precalc_src")
    assert relmax(g.u_fest, z["u_fest"]) <= RTOL
    g.rt_prepare("prep_part2")
    for rt in ("dir_cell", "dir_src"):
        g.rt_loop(rt); launches[rt] = g.counters()["launches"]; retries[rt] = g.counters()["retries"]
    g.rt_mpi("dir_src")
    assert launches["dir_cell"] > 1, launches     # at least one undo + retry with fewer sectors per task
    assert retries["dir_cell"] >= 1 and sum(retries.values()) >= 2, retries   # and the undone attempts are counted (drt_counters.retries)
    assert relmax(g.u_final, z["u_final_dir"]) <= RTOL and relelem(g.u_final, z["u_final_dir"]) <= RTOL_ELEM
    assert relmax(g.lum_lost, z["lum_lost_dir"]) <= 1e-7
    for il in range(g.L):
        golden_scaspe_check(z, "scaspe_dir", il, g.scaspe(il), RTOL, name, RTOL_ELEM)
    while not g.scatt_prepare():
        g.rt_loop("scatt"); g.rt_mpi("scatt")
    assert g.iterations == int(z["iterations"])
    assert relmax(g.u_final, z["u_final"]) <= RTOL and relelem(g.u_final, z["u_final"]) <= RTOL_ELEM
    assert relmax(g.lum_lost, z["lum_lost"]) <= 1e-7
    for il in range(g.L):
        golden_scaspe_check(z, "scaspe_tot", il, g.scaspe_tot(il), RTOL, name, RTOL_ELEM)
    g.close()


def test_measured_cost_scheduling_changes_the_tasks_not_the_result(golden, pkg, monkeypatch):
    """The expected cost of a cell source - the wave time it took in the previous launch of the stage (DRT_A_SRC_COST) scaled
    with its luminosity, sqrt(luminosity) before anything was measured - cuts the expensive sources into smaller tasks
    (fewer sectors per task, listed in front) or runs them as sub-tree tasks like point sources. The sectors of a source are independent (rt_routines.f90:1356-1358) and the ladders below distinct
    nside-8 pixels never exchange rays (:2494), so the rays, intersections and deposits must stay EXACTLY those of the
    reference whatever the cut, and the fields within the usual tolerance. On this small grid the default limit
    (DRT_HEAVY_DIV = 16) already gives a mix of deep and heavy sources, a very fine one makes every source deep (the
    whole scattering stage then runs through sub-tree tasks), 0 switches the cutting off."""
    z, path = golden("trusti32")
    case = pkg.grids.read_case(path, "trusti32")
    seen = {}
    for div in ("16", "100000", "0"):
        monkeypatch.setenv("DRT_HEAVY_DIV", div)
        g = pkg.DartRayGPU(case)
        used = []
        g.rt_prepare("start")
        g.rt_loop("precalc_cell"); g.rt_loop("precalc_src"); g.rt_mpi("precalc_src")
        g.rt_prepare("prep_part2")
        g.rt_loop("dir_cell"); g.rt_loop("dir_src"); g.rt_mpi("dir_src")
        while not g.scatt_prepare():
            g.rt_loop("scatt")
            c = g.counters()
            used.append((int(c["heavy_tasks"]), int(c["deep_sources"]), int(c["sources"])))
            cost = g.get_array("src_cost")
            assert int((cost > 0).sum()) == int(c["sources"])     # every source of the launch has a measured cost
            g.rt_mpi("scatt")
        assert g.iterations == int(z["iterations"])
        c = g.counters()
        seen[div] = (int(c["rays"]), int(c["intersections"]), int(c["deposits"]))
        assert relmax(g.u_final, z["u_final"]) <= RTOL and relelem(g.u_final, z["u_final"]) <= RTOL_ELEM
        assert relmax(g.lum_lost, z["lum_lost"]) <= 1e-7
        for il in range(0, g.L, 5):
            golden_scaspe_check(z, "scaspe_tot", il, g.scaspe_tot(il), RTOL, "trusti32", RTOL_ELEM)
        if div == "100000":
            assert all(u[1] == u[2] for u in used[1:]), used      # all deep
        elif div == "0":
            assert all(u[:2] == (0, 0) for u in used), used
        else:
            assert any(0 < u[1] < u[2] and u[0] > 0 for u in used[1:]), used   # a mix
        g.close()
    assert seen["16"] == seen["0"] == seen["100000"], seen


def test_point_source_overflow_is_a_clean_capacity_error(golden, pkg, monkeypatch):
    """Sub-tree tasks of a point source with 4-ray queues (DRT_TEST_MAX_CAP, a test aid of carve_arena; with the 64 rays
    of the smallest real arena the sub-tree ladders of the test cases never overflow): the
    direct-light stage of the TRUST-I case overflows in every attempt. Each attempt must be undone, the stage must end
    in DRT_E_CAPACITY with the accumulators as they were - the failure mode of round 1 was a memory fault - and with
    full-depth queues the same context must then reproduce the golden result."""
    z, path = golden("slab1801")
    case = pkg.grids.read_case(path, "slab1801")
    g = pkg.DartRayGPU(case, waves=64)
    g.rt_prepare("start")
    g.rt_loop("precalc_cell"); g.rt_loop("precalc_src"); g.rt_mpi("precalc_src")
    g.rt_prepare("prep_part2")
    g.rt_loop("dir_cell")
    monkeypatch.setenv("DRT_TEST_MAX_CAP", "4")
    with pytest.raises(pkg.host.DartRayError, match="-5"):
        g.rt_loop("dir_src")
    assert g.counters()["launches"] > 2   # more than one attempt (phase 1 + phase 2 each)
    assert float(np.abs(g.u_final).max()) == 0.0 and float(np.abs(g.scaspe(0)).max()) == 0.0 and float(g.lum_lost.max()) == 0.0
    monkeypatch.delenv("DRT_TEST_MAX_CAP")
    g.rt_loop("dir_src"); g.rt_mpi("dir_src")
    assert relmax(g.u_final, z["u_final_dir"]) <= RTOL
    golden_scaspe_check(z, "scaspe_dir", 0, g.scaspe(0), RTOL, "slab1801", RTOL_ELEM)
    g.close()


def test_hopeless_arena_is_a_clean_capacity_error(golden, pkg):
    """One block of four waves with 64-ray queues cannot hold blobB's direct-light ladders even at one sector per task:
    the stage must come back as DRT_E_CAPACITY (-5) with the accumulators untouched - not as a memory fault - and the
    context must go on working once it is given room."""
    z, path = golden("blobB")
    case = pkg.grids.read_case(path, "blobB")
    g = pkg.DartRayGPU(case, arena_bytes=_arena_for_cap(pkg, 4, 4, 64), waves=4)
    g.rt_prepare("start")
    g.rt_loop("precalc_cell"); g.rt_loop("precalc_src"); g.rt_mpi("precalc_src")
    g.rt_prepare("prep_part2")
    with pytest.raises(pkg.host.DartRayError, match="-5"):
        g.rt_loop("dir_cell")
    assert float(np.abs(g.u_final).max()) == 0.0 and all(float(np.abs(g.scaspe(il)).max()) == 0.0 for il in range(g.L))
    g.L_.drt_configure(g.h, 1 << 30, 1024)
    g.rt_loop("dir_cell"); g.rt_loop("dir_src"); g.rt_mpi("dir_src")
    assert relmax(g.u_final, z["u_final_dir"]) <= RTOL
    g.close()


@pytest.mark.parametrize("name", ["blobA", "blobD", "trusti32"])
def test_gpu_counters_match_oracle(name, golden, pkg):
    """Intersections / rays / deposits are the reference's own definitions: the GPU
    path must count exactly what the sequential algorithm does - including on the 32-wavelength TRUST-I case,
    i.e. in the kernel instantiation bench.py times (rt_stage_wide<32,4>), and through one scattering iteration."""
    import os
    z, path = golden(name)
    case = pkg.grids.read_case(path, name)
    o = op.Oracle(path, nsrc=case.p_src_pos.shape[0], ndir=case.dir_out.shape[0])
    o.set_parallel(1, 0, min(8, os.cpu_count() or 1))
    g = pkg.DartRayGPU(case)
    g.rt_prepare("start")
    for rt in ("precalc_cell", "precalc_src"):
        o.reset_counters(); o.rt_loop(rt)
        g.reset_counters(); g.rt_loop(rt)
        co, cg = o.counters(), g.counters()
        assert (cg["intersections"], cg["rays"], cg["deposits"]) == (co["intersections"], co["rays"], co["deposits"]), rt
    g.rt_mpi("precalc_src")
    g.rt_prepare("prep_part2")
    for rt in ("dir_cell", "dir_src"):
        o.reset_counters(); o.rt_loop(rt)
        g.reset_counters(); g.rt_loop(rt)
        co, cg = o.counters(), g.counters()
        assert (cg["intersections"], cg["rays"], cg["deposits"]) == (co["intersections"], co["rays"], co["deposits"]), rt
    o.rt_mpi("dir_src"); g.rt_mpi("dir_src")
    assert not o.scatt_prepare() and not g.scatt_prepare()
    o.reset_counters(); o.rt_loop("scatt")
    g.reset_counters(); g.rt_loop("scatt")
    co, cg = o.counters(), g.counters()
    assert (cg["intersections"], cg["rays"], cg["deposits"]) == (co["intersections"], co["rays"], co["deposits"]), "scatt"
    # every (ray, wavelength) deposit leaves the wave inside an atomic instruction that carries at least one lane
    assert 0 < cg["atomic_instructions"] <= cg["scaspe_adds"] + cg["u_adds"] and cg["atomic_lanes"] >= cg["atomic_instructions"]
    g.close(); o.close()


@pytest.mark.parametrize("name,partition", [("blobB", "block"), ("blobD", "mod"), ("trusti32", "cyclic"), ("trusti32", "cost"), ("blobE", "cost"), ("trusti32", "cost-unsynced")])
def test_two_ranks_in_one_process_match_golden(name, partition, golden, pkg, monkeypatch):
    """The product's multi-rank path on ONE GPU: two library contexts (np_mpi = 2, id_mpi = 0 / 1) run every stage on
    their share of the work - cell sources partitioned by build_src_list (chunks of consecutive cells dealt by expected
    cost - the default, "cost" -, or round the ranks, one contiguous block per rank, or the reference's
    i = im*np_mpi + id_mpi with DRT_PARTITION=mod), the (source, sector) tasks of point sources dealt round the ranks -
    and rt_mpi's arithmetic (io_routines.f90:3667-3855: u_final - u_fest clamp, (np-1)/np lum_lost correction, sums)
    runs on zero-copy aliases of the contexts' device buffers with the all-reduce replaced by an in-process sum
    (parallel.rt_mpi_local; over RCCL the same step list runs through torch.distributed). Both ranks must end with the
    reference's single-rank result."""
    import torch
    from dartray_amd import parallel
    monkeypatch.setenv("DRT_PARTITION", partition)
    monkeypatch.setenv("DRT_PARTITION_CHUNK", "16")  # several chunks per rank on these small grids
    z, path = golden(name)
    case = pkg.grids.read_case(path, name)
    ranks = [pkg.DartRayGPU(case, np_mpi=2, id_mpi=r) for r in (0, 1)]
    one = pkg.DartRayGPU(case)
    for g in ranks + [one]:
        g.rt_prepare("start")

    def mpi(rt):
        torch.cuda.synchronize()
        fields = [g.mpi_fields(rt) for g in ranks]
        if partition == "cost-unsynced":   # a host that reduces the fields itself and knows nothing of DRT_A_SRC_COST: every
            for f in fields:               # rank keeps the costs of its own sources only, the deal must not use them
                f.pop("src_cost"); f.pop("src_cost_done")
        parallel.rt_mpi_local(pkg.host.RTT[rt], fields)
        torch.cuda.synchronize()

    def check_partition(rt):
        all_src = one.sources(rt)
        parts = [g.sources(rt) for g in ranks]
        if rt.endswith("_src"):     # point sources: every rank lists them all, the sector tasks are shared
            assert all(list(p) == list(all_src) for p in parts), rt
        else:                       # cell sources: a partition of the emitting leaves, balanced to one cell
            assert sorted(np.concatenate(parts)) == sorted(all_src) and len(set(parts[0]) & set(parts[1])) == 0, rt
            assert abs(len(parts[0]) - len(parts[1])) <= (1 if partition == "block" else 16 if partition == "cyclic" else len(all_src)), rt

    for rt in ("precalc_cell", "precalc_src"):
        check_partition(rt)
        for g in ranks:
            g.rt_loop(rt)
    mpi("precalc_src")
    for g in ranks:
        assert relmax(g.u_fest, z["u_fest"]) <= RTOL and relelem(g.u_fest, z["u_fest"]) <= RTOL_ELEM
        g.rt_prepare("prep_part2")
    one.rt_loop("precalc_cell"); one.rt_loop("precalc_src"); one.rt_mpi("precalc_src"); one.rt_prepare("prep_part2")
    for rt in ("dir_cell", "dir_src"):
        check_partition(rt)
        for g in ranks:
            g.rt_loop(rt)
    mpi("dir_src")
    for g in ranks:
        assert relmax(g.u_final, z["u_final_dir"]) <= RTOL and relelem(g.u_final, z["u_final_dir"]) <= RTOL_ELEM
        assert relmax(g.lum_lost, z["lum_lost_dir"]) <= 1e-7
        for il in range(g.L):
            golden_scaspe_check(z, "scaspe_dir", il, g.scaspe(il), RTOL, name, RTOL_ELEM)
    while True:
        cn = [g.scatt_prepare() for g in ranks]
        assert cn[0] == cn[1]
        if cn[0]:
            break
        # whatever the deal is made from (expected costs from the second iteration on), it is a partition of the emitting leaves
        parts = [g.sources("scatt") for g in ranks]
        emitting = np.flatnonzero((ranks[0].get_array("lumcell").sum(axis=1) > 0) & (case.tree.cchild == -1))
        assert sorted(np.concatenate(parts)) == sorted(emitting) and len(set(parts[0]) & set(parts[1])) == 0
        for g in ranks:
            g.rt_loop("scatt")
        mpi("scatt")
    for g in ranks:
        assert g.iterations == int(z["iterations"])
        assert relmax(g.u_final, z["u_final"]) <= RTOL and relelem(g.u_final, z["u_final"]) <= RTOL_ELEM
        assert relmax(g.lum_lost, z["lum_lost"]) <= 1e-7
        for il in range(g.L):
            golden_scaspe_check(z, "scaspe_tot", il, g.scaspe_tot(il), RTOL, name, RTOL_ELEM)
    # observer pass: the main rank only (rt_routines.f90:1483)
    if ranks[0].ndir:
        for g in ranks:
            g.rt_prepare("i_obs"); g.rt_loop_iobs("i_obs")
        assert float(np.abs(ranks[1].i_obs).max()) == 0.0
    for g in ranks + [one]:
        g.close()


@pytest.mark.parametrize("name,log_mb", [("blobD", 0), ("blobE", 0), ("trusti32", 0), ("trusti32", 64)])
def test_deposit_log_matches_golden(name, log_mb, golden, pkg, monkeypatch):
    """DRT_LOG=1 (rt_wide.hpp MODE 2 + rt_log.hpp): the tracing kernel appends one record per (ray, cell) that scatters and a
    reduce pass adds the Henyey-Greenstein rows of every cell once, instead of atomic row adds from inside the tracing
    kernel. Same rays, intersections and deposits; the fields agree with the reference as they do in the default mode - on
    7 wavelengths with kp 1 / -1 / 0 interleaved, on slot tables with more slots than lanes (kp 2) and on the 32 wavelengths
    of TRUST-I; with a 64 MiB log the stages do not fit at once: launches that overflow the log are undone and repeated in
    more batches (drt_counters.retries), few waves run (each holds a chunk of the log), and the result is the same."""
    monkeypatch.setenv("DRT_LOG", "1")
    if log_mb:
        monkeypatch.setenv("DRT_LOG_MB", str(log_mb))
    z, path = golden(name)
    case = pkg.grids.read_case(path, name)
    g = pkg.DartRayGPU(case)
    retries = 0
    g.rt_prepare("start")
    g.rt_loop("precalc_cell"); g.rt_loop("precalc_src"); g.rt_mpi("precalc_src")
    assert relmax(g.u_fest, z["u_fest"]) <= RTOL and relelem(g.u_fest, z["u_fest"]) <= RTOL_ELEM
    g.rt_prepare("prep_part2")
    for rt in ("dir_cell", "dir_src"):
        g.rt_loop(rt); retries += g.counters()["retries"]
    g.rt_mpi("dir_src")
    assert relmax(g.u_final, z["u_final_dir"]) <= RTOL and relelem(g.u_final, z["u_final_dir"]) <= RTOL_ELEM
    assert relmax(g.lum_lost, z["lum_lost_dir"]) <= 1e-7
    for il in range(g.L):
        golden_scaspe_check(z, "scaspe_dir", il, g.scaspe(il), RTOL, name, RTOL_ELEM)
    if g.ndir:
        g.rt_loop_iobs("i_obs_dir_cell"); g.rt_loop_iobs("i_obs_dir_src")
    while not g.scatt_prepare():
        g.rt_loop("scatt"); retries += g.counters()["retries"]; g.rt_mpi("scatt")
    assert g.iterations == int(z["iterations"])
    assert relmax(g.u_final, z["u_final"]) <= RTOL and relelem(g.u_final, z["u_final"]) <= RTOL_ELEM
    assert relmax(g.lum_lost, z["lum_lost"]) <= 1e-7
    for il in range(g.L):
        golden_scaspe_check(z, "scaspe_tot", il, g.scaspe_tot(il), RTOL, name, RTOL_ELEM)
    if g.ndir:
        g.rt_prepare("i_obs"); g.rt_loop_iobs("i_obs")
        assert relmax(g.sed(), z["sed"]) <= 1e-9 and relelem(g.sed(), z["sed"]) <= RTOL_ELEM
    if log_mb:
        assert retries >= 1, retries    # the small log overflowed at least once and the stage was repeated in more batches
    g.close()


def test_in_library_rccl_path_on_one_rank(golden, pkg):
    """drt_comm_init / drt_rt_mpi reach RCCL through dlopen; with one GPU per box that path had never run. drt_comm_selftest
    runs all of it on one rank: library found, ncclGetUniqueId, ncclCommInitRank (id by value), ncclAllReduce(double, sum) in
    place on a device buffer on the library's stream, result unchanged, ncclCommDestroy."""
    z, path = golden("cube65")
    case = pkg.grids.read_case(path, "cube65")
    g = pkg.DartRayGPU(case)
    g.comm_selftest()
    g.close()


def test_rt_mpi_without_communicator_is_an_error(golden, pkg):
    """np_mpi > 1 set through drt_set_params but no communicator of that size (drt_comm_init): the fields of the rank
    are partial sums, drt_rt_mpi must refuse (DRT_E_COMM = -6) instead of passing them on; and a communicator whose
    size or rank contradicts np_mpi / id_mpi is refused at drt_comm_init."""
    z, path = golden("cube65")
    case = pkg.grids.read_case(path, "cube65")
    g = pkg.DartRayGPU(case, np_mpi=2, id_mpi=1)
    g.rt_prepare("start")
    g.rt_loop("precalc_cell"); g.rt_loop("precalc_src")
    with pytest.raises(pkg.host.DartRayError, match="-6"):
        g.rt_mpi("precalc_src")
    with pytest.raises(pkg.host.DartRayError, match="-2"):
        g.comm_init(4, 1, b"\0" * 128)
    g.close()


def test_energy_budget_slab(pkg):
    """Known-answer check (SURVEY.md section 8c ii): with albedo 0 nothing is scattered; what the slab
    absorbs is below the source luminosity and the luminosity written off as lum_lost stays under the
    1 % the reference warns about (io_routines.f90:2211-2222)."""
    case = pkg.grids.slab_case(tau_z=0.5, max_lvl=2, albedo=0.0, g=0.0, ndir=1)
    g = pkg.DartRayGPU(case)
    g.run_all(3)
    u = g.u_final[:, 0]
    vol = case.tree.csize ** 3
    absorbed = float((u * case.dens[:, 0] * pkg.grids.CS_PC * vol).sum())
    lost = float(g.lum_lost[0])
    lum = float(case.p_src_lum[0, 0])
    assert 0 < absorbed < lum
    assert lost / lum < 0.01
    g.close()


def test_fortran_driver_matches_golden(golden, pkg):
    """The Fortran host (ISO_C_BINDING facade + dartray_main state machine,
    dart-ray_amd/fortran/) drives the library through the same stage order as dartray_hub.f90 and
    reproduces the reference's golden vectors."""
    import os
    import subprocess
    import tempfile
    exe = os.path.join(os.path.dirname(pkg.__file__), "fortran", "dartray_gpu_driver")
    if not os.path.exists(exe):
        pytest.skip("Fortran driver not built (amdflang missing)")
    z, path = golden("blobA")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "f")
        r = subprocess.run([exe, path, out], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        got = op.read_ref_output(out + ".out")
    assert got["iterations"] == int(z["iterations"])
    for k in ("u_fest", "u_final_dir", "u_final", "i_obs_dir", "i_obs", "sed_dir", "sed"):
        assert relmax(got[k], z[k]) <= RTOL, k
    for k in ("lum_lost_dir", "lum_lost"):
        assert relmax(got[k], z[k]) <= 1e-7, k
    for il in range(got["L"]):
        golden_scaspe_check(z, "scaspe_dir", il, got["scaspe_dir"][il], RTOL, "fortran")
        golden_scaspe_check(z, "scaspe_tot", il, got["scaspe_tot"][il], RTOL, "fortran")


def _baseline_config_cases(grids):
    """Small versions of BASELINE configs 3-5: MAGTAR shell (base 10,3, one point
    source), GALAXY example (27-ary tree, every emitting leaf a source, z walls), N-body (irregular tree)."""
    return {
        "magtar_shell": lambda: grids.shell_case(max_lvl=2, nlambda=3, ndir=2),
        "galaxy": lambda: grids.galaxy_case(max_lvl=3, nlambda=3, max_sca_iterations=2),
        "nbody": lambda: grids.nbody_case(max_lvl=3, nlambda=2, max_sca_iterations=2),
    }


@pytest.mark.parametrize("name", ["magtar_shell", "galaxy", "nbody"])
def test_gpu_matches_oracle_on_baseline_configs(name, pkg, tmp_path):
    """The whole dartray_main sequence (precalculation, direct light, scattering iterations, observer
    passes) on the geometries of BASELINE configs 3-5 at sizes the oracle finishes in seconds: same
    number of scattering iterations, same fields and SEDs (tolerance 1e-9 of the array maximum; the oracle
    runs its sources on 8 threads, so its own sums are ordered differently from run to run)."""
    import os
    case = _baseline_config_cases(pkg.grids)[name]()
    path = str(tmp_path / "case.bin")
    case.write(path)
    nsrc, ndir = case.p_src_pos.shape[0], case.dir_out.shape[0]
    o = op.Oracle(path, nsrc=nsrc, ndir=ndir)
    o.set_parallel(1, 0, min(8, os.cpu_count() or 1))
    o.run_all(31)
    g = pkg.DartRayGPU(case)
    g.run_all(31)
    assert g.iterations == o.iterations
    assert relmax(g.u_fest, o.u_fest) <= RTOL
    assert relmax(g.u_final, o.u_final) <= RTOL
    assert relmax(g.lum_lost, o.lum_lost) <= 1e-7
    for il in range(g.L):
        assert relmax(g.scaspe_tot(il), o.scaspe_tot(il)) <= RTOL
    assert relmax(g.i_obs, o.i_obs[:ndir]) <= RTOL
    assert relmax(g.sed(), o.sed()[:ndir]) <= 1e-9   # north_star: 1e-4
    g.close(); o.close()


@pytest.mark.parametrize("name", ["cube65", "blobB"])
def test_i_obs_algorithm_matches_reference(name, golden, pkg):
    """dartray_i_obs (dartray_hub.f90:125-168) through the C ABI: a main run leaves scaspe_tot; a second context with
    OTHER observer directions and drt_set_algorithm(DRT_RTA_I_OBS) takes it over (drt_set_array, the HEALPix part is what
    counts) and runs the three observer passes. Fixture: the unmodified reference modules driven through this sequence."""
    import copy
    import os
    z, path = golden(name)
    a = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", name + "_iobsalg.npz")))
    case = pkg.grids.read_case(path, name)
    main = pkg.DartRayGPU(case)
    main.run_all(31)
    assert relmax(main.i_obs, z["i_obs"]) <= RTOL
    tot = [main.scaspe_tot(il) for il in range(main.L)]
    main.close()
    case_b = copy.copy(case)
    case_b.dir_out = np.ascontiguousarray(a["dir_alg"])
    g = pkg.DartRayGPU(case_b)
    g.set_algorithm("i_obs")
    g.rt_prepare("start")
    g.rt_prepare("prep_part2")
    with pytest.raises(pkg.host.DartRayError):
        g.rt_prepare("i_obs")                      # no scaspe_tot yet
    for il in range(g.L):
        t = tot[il].copy()
        t[:, g.npix_hp[il]:] = 0.0                 # the observer slots of the main run mean nothing here
        g.set_array("scaspe_tot", t, il)
    g.rt_loop_iobs("i_obs_dir_src"); g.rt_loop_iobs("i_obs_dir_cell")
    assert relmax(g.i_obs, a["i_obs_alg_dir"]) <= RTOL
    assert relmax(g.sed(), a["sed_alg_dir"]) <= 1e-9
    g.rt_prepare("i_obs"); g.rt_loop_iobs("i_obs")
    assert relmax(g.i_obs, a["i_obs_alg"]) <= RTOL and relelem(g.i_obs, a["i_obs_alg"]) <= RTOL_ELEM
    assert relmax(g.sed(), a["sed_alg"]) <= 1e-9
    g.close()


@pytest.mark.parametrize("name", ["cube65", "blobB"])
def test_psel_av_matches_reference(name, golden, pkg):
    """psel_av_arr (print_psel_av, rt_routines.f90:1456-1465): with drt_set_psel_av the library keeps, per rt_loop, the
    average path of the rays of every source that ended (given up below en_lim or at a wall, or leaving the model) -
    whatever way the source's sectors were cut into tasks. Fixture from the unmodified reference modules; the same
    sources have rays, and the averages agree to the accuracy of a sum taken in another order."""
    import os
    z, path = golden(name)
    ref = np.load(os.path.join(os.path.dirname(__file__), "golden", name + "_pselav.npz"))["psel_av"]
    case = pkg.grids.read_case(path, name)
    g = pkg.DartRayGPU(case)
    with pytest.raises(pkg.host.DartRayError):
        g.get_array("psel_av")                    # not enabled
    g.set_psel_av(True)
    g.rt_prepare("start")
    g.rt_loop("precalc_cell"); g.rt_loop("precalc_src"); g.rt_mpi("precalc_src")
    g.rt_prepare("prep_part2")
    g.rt_loop("dir_cell"); g.rt_loop("dir_src"); g.rt_mpi("dir_src")
    rows = [g.get_array("psel_av")]     # cells and point sources of the direct light share row 0
    while not g.scatt_prepare():
        g.rt_loop("scatt"); rows.append(g.get_array("psel_av")); g.rt_mpi("scatt")
    got = np.array(rows)
    assert got.shape == ref.shape
    assert np.array_equal(got > 0, ref > 0)
    assert relmax(got, ref) <= 1e-12 and relelem(got, ref) <= 1e-10
    assert relmax(g.u_final, z["u_final"]) <= RTOL     # and the run itself is what it was
    g.close()


def test_psel_av_with_a_rank_that_has_no_cell_sources(golden, pkg, monkeypatch):
    """Two ranks, psel_av on, and a deal that leaves rank 1 without a single cell source (one chunk holds them all): that rank
    returns early from drt_rt_loop, yet it must own zeroed psel sums and counts - rt_mpi reduces them on EVERY rank (a rank that
    skipped the two collectives would hang RCCL; one that kept the previous stage's sums would add them to everybody's) - and end
    with the same averages and fields as the rank that traced everything (reduce_psel_av_arr, io_routines.f90:3859-3881)."""
    import os
    import torch
    from dartray_amd import parallel
    monkeypatch.setenv("DRT_PARTITION", "cyclic")
    monkeypatch.setenv("DRT_PARTITION_CHUNK", "1000000")
    name = "blobB"
    z, path = golden(name)
    ref = np.load(os.path.join(os.path.dirname(__file__), "golden", name + "_pselav.npz"))["psel_av"]
    case = pkg.grids.read_case(path, name)
    ranks = [pkg.DartRayGPU(case, np_mpi=2, id_mpi=r) for r in (0, 1)]
    for g in ranks:
        g.set_psel_av(True)
        g.rt_prepare("start")
    assert len(ranks[1].sources("precalc_cell")) == 0 and len(ranks[0].sources("precalc_cell")) > 0

    def mpi(rt):
        torch.cuda.synchronize()
        fields = [g.mpi_fields(rt) for g in ranks]
        if rt != "precalc_src":
            assert all(len(f["psel"]) == 2 for f in fields)   # both ranks bring both arrays to the reduction
        parallel.rt_mpi_local(pkg.host.RTT[rt], fields)
        torch.cuda.synchronize()

    for rt in ("precalc_cell", "precalc_src"):
        for g in ranks:
            g.rt_loop(rt)
    mpi("precalc_src")
    for g in ranks:
        g.rt_prepare("prep_part2")
    for rt in ("dir_cell", "dir_src"):
        for g in ranks:
            g.rt_loop(rt)
    mpi("dir_src")
    rows = [[g.get_array("psel_av")] for g in ranks]
    while True:
        cn = [g.scatt_prepare() for g in ranks]
        assert cn[0] == cn[1]
        if cn[0]:
            break
        assert len(ranks[1].sources("scatt")) == 0
        for g in ranks:
            g.rt_loop("scatt")
        mpi("scatt")
        for r, g in zip(rows, ranks):
            r.append(g.get_array("psel_av"))
    for r, g in zip(rows, ranks):
        got = np.array(r)
        assert got.shape == ref.shape
        assert np.array_equal(got > 0, ref > 0)
        assert relmax(got, ref) <= 1e-12 and relelem(got, ref) <= 1e-10
        assert relmax(g.u_final, z["u_final"]) <= RTOL
        g.close()


def test_fortran_main_prints_psel_av(pkg, tmp_path):
    """print_psel_av = .TRUE. in dartray_gpu_main: 'grid_<label>_psel_av_part2.h5' (1, 1, sources) after the direct light and
    'grid_<label>_psel_av.h5' (1, iterations, sources) at the end (print_psel_av_arr, io_routines.f90:1337-1361), with the
    rows a run through the C ABI collects stage by stage."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(pkg.__file__), "fortran", "dartray_gpu_main")
    if not os.path.exists(exe) or not os.path.exists(pkg.io.LIB_PATH):
        pytest.skip("Fortran main program / HDF5 library not built")
    idx = [8, 14]
    lambdas = [float(pkg.grids.TRUSTI_TABLE[i, 0]) for i in idx]
    case = pkg.grids.trusti_case(tau_z=1.0, max_lvl=2, wavelengths=idx, max_sca_iterations=2)
    root = str(tmp_path)
    inp = pkg.io.write_reference_inputs(case, lambdas, root, label="slab", print_psel_av=True)
    out = subprocess.run([exe, inp], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    g = pkg.DartRayGPU(case)
    g.set_psel_av(True)
    g.rt_prepare("start")
    g.rt_loop("precalc_cell"); g.rt_loop("precalc_src"); g.rt_mpi("precalc_src")
    g.rt_prepare("prep_part2")
    g.rt_loop("dir_cell"); g.rt_loop("dir_src"); g.rt_mpi("dir_src")
    rows = [g.get_array("psel_av")]
    while not g.scatt_prepare():
        g.rt_loop("scatt"); g.rt_mpi("scatt"); rows.append(g.get_array("psel_av"))
    rows = np.array(rows)
    M = g.N + g.nsrc
    assert g.iterations == rows.shape[0]
    f = lambda what: os.path.join(root, "RUNS", pkg.io.filename("slab", -1.0, what))
    p2 = pkg.io.read_array(f("psel_av_part2"), "psel_av_arr", [1, 1, M])
    assert relmax(p2.reshape(M), rows[0]) <= 1e-12
    fin = pkg.io.read_array(f("psel_av"), "psel_av_arr", [1, g.iterations, M])      # C order: [M][iterations][1]
    assert relmax(fin.reshape(M, g.iterations).T, rows) <= 1e-12
    assert (rows[0] > 0).sum() == 1 and rows[0][g.N] > 0      # the direct light comes from the star alone
    g.close()


def test_fortran_main_runs_the_i_obs_algorithm_on_the_files_of_a_main_run(pkg, tmp_path):
    """rt_algorithm = 'i_obs' in dartray_gpu_main (the reference's dartray_i_obs.f90 / dartray_hub.f90:125-168): a main run
    leaves its scaspe_tot files (print_scaspe_tot); a second run with other observer directions and
    label_model_out_i_obs reads their HEALPix rows back (more rows per cell in the file than this run has: the main
    run's own directions), runs the three observer passes and writes 'grid_<label>_..._int.h5' (reset_filenames). Its
    arrays must equal what the same sequence gives through the C ABI (checked against the reference's fixture in
    test_i_obs_algorithm_matches_reference)."""
    import copy
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(pkg.__file__), "fortran", "dartray_gpu_main")
    if not os.path.exists(exe) or not os.path.exists(pkg.io.LIB_PATH):
        pytest.skip("Fortran main program / HDF5 library not built")
    idx = [8, 14, 27]       # HG with 55 and 19 slots and an isotropic wavelength (nside_sca 2, 1, 0 in the files)
    lambdas = [float(pkg.grids.TRUSTI_TABLE[i, 0]) for i in idx]
    case = pkg.grids.trusti_case(tau_z=1.0, max_lvl=2, wavelengths=idx, max_sca_iterations=2)
    root = str(tmp_path)
    inp = pkg.io.write_reference_inputs(case, lambdas, root, label="slab")
    out = subprocess.run([exe, inp], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    # other directions, fewer of them than the main run had
    new_dirs = np.array([[0.3 + 0.5 * th, 0.4 + 0.5 * ph] for th, ph in case.dir_out[:3]])
    case_b = copy.copy(case)
    case_b.dir_out = new_dirs
    inp_b = pkg.io.write_reference_inputs(case_b, lambdas, root, label="slab", rt_algorithm="i_obs", label_i_obs="newdirs",
                                          input_name="input_iobs.in", dir_out_name="dir_out_iobs.dat")
    out = subprocess.run([exe, inp_b], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "i_obs algorithm" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
    # the same through the C ABI
    main = pkg.DartRayGPU(case)
    main.run_all(31)
    tot = [main.scaspe_tot(il) for il in range(main.L)]
    main.close()
    g = pkg.DartRayGPU(case_b)
    g.set_algorithm("i_obs")
    g.rt_prepare("start"); g.rt_prepare("prep_part2")
    for il in range(g.L):
        t = np.zeros((g.N, g.npix[il]))
        t[:, :g.npix_hp[il]] = tot[il][:, :g.npix_hp[il]]
        g.set_array("scaspe_tot", t, il)
    g.rt_loop_iobs("i_obs_dir_src"); g.rt_loop_iobs("i_obs_dir_cell")
    io_dir, sed_dir = g.i_obs.copy(), g.sed().copy()
    g.rt_prepare("i_obs"); g.rt_loop_iobs("i_obs")
    N, nsrc, ndir = g.N, g.nsrc, 3
    f = lambda lam, what: os.path.join(root, "RUNS", pkg.io.filename("newdirs", lam, what, "_int"))
    for il, lam in enumerate(lambdas):
        assert relmax(pkg.io.read_array(f(lam, "i_obs"), "i_obs", [N + nsrc, ndir]), g.i_obs[:, :, il]) <= RTOL, il
        assert relmax(pkg.io.read_array(f(lam, "i_obs_part2"), "i_obs", [N + nsrc, ndir]), io_dir[:, :, il]) <= RTOL, il
    assert relmax(pkg.io.read_array(f(-1.0, "sed"), "sed_arr", [g.L, ndir]), g.sed()) <= 1e-9
    assert relmax(pkg.io.read_array(f(-1.0, "sed_dir"), "sed_arr_dir", [g.L, ndir]), sed_dir) <= 1e-9
    assert float((g.i_obs - io_dir).max()) > 0      # there is scattered light in it
    g.close()


@pytest.mark.parametrize("mapping", ["ray_major", "lane_per_wavelength"])
def test_more_than_32_wavelengths_match_the_oracle(mapping, pkg, tmp_path, monkeypatch):
    """40 wavelengths per ray (the 32 of the TRUST-I list and eight of them again): the <64> instantiations of both
    kernels - 64-bit wavelength masks, two 32-bit halves in the wave-OR of the live wavelengths, ten operand chunks, rows
    of 64 doubles in the ray queues - which no BASELINE config reaches (they carry at most 32). Whole run against the C
    oracle on the 1 801-cell slab: same counts, same number of scattering iterations, fields to 1e-9."""
    import os
    monkeypatch.setenv("DRT_RAY_MAJOR", "1" if mapping == "ray_major" else "0")
    wl = list(range(32)) + [0, 5, 9, 13, 17, 21, 25, 29]
    case = pkg.grids.trusti_case(tau_z=1.0, max_lvl=2, wavelengths=wl, max_sca_iterations=2)
    assert case.lnum == 40
    path = str(tmp_path / "case.bin")
    case.write(path)
    nsrc, ndir = case.p_src_pos.shape[0], case.dir_out.shape[0]
    o = op.Oracle(path, nsrc=nsrc, ndir=ndir)
    o.set_parallel(1, 0, min(8, os.cpu_count() or 1))
    g = pkg.DartRayGPU(case)
    g.rt_prepare("start")
    for rt in ("precalc_cell", "precalc_src"):
        o.rt_loop(rt); g.rt_loop(rt)
    o.rt_mpi("precalc_src"); g.rt_mpi("precalc_src")
    g.rt_prepare("prep_part2")
    for rt in ("dir_cell", "dir_src"):
        o.reset_counters(); o.rt_loop(rt)
        g.reset_counters(); g.rt_loop(rt)
        co, cg = o.counters(), g.counters()
        assert (cg["intersections"], cg["rays"], cg["deposits"]) == (co["intersections"], co["rays"], co["deposits"]), rt
    o.rt_mpi("dir_src"); g.rt_mpi("dir_src")
    while True:
        co_, cg_ = o.scatt_prepare(), g.scatt_prepare()
        assert co_ == cg_
        if cg_:
            break
        o.reset_counters(); o.rt_loop("scatt")
        g.reset_counters(); g.rt_loop("scatt")
        co, cg = o.counters(), g.counters()
        assert (cg["intersections"], cg["rays"], cg["deposits"]) == (co["intersections"], co["rays"], co["deposits"]), "scatt"
        o.rt_mpi("scatt"); g.rt_mpi("scatt")
    assert g.iterations == o.iterations
    assert relmax(g.u_final, o.u_final) <= RTOL and relelem(g.u_final, o.u_final) <= RTOL_ELEM
    assert relmax(g.lum_lost, o.lum_lost) <= 1e-7
    for il in range(g.L):
        assert relmax(g.scaspe_tot(il), o.scaspe_tot(il)) <= RTOL, il
    # the repeated wavelengths are copies of their originals
    for k, il in enumerate([0, 5, 9, 13, 17, 21, 25, 29]):
        assert relmax(g.u_final[:, 32 + k], g.u_final[:, il]) <= 1e-12
    g.close(); o.close()


def test_fortran_main_on_reference_style_inputs(pkg, tmp_path):
    """dartray_gpu_main reads what the reference's programs read (input namelists, wavelength list, HDF5 main
    and lambda grids, dir_out / point-source files, effective-opacity table) and writes make_output's HDF5 files
    (io_routines.f90:1743); the fields in those files equal the ones a direct run through the C ABI produces."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(pkg.__file__), "fortran", "dartray_gpu_main")
    if not os.path.exists(exe) or not os.path.exists(pkg.io.LIB_PATH):
        pytest.skip("Fortran main program / HDF5 library not built")
    lambdas = [0.443, 1.0, 2.2]
    case = pkg.grids.blob_case(base=(4, 2), max_lvl=2, nlambda=3, walls=True, max_sca_iterations=2)
    inp = pkg.io.write_reference_inputs(case, lambdas, str(tmp_path), label="blob")
    out = subprocess.run([exe, inp], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    g = pkg.DartRayGPU(case)
    g.run_all(31)
    N, nsrc, ndir = g.N, g.nsrc, g.ndir
    rd = str(tmp_path / "RUNS")
    f = lambda lam, what: os.path.join(rd, pkg.io.filename("blob", lam, what))
    for il, lam in enumerate(lambdas):
        assert relmax(pkg.io.read_array(f(lam, "ufield"), "u_final", [N]), g.u_final[:, il]) <= RTOL
        assert relmax(pkg.io.read_array(f(lam, "scaspe_tot"), "scaspe_tot", [g.npix[il], N]), g.scaspe_tot(il)) <= RTOL
        assert pkg.io.read_int(f(lam, "scaspe_tot"), "nside_sca") == (2 ** g.kp[il] if g.kp[il] >= 0 else 0)   # Fortran integer 2**kp
        assert relmax(pkg.io.read_array(f(lam, "i_obs"), "i_obs", [N + nsrc, ndir]), g.i_obs[:, :, il]) <= RTOL
        assert os.path.exists(f(lam, "u_fest_part1")) and os.path.exists(f(lam, "ufield_part2")) and os.path.exists(f(lam, "scaspe_part2"))
    assert relmax(pkg.io.read_array(f(-1.0, "lum_lost"), "lum_lost", [g.L]), g.lum_lost) <= 1e-7
    assert relmax(pkg.io.read_array(f(-1.0, "sed"), "sed_arr", [g.L, ndir]), g.sed()) <= 1e-9
    g.close()


def test_fortran_main_scales_the_main_grid_like_scale_dens_arr(pkg, tmp_path):
    """The shipped TRUST-I input (TRUSTI_GRIDS/input_trustI_tau0.01_LD.in) has no lambda grids: grid_initialize scales the
    main grid's dens with kext(lambda)/kext(lambda_ref) from the effective-opacity table (scale_dens_arr,
    rt_routines.f90:5018-5071). dartray_gpu_main on such an input set (use_lambda_grid unset, opacity table carrying the
    extinction curve) must give the fields of the run that reads one grid per wavelength and of a direct run through the
    C ABI; with lambda grids the 5E-3 consistency check of scale_dens_arr is exercised on the way."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(pkg.__file__), "fortran", "dartray_gpu_main")
    if not os.path.exists(exe) or not os.path.exists(pkg.io.LIB_PATH):
        pytest.skip("Fortran main program / HDF5 library not built")
    idx = [8, 11, 14]
    lambdas = [float(pkg.grids.TRUSTI_TABLE[i, 0]) for i in idx]
    case = pkg.grids.trusti_case(tau_z=1.0, max_lvl=2, wavelengths=idx, max_sca_iterations=2)
    g = pkg.DartRayGPU(case)
    g.run_all(31)
    N, nsrc, ndir = g.N, g.nsrc, g.ndir
    for sub, lg in (("main_only", False), ("lambda_grids", True)):
        root = str(tmp_path / sub)
        os.makedirs(root)
        # as in the shipped input, the reference wavelength (1 um) is not the first of the list
        inp = pkg.io.write_reference_inputs(case, lambdas, root, label="slab", use_lambda_grid=lg, iref=1)
        out = subprocess.run([exe, inp], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
        f = lambda lam, what: os.path.join(root, "RUNS", pkg.io.filename("slab", lam, what))
        for il, lam in enumerate(lambdas):
            assert relmax(pkg.io.read_array(f(lam, "ufield"), "u_final", [N]), g.u_final[:, il]) <= RTOL, (sub, il)
            assert relmax(pkg.io.read_array(f(lam, "scaspe_tot"), "scaspe_tot", [g.npix[il], N]), g.scaspe_tot(il)) <= RTOL, (sub, il)
        assert relmax(pkg.io.read_array(f(-1.0, "sed"), "sed_arr", [g.L, ndir]), g.sed()) <= 1e-9
    # a lambda grid that contradicts the opacity table by 1 % is refused, as in the reference
    bad = case.dens.copy(); bad[:, 2] *= 1.01
    pkg.io.write_lambda_grid(os.path.join(root, "GRIDS", pkg.io.lambda_grid_filename("slab", lambdas[2])), bad[:, 2], case.dens_stars[:, 2])
    out = subprocess.run([exe, inp], capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "scale_dens_arr" in out.stdout
    g.close()


def test_fortran_main_restarts_from_the_files_of_an_earlier_run(pkg, tmp_path):
    """Stage restart (rt_prepare, dartray_hub.f90:397-417; check_files, io_routines.f90:893; read_output :1900): with
    restore_file_mpi set, a run that finds the part-2 files (u_final, scaspe, i_obs, lum_lost after the direct light)
    starts at the scattering iterations, one that finds only the part-1 files (u_fest) starts at the direct light, and
    both end with the fields of the uninterrupted run; a run that finds final output files stops before it overwrites
    them."""
    import glob
    import os
    import shutil
    import subprocess
    exe = os.path.join(os.path.dirname(pkg.__file__), "fortran", "dartray_gpu_main")
    if not os.path.exists(exe) or not os.path.exists(pkg.io.LIB_PATH):
        pytest.skip("Fortran main program / HDF5 library not built")
    idx = [8, 14]
    lambdas = [float(pkg.grids.TRUSTI_TABLE[i, 0]) for i in idx]
    case = pkg.grids.trusti_case(tau_z=1.0, max_lvl=2, wavelengths=idx, max_sca_iterations=2)
    N, ndir, L = case.tree.ncell, case.dir_out.shape[0], case.lnum

    def run(sub, restore, keep=()):
        root = str(tmp_path / sub)
        inp = pkg.io.write_reference_inputs(case, lambdas, root, label="slab", restore_file_mpi=restore)
        for pat in keep:
            for f in glob.glob(os.path.join(str(tmp_path / "full"), "RUNS", pat)):
                shutil.copy(f, os.path.join(root, "RUNS"))
        out = subprocess.run([exe, inp], capture_output=True, text=True, timeout=300)
        return root, out

    def fields(root):
        f = lambda lam, what: os.path.join(root, "RUNS", pkg.io.filename("slab", lam, what))
        r = {"sed": pkg.io.read_array(f(-1.0, "sed"), "sed_arr", [L, ndir]), "lum_lost": pkg.io.read_array(f(-1.0, "lum_lost"), "lum_lost", [L])}
        for il, lam in enumerate(lambdas):
            r["u", il] = pkg.io.read_array(f(lam, "ufield"), "u_final", [N])
        return r, f

    root, out = run("full", False)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "restor" not in out.stdout
    ref, f = fields(root)
    # a second run on the same directory must not overwrite the final files
    out = subprocess.run([exe, os.path.join(root, "input.in")], capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "exists already" in out.stdout
    for sub, keep, msg in (("from_part2", ("*part2*", "*part1*"), "Intermediate files will be restored"),
                           ("from_part1", ("*part1*",), "Output part 1 will be restored")):
        root, out = run(sub, True, keep)
        assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
        assert msg in out.stdout, out.stdout[-2000:]
        got, _ = fields(root)
        for k in ref:
            tol = 1e-7 if k == "lum_lost" else 1e-9 if k == "sed" else RTOL
            assert relmax(got[k], ref[k]) <= tol, (sub, k)
    # restore_file_mpi unset: the files are there but the run starts from the beginning (and still ends the same)
    root, out = run("no_restore", False, ("*part1*",))
    assert out.returncode == 0 and "restored" not in out.stdout
    got, _ = fields(root)
    assert relmax(got["u", 0], ref["u", 0]) <= RTOL


@pytest.mark.parametrize("name", ["magtar_shell", "galaxy", "nbody"])
def test_baseline_configs_at_their_wavelength_counts(name, pkg):
    """BASELINE configs 3-5 with as many wavelengths per ray as the survey gives them, at sizes the oracle cannot reach:
      * config 3, MAGTAR shell: FULL size, 128 009 cells (base 10,3, max_lvl 3), 28 wavelengths, one point source;
      * config 4, GALAXY example: 16 wavelengths (the shipped list below 5 um) on the max_lvl-4 tree, 33 211 cells -
        REDUCED from the ~10^6 cells of the shipped grid (max_lvl 5 gives 315 334 cells and takes ~5 min per stage here;
        tools/run_config.py runs it by hand, profiles/r01_g_baseline_configs.txt);
      * config 5, N-body: 28 wavelengths, 4x10^5 pseudo-particles, 35 749 cells (irregular 27-ary tree) - the snapshot
        of the reference is not shipped, size as in SURVEY.md section 8d.
    One scattering iteration each. Size-independent properties: the dust absorbs a fraction in (0, 1) of what the
    sources emit at every wavelength, more where it is more opaque; the luminosity written off in lum_lost stays
    small; a second run reproduces fields and ray counts."""
    g = pkg.grids
    case = {"magtar_shell": lambda: g.shell_case(max_lvl=3, nlambda=28, max_sca_iterations=1),
            "galaxy": lambda: g.galaxy_case(max_lvl=4, nlambda=16, max_sca_iterations=1),
            "nbody": lambda: g.nbody_case(max_lvl=4, nlambda=28, npart=400000, max_sca_iterations=1)}[name]()
    t = case.tree
    assert (t.ncell, case.lnum) == {"magtar_shell": (128009, 28), "galaxy": (33211, 16), "nbody": (35749, 28)}[name]
    vol = (t.csize ** 3)[:, None]
    emitted = case.p_src_lum.sum(axis=0) + (case.dens_stars * vol).sum(axis=0)

    def run():
        gpu = pkg.DartRayGPU(case, arena_bytes=64 << 30)
        inter = []
        gpu.run_all(11, on_stage=lambda s: inter.append(gpu.counters()["intersections"]))  # precalculation, direct light, scattering
        u, lost, it = gpu.u_final, gpu.lum_lost, gpu.iterations
        gpu.close()
        return u, lost, it, inter

    u, lost, it, inter = run()
    assert it >= 1 and np.isfinite(u).all() and (u >= 0).all()
    kabs = case.dens * (1.0 - case.ksca_norm)[None, :]
    frac = (u * kabs * g.CS_PC * vol).sum(axis=0) / emitted
    assert (frac > 0).all() and (frac < 1).all()
    kmax = kabs.max(axis=0)
    for i in range(case.lnum):          # pairwise where the absorption opacity differs by more than a factor of two
        for j in range(case.lnum):
            if kmax[i] > 2.0 * kmax[j]:
                assert frac[i] > frac[j], (i, j, frac[i], frac[j])
    assert (lost / emitted < 0.08).all()
    u2, lost2, it2, inter2 = run()
    assert it2 == it and inter2 == inter and relmax(u2, u) <= 1e-11 and relmax(lost2, lost) <= 1e-9


def test_bench_workload_properties(pkg):
    """The bench workload itself (BASELINE configs[1]: TRUST-I tau_z = 1, 32 wavelengths, 85 801 cells) through the
    direct-light stage and one scattering iteration: the oracle would need hours here, so size-independent
    properties are checked. Every wavelength absorbs a fraction in (0, 1) of what the star emits, the absorbed
    fraction follows the extinction curve (more opaque -> more absorbed, compared pairwise where the opacity
    differs by more than a factor of two), the scattering iteration only adds energy, lum_lost stays below the
    1 % the reference warns about, and the counters of the scattering step equal those of a second, identical step
    (rays depend only on u_fest and scaspe_prev, both frozen during the step)."""
    g = pkg.grids
    case = g.trusti_case(tau_z=1.0, max_lvl=5, wavelengths=list(range(32)), max_sca_iterations=1)
    t = case.tree
    assert t.ncell == 85801
    gpu = pkg.DartRayGPU(case, arena_bytes=96 << 30)
    gpu.rt_prepare("start")
    gpu.rt_loop("precalc_cell"); gpu.rt_loop("precalc_src"); gpu.rt_mpi("precalc_src")
    gpu.rt_prepare("prep_part2")
    gpu.rt_loop("dir_cell"); gpu.rt_loop("dir_src"); gpu.rt_mpi("dir_src")
    u_dir = gpu.u_final.copy()
    assert not gpu.scatt_prepare()
    gpu.reset_counters(); gpu.rt_loop("scatt"); c1 = gpu.counters()
    u_sc = gpu.u_final.copy()
    gpu.reset_counters(); gpu.rt_loop("scatt"); c2 = gpu.counters()
    for k in ("intersections", "rays", "deposits", "scaspe_adds", "lambda_deposits"):
        assert c1[k] == c2[k], k
    lost = gpu.lum_lost.copy()
    gpu.close()
    assert np.isfinite(u_sc).all() and (u_sc >= u_dir * (1 - 1e-12)).all()
    vol = (t.csize ** 3)[:, None]
    kabs = case.dens * (1.0 - case.ksca_norm)[None, :]
    emitted = case.p_src_lum.sum(axis=0)
    frac = (u_dir * kabs * g.CS_PC * vol).sum(axis=0) / emitted
    assert (frac > 0).all() and (frac < 1).all()
    kext = g.TRUSTI_TABLE[:, 1] * (1.0 - g.TRUSTI_TABLE[:, 2])  # absorption opacity relative to 1 um
    for i in range(32):
        for j in range(32):
            if kext[i] > 2.0 * kext[j]:
                assert frac[i] > frac[j], (i, j)
    assert (lost / emitted < 0.01).all()


def test_bench_grid_sample_of_sources_matches_oracle(pkg, tmp_path, monkeypatch):
    """Parity at the size bench.py times: BASELINE configs[1] (TRUST-I tau_z = 1, 32 wavelengths, 85 801 cells), scattering
    iteration 1, on the sources i = 0 (mod 256) - the reference's own deal of sources to rank 0 of 256 ranks
    (rt_routines.f90:1003-1005), which both the oracle (set_parallel) and the library (np_mpi = 256, DRT_PARTITION=mod) follow.
    The oracle starts from the state the library's direct-light stages left (as bench.py's cpu_baseline leg does); a second
    library context is given the same state through drt_set_array. Rays, intersections and deposits must be equal and what the
    sample adds to u_final, scaspe and lum_lost must agree to the tolerances of the golden tests."""
    import os
    m = 256
    grids = pkg.grids
    case = grids.trusti_case(tau_z=1.0, max_lvl=5, wavelengths=list(range(32)), max_sca_iterations=1)
    assert case.tree.ncell == 85801
    full = pkg.DartRayGPU(case, arena_bytes=64 << 30)
    full.rt_prepare("start")
    full.rt_loop("precalc_cell"); full.rt_loop("precalc_src"); full.rt_mpi("precalc_src")
    full.rt_prepare("prep_part2")
    full.rt_loop("dir_cell"); full.rt_loop("dir_src"); full.rt_mpi("dir_src")
    state = {"u_final": full.u_final.copy(), "u_fest": full.u_fest.copy(), "lum_lost": full.lum_lost.copy(),
             "scaspe": [full.scaspe(il).copy() for il in range(full.L)]}
    L = full.L
    full.close()

    path = str(tmp_path / "bench_case.bin")
    case.write(path)
    o = op.Oracle(path, nsrc=case.p_src_pos.shape[0], ndir=case.dir_out.shape[0])
    np.copyto(o.u_final, state["u_final"]); np.copyto(o.u_fest, state["u_fest"]); np.copyto(o.lum_lost, state["lum_lost"])
    for il in range(L):
        np.copyto(o.scaspe(il), state["scaspe"][il])
    assert not o.scatt_prepare()
    u0_o = np.array(o.u_final).copy(); lost0_o = np.array(o.lum_lost).copy()
    o.set_parallel(m, 0, min(16, os.cpu_count() or 1))
    o.reset_counters(); o.rt_loop("scatt")
    co = o.counters()

    monkeypatch.setenv("DRT_PARTITION", "mod")
    g = pkg.DartRayGPU(case, np_mpi=m, id_mpi=0, arena_bytes=64 << 30)
    g.rt_prepare("start"); g.rt_prepare("prep_part2")
    g.set_array("u_fest", state["u_fest"]); g.set_array("u_final", state["u_final"]); g.set_array("lum_lost", state["lum_lost"])
    for il in range(L):
        g.set_array("scaspe", state["scaspe"][il], il)
    assert not g.scatt_prepare()
    u0_g = g.u_final.copy(); lost0_g = g.lum_lost.copy()
    assert relmax(u0_g, u0_o) <= 1e-14
    src = g.sources("scatt")
    assert len(src) > 100 and all(int(i) % m == 0 for i in src)
    g.reset_counters(); g.rt_loop("scatt")
    cg = g.counters()
    assert (cg["intersections"], cg["rays"], cg["deposits"]) == (co["intersections"], co["rays"], co["deposits"])
    assert cg["intersections"] > 10_000_000
    du_g, du_o = g.u_final - u0_g, np.array(o.u_final) - u0_o
    # the increments relative to their maximum; element by element the fields themselves (an increment is the difference of two
    # fields ~1e11 times larger in places: its last digits are rounding of the subtraction, not of the kernels)
    assert relmax(du_g, du_o) <= RTOL and relelem(g.u_final, np.array(o.u_final)) <= RTOL_ELEM
    assert relmax(g.lum_lost - lost0_g, np.array(o.lum_lost) - lost0_o) <= 1e-7
    for il in range(L):
        a, b = g.scaspe(il), np.array(o.scaspe(il))
        assert relmax(a, b) <= RTOL and relelem(a, b) <= RTOL_ELEM, il
    g.close(); o.close()


def test_a_repeated_stage_is_not_undone_again(pkg):
    """Queue-aware planning (plan_tasks, CNT_QMAX): once a kind of stage has run, its next launches are sized by the longest ray
    queue the last one held - sectors per task and task limit grow only where that queue left room. On TRUST-I tau = 1 with 32
    wavelengths (21 801 cells) and a 16 GiB arena (4480 rays per queue) the scattering step is repeated three times: whatever the first,
    unmeasured call needs, calls two and three must go through without an undone attempt, and all three trace the same rays."""
    case = pkg.grids.trusti_case(tau_z=1.0, max_lvl=4, wavelengths=list(range(32)), max_sca_iterations=1)
    g = pkg.DartRayGPU(case, arena_bytes=16 << 30)
    g.rt_prepare("start")
    g.rt_loop("precalc_cell"); g.rt_loop("precalc_src"); g.rt_mpi("precalc_src")
    g.rt_prepare("prep_part2")
    g.rt_loop("dir_cell"); g.rt_loop("dir_src"); g.rt_mpi("dir_src")
    assert not g.scatt_prepare()
    seen = []
    for k in range(3):
        g.reset_counters(); g.rt_loop("scatt")
        c = g.counters()
        seen.append((c["intersections"], c["rays"], c["deposits"], c["retries"], c["launches"]))
    assert seen[0][:3] == seen[1][:3] == seen[2][:3], seen
    assert seen[1][3] == 0 and seen[2][3] == 0, seen
    g.close()


def test_probe_part_changes_the_launches_not_the_result(pkg, monkeypatch):
    """A stage of which no queue length is known (8192 sources or more) runs a sixteenth of them first, one sector per task, and plans the others by
    what those showed (rt_loop: the probe). Sums over sources commute: on TRUST-I tau = 1 with 32 wavelengths (21 801 cells, 15 000
    sources) the first scattering call with the probe and without it (DRT_NO_PROBE) traces the same rays and leaves the same
    fields; only the launches differ."""
    case = pkg.grids.trusti_case(tau_z=1.0, max_lvl=4, wavelengths=list(range(32)), max_sca_iterations=1)
    out = {}
    for tag in ("probe", "whole"):
        if tag == "whole":
            monkeypatch.setenv("DRT_NO_PROBE", "1")
        g = pkg.DartRayGPU(case)
        g.set_psel_av(True)   # the average path of every source's rays: sums that must survive the stage running in parts
        g.rt_prepare("start")
        g.rt_loop("precalc_cell"); g.rt_loop("precalc_src"); g.rt_mpi("precalc_src")
        g.rt_prepare("prep_part2")
        g.rt_loop("dir_cell"); g.rt_loop("dir_src"); g.rt_mpi("dir_src")
        assert not g.scatt_prepare()
        g.reset_counters(); g.rt_loop("scatt")
        c = g.counters()
        psel = g.get_array("psel_av")
        assert (psel[:g.N] > 0).sum() == c["sources"]
        out[tag] = (c, g.u_final.copy(), g.scaspe(0).copy(), g.scaspe(20).copy(), g.lum_lost.copy(), psel)
        g.close()
    cp, cw = out["probe"][0], out["whole"][0]
    assert (cp["intersections"], cp["rays"], cp["deposits"]) == (cw["intersections"], cw["rays"], cw["deposits"])
    assert cp["launches"] > cw["launches"]
    for k in (1, 2, 3, 4, 5):
        assert relmax(out["probe"][k], out["whole"][k]) < 1e-12
