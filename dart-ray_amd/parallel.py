"""Stage-end reductions of rt_mpi (io_routines.f90:3667-3881) on torch tensors.

The reference runs one MPI rank per node in "no communication" mode: every rank holds the whole
grid, processes the sources i = im*np_mpi + id_mpi (rt_routines.f90:1003-1005) and the per-cell
arrays are summed over ranks at the end of each stage. Here one process drives one MI355X and the
sums are torch.distributed all-reduces (backend "nccl" = RCCL over xGMI) issued directly on the
library's device buffers; the same functions run on CPU tensors with the "gloo" backend, which is
how the multi-rank logic is tested without GPUs (tests/test_parallel_gloo.py).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.distributed as dist

RTT_PRECALC_SRC, RTT_DIR_SRC, RTT_SCATT = 2, 7, 10


def _steps(rt_type: int, f: Dict[str, object], npr: int):
    """rt_mpi (io_routines.f90:3667-3704) as a list of (tensor to sum over the ranks, what to do afterwards): the
    arithmetic in front of each sum is applied here, in place. One list per rank; the lists of all ranks line up."""
    steps = []
    if rt_type == RTT_PRECALC_SRC:
        # reduce_u_fest_arr (:3708-3729): MPI_Reduce(SUM) + MPI_Bcast == all-reduce
        steps.append((f["u_fest"], None))
    elif rt_type in (RTT_DIR_SRC, RTT_SCATT):
        # reduce_u_final_arr (:3732-3761): in a scattering iteration every rank starts from the same u_fest = u_final of
        # the previous stage, so only the increments are summed
        u_final, u_fest = f["u_final"], f["u_fest"]
        if rt_type == RTT_SCATT:
            u_final.sub_(u_fest).clamp_(min=0.0)
            steps.append((u_final, lambda: u_final.add_(u_fest)))
        else:
            steps.append((u_final, None))
        # reduce_scaspe_arr (:3806-3834): one wavelength at a time
        for s in f["scaspe"]:
            steps.append((s, None))
        # reduce_lum_lost (:3837-3855): lum_lost enters a scattering iteration holding the global value on every rank,
        # hence the (np-1)/np correction before the sum
        if rt_type == RTT_SCATT:
            f["lum_lost"].sub_(f["lum_lost_prev"] * ((npr - 1.0) / npr))
        steps.append((f["lum_lost"], None))
    else:
        raise ValueError(f"rt_mpi: rt_type {rt_type} has no reduction (io_routines.f90:3697)")
    # reduce_psel_av_arr (io_routines.f90:3859-3881; rt_mpi calls it after dir_src and scatt only), as sums and counts: a
    # source's tasks may be spread over the ranks here. Whether these two sums are in the list depends on the print_psel_av
    # flag alone, which is the same on every rank - never on whether this rank traced anything.
    if rt_type != RTT_PRECALC_SRC:
        for t in f.get("psel", []):
            steps.append((t, None))
    # not in the reference: the measured wave time per cell source (DRT_A_SRC_COST), known to the rank that traced it, is
    # what every rank deals the next stage's sources by
    if f.get("src_cost") is not None:
        steps.append((f["src_cost"], f.get("src_cost_done")))
    return steps


def rt_mpi(rt_type: int, fields: Dict[str, object], group=None) -> None:
    """rt_mpi over torch.distributed (backend nccl = RCCL on the GPU buffers, gloo on CPU tensors). `fields`: u_fest,
    u_final, lum_lost, lum_lost_prev (tensors) and scaspe (list of tensors) of THIS rank, all updated in place."""
    for t, after in _steps(rt_type, fields, dist.get_world_size(group)):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        if after:
            after()


def rt_mpi_local(rt_type: int, rank_fields: List[Dict[str, object]]) -> None:
    """The same reductions for several ranks that live in ONE process (e.g. two library contexts on one GPU, each with
    its own id_mpi): the all-reduce is a sum over the list. Used to test the product's multi-rank path - source
    partition, device-buffer aliasing, the arithmetic around the sums - where only one GPU is available."""
    npr = len(rank_fields)
    per_rank = [_steps(rt_type, f, npr) for f in rank_fields]
    for k in range(len(per_rank[0])):
        total = per_rank[0][k][0].clone()
        for r in range(1, npr):
            total.add_(per_rank[r][k][0])
        for r in range(npr):
            per_rank[r][k][0].copy_(total)
            if per_rank[r][k][1]:
                per_rank[r][k][1]()


class _DevAlias:
    """A raw device buffer presented through __cuda_array_interface__ (zero-copy into torch)."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def alias_device(ptr: int, n: int, device: Optional[int] = None) -> torch.Tensor:
    dev = "cuda" if device is None else f"cuda:{device}"
    t = torch.as_tensor(_DevAlias(ptr, n), device=dev)
    assert t.data_ptr() == ptr, "torch copied the buffer instead of aliasing it"
    return t
