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


def reduce_u_fest(u_fest: torch.Tensor, group=None) -> None:
    """reduce_u_fest_arr (io_routines.f90:3708-3729): MPI_Reduce(SUM) + MPI_Bcast == all-reduce."""
    dist.all_reduce(u_fest, op=dist.ReduceOp.SUM, group=group)


def reduce_u_final(u_final: torch.Tensor, u_fest: torch.Tensor, rt_type: int, group=None) -> None:
    """reduce_u_final_arr (io_routines.f90:3732-3761). In a scattering iteration every rank starts from
    the same u_fest = u_final of the previous stage, so only the increments are summed."""
    if rt_type == RTT_SCATT:
        u_final.sub_(u_fest).clamp_(min=0.0)
    dist.all_reduce(u_final, op=dist.ReduceOp.SUM, group=group)
    if rt_type == RTT_SCATT:
        u_final.add_(u_fest)


def reduce_scaspe(scaspe: List[torch.Tensor], group=None) -> None:
    """reduce_scaspe_arr (io_routines.f90:3806-3834): one wavelength at a time."""
    for s in scaspe:
        dist.all_reduce(s, op=dist.ReduceOp.SUM, group=group)


def reduce_lum_lost(lum_lost: torch.Tensor, lum_lost_prev: torch.Tensor, rt_type: int, group=None) -> None:
    """reduce_lum_lost (io_routines.f90:3837-3855): lum_lost enters a scattering iteration holding the
    global value on every rank, hence the (np-1)/np correction before the sum."""
    npr = dist.get_world_size(group)
    if rt_type == RTT_SCATT:
        lum_lost.sub_(lum_lost_prev * ((npr - 1.0) / npr))
    dist.all_reduce(lum_lost, op=dist.ReduceOp.SUM, group=group)


def rt_mpi(rt_type: int, fields: Dict[str, object], group=None) -> None:
    """rt_mpi (io_routines.f90:3667-3704). `fields`: u_fest, u_final, lum_lost, lum_lost_prev (tensors)
    and scaspe (list of tensors), all updated in place."""
    if rt_type == RTT_PRECALC_SRC:
        reduce_u_fest(fields["u_fest"], group)
    elif rt_type in (RTT_DIR_SRC, RTT_SCATT):
        reduce_u_final(fields["u_final"], fields["u_fest"], rt_type, group)
        reduce_scaspe(fields["scaspe"], group)
        reduce_lum_lost(fields["lum_lost"], fields["lum_lost_prev"], rt_type, group)
    else:
        raise ValueError(f"rt_mpi: rt_type {rt_type} has no reduction (io_routines.f90:3697)")


class _DevAlias:
    """A raw device buffer presented through __cuda_array_interface__ (zero-copy into torch)."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def alias_device(ptr: int, n: int, device: Optional[int] = None) -> torch.Tensor:
    dev = "cuda" if device is None else f"cuda:{device}"
    t = torch.as_tensor(_DevAlias(ptr, n), device=dev)
    assert t.data_ptr() == ptr, "torch copied the buffer instead of aliasing it"
    return t
