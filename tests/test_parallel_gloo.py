"""The N>1 path on CPU: two `gloo` ranks, each running the oracle on its share of the sources
(i = im*np_mpi + id_mpi, rt_routines.f90:1003-1005) and the stage-end reductions of
dart-ray_amd/parallel.py (the very functions bench.py runs over RCCL on the GPU buffers). The
result must equal a single-rank run."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
import torch, torch.distributed as dist
ROOT = sys.argv[1]; case_path = sys.argv[2]; out = sys.argv[3]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as ge
import oracle_py as op
pkg = ge._load_pkg()
from dartray_amd import parallel
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
hdr = np.fromfile(case_path, dtype=np.int32, count=16)
o = op.Oracle(case_path, nsrc=int(hdr[6]), ndir=int(hdr[7]))
o.set_parallel(world, rank, 1)
# the per-source cost vector the GPU library keeps (DRT_A_SRC_COST): this rank knows the costs of its own sources only; rt_mpi
# sums it and then tells the library (src_cost_done). Here: cost = 1 + cell index for the cells i = rank (mod world).
marks = []
def fields():
    cost = np.zeros(o.N); cost[rank::world] = 1.0 + np.arange(o.N)[rank::world]
    f = {"u_fest": torch.from_numpy(o.u_fest), "u_final": torch.from_numpy(o.u_final),
         "lum_lost": torch.from_numpy(o.lum_lost), "lum_lost_prev": torch.from_numpy(o.lum_lost_prev),
         "scaspe": [torch.from_numpy(o.scaspe(il)) for il in range(o.L)],
         "src_cost": torch.from_numpy(cost), "src_cost_done": lambda: marks.append(1)}
    fields.cost = cost
    return f
o.rt_loop("precalc_cell"); o.rt_loop("precalc_src"); parallel.rt_mpi(op.RTT["precalc_src"], fields())
assert len(marks) == 1 and np.array_equal(fields.cost, 1.0 + np.arange(o.N)), "src_cost was not summed over the ranks"
o.rt_loop("dir_cell"); o.rt_loop("dir_src"); parallel.rt_mpi(op.RTT["dir_src"], fields())
while not o.scatt_prepare():
    o.rt_loop("scatt"); parallel.rt_mpi(op.RTT["scatt"], fields())
if rank == 0:
    np.savez(out, u_final=np.array(o.u_final), lum_lost=np.array(o.lum_lost), iterations=o.iterations,
             **{f"scaspe_tot{il}": np.array(o.scaspe_tot(il)) for il in range(o.L)})
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("name", ["cube65"])
def test_two_ranks_equal_one(name, golden):
    z, path = golden(name)
    with tempfile.TemporaryDirectory() as tmp:
        worker = os.path.join(tmp, "worker.py")
        open(worker, "w").write(WORKER)
        out = os.path.join(tmp, "out.npz")
        env = dict(os.environ, OMP_NUM_THREADS="1")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
               "--master-port", "29611", worker, ROOT, path, out]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        got = np.load(out)
    assert int(got["iterations"]) == int(z["iterations"])
    for k in ("u_final", "lum_lost"):
        ref = z[k]
        assert np.abs(got[k] - ref).max() <= 1e-10 * np.abs(ref).max(), k
    from conftest import golden_scaspe_check
    L = len([k for k in got.files if k.startswith("scaspe_tot")])
    for il in range(L):
        golden_scaspe_check(z, "scaspe_tot", il, got[f"scaspe_tot{il}"], 1e-10, name)
