#!/usr/bin/env python3
"""Developer tool: diff the multiset of ray launches of one stage between the oracle and the GPU.
usage: python tools/trace_diff.py <golden-name> <stage>   (stage: precalc_cell precalc_src dir_cell dir_src)"""
import os, sys, tempfile, ctypes as C, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as ge
import oracle_py as op
pkg = ge._load_pkg()
name, stage = sys.argv[1], sys.argv[2]
z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
tmp = tempfile.mkdtemp()
path = os.path.join(tmp, name + ".bin"); z["case_bin"].tofile(path)
case = pkg.grids.read_case(path, name)
order = ["precalc_cell", "precalc_src", "dir_cell", "dir_src"]
tr = os.path.join(tmp, "orc.trace")
o = op.Oracle(path, nsrc=case.p_src_pos.shape[0], ndir=case.dir_out.shape[0])
o.L_.orc_trace_open.argtypes = [C.c_void_p, C.c_char_p]
g = pkg.DartRayGPU(case)
g.rt_prepare("start")
L = g.L_
L.drt_debug_trace.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
CAP = 6000000
for st in order:
    if st == "dir_cell": g.rt_prepare("prep_part2")
    if st == stage:
        o.L_.orc_trace_open(o.h, tr.encode())
        o.rt_loop(st)
        o.L_.orc_trace_open(o.h, None)
        L.drt_debug_trace(g.h, CAP, None, None)
        g.rt_loop(st)
        break
    o.rt_loop(st); g.rt_loop(st)
    if st == "precalc_src": g.rt_mpi(st)
    if st == "dir_src": o.rt_mpi(st); g.rt_mpi(st)
buf = np.zeros(CAP * 6, dtype=np.int32); cnt = C.c_int()
L.drt_debug_trace(g.h, 0, buf.ctypes.data_as(C.POINTER(C.c_int)), C.byref(cnt))
gp = buf[: min(cnt.value, CAP) * 6].reshape(-1, 6)
orc = np.loadtxt(tr, dtype=np.int64).reshape(-1, 5)
print("oracle launches", len(orc), "gpu launches", cnt.value)
# key: (src, nside, ipix or -1 for reco, cc_old, type)
co = collections.Counter(map(tuple, orc.tolist()))
gk = gp[:, [0, 1, 2, 3, 4]].astype(np.int64)
cg = collections.Counter(map(tuple, gk.tolist()))
for label, extra in (("GPU-only", cg - co), ("oracle-only", co - cg)):
    print(label, "launches:", sum(extra.values()))
    bt = collections.Counter()
    for k, v in extra.items(): bt[(k[1], k[4])] += v
    print("   by (nside, type):", dict(sorted(bt.items())))
    for k, v in list(extra.items())[:12]:
        print("     src %d nside %d ipix %d cc_old %d type %d x%d lvl(cc_old)=%d ccoord %s" % (k[0], k[1], k[2], k[3], k[4], v, case.tree.lvl[k[3]], case.tree.ccoord[k[3]]))
