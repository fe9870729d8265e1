#!/usr/bin/env python3
"""Developer tool: run a golden/synthetic case stage by stage on the GPU and in the oracle and
print counters and field differences. usage: python tools/compare.py <golden-name> [stop-stage]"""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as ge
import oracle_py as op
pkg = ge._load_pkg()

def rel(a, b):
    s = np.abs(b).max()
    return 0.0 if s == 0 else float(np.abs(a - b).max() / s)

def main():
    name = sys.argv[1]
    z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, name + ".bin"); z["case_bin"].tofile(path)
    case = pkg.grids.read_case(path, name)
    o = op.Oracle(path, nsrc=case.p_src_pos.shape[0], ndir=case.dir_out.shape[0])
    g = pkg.DartRayGPU(case)
    g.rt_prepare("start")
    def stage(rt, field):
        o.reset_counters(); o.rt_loop(rt); g.reset_counters(); g.rt_loop(rt)
        co, cg = o.counters(), g.counters()
        a, b = getattr(g, field), np.array(getattr(o, field))
        print(f"{rt:13s} oracle i/r/d {co['intersections']:>9d} {co['rays']:>8d} {co['deposits']:>9d} | gpu {cg['intersections']:>9d} {cg['rays']:>8d} {cg['deposits']:>9d} "
              f"| {field} rel {rel(a, b):.2e} | ms {cg['kernel_ms']:.3f} maxq {co['max_high']}/{co['max_low']} nside {co['max_nside']}")
        if rel(a, b) > 1e-9:
            d = np.abs(a - b); idx = np.argsort(d.ravel())[::-1][:5]
            for k in idx:
                i, il = divmod(int(k), a.shape[1])
                print(f"    cell {i} il {il} lvl {case.tree.lvl[i]} gpu {a[i, il]:.10e} orc {b[i, il]:.10e} ccoord {case.tree.ccoord[i]}")
    stage("precalc_cell", "u_fest"); stage("precalc_src", "u_fest")
    g.rt_prepare("prep_part2")
    stage("dir_cell", "u_final"); stage("dir_src", "u_final")
    for il in range(g.L):
        a, b = g.scaspe(il), np.array(o.scaspe(il))
        print(f"  scaspe[{il}] rel {rel(a, b):.2e}")
        if rel(a, b) > 1e-9:
            d = np.abs(a - b); idx = np.argsort(d.ravel())[::-1][:8]
            for k in idx:
                i, p = divmod(int(k), a.shape[1])
                print(f"      cell {i} pix {p} lvl {case.tree.lvl[i]} gpu {a[i, p]:.10e} orc {b[i, p]:.10e} rowsum gpu {a[i, :g.npix_hp[il]].sum():.10e} orc {b[i, :g.npix_hp[il]].sum():.10e} ccoord {case.tree.ccoord[i]}")
            ncell_bad = int((np.abs(a - b).max(axis=1) > 1e-9 * np.abs(b).max()).sum())
            print(f"      cells with differences: {ncell_bad}")
    print("  lum_lost", g.lum_lost, np.array(o.lum_lost))
    o.rt_mpi("dir_src"); g.rt_mpi("dir_src")
    it = 0
    while True:
        fo, fg = o.scatt_prepare(), g.scatt_prepare()
        if fo != fg: print("cnflag differs", fo, fg)
        if fo or fg: break
        stage("scatt", "u_final"); o.rt_mpi("scatt"); g.rt_mpi("scatt")
        for il in range(g.L):
            print(f"  scaspe[{il}] rel {rel(g.scaspe(il), np.array(o.scaspe(il))):.2e}")
    print("iterations", o.iterations, g.iterations)

main()
