#!/usr/bin/env python3
"""Developer tool (no GPU needed): where in a kernel's loops do the spills sit?

The stage kernels run at 168 vector registers with a few hundred bytes of scratch, and every change to them re-deals which values
are spilled. What costs is a scratch load on the path every step takes: a memory round trip in a latency-bound loop. Round 3: five
more scratch loads in the ~1120-instruction step loop of rt_stage_wide<1,0> were 558 against 507 ms on the one-wavelength step,
found only after three A/B runs on the GPU; the counts below showed it in every one of those builds, without a GPU.

    hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function -S --cuda-device-only \
          dart-ray_amd/csrc/dartray_rt.hip -o /tmp/k.s
    python tools/loop_spills.py _Z13rt_stage_wideILi1ELi0EEvPK10DevProblem9StageArgs /tmp/k.s [/tmp/other_build.s ...]

For every loop (a branch back to an earlier label) of the kernel: first and last instruction index, size, scratch loads / stores,
v_writelane / v_readlane (scalar registers kept in vector lanes), global loads, f64 instructions - smallest loop first. Compare
the lines of the same loop between two builds (sizes shift by a few instructions). Landmarks, round 3's final code:
rt_stage_wide<1,0>: step loop n ~ 1124 with 6 / 1 scratch loads / stores, its parent n ~ 1712 with 12 / 3;
rt_stage_wide<32,0>: the deposit loop n ~ 857 with 0 / 0.
"""
import re
import sys


def loops(path, kern):
    txt = open(path).read()
    m = re.search(r'^' + re.escape(kern) + r':(.*?)\.end_amdhsa_kernel', txt, re.S | re.M)
    if not m:
        raise SystemExit(f"{path}: no kernel {kern}")
    ins, labels = [], {}
    for l in (x.strip() for x in m.group(1).split('\n')):
        if not l or l.startswith((';', '//')):
            continue
        ml = re.match(r'(\.LBB\w+):', l)
        if ml:
            labels[ml.group(1)] = len(ins)
            continue
        if l.startswith('.'):
            continue
        ins.append(l)
    res = []
    for i, l in enumerate(ins):
        mm = re.match(r's_c?branch\S*\s+(\.LBB\w+)', l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] <= i:
            a = labels[mm.group(1)]
            body = ins[a:i + 1]
            cnt = lambda *p: sum(1 for x in body if x.startswith(p))
            res.append((a, i, len(body), cnt('scratch_load'), cnt('scratch_store'), cnt('v_writelane'), cnt('v_readlane'),
                        cnt('global_load', 'flat_load'), sum(1 for x in body if '_f64' in x.split()[0])))
    return res


if __name__ == "__main__":
    if len(sys.argv) < 3:
        raise SystemExit(__doc__)
    kern = sys.argv[1]
    for p in sys.argv[2:]:
        print(p)
        for r in sorted(loops(p, kern), key=lambda r: r[2]):
            print('  loop [%5d,%5d] n=%4d scratch ld/st %3d/%3d  writelane/readlane %3d/%3d  gloads %3d f64 %3d' % r)
