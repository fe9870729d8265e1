#!/bin/bash
# Developer tool: gfx950 ISA of one kernel of dartray_rt.hip. usage: tools/isa.sh [mangled-name-substring] [extra -D flags] -> /tmp/asm/k.s
set -e
K=${1:-_Z13rt_stage_wideILi32ELi4EEvPK10DevProblem9StageArgs}; shift || true
mkdir -p /tmp/asm
cd "$(dirname "$0")/../dart-ray_amd/csrc"
hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function "$@" -S --cuda-device-only dartray_rt.hip -o /tmp/asm/all.s 2>/dev/null
a=$(grep -n "^$K:" /tmp/asm/all.s | cut -d: -f1)
b=$(grep -n "\.end_amdhsa_kernel" /tmp/asm/all.s | awk -F: -v a=$a '$1>a{print $1; exit}')
sed -n "${a},${b}p" /tmp/asm/all.s > /tmp/asm/k.s
wc -l /tmp/asm/k.s
grep -E "vgpr_count|sgpr_count|scratch|\.amdhsa_next_free_vgpr|amdhsa_accum_offset|private_segment_fixed_size" /tmp/asm/k.s | head
