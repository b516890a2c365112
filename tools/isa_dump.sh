#!/bin/bash
# ISA of one translation unit with source lines attached (for tools/isa_budget.py): tools/isa_dump.sh nn_grid [extra -D flags]  ->  /tmp/mislam_isa/<unit>.s
u=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/mislam_isa/$u && cd /tmp/mislam_isa/$u || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -gline-tables-only -w -I/opt/rocm/include -I$root/cuda-slam_amd/csrc -I$root/include "$@" \
    --save-temps -c $root/cuda-slam_amd/csrc/$u.hip -o /tmp/mislam_isa/$u/$u.o 2>/dev/null
cp /tmp/mislam_isa/$u/$u-hip-amdgcn-amd-amdhsa-gfx950.s /tmp/mislam_isa/$u.s && echo /tmp/mislam_isa/$u.s
