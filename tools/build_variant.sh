#!/bin/bash
# usage: tools/build_variant.sh <name> [-D...]   ->  build_variants/libpte_v_<name>.so
# A development build of libpte for A/B runs (PTE_LIB=... python tools/bench_variant.py): only the tree depths of d = 1024 / 4096 and no
# Langevin-family kernels (PTE_DEV_*: ~25 s instead of 2.5 min).  Never shipped: pigeons.jl_amd/lib/ is built by __graft_entry__.build().
cd "$(dirname "$0")/.." && mkdir -p build_variants
n=$1; shift
exec /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -ffp-contract=off -fPIC -shared -mllvm -align-all-nofallthru-blocks=6 -mllvm -amdgpu-sched-strategy=max-ilp -Wno-unused-value -DPTE_DEV_FEW_NLU -DPTE_DEV_NO_LANGEVIN "$@" \
     -o build_variants/libpte_v_$n.so pigeons.jl_amd/csrc/pte.hip -ldl
