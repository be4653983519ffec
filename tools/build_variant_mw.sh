#!/bin/bash
# usage: [PROF=1] tools/build_variant_mw.sh <name> [-D...]   ->  build_variants/libpte_mw_<name>.so
# Development build for the four-waves-per-replica Langevin kernel (pte_automala_mw.hpp): pte.hip with the tree depths of d = 1024 / 4096 only,
# pte_langevin.hip with k_explore_langevin_mw only (PTE_DEV_ONLY_MW) -- 40 s instead of 2.5 min.  PTE_LIB=<path> python tools/bench_mw.py (BM_ONLY=mw)
# PROF=1: both units with -DPTE_PROFILE_AM (section stamps, tools/prof_mw.py)
cd "$(dirname "$0")/.." && mkdir -p build_variants
n=$1; shift
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -mllvm -align-all-nofallthru-blocks=6"
P=""; O=build_variants/mw_pte.o
if [ -n "$PROF" ]; then P="-DPTE_PROFILE_AM"; O=build_variants/mw_pte_prof.o; fi
[ -f $O ] || /opt/rocm/bin/hipcc $F -O2 -mllvm -amdgpu-sched-strategy=max-ilp -DPTE_SPLIT_LANGEVIN -DPTE_DEV_FEW_NLU $P -c -o $O pigeons.jl_amd/csrc/pte.hip || exit 1
/opt/rocm/bin/hipcc $F -DPTE_DEV_ONLY_MW $P "$@" -c -o build_variants/mw_lang_$n.o pigeons.jl_amd/csrc/pte_langevin.hip || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o build_variants/libpte_mw_$n.so $O build_variants/mw_lang_$n.o -ldl
