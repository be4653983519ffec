// pte_langevin.hip -- the second translation unit of libpte.so: the Langevin-family kernels (AutoMALA, MALA, SliceSampler on the interpolated
// funnel path; pte_automala.hpp) behind langevin_launch.  Compiled WITHOUT -amdgpu-sched-strategy=max-ilp (pte_automala_params.hpp says why).
#define PTE_TU_LANGEVIN 1          // pte_kernels.hpp: leave the engine's non-template kernels to pte.hip
#include "pte_langevin_launch.hpp"
