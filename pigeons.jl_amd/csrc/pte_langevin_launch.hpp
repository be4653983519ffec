// pte_langevin_launch.hpp -- langevin_launch / langevin_set_rng_policy (pte_automala_params.hpp): included by exactly ONE translation unit,
// pte_langevin.hip in the product build, pte.hip when it is compiled alone.
#pragma once
#include <hip/hip_ext.h>
#include "pte_automala.hpp"

namespace pte {

template <typename K>
static inline void langevin_launch_one(K kernel, const LangevinLaunch &L, const EngineDev &dev, const AmParams &ap) {
    if (L.ext) hipExtLaunchKernelGGL(kernel, dim3(L.N), dim3(64), 0, L.stream, L.ev_a, L.ev_b, 0, dev, ap);
    else hipLaunchKernelGGL(kernel, dim3(L.N), dim3(64), 0, L.stream, dev, ap);
}

int langevin_launch(const LangevinLaunch &L, const EngineDev &dev, const AmParams &ap) {
#ifdef PTE_DEV_NO_LANGEVIN      // development builds only (tools/build_variant.sh): three quarters of the compile time are these instantiations
    (void)L; (void)dev; (void)ap; return 1;
#else
#define AM_ONE(EE)                                                                                                              \
    if (L.slice) langevin_launch_one(k_explore_automala<EE, TGT_FUNNEL, true>, L, dev, ap);                                     \
    else if (L.target == TGT_FUNNEL && L.full) langevin_launch_one(k_explore_automala<EE, TGT_FUNNEL, false, true>, L, dev, ap); \
    else if (L.target == TGT_FUNNEL) langevin_launch_one(k_explore_automala<EE, TGT_FUNNEL>, L, dev, ap);                        \
    else if (L.full) langevin_launch_one(k_explore_automala<EE, TGT_MVN, false, true>, L, dev, ap);                              \
    else langevin_launch_one(k_explore_automala<EE, TGT_MVN>, L, dev, ap);
    switch (L.E) { case 1: AM_ONE(1) break; case 2: AM_ONE(2) break; case 4: AM_ONE(4) break; case 8: AM_ONE(8) break; default: AM_ONE(16) break; }
#undef AM_ONE
    return 0;
#endif
}

void langevin_refresh_funnel_stats(int E, unsigned N, hipStream_t stream, const EngineDev &dev, double log3) {
    switch (E) {
    case 1: hipLaunchKernelGGL(k_refresh_funnel_stats<1>, dim3(N), dim3(64), 0, stream, dev, log3); break;
    case 2: hipLaunchKernelGGL(k_refresh_funnel_stats<2>, dim3(N), dim3(64), 0, stream, dev, log3); break;
    case 4: hipLaunchKernelGGL(k_refresh_funnel_stats<4>, dim3(N), dim3(64), 0, stream, dev, log3); break;
    case 8: hipLaunchKernelGGL(k_refresh_funnel_stats<8>, dim3(N), dim3(64), 0, stream, dev, log3); break;
    default: hipLaunchKernelGGL(k_refresh_funnel_stats<16>, dim3(N), dim3(64), 0, stream, dev, log3); break;
    }
}

int langevin_set_rng_policy(unsigned policy) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_rng_policy), &policy, sizeof policy); }

}  // namespace pte
