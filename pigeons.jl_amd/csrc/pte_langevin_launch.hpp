// pte_langevin_launch.hpp -- langevin_launch / langevin_set_rng_policy (pte_automala_params.hpp): included by exactly ONE translation unit,
// pte_langevin.hip in the product build, pte.hip when it is compiled alone.
#pragma once
#include <hip/hip_ext.h>
#include "pte_automala.hpp"
#include "pte_automala_mw.hpp"

namespace pte {

template <typename K>
static inline void langevin_launch_one(K kernel, const LangevinLaunch &L, const EngineDev &dev, const AmParams &ap) {
    if (L.ext) hipExtLaunchKernelGGL(kernel, dim3(L.N), dim3(64), 0, L.stream, L.ev_a, L.ev_b, 0, dev, ap);
    else hipLaunchKernelGGL(kernel, dim3(L.N), dim3(64), 0, L.stream, dev, ap);
}

template <typename K>
static inline void langevin_launch_mw(K kernel, const LangevinLaunch &L, const EngineDev &dev, const AmParams &ap) {      // one 256-thread workgroup per replica
    if (L.ext) hipExtLaunchKernelGGL(kernel, dim3(L.N), dim3(64 * MW_NWV), 0, L.stream, L.ev_a, L.ev_b, 0, dev, ap);
    else hipLaunchKernelGGL(kernel, dim3(L.N), dim3(64 * MW_NWV), 0, L.stream, dev, ap);
}

template <typename K>
static inline void langevin_launch_scans(K kernel, const LangevinLaunch &L, const EngineDev &dev, const AmParams &ap) {
    const unsigned wg = (unsigned)(L.scan_wg > 1 ? L.scan_wg : 1), grid = (L.N + wg - 1) / wg;
    if (L.ext) hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(64 * wg), 0, L.stream, L.ev_a, L.ev_b, 0, dev, ap, *L.scans);
    else hipLaunchKernelGGL(kernel, dim3(grid), dim3(64 * wg), 0, L.stream, dev, ap, *L.scans);
}

// the fused scan loop: one wave per chain for the shapes one wave holds comfortably (E <= 8: d <= 512), four waves per chain beyond (E = 16: k_scans_langevin_mw)
#define AM_SCANS_ONE(EE, WHAT)                                                                                                   \
    if (target == TGT_FUNNEL && full) { WHAT((k_scans_automala<EE, TGT_FUNNEL, true>)); }                                        \
    else if (target == TGT_FUNNEL) { WHAT((k_scans_automala<EE, TGT_FUNNEL, false>)); }                                          \
    else if (full) { WHAT((k_scans_automala<EE, TGT_MVN, true>)); }                                                              \
    else { WHAT((k_scans_automala<EE, TGT_MVN, false>)); }
#define AM_SCANS_WG_ONE(EE, WHAT)                                                                                                \
    if (target == TGT_FUNNEL && full) { WHAT((k_scans_automala_wg<EE, TGT_FUNNEL, true>)); }                                     \
    else if (target == TGT_FUNNEL) { WHAT((k_scans_automala_wg<EE, TGT_FUNNEL, false>)); }                                       \
    else if (full) { WHAT((k_scans_automala_wg<EE, TGT_MVN, true>)); }                                                           \
    else { WHAT((k_scans_automala_wg<EE, TGT_MVN, false>)); }

template <typename K>
static inline void langevin_launch_scans_mw(K kernel, const LangevinLaunch &L, const EngineDev &dev, const AmParams &ap) {      // one 256-thread workgroup per chain
    if (L.ext) hipExtLaunchKernelGGL(kernel, dim3(L.N), dim3(64 * MW_NWV), 0, L.stream, L.ev_a, L.ev_b, 0, dev, ap, *L.scans);
    else hipLaunchKernelGGL(kernel, dim3(L.N), dim3(64 * MW_NWV), 0, L.stream, dev, ap, *L.scans);
}
// (the scaled-precision MVN path only: on the funnel path the loop measured 1.905 ms per scan against 1.875-1.90 of the per-scan launches -- its
// body as a called function is 7 % slower than inlined, which eats what the loop gains; profiles/r06_langevin_mw.txt -- and is not instantiated)
#ifdef PTE_DEV_MW_FUNNEL_LOOP    // development builds only: the funnel's loop, to measure it again
#define AM_SCANS_MW(WHAT)                                                                                                        \
    if (target == TGT_FUNNEL && full) { WHAT((k_scans_langevin_mw<TGT_FUNNEL, true>)); }                                         \
    else if (target == TGT_FUNNEL) { WHAT((k_scans_langevin_mw<TGT_FUNNEL, false>)); }                                           \
    else if (full) { WHAT((k_scans_langevin_mw<TGT_MVN, true>)); }                                                               \
    else { WHAT((k_scans_langevin_mw<TGT_MVN, false>)); }
#else
#define AM_SCANS_MW(WHAT)                                                                                                        \
    if (target == TGT_FUNNEL) { return MW_NO_FUNNEL_LOOP; }                                                                      \
    else if (full) { WHAT((k_scans_langevin_mw<TGT_MVN, true>)); }                                                               \
    else { WHAT((k_scans_langevin_mw<TGT_MVN, false>)); }
#endif

int langevin_scan_wg() { return PTE_SCAN_WG; }
int langevin_scan_loop_blocks_per_cu(int E, int target, bool full, int scan_wg) {
#if defined(PTE_DEV_NO_LANGEVIN)
    (void)E; (void)target; (void)full; (void)scan_wg; return 0;
#else
    int n = 0;
    if (E == 16) {                                      // 512 < d <= 1024: four waves per chain (k_scans_langevin_mw)
        if (scan_wg > 1) return 0;
#define MW_NO_FUNNEL_LOOP 0
#define AM_OCC(KERNEL) hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, KERNEL, 64 * MW_NWV, 0)
        AM_SCANS_MW(AM_OCC)
#undef AM_OCC
#undef MW_NO_FUNNEL_LOOP
        return n;
    }
#ifdef PTE_DEV_ONLY_MW
    return 0;
#else
    if (scan_wg > 1) {
        if (scan_wg != PTE_SCAN_WG) return 0;
#define AM_OCC(KERNEL) hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, KERNEL, 64 * PTE_SCAN_WG, 0)
        switch (E) { case 1: AM_SCANS_WG_ONE(1, AM_OCC) break; case 2: AM_SCANS_WG_ONE(2, AM_OCC) break; case 4: AM_SCANS_WG_ONE(4, AM_OCC) break; case 8: AM_SCANS_WG_ONE(8, AM_OCC) break; default: return 0; }
#undef AM_OCC
        return n;
    }
#define AM_OCC(KERNEL) hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, KERNEL, 64, 0)
    switch (E) { case 1: AM_SCANS_ONE(1, AM_OCC) break; case 2: AM_SCANS_ONE(2, AM_OCC) break; case 4: AM_SCANS_ONE(4, AM_OCC) break; case 8: AM_SCANS_ONE(8, AM_OCC) break; default: return 0; }
#undef AM_OCC
    return n;
#endif
#endif
}

int langevin_launch(const LangevinLaunch &L, const EngineDev &dev, const AmParams &ap) {
#ifdef PTE_DEV_NO_LANGEVIN      // development builds only (tools/build_variant.sh): three quarters of the compile time are these instantiations
    (void)L; (void)dev; (void)ap; return 1;
#else
#ifdef PTE_DEV_ONLY_MW           // development builds only (tools/build_variant_mw.sh): the four-waves-per-replica kernels and nothing else of the family
    if (L.slice || L.E != 16) return 1;
    if (L.scans) {
        const int target = L.target; const bool full = L.full;
        if (L.scan_wg > 1) return 1;
#define MW_NO_FUNNEL_LOOP 1
#define AM_GO(KERNEL) langevin_launch_scans_mw(KERNEL, L, dev, ap)
        AM_SCANS_MW(AM_GO)
#undef AM_GO
#undef MW_NO_FUNNEL_LOOP
        return 0;
    }
    if (L.target == TGT_FUNNEL && L.full) langevin_launch_mw(k_explore_langevin_mw<TGT_FUNNEL, true>, L, dev, ap);
    else if (L.target == TGT_FUNNEL) langevin_launch_mw(k_explore_langevin_mw<TGT_FUNNEL, false>, L, dev, ap);
    else if (L.full) langevin_launch_mw(k_explore_langevin_mw<TGT_MVN, true>, L, dev, ap);
    else langevin_launch_mw(k_explore_langevin_mw<TGT_MVN, false>, L, dev, ap);
    return 0;
#else
    if (L.scans) {
        const int target = L.target; const bool full = L.full;
#define AM_GO(KERNEL) langevin_launch_scans(KERNEL, L, dev, ap)
        if (L.E == 16) {                                 // 512 < d <= 1024: four waves per chain
            if (L.scan_wg > 1) return 1;
#define MW_NO_FUNNEL_LOOP 1
#define AM_GO_MW(KERNEL) langevin_launch_scans_mw(KERNEL, L, dev, ap)
            AM_SCANS_MW(AM_GO_MW)
#undef AM_GO_MW
#undef MW_NO_FUNNEL_LOOP
            return 0;
        }
        if (L.scan_wg > 1) {
            if (L.scan_wg != PTE_SCAN_WG) return 1;
            switch (L.E) { case 1: AM_SCANS_WG_ONE(1, AM_GO) break; case 2: AM_SCANS_WG_ONE(2, AM_GO) break; case 4: AM_SCANS_WG_ONE(4, AM_GO) break; case 8: AM_SCANS_WG_ONE(8, AM_GO) break; default: return 1; }
            return 0;
        }
        switch (L.E) { case 1: AM_SCANS_ONE(1, AM_GO) break; case 2: AM_SCANS_ONE(2, AM_GO) break; case 4: AM_SCANS_ONE(4, AM_GO) break; case 8: AM_SCANS_ONE(8, AM_GO) break; default: return 1; }
#undef AM_GO
        return 0;
    }
#define AM_ONE(EE)                                                                                                              \
    if (L.slice) langevin_launch_one(k_explore_automala<EE, TGT_FUNNEL, true>, L, dev, ap);                                     \
    else if (L.target == TGT_FUNNEL && L.full) langevin_launch_one(k_explore_automala<EE, TGT_FUNNEL, false, true>, L, dev, ap); \
    else if (L.target == TGT_FUNNEL) langevin_launch_one(k_explore_automala<EE, TGT_FUNNEL>, L, dev, ap);                        \
    else if (L.full) langevin_launch_one(k_explore_automala<EE, TGT_MVN, false, true>, L, dev, ap);                              \
    else langevin_launch_one(k_explore_automala<EE, TGT_MVN>, L, dev, ap);
    switch (L.E) {
    case 1: AM_ONE(1) break; case 2: AM_ONE(2) break; case 4: AM_ONE(4) break; case 8: AM_ONE(8) break;
    default:
        // 512 < d <= 1024.  SliceSampler on the interpolated path: the one-wave kernel's slice instantiation (it holds no momentum, gradient or
        // trial copies and does not spill).  AutoMALA / MALA: four waves per replica (pte_automala_mw.hpp, round 6) -- the one-wave
        // instantiations with sixteen blocks per lane (250-300 spilled VGPRs) exist in the test build only, as the A/B reference of the new kernel
        if (L.slice) { langevin_launch_one(k_explore_automala<16, TGT_FUNNEL, true>, L, dev, ap); break; }
#ifdef PTE_TEST_KERNELS
        if (L.one_wave16) {
            if (L.target == TGT_FUNNEL && L.full) langevin_launch_one(k_explore_automala<16, TGT_FUNNEL, false, true>, L, dev, ap);
            else if (L.target == TGT_FUNNEL) langevin_launch_one(k_explore_automala<16, TGT_FUNNEL>, L, dev, ap);
            else if (L.full) langevin_launch_one(k_explore_automala<16, TGT_MVN, false, true>, L, dev, ap);
            else langevin_launch_one(k_explore_automala<16, TGT_MVN>, L, dev, ap);
            break;
        }
#endif
        if (L.target == TGT_FUNNEL && L.full) langevin_launch_mw(k_explore_langevin_mw<TGT_FUNNEL, true>, L, dev, ap);
        else if (L.target == TGT_FUNNEL) langevin_launch_mw(k_explore_langevin_mw<TGT_FUNNEL, false>, L, dev, ap);
        else if (L.full) langevin_launch_mw(k_explore_langevin_mw<TGT_MVN, true>, L, dev, ap);
        else langevin_launch_mw(k_explore_langevin_mw<TGT_MVN, false>, L, dev, ap);
        break;
    }
#undef AM_ONE
    return 0;
#endif
#endif
}

void langevin_refresh_funnel_stats(int E, unsigned N, hipStream_t stream, const EngineDev &dev, double log3) {
#ifdef PTE_DEV_ONLY_MW
    hipLaunchKernelGGL(k_refresh_funnel_stats<16>, dim3(N), dim3(64), 0, stream, dev, log3); (void)E; return;
#endif
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
