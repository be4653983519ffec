// pte_automala_params.hpp -- what the launcher (pte.hip) and the Langevin-family kernels (pte_automala.hpp) share: kernel parameters and
// the one entry point through which the kernels are launched.  The product library is built from TWO translation units -- pte.hip
// (everything else, scheduled with -amdgpu-sched-strategy=max-ilp: the one-wave-per-SIMD slice kernels gain 1.3-2.3 %) and
// pte_langevin.hip (these kernels with the default scheduler: max-ilp costs their d >= 512 instantiations up to 10 %).  Tools and
// development builds compile pte.hip alone (no -DPTE_SPLIT_LANGEVIN): it then includes the kernels and this entry point itself.
#pragma once
#include "pte_kernels.hpp"

namespace pte {

enum { TGT_MVN = 0, TGT_FUNNEL = 2 };
enum { ERR_AM_DENSITY = 5, ERR_AM_STEP = 6 };

struct AmParams {
    double step_size;
    int n_refresh;
    int precond;            // 0 identity, 1 diagonal, 2 mix-diagonal
    double p0, p1;          // mix proportions
    const double *target_std;   // [d] or nullptr (== `nothing`: identity, no draw)
    int use_mh;             // scan != 1
    int mala;               // 1: MALA (src/explorers/MALA.jl:74-97) -- fixed step size, one leapfrog, always MH
    int slice;              // 1: SliceSampler on this path (SliceSampler.jl:24-237) with the full log potential per evaluation
    double slice_w; int slice_p, slice_n_passes, slice_max_iter;
    double ref_prec;        // funnel: precision of the normal reference
    double log3;            // log(3.0) from the host libm
    int pace;               // k_*_langevin_mw: 1 = steer the waves' priorities by the replicas' pace (several workgroups share a compute unit and all are resident)
};

// one launch of k_explore_automala<E, target, slice mode, whole blocks>: N workgroups of one wave on `stream`; `ext`: the launch carries
// the start / stop events (hipExtLaunchKernelGGL: the kernel's own begin and end, see PTE_LAUNCH1 in pte.hip)
struct LangevinLaunch { int E; int target; bool slice; bool full; unsigned N; hipStream_t stream; bool ext; hipEvent_t ev_a, ev_b;
                        const ScanLoop *scans = nullptr;        // scans != nullptr: k_scans_automala, all the scans of a pte_run_scans call in one launch
                        int scan_wg = 1;                         // ... > 1: k_scans_automala_wg, that many consecutive chains (waves) per workgroup (must equal langevin_scan_wg())
                        bool one_wave16 = false; };              // test build only (PTE_KERNEL_TEST_LANGEVIN_ONE_WAVE): 512 < d <= 1024 on the one-wave kernel with sixteen blocks per lane
int langevin_launch(const LangevinLaunch &L, const EngineDev &dev, const AmParams &ap);     // 0, or 1 if this build holds no such kernel
int langevin_scan_loop_blocks_per_cu(int E, int target, bool full, int scan_wg = 1);         // occupancy of k_scans_automala[_wg]<E, target, full> (0: not in this build)
int langevin_scan_wg();                                                                      // PTE_SCAN_WG of the Langevin translation unit
void langevin_refresh_funnel_stats(int E, unsigned N, hipStream_t stream, const EngineDev &dev, double log3);      // k_refresh_funnel_stats<E>
int langevin_set_rng_policy(unsigned policy);                                                // the translation unit's own copy of g_rng_policy (hipError_t as int)

}  // namespace pte
