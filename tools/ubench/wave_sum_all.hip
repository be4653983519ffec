// Microbenchmark 13 (round 3): the fixed-tree wave sum with the total delivered to ALL lanes in a VGPR (gfx950 v_permlane16_swap /
// v_permlane32_swap for the two cross-row levels) against wave_sum_dpp (row_bcast + v_readlane: the total arrives in SGPRs, i.e.
// every use of it is a trip vector -> scalar -> vector).  Checks bit-identity on random data, then times a dependent chain
// sum -> scale -> sum ..., one wave per SIMD.
#include "../../pigeons.jl_amd/csrc/pte_device.hpp"
#include <cstdio>
#include <cstdlib>
using namespace pte;
__device__ __forceinline__ double wave_sum_all(double v) {
    v = dpp_add_step<0xB1, 0xF>(v); v = dpp_add_step<0x4E, 0xF>(v); v = dpp_add_step<0x141, 0xF>(v); v = dpp_add_step<0x140, 0xF>(v);
    {   // rows 0 + 1 and 2 + 3
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
    }
    {   // halves
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
    }
    return v;
}
__global__ void k_check(const double *x, double *a, double *b) {
    const double v = x[blockIdx.x * 64 + threadIdx.x];
    a[blockIdx.x * 64 + threadIdx.x] = wave_sum_dpp(v);
    b[blockIdx.x * 64 + threadIdx.x] = wave_sum_all(v);
}
template <int MODE> __global__ __launch_bounds__(64) void k_time(double *out, unsigned long long *cyc, double c) {
    double v = 1.0 + threadIdx.x * 1e-3;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < 10000; ++it) {
        const double s = MODE ? wave_sum_all(v) : wave_sum_dpp(v);
        v = v * c + s * 1e-9;                      // the total is consumed by vector arithmetic, as in the Langevin kernels
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    const int nb = 4096; double *x, *a, *b; (void)hipMalloc(&x, nb * 64 * 8); (void)hipMalloc(&a, nb * 64 * 8); (void)hipMalloc(&b, nb * 64 * 8);
    double *hx = (double *)malloc(nb * 64 * 8), *ha = (double *)malloc(nb * 64 * 8), *hb = (double *)malloc(nb * 64 * 8);
    srand(1); for (int i = 0; i < nb * 64; ++i) hx[i] = (rand() / (double)RAND_MAX - 0.5) * exp2((double)(rand() % 40 - 20));
    (void)hipMemcpy(x, hx, nb * 64 * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_check, dim3(nb), dim3(64), 0, 0, x, a, b); (void)hipDeviceSynchronize();
    (void)hipMemcpy(ha, a, nb * 64 * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(hb, b, nb * 64 * 8, hipMemcpyDeviceToHost);
    long bad = 0; for (int i = 0; i < nb * 64; ++i) bad += !(ha[i] == hb[i]);
    printf("bit-identity of the all-lanes sum with wave_sum_dpp over %d waves of random data: %ld mismatching lanes\n", nb, bad);
    unsigned long long *cyc, c; double *out; (void)hipMalloc(&cyc, 8); (void)hipMalloc(&out, 512);
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) { if (mode) hipLaunchKernelGGL(k_time<1>, dim3(1024), dim3(64), 0, 0, out, cyc, 0.999); else hipLaunchKernelGGL(k_time<0>, dim3(1024), dim3(64), 0, 0, out, cyc, 0.999); (void)hipDeviceSynchronize(); }
        (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%s: %.1f cycles per dependent sum + use\n", mode ? "all-lanes sum (permlane swaps)          " : "wave_sum_dpp (row_bcast + v_readlane)   ", c / 10000.0);
    }
    return bad != 0;
}
