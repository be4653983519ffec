// Microbenchmark 14 (round 3): wave_sum_pairs<4|8> (gfx950 permlane swaps packing two chains' partial sums per register from the row
// level upwards) against wave_sum_dpp per chain + the block add: bit-identity on random data, then cycles per 8-chain reduction, one
// wave per SIMD.
#include "../../pigeons.jl_amd/csrc/pte_device.hpp"
#include <cstdio>
#include <cstdlib>
using namespace pte;
template <int M> __global__ void k_check(const double *x, double *a, double *b) {
    double v[M], o[M / 2];
    for (int j = 0; j < M; ++j) v[j] = x[(blockIdx.x * M + j) * 64 + threadIdx.x];
    for (int i = 0; i < M / 2; ++i) a[blockIdx.x * (M / 2) + i] = wave_sum_dpp(v[2 * i]) + wave_sum_dpp(v[2 * i + 1]);
    wave_sum_pairs<M>(v, o);
    for (int i = 0; i < M / 2; ++i) b[blockIdx.x * (M / 2) + i] = o[i];
}
template <int MODE> __global__ __launch_bounds__(64) void k_time(double *out, unsigned long long *cyc, double c) {
    double v[8];
    for (int j = 0; j < 8; ++j) v[j] = 1.0 + threadIdx.x * 1e-3 + j;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 2000; ++it) {
        double o[4];
        if (MODE == 0) { double w[8]; for (int j = 0; j < 8; ++j) w[j] = v[j]; wave_sum_dpp_multi<8>(w); for (int i = 0; i < 4; ++i) o[i] = w[2 * i] + w[2 * i + 1]; }
        else { double w[8]; for (int j = 0; j < 8; ++j) w[j] = v[j]; wave_sum_pairs<8>(w, o); }
        for (int j = 0; j < 8; ++j) v[j] = v[j] * c + o[j & 3] * 1e-9;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = v[0] + v[7];
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}
template <int M> static int check() {
    const int B = 256; const size_t n = (size_t)B * M * 64;
    double *hx = (double *)malloc(n * 8), *x, *a, *b, *ha = (double *)malloc(B * M * 4), *hb = (double *)malloc(B * M * 4);
    srand(7 + M);
    for (size_t i = 0; i < n; ++i) hx[i] = ((double)rand() / RAND_MAX - 0.5) * ((i % 7) ? 1.0 : 1e6);
    (void)hipMalloc(&x, n * 8); (void)hipMalloc(&a, B * M * 4); (void)hipMalloc(&b, B * M * 4);
    (void)hipMemcpy(x, hx, n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_check<M>, dim3(B), dim3(64), 0, 0, x, a, b);
    (void)hipMemcpy(ha, a, B * M * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(hb, b, B * M * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < B * M / 2; ++i) bad += ha[i] != hb[i];
    printf("wave_sum_pairs<%d>: %d of %d pair sums differ from wave_sum_dpp + wave_sum_dpp\n", M, bad, B * M / 2);
    return bad;
}
int main() {
    int bad = check<4>() + check<8>();
    double *out; unsigned long long *cyc, c;
    (void)hipMalloc(&out, 1024 * 64 * 8); (void)hipMalloc(&cyc, 8);
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) { if (mode) hipLaunchKernelGGL(k_time<1>, dim3(1024), dim3(64), 0, 0, out, cyc, 0.999); else hipLaunchKernelGGL(k_time<0>, dim3(1024), dim3(64), 0, 0, out, cyc, 0.999); (void)hipDeviceSynchronize(); }
        (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%s: %.1f ticks per 8-chain reduction + block adds\n", mode ? "wave_sum_pairs<8> (permlane swaps)        " : "wave_sum_dpp_multi<8> + 4 uniform adds    ", c / 2000.0);
    }
    return bad != 0;
}
