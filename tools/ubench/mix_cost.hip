// Microbenchmark 3: cycles of instruction MIXES as they occur in the slice kernel's proposal chain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 2048
__device__ __forceinline__ int bfi(int m, int a, int b) { int r; asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(a), "v"(b)); return r; }
__device__ __forceinline__ int ashr31(int x) { int r; asm volatile("v_ashrrev_i32 %0, 31, %1" : "=v"(r) : "v"(x)); return r; }
__device__ __forceinline__ double sel(int m, double a, double b) {
    return __hiloint2double(bfi(m, __double2hiint(a), __double2hiint(b)), bfi(m, __double2loint(a), __double2loint(b))); }
__device__ __forceinline__ double rl(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane)); }

template <int MODE>
__global__ __launch_bounds__(64) void k(double *out, uint64_t *cyc, double a, double b) {
    double Lb = a, Rb = a + 10.0, xold = a + 3.0, U = b + threadIdx.x * 1e-3, cand = 0;
    int lm = threadIdx.x == 5 ? -1 : 0;
    int p = 0;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
        if (MODE == 0) {            // 4 dependent fp64 ops
            double W = Rb - Lb; double t = W * b; double v = Lb + t; Lb = v - xold;
        } else if (MODE == 1) {     // fp64 chain + sign mask + 4 bfi (bracket update)
            double W = Rb - Lb; double t = W * b; double v = Lb + t; double s = v - xold;
            int m = ashr31(__double2hiint(s)); Lb = sel(m, v, Lb); Rb = sel(m, Rb, v);
        } else if (MODE == 2) {     // + readlane of u
            double u = rl(U, p & 63); p++;
            double W = Rb - Lb; double t = W * u; double v = Lb + t; double s = v - xold;
            int m = ashr31(__double2hiint(s)); Lb = sel(m, v, Lb); Rb = sel(m, Rb, v);
        } else if (MODE == 3) {     // + candidate placement
            double u = rl(U, p & 63); p++;
            double W = Rb - Lb; double t = W * u; double v = Lb + t; double s = v - xold;
            cand = sel(lm, v, cand);
            int m = ashr31(__double2hiint(s)); Lb = sel(m, v, Lb); Rb = sel(m, Rb, v);
        } else if (MODE == 4) {     // same with compare + select (what the compiler makes of ?:)
            double u = rl(U, p & 63); p++;
            double W = Rb - Lb; double t = W * u; double v = Lb + t;
            cand = (threadIdx.x == 5) ? v : cand;
            bool below = v < xold; Lb = below ? v : Lb; Rb = below ? Rb : v;
        } else if (MODE == 5) {     // 4 bfi on independent data, dependent chain through one of them
            int m = lm; Lb = sel(m, Rb, Lb); Rb = sel(m, Lb, Rb);
        } else if (MODE == 6) {     // fp64 op followed by a 32-bit op on its result, chained
            double s = Lb - xold; int m = ashr31(__double2hiint(s)); Lb = __hiloint2double(__double2hiint(Lb) ^ (m & 1), __double2loint(Lb));
        } else if (MODE == 7) {     // tree path: mul + 10 dependent adds + mul + cmp->ballot
            double t = Lb * Lb;
#pragma unroll
            for (int k = 0; k < 10; ++k) t = t + b;
            double lp = t * a;
            unsigned long long ins = __ballot(xold < lp);
            Lb = Lb + (double)(int)(ins & 1);
        } else if (MODE == 8) {     // ballot -> scalar ffs -> dynamic readlane x2 (accept extraction)
            unsigned long long ins = __ballot(Lb < xold + threadIdx.x);
            int n = __builtin_ctzll(ins | (1ull << 63));
            Lb = rl(U, n) + Lb;
        }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = Lb + Rb + cand;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE> void run(const char *name) {
    double *out; uint64_t *cyc; hipMalloc(&out, 8 * 64); hipMalloc(&cyc, 8);
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0, 0.3); hipDeviceSynchronize();
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0, 0.3); hipDeviceSynchronize();
    uint64_t h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-70s %7.1f cycles/iter\n", name, (double)h / ITER);
}
int main() {
    run<0>("4 dependent fp64 ops (sub,mul,add,sub)");
    run<1>("fp64 x4 + ashr + 4 bfi (bracket update, VALU only)");
    run<2>("  + 2 readlane (u)");
    run<3>("  + 2 bfi (candidate placement) = one proposal step");
    run<4>("one proposal step written with ?: (v_cmp + v_cndmask)");
    run<5>("4 bfi dependent");
    run<6>("fp64 sub -> ashr -> and/xor on hi word -> fp64 (mixed 64/32 chain)");
    run<7>("tree path: mul + 10 adds + mul + cmp/ballot -> use");
    run<8>("ballot -> ctz -> 2 dynamic readlane -> add");
    return 0;
}
