// Dev microbenchmark (not part of the product): cycles of one dependent wave-level sum of a
// double, DPP butterfly + readlanes (the product's wave_sum) against an f64-MFMA formulation.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../pybmc_amd/csrc/bmc_dev.h"
using namespace bmc;
typedef double d4 __attribute__((ext_vector_type(4)));

// sum over 64 lanes with v_mfma_f64_16x16x4_f64: A = v (lane l -> row l%16, k = l/16), B = 1:
// D[i][j] = sum_k A[i][k] (4 lanes each); then B' = D's register r (rows 4k+r), A' = 1, accumulated
// over r: every element of the result is the sum of the 16 row partials = the wave total.
__device__ __forceinline__ double wave_sum_mfma(double v) {
    d4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v, 1.0, acc, 0, 0, 0);
    d4 t = {0, 0, 0, 0};
    t = __builtin_amdgcn_mfma_f64_16x16x4f64(1.0, acc[0], t, 0, 0, 0);
    t = __builtin_amdgcn_mfma_f64_16x16x4f64(1.0, acc[1], t, 0, 0, 0);
    t = __builtin_amdgcn_mfma_f64_16x16x4f64(1.0, acc[2], t, 0, 0, 0);
    t = __builtin_amdgcn_mfma_f64_16x16x4f64(1.0, acc[3], t, 0, 0, 0);
    return t[0];
}

template <int MODE>
__global__ void chain(double* out, long long* ticks, int n) {
    double v = 1.0 + threadIdx.x * 1e-3;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        double s;
        if (MODE == 0) s = wave_sum(v);
        else s = wave_sum_mfma(v);
        v = v * 1e-3 + s * 1e-9 + threadIdx.x * 1e-3;   // dependent, lane-varying again
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) ticks[0] = t1 - t0;
}

int main() {
    double* out; long long* ticks; long long h;
    (void)hipMalloc(&out, 4096); (void)hipMalloc(&ticks, 64);
    const int n = 20000;
    double hv[64];
    for (int mode = 0; mode < 2; ++mode) {
        if (mode == 0) hipLaunchKernelGGL(chain<0>, dim3(1), dim3(64), 0, 0, out, ticks, n);
        else hipLaunchKernelGGL(chain<1>, dim3(1), dim3(64), 0, 0, out, ticks, n);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hv, out, 512, hipMemcpyDeviceToHost);
        printf("%s: %.1f cycles per dependent (sum + 2 flops) step, v[0]=%.17g v[63]=%.17g\n",
               mode ? "mfma f64 16x16x4 x5" : "dpp butterfly + readlanes", (double)h / n, hv[0], hv[63]);
    }
    return 0;
}
