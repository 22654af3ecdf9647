// Dev microbenchmark (not part of the product): sustained v_mfma_f64_16x16x4_f64 rate,
// register operands only, 4 independent accumulators per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void peak(double* out, int n) {
    d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    double x = 1.0 + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9;
    for (int i = 0; i < n; ++i) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, a3, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
int main() {
    double* out; (void)hipMalloc(&out, 8 * 256 * 4096);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int wgs_per_cu : {1, 2, 4}) {
        const int n = 20000, blocks = 256 * wgs_per_cu;
        hipLaunchKernelGGL(peak, dim3(blocks), dim3(256), 0, 0, out, 100);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(peak, dim3(blocks), dim3(256), 0, 0, out, n);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)blocks * 4 /*waves*/ * n * 4 /*mfma*/ * 2048.0;
        printf("%d workgroups of 4 waves per CU: %.1f TFLOP/s f64 (%.2f ms)\n", wgs_per_cu, flop / ms / 1e9, ms);
    }
    return 0;
}
