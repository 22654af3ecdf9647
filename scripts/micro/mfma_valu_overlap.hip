// Dev microbenchmark (not part of the product): do f64 VALU / integer VALU instructions of one
// wave overlap with the v_mfma_f64_16x16x4_f64 of another wave on the same SIMD?  Waves of a
// workgroup alternate roles (even: MFMA loop, odd: VALU loop); the kernel is timed with only the
// MFMA waves working, only the VALU waves working, and both.  both ~ max(...) -> they overlap;
// both ~ sum -> they share the unit (or the issue port).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int VKIND>   // 0: f64 FMA chain x4, 1: 32-bit integer multiplies (Philox-like), 2: f32 FMA
__global__ __launch_bounds__(512) void mix(double* out, int n_mfma, int n_valu) {
    const int wave = threadIdx.x >> 6;
    double r = 0.0;
    if ((wave & 1) == 0) {
        d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        double x = 1.0 + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9;
        for (int i = 0; i < n_mfma; ++i) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, a3, 0, 0, 0);
        }
        r = a0[0] + a1[1] + a2[2] + a3[3];
    } else if (VKIND == 0) {
        double a = threadIdx.x, b = 1.0, c = 2.0, d = 3.0;
        const double m = 1.0000001, k = 1e-9;
        for (int i = 0; i < n_valu; ++i) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                a = fma(a, m, k); b = fma(b, m, k); c = fma(c, m, k); d = fma(d, m, k);
            }
        }
        r = a + b + c + d;
    } else if (VKIND == 1) {
        unsigned a = threadIdx.x, b = 17, c = 29, d = 31;
        for (int i = 0; i < n_valu; ++i) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                a = __umulhi(a, 0xD2511F53u) ^ b; b = b * 0xCD9E8D57u + c;
                c = __umulhi(c, 0xCD9E8D57u) ^ d; d = d * 0xD2511F53u + a;
            }
        }
        r = (double)(a ^ b ^ c ^ d);
    } else {
        float a = threadIdx.x, b = 1.0f, c = 2.0f, d = 3.0f;
        for (int i = 0; i < n_valu; ++i) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                a = fmaf(a, 1.0000001f, 1e-9f); b = fmaf(b, 1.0000001f, 1e-9f);
                c = fmaf(c, 1.0000001f, 1e-9f); d = fmaf(d, 1.0000001f, 1e-9f);
            }
        }
        r = a + b + c + d;
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int VKIND>
static float run(double* out, int blocks, int nm, int nv) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(mix<VKIND>, dim3(blocks), dim3(512), 0, 0, out, nm / 10 + 1, nv / 10 + 1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(mix<VKIND>, dim3(blocks), dim3(512), 0, 0, out, nm, nv);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

// all eight waves of a workgroup on MFMAs (mask 0) or only the even ones (mask 1, as above)
__global__ __launch_bounds__(512) void mfma_only(double* out, int n, int mask) {
    const int wave = threadIdx.x >> 6;
    double r = 0.0;
    if ((wave & mask) == 0) {
        d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        double x = 1.0 + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9;
        for (int i = 0; i < n; ++i) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, a3, 0, 0, 0);
        }
        r = a0[0] + a1[1] + a2[2] + a3[3];
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

int main() {
    double* out; (void)hipMalloc(&out, 8 * 512 * 4096);
    {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int mask : {1, 0})
            for (int blocks : {256, 512}) {
                const int n = 4000;
                hipLaunchKernelGGL(mfma_only, dim3(blocks), dim3(512), 0, 0, out, 400, mask);
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(mfma_only, dim3(blocks), dim3(512), 0, 0, out, n, mask);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                const int mw = mask ? 4 : 8;
                printf("%d workgroups of 8 waves, %d of them on MFMAs: %.3f ms = %.1f TFLOP/s f64\n", blocks, mw, ms,
                       (double)blocks * mw * n * 4 * 2048.0 / ms / 1e9);
            }
    }
    const char* names[3] = {"f64 FMA", "u32 mul/mulhi", "f32 FMA"};
    for (int wg_per_cu : {1, 2}) {
        const int blocks = 256 * wg_per_cu;   // 8 waves per workgroup: 4 MFMA + 4 VALU waves
        const int nm = 4000;
        for (int kind = 0; kind < 3; ++kind) {
            // size the VALU loop to about the MFMA loop's duration
            auto go = [&](int a, int b) {
                return kind == 0 ? run<0>(out, blocks, a, b) : kind == 1 ? run<1>(out, blocks, a, b)
                                                                      : run<2>(out, blocks, a, b);
            };
            const float tm = go(nm, 0);
            int nv = 4000;
            float tv = go(0, nv);
            nv = (int)(nv * tm / tv);
            tv = go(0, nv);
            const float tb = go(nm, nv);
            printf("%d WG/CU (%d MFMA + %d VALU waves per SIMD), VALU = %-14s: mfma only %.3f ms, valu only %.3f ms, "
                   "both %.3f ms  (sum %.3f, max %.3f)\n", wg_per_cu, wg_per_cu, wg_per_cu, names[kind], tm, tv, tb,
                   tm + tv, tm > tv ? tm : tv);
        }
    }
    return 0;
}
