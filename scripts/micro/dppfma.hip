// Dev microbenchmark (not part of the product): issue rate and dependent latency of
// v_fmac_f64 with a plain operand, with a DPP row_newbcast operand, and of an LDS broadcast
// read (ds_read_b64 / ds_read_b128 at a lane-uniform address) per FMA -- the three ways the
// residual pass can get u_j into every lane (bmc_loop.h, fmac_rowbcast_neg).
// Usage: dppfma [waves_per_block]   (1 block; 4 waves = one per SIMD, 8 = two per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define FM_PLAIN(a, u, x) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a) : "v"(u), "v"(x))
#define FM_DPP(a, u, x, n) \
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #n " row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(u), "v"(x))

template <int MODE>
__global__ void k(double* out, long long* ticks, int n, const double* uin) {
    __shared__ double ul[64];
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 64) ul[threadIdx.x] = uin[threadIdx.x];
    __syncthreads();
    double a[8], x[8];
    for (int i = 0; i < 8; ++i) { a[i] = lane * 1e-3 + i; x[i] = 1e-9 * (lane + i + 1); }
    double u = ul[lane & 15];
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; ++it) {
        if (MODE == 0) {          // 16 independent-ish plain FMAs (8 accumulators x 2)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) FM_PLAIN(a[i], u, x[i]);
        } else if (MODE == 1) {   // the same with DPP row broadcasts
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                FM_DPP(a[0], u, x[0], 0); FM_DPP(a[1], u, x[1], 1); FM_DPP(a[2], u, x[2], 2); FM_DPP(a[3], u, x[3], 3);
                FM_DPP(a[4], u, x[4], 4); FM_DPP(a[5], u, x[5], 5); FM_DPP(a[6], u, x[6], 6); FM_DPP(a[7], u, x[7], 7);
            }
        } else if (MODE == 2) {   // dependent chain, plain
#pragma unroll
            for (int i = 0; i < 16; ++i) FM_PLAIN(a[0], u, x[i & 7]);
        } else if (MODE == 3) {   // dependent chain, DPP
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                FM_DPP(a[0], u, x[0], 0); FM_DPP(a[0], u, x[1], 1); FM_DPP(a[0], u, x[2], 2); FM_DPP(a[0], u, x[3], 3);
                FM_DPP(a[0], u, x[4], 4); FM_DPP(a[0], u, x[5], 5); FM_DPP(a[0], u, x[6], 6); FM_DPP(a[0], u, x[7], 7);
            }
        } else if (MODE == 4) {   // 16 FMAs fed by 16 LDS broadcast reads (8 x ds_read_b128)
            const volatile double* vl = ul;
#pragma unroll
            for (int i = 0; i < 16; ++i) { const double uu = vl[(i + it) & 63]; a[i & 7] = fma(uu, x[i & 7], a[i & 7]); }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

int main(int argc, char** argv) {
    const int waves = argc > 1 ? atoi(argv[1]) : 4;
    double *out, *u; long long* ticks; long long h;
    (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&ticks, 4096); (void)hipMalloc(&u, 512);
    double hu[64]; for (int i = 0; i < 64; ++i) hu[i] = 1.0 + i * 1e-6;
    (void)hipMemcpy(u, hu, 512, hipMemcpyHostToDevice);
    const int n = 20000;
    const char* names[5] = {"plain fmac, 16 independent", "dpp row_newbcast fmac, 16 independent",
                            "plain fmac, dependent chain of 16", "dpp fmac, dependent chain of 16",
                            "fma fed by LDS broadcast reads, 16"};
    for (int mode = 0; mode < 5; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(1), dim3(64 * waves), 0, 0, out, ticks, n, u); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(1), dim3(64 * waves), 0, 0, out, ticks, n, u); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(1), dim3(64 * waves), 0, 0, out, ticks, n, u); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(1), dim3(64 * waves), 0, 0, out, ticks, n, u); break;
                default: hipLaunchKernelGGL(k<4>, dim3(1), dim3(64 * waves), 0, 0, out, ticks, n, u); break;
            }
            (void)hipDeviceSynchronize();
        }
        (void)hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
        printf("%d waves/CU  %-42s %.2f cycles per FMA (wave 0)\n", waves, names[mode], (double)h / n / 16);
    }
    return 0;
}
