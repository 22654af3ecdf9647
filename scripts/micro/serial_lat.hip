// Dev microbenchmark (not part of the product): dependent-chain latencies of the instruction
// kinds on the loop kernel's serial path, one wave alone on its SIMD (cycles per step).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../pybmc_amd/csrc/bmc_dev.h"
using namespace bmc;

template <int MODE>
__global__ void k(double* out, long long* ticks, int n) {
    __shared__ double lds[512];
    const int lane = threadIdx.x & 63;
    double v = 1.0 + lane * 1e-3, w = 0.5;
    lds[threadIdx.x] = v;
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        if (MODE == 0) {            // v_rsq_f64 dependent chain
            v = __builtin_amdgcn_rsq(v) + 1.0;
        } else if (MODE == 1) {     // add only (baseline for mode 0)
            v = v + 1.0;
        } else if (MODE == 2) {     // one DPP butterfly step: 2 dpp movs + add
            v += dpp_mov_f64<0xB1>(v);
        } else if (MODE == 3) {     // readlane -> scalar operand -> add
            v = v + readlane_f64(v, 16);
        } else if (MODE == 4) {     // full wave_sum
            v = wave_sum(v) * 1e-3 + lane;
        } else if (MODE == 5) {     // LDS write -> read back (same wave), dependent
            lds[lane] = v;
            v = lds[(lane + 1) & 63] + 1.0;
        } else if (MODE == 6) {     // LDS write, barrier, read (4 waves)
            lds[threadIdx.x] = v;
            __syncthreads();
            v = lds[(threadIdx.x + 64) & 255] + 1.0;
            __syncthreads();
        } else if (MODE == 7) {     // library rsqrt (rsq + Newton + class checks)
            v = rsqrt(v) + 1.0;
        } else if (MODE == 8) {     // library sqrt
            v = sqrt(v) + 1.0;
        } else if (MODE == 9) {     // s_barrier alone (4 waves)
            __syncthreads();
            v = v + 1.0;
        } else if (MODE == 10) {    // v_cndmask pair on a compare
            v = (v < w) ? v + 1.0 : v - 1.0;
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) ticks[0] = t1 - t0;
}

int main() {
    double* out; long long* ticks; long long h;
    (void)hipMalloc(&out, 1 << 16); (void)hipMalloc(&ticks, 64);
    const int n = 20000;
    const char* names[11] = {"v_rsq_f64 + add", "add f64", "dpp step (2 mov_dpp + add)", "readlane x2 + add",
                             "wave_sum + 2 flops", "LDS write -> read (one wave) + add",
                             "LDS write, barrier, read, barrier (4 waves)", "rsqrt() + add", "sqrt() + add",
                             "barrier + add (4 waves)", "cmp + select + add"};
#define RUN(M, W) hipLaunchKernelGGL(k<M>, dim3(1), dim3(64 * W), 0, 0, out, ticks, n)
    for (int mode = 0; mode < 11; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            switch (mode) {
                case 0: RUN(0, 1); break; case 1: RUN(1, 1); break; case 2: RUN(2, 1); break;
                case 3: RUN(3, 1); break; case 4: RUN(4, 1); break; case 5: RUN(5, 1); break;
                case 6: RUN(6, 4); break; case 7: RUN(7, 1); break; case 8: RUN(8, 1); break;
                case 9: RUN(9, 4); break; default: RUN(10, 1); break;
            }
            (void)hipDeviceSynchronize();
        }
        (void)hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
        printf("%-48s %.1f cycles per step\n", names[mode], (double)h / n);
    }
    return 0;
}
