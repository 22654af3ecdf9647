// Dev microbenchmark (not part of the product): which XCD (HW_REG_XCC_ID) and which CU the
// workgroups of a 196 x 512-thread launch land on, per HIP stream.  The loop kernel's teams are
// g mod 8; is block b always on XCD (b + c) mod 8, and does c depend on the stream / queue?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void where(int* xcc, int* hwid, long long spin) {
    if (threadIdx.x == 0) {
        xcc[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf;       // XCC_ID[3:0]
        hwid[blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 4);              // HW_ID
    }
    // stay resident for a while so that all 196 workgroups are on the chip together
    const long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < spin) {}
}
int main() {
    int *xcc, *hw;
    hipMalloc(&xcc, 4096); hipMalloc(&hw, 4096);
    int hx[196], hh[196];
    for (int s = 0; s < 10; ++s) {
        hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(where, dim3(196), dim3(512), 0, st, xcc, hw, 200000LL);
            hipStreamSynchronize(st);
            hipMemcpy(hx, xcc, sizeof(hx), hipMemcpyDeviceToHost);
            hipMemcpy(hh, hw, sizeof(hh), hipMemcpyDeviceToHost);
            printf("stream %d rep %d: xcc of blocks 0..15:", s, rep);
            for (int b = 0; b < 16; ++b) printf(" %d", hx[b]);
            int rot_ok = 1;
            for (int b = 0; b < 196; ++b) rot_ok &= (hx[b] == (hx[0] + b) % 8);
            printf("  | rotation of b mod 8: %s | SE/CU of blocks 0,8,16,24:", rot_ok ? "yes" : "NO");
            for (int b = 0; b < 32; b += 8) printf(" se%d.cu%d", (hh[b] >> 13) & 7, (hh[b] >> 8) & 15);
            printf("\n");
        }
    }
    return 0;
}
