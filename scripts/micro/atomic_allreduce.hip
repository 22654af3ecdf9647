// Dev microbenchmark (not part of the product): one all-reduce round among the workgroups of
// one XCD, three ways.  32 workgroups of 5 waves (the C2 geometry of gibbs_loop_kernel), each
// round depends on the previous total (as a Gibbs iteration does).
//   granule   wave 0 of every group publishes its value as two {epoch, 32 bits} words and
//             gathers the 64 words of the chain, one per lane (what the product does)
//   atomic32  wave 0 of every group adds {fixed-point value, arrival count} words to one shared
//             pair of 64-bit accumulators (L2 atomics) and polls that pair
//   atomic160 every wave adds; wave 0 polls (no LDS hop, no barrier in front of the publish)
//   atomic160x4 the same over 4 accumulator pairs (wave w adds to pair w & 3)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;

__device__ __forceinline__ u64 ld(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int MODE>   // 0 granule, 1 atomic32, 2 atomic160, 3 atomic160x4
__global__ __launch_bounds__(320) void ring(u64* words, int* xcc, int rounds, long long* ticks, double* sink) {
    __shared__ double bc[2];
    const int b = blockIdx.x;
    if (threadIdx.x == 0) xcc[b] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf;
    if (b % 8 != 0) return;
    const int g = b / 8, G = gridDim.x / 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int NARR = MODE == 1 ? G : 5 * G;
    double total = 1.0;
    u64 prev[2][4] = {};
    long long t0 = 0;
    for (int r = 1; r <= rounds; ++r) {
        if (r == 3) t0 = __builtin_amdgcn_s_memtime();
        const int par = r & 1;
        double x = 1.0 + 1e-3 * g + 1e-6 * wave + 1e-9 * total;   // depends on the last round
        if (MODE == 0) {
            // group value through LDS as the product does: skipped here, wave 0's x stands for it
            if (wave == 0) {
                u64* slot = words + par * 1024;
                if (lane < 2) {
                    const unsigned w = lane == 0 ? (unsigned)__double2hiint(x) : (unsigned)__double2loint(x);
                    __hip_atomic_store(slot + g * 8 + lane, ((u64)r << 32) | w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                const bool have = lane < 2 * G;
                const u64* p = slot + (have ? (lane >> 1) * 8 + (lane & 1) : 0);
                u64 v;
                for (long sp = 0;; ++sp) {
                    v = ld(p);
                    if (__all(!have || (unsigned)(v >> 32) == (unsigned)r)) break;
                    if (sp > 50000000) return;
                }
                const int w = have ? (int)(unsigned)v : 0;
                const int other = __builtin_amdgcn_update_dpp(0, w, 0xB1, 0xf, 0xf, true);
                const int swap = (w ^ other) & -(lane & 1);
                double d = __hiloint2double(w ^ swap, other ^ swap);
                for (int o = 2; o < 64; o <<= 1) d += __shfl_xor(d, o);
                total = d;
            }
        } else {
            const int NP = MODE == 3 ? 4 : 1;
            u64* slot = words + par * 1024;
            if ((MODE == 1 ? wave == 0 : true) && lane < 2) {
                // fixed point: hi = floor(x 2^40), lo = the next 40 bits
                const double xs = ldexp(x, 40);
                const double qh = floor(xs);
                const double ql = floor(ldexp(xs - qh, 40));
                const double q = lane == 0 ? qh : ql;
                const u64 bits = (u64)__double_as_longlong(q + 4503599627370496.0) & 0xFFFFFFFFFFFFFull;
                const u64 word = (bits << 8) | 1ull;
                __hip_atomic_fetch_add(slot + (MODE == 3 ? (wave & 3) * 16 : 0) + lane, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (wave == 0) {
                const bool have = lane < 2 * NP;
                const u64* p = slot + (have ? (lane >> 1) * 16 + (lane & 1) : 0);
                const int want = MODE == 3 ? ((lane >> 1) == 0 ? 2 * G : G) : NARR;   // waves 0,4 -> pair 0
                u64 v, d;
                const u64 pv = prev[par][0];
                for (long sp = 0;; ++sp) {
                    v = ld(p);
                    d = v - pv;
                    if (__all(!have || (int)(d & 255) == want)) break;
                    if (sp > 50000000) return;
                }
                prev[par][0] = v;
                double f = have ? (double)(d >> 8) : 0.0;
                if (lane & 1) f = ldexp(f, -40);
                f += __shfl_xor(f, 1);
                if (NP > 1) { f += __shfl_xor(f, 2); f += __shfl_xor(f, 4); }
                total = ldexp(__shfl(f, 0), -40);
            }
        }
        // the leader hands the total to the other waves (the product's B1 barrier + LDS)
        if (wave == 0 && lane == 0) bc[par] = total;
        __syncthreads();
        total = bc[par];
    }
    if (g == 0 && threadIdx.x == 0) { ticks[0] = __builtin_amdgcn_s_memtime() - t0; sink[0] = total; }
}

int main() {
    u64* words; int* xcc; long long* ticks; double* sink;
    hipMalloc(&words, 2048 * 8 * 2); hipMalloc(&xcc, 4096); hipMalloc(&ticks, 64); hipMalloc(&sink, 64);
    const int rounds = 20000;
    const char* names[4] = {"granule", "atomic32", "atomic160", "atomic160x4"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 4; ++mode) {
            hipMemset(words, 0, 2048 * 8 * 2);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            switch (mode) {
                case 0: hipLaunchKernelGGL(ring<0>, dim3(256), dim3(320), 0, 0, words, xcc, rounds, ticks, sink); break;
                case 1: hipLaunchKernelGGL(ring<1>, dim3(256), dim3(320), 0, 0, words, xcc, rounds, ticks, sink); break;
                case 2: hipLaunchKernelGGL(ring<2>, dim3(256), dim3(320), 0, 0, words, xcc, rounds, ticks, sink); break;
                default: hipLaunchKernelGGL(ring<3>, dim3(256), dim3(320), 0, 0, words, xcc, rounds, ticks, sink); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            int hx[256]; long long ht; double hs;
            hipMemcpy(hx, xcc, sizeof(int) * 256, hipMemcpyDeviceToHost);
            hipMemcpy(&ht, ticks, 8, hipMemcpyDeviceToHost); hipMemcpy(&hs, sink, 8, hipMemcpyDeviceToHost);
            int same = 1; for (int i = 0; i < 256; i += 8) same &= hx[i] == hx[0];
            printf("%-12s %.3f us/round (events), %.0f ticks/round, one XCD: %d, total %.12f\n", names[mode],
                   ms * 1e3 / rounds, (double)ht / (rounds - 2), same, hs);
        }
    return 0;
}
