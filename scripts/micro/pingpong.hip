// Dev microbenchmark (not part of the product): one-way latency of a store -> polled load
// between two workgroups, on the same XCD (through its L2) and across XCDs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;

template <int SCOPE_LOCAL>
__global__ void pingpong(u64* flag, int* xcc, int partner_stride, int rounds, long long* ticks) {
    const int b = blockIdx.x;
    if (threadIdx.x == 0) xcc[b] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf;
    if (b != 0 && b != partner_stride) return;
    if (threadIdx.x >= 64) return;
    const bool first = b == 0;
    u64* mine = flag + (first ? 0 : 64);
    u64* theirs = flag + (first ? 64 : 0);
    const int lane = threadIdx.x;
    long long t0 = 0;
    for (int r = 1; r <= rounds; ++r) {
        if (r == 2 && first) t0 = __builtin_amdgcn_s_memtime();
        if (first) {
            if (lane == 0) {
                if (SCOPE_LOCAL) __hip_atomic_store(mine, (u64)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else __hip_atomic_store(mine, (u64)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            for (long sp = 0; __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (u64)r; ++sp) if (sp > 20000000) return;
        } else {
            for (long sp = 0; __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (u64)r; ++sp) if (sp > 20000000) return;
            if (lane == 0) {
                if (SCOPE_LOCAL) __hip_atomic_store(mine, (u64)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else __hip_atomic_store(mine, (u64)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (first && lane == 0) ticks[0] = __builtin_amdgcn_s_memtime() - t0;
}

// load round trip alone: dependent chain of sc1 loads from one address
__global__ void load_rtt(const u64* p, int n, long long* ticks, u64* sink) {
    u64 acc = 0;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        u64 v = __hip_atomic_load(p + (acc & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        acc += v;
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { ticks[0] = t1 - t0; sink[0] = acc; }
}

int main() {
    u64* flag; int* xcc; long long* ticks; u64* sink;
    hipMalloc(&flag, 4096); hipMalloc(&xcc, 4096); hipMalloc(&ticks, 64); hipMalloc(&sink, 64);
    const int rounds = 20000;
    int hx[64]; long long ht;
    for (int local = 0; local < 2; ++local)
        for (int stride : {8, 1}) {
            if (local && stride == 1) continue;   // workgroup-scope store is only valid on one XCD
            hipMemset(flag, 0, 4096);
            if (local) hipLaunchKernelGGL(pingpong<1>, dim3(16), dim3(64), 0, 0, flag, xcc, stride, rounds, ticks);
            else hipLaunchKernelGGL(pingpong<0>, dim3(16), dim3(64), 0, 0, flag, xcc, stride, rounds, ticks);
            hipDeviceSynchronize();
            hipMemcpy(hx, xcc, sizeof(int) * 16, hipMemcpyDeviceToHost);
            hipMemcpy(&ht, ticks, 8, hipMemcpyDeviceToHost);
            printf("store scope %s, groups 0 (xcc %d) <-> %d (xcc %d): round trip %.0f ticks (memtime, 100 MHz => %.0f ns), one way %.0f ns\n",
                   local ? "workgroup" : "agent", hx[0], stride, hx[stride], (double)ht / (rounds - 1),
                   (double)ht / (rounds - 1) * 10.0, (double)ht / (rounds - 1) * 5.0);
        }
    hipMemset(flag, 0, 4096);
    hipLaunchKernelGGL(load_rtt, dim3(1), dim3(64), 0, 0, flag, 10000, ticks, sink);
    hipDeviceSynchronize();
    hipMemcpy(&ht, ticks, 8, hipMemcpyDeviceToHost);
    printf("dependent sc1 load round trip: %.1f ticks = %.0f ns\n", (double)ht / 10000, (double)ht / 10000 * 10.0);
    return 0;
}
