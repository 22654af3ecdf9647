// Device-side helpers shared by the gfx950 kernels (wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bmc_math.h"

namespace bmc {

constexpr int WAVE = 64;

// ---- panel layout ----------------------------------------------------------
// X is kept on the device as row panels of RP = 64*VEC rows:
//   element (row n, col j)  ->  ((n / RP) * K + j) * RP + (n % RP)
// so that one wave-instruction reads VEC consecutive rows per lane of ONE column
// (64*VEC*sizeof(T) contiguous bytes), and a whole panel (K*RP elements) is one
// contiguous block that can be pinned in LDS verbatim.  y is a plain zero-padded
// vector of NP*RP elements.  Rows >= N are zero in both, so they add 0 to rss.
__host__ __device__ inline int64_t panel_offset(int64_t n, int32_t j, int32_t K, int32_t RP) {
    return ((n / RP) * K + j) * RP + (n % RP);
}

// ---- wave-level sum of a double, fixed order ---------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_rows_f64(double v) {  // 0.0 in the rows not selected
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}

// Sum over the 64 lanes; every lane returns the same bits.  The order is fixed (quad xor 1,
// quad xor 2, half-row mirror, row mirror -> every lane holds its row total; then the four
// row totals in order through the scalar unit), so the result does not depend on timing or
// placement.  (A/B on one MI355X: finishing with row_bcast15/row_bcast31 DPP steps instead of
// the four readlanes is 2 % slower per Gibbs iteration.)
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_mov_f64<0xB1>(v);          // quad_perm [1,0,3,2]
    v += dpp_mov_f64<0x4E>(v);          // quad_perm [2,3,0,1]
    v += dpp_mov_f64<0x141>(v);         // row_half_mirror
    v += dpp_mov_f64<0x140>(v);         // row_mirror
    const double r0 = readlane_f64(v, 0), r1 = readlane_f64(v, 16);
    const double r2 = readlane_f64(v, 32), r3 = readlane_f64(v, 48);
    return ((r0 + r1) + r2) + r3;
}

// ---- inter-workgroup granules (cdna guide, Guideline 16 form R2) -------------
// One naturally aligned 8-byte {tag = epoch, value = 32 data bits} written by ONE
// agent-scope relaxed atomic store (global_store_dwordx2 sc1) and read by
// agent-scope relaxed atomic loads (global_load_dwordx2 sc1): the data is the
// flag, no fence on either side.  A double travels as two granules.
using gu64 = unsigned long long;

__device__ __forceinline__ void granule_store(gu64* g, unsigned epoch, unsigned value) {
    __hip_atomic_store(g, ((gu64)epoch << 32) | (gu64)value, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ gu64 granule_load(const gu64* g) {
    return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11) -------------------------
struct u32x4 { uint32_t x, y, z, w; };

__host__ __device__ inline u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c.x;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c.z;
        u32x4 n;
        n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
        n.y = (uint32_t)p1;
        n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
        n.w = (uint32_t)p0;
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// 53 random bits -> (0, 1]
__host__ __device__ inline double u53_open0(uint32_t hi, uint32_t lo) {
    const uint64_t m = (((uint64_t)hi << 32) | lo) >> 11;
    return (double)(m + 1) * (1.0 / 9007199254740992.0);
}

enum : uint32_t { STREAM_NORMAL = 0x4e4f524du, STREAM_GAMMA = 0x47414d4du,
                  STREAM_PRED_NORMAL = 0x50524544u, STREAM_UNIFORM = 0x554e4946u };

}  // namespace bmc
