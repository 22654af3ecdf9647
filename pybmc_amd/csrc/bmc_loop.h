// Building blocks shared by the persistent loop kernels (Gibbs and simplex samplers):
// on-chip panel store, XCD placement check, group all-reduce of the partial rss.
#pragma once
#include <utility>

#include "bmc_dev.h"
#include "bmc_launch.h"

namespace bmc {

constexpr int MAX_KCH = 4;        // K <= 256 columns (64 per lane-chunk)
constexpr int RED_DOUBLES = 512;  // LDS doubles of the group-level sum (group_allreduce)
constexpr int MAX_GROUPS = 256;   // 8 teams of <= 32 groups (exchange_sum)
constexpr unsigned long long SPIN_TIMEOUT_TICKS = 400000000ull;  // 4 s of s_memrealtime (100 MHz)

enum { MODE_REG = 0, MODE_LDS = 1, MODE_STREAM = 2 };

struct LdsPlan {
    size_t u, red, ctl, aux, y, x, total;
};

// u: K doubles zero-padded to a multiple of 64 (MODE_REG reads KMAX of them);
// aux: kernel-specific doubles (the simplex kernel keeps Vt_hat there when it fits).
__host__ __device__ inline LdsPlan lds_plan(int K, int elem, int RP, int ppg, bool lds_resident,
                                            int aux_doubles = 0, int u_slices = 1, int red_slices = 1) {
    LdsPlan L;
    const size_t kp = (size_t)((K + 63) & ~63) * sizeof(double) * (size_t)u_slices;
    size_t o = 0;
    L.u = o;   o += kp;
    L.red = o; o += (size_t)RED_DOUBLES * sizeof(double) * (size_t)red_slices;
                                                    // one chain: [waves <= 8][64 lanes]; several
                                                    // chains per pass: [chains <= 8][waves <= 8], or
                                                    // (lane-wise form) [chains][waves <= 8][64 lanes]
    L.ctl = o; o += 8 * sizeof(double);
    L.aux = o; o += (size_t)((aux_doubles + 1) & ~1) * sizeof(double);
    L.y = o;
    if (lds_resident) o += (size_t)ppg * RP * elem;
    o = (o + 15) & ~(size_t)15;
    L.x = o;
    if (lds_resident) o += (size_t)ppg * K * RP * elem;
    L.total = o;
    return L;
}

// Diagnostic build only (-DBMC_STAMPS, scripts/dev_*): phase shares of one iteration as
// seen by wave 0 of group 0 of chain 0.  The product build contains no stamp.
#ifdef BMC_STAMPS
#define STAMP(i)                                                                      \
    do {                                                                              \
        if (stamping) {                                                               \
            __builtin_amdgcn_sched_barrier(0);                                        \
            unsigned long long now_;                                                  \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory"); \
            __builtin_amdgcn_sched_barrier(0);                                        \
            acc_[i] += now_ - last_;                                                  \
            last_ = now_;                                                             \
        }                                                                             \
    } while (0)
#define GSTAMP(i) STAMP(i)
#define STAMP_PARAMS , bool stamping, unsigned long long (&acc_)[12], unsigned long long& last_
#define STAMP_ARGS , stamping, acc_, last_
#else
#define GSTAMP(i)
#define STAMP_PARAMS
#define STAMP_ARGS
// Product build: no stamp; the phase boundary stays a scheduling fence (same-box A/B: neutral
// at C2, 2.6 % faster for single-workgroup chains than hipcc's own interleaving of the phases).
#define STAMP(i) __builtin_amdgcn_sched_barrier(0)
#endif

// ---- granule exchange ----------------------------------------------------------
// LOCAL = the chain's groups were verified to share one XCD: the store stays in that
// XCD's L2 (workgroup scope: global_store sc0) and the L1-bypassing agent-scope load
// (global_load sc1) is served by the same L2.  Otherwise the store is agent scope
// (sc1, write-through) and visible to every XCD.
// Granule i of a (chain, parity) slot: the pair (high word, low word) of group i/2 sits
// GRAN_PAIR_STRIDE words after the pair of group i/2 - 1.
template <int STRIDE = GRAN_PAIR_STRIDE>
__device__ __forceinline__ size_t gran_at(int i) {
    return (size_t)(i >> 1) * STRIDE + (i & 1);
}

template <bool LOCAL>
__device__ __forceinline__ void granule_put(gu64* g, unsigned epoch, unsigned value) {
    const gu64 w = ((gu64)epoch << 32) | (gu64)value;
    if constexpr (LOCAL)
        __hip_atomic_store(g, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else
        __hip_atomic_store(g, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#ifndef BMC_REGMULTI_DEPTH_BIG
#define BMC_REGMULTI_DEPTH_BIG 3
#endif
#ifndef BMC_POLL_DEPTH2
#define BMC_POLL_DEPTH2 2
#endif
// Gather n2 <= 64 granules of `epoch`, one per lane (the usual case: G <= 32).  Two reads are
// kept in flight, so the epoch is seen half a load round trip after it lands instead of up
// to a full one (three or four in flight were slower: same-box A/B).  Returns false when the
// bounded spin expired.  Lanes >= n2 get 0.
template <int DEPTH = 2, int STRIDE = GRAN_PAIR_STRIDE>
__device__ __forceinline__ bool granule_gather1(const gu64* gp, int n2, unsigned epoch, int lane,
                                                gu64& x STAMP_PARAMS) {
    unsigned long long t_start = 0;
    const bool have = lane < n2;
    const gu64* p = gp + gran_at<STRIDE>(have ? lane : 0);
    gu64 q[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) q[d] = granule_load(p);
    for (unsigned spins = 0;; ++spins) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            q[(d + DEPTH - 1) % DEPTH] = granule_load(p);
#ifdef BMC_STAMPS
            if (stamping) acc_[9] += 1;
#endif
            const gu64 w = q[d];
            if (__all(!have || (unsigned)(w >> 32) == epoch)) { x = have ? w : 0; return true; }
        }
        if ((spins & 0x7f) == 0x7f) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (t_start == 0) t_start = now;
            else if (now - t_start > SPIN_TIMEOUT_TICKS) return false;
        }
    }
}

// one granule per lane: even lane 2g' holds the high word of group g', odd lane the low
// word; lanes past the last group hold 0.  Both lanes of a pair assemble group g's double, so
// the sum over g' is wave_sum without its first step (which adds the two lanes of a pair).
__device__ __forceinline__ double granule_sum1(gu64 x, int lane) {
    const int w = (int)(unsigned)x;
    const int other = __builtin_amdgcn_update_dpp(0, w, 0xB1, 0xf, 0xf, true);
    // even lane: (hi, lo) = (w, other); odd lane: (other, w).  Branch-free (xor swap under a
    // lane-parity mask): as a ?: hipcc built it from exec-mask branches on the serial path.
    const int swap = (w ^ other) & -(lane & 1);
    double v = __hiloint2double(w ^ swap, other ^ swap);
    v += dpp_mov_f64<0x4E>(v);          // quad_perm [2,3,0,1]
    v += dpp_mov_f64<0x141>(v);         // row_half_mirror
    v += dpp_mov_f64<0x140>(v);         // row_mirror
    const double r0 = readlane_f64(v, 0), r1 = readlane_f64(v, 16);
    const double r2 = readlane_f64(v, 32), r3 = readlane_f64(v, 48);
    return ((r0 + r1) + r2) + r3;
}

// Publish this group's XCC id, gather the chain's G ids (agent scope, always valid) and
// decide.  G <= 32: 1 = all groups on one XCD.  G > 32 (two-level exchange, teams g mod 8):
// 1 = every group shares its XCD with the first member of its team, i.e. each team is
// XCD-local.  Run by wave 0; returns -1 when the spin expired.
// `nonce` (the launch's epoch base, below) tags the words: a word of an earlier launch that a
// cache still held would not be taken for this launch's.
__device__ __forceinline__ int detect_placement(gu64* xcc_words, int G, int g, int lane,
                                                unsigned nonce = 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf;  // HW_REG_XCC_ID[3:0]
    const gu64 tagv = (gu64)nonce << 32;
    if (lane == 0)
        __hip_atomic_store(xcc_words + g, tagv | (gu64)(xcc + 1), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    const bool teams = G > 32;
    bool same = true;
    unsigned long long t_start = 0;
    for (unsigned spins = 0;; ++spins) {
        bool ok = true;
        same = true;
        for (int b = 0; b < G; b += 64) {
            const int idx = b + lane;
            gu64 w = tagv | (xcc + 1), lead = tagv | (xcc + 1);
            if (idx < G) {
                w = __hip_atomic_load(xcc_words + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (teams)
                    lead = __hip_atomic_load(xcc_words + (idx & 7), __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
            }
            // ready = this launch's tag and a non-zero id
            ok = ok && (w >> 32) == nonce && (unsigned)w != 0 && (lead >> 32) == nonce && (unsigned)lead != 0;
            same = same && (w == lead);
        }
        if (__all(ok)) break;
        if ((spins & 0xff) == 0xff) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (t_start == 0) t_start = now;
            else if (now - t_start > SPIN_TIMEOUT_TICKS) return -1;
        }
    }
    return __all(same) ? 1 : 0;
}

// Which XCD did the hardware start this launch's round robin on?  Workgroup b of a launch runs
// on XCD (b + c) mod 8, with c depending on the queue (HIP stream) and its history
// (scripts/micro/xcc_map.hip).  Every workgroup publishes the XCC id it runs on under its BLOCK
// index and reads all G of them; the answer is c if the placement really is that rotation, -1
// otherwise (or when the bounded spin expired) -- data every workgroup reads identically, so all
// take the same decision.  Run by wave 0.
__device__ __forceinline__ int detect_rotation(gu64* words, int G, int b, int lane, unsigned nonce = 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf;  // HW_REG_XCC_ID[3:0]
    const gu64 tagv = (gu64)nonce << 32;
    if (lane == 0)
        __hip_atomic_store(words + b, tagv | (gu64)(xcc + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long t_start = 0;
    for (unsigned spins = 0;; ++spins) {
        bool ok = true, rot = true;
        const gu64 w0 = __hip_atomic_load(words, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int base = 0; base < G; base += 64) {
            const int idx = base + lane;
            gu64 w = tagv | 1;
            if (idx < G) w = __hip_atomic_load(words + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = ok && (w >> 32) == nonce && (unsigned)w != 0 && (w0 >> 32) == nonce && (unsigned)w0 != 0;
            rot = rot && (idx >= G || (((unsigned)w - 1) & 7) == (((unsigned)w0 - 1 + (unsigned)idx) & 7));
        }
        if (__all(ok)) return __all(rot) ? (int)(((unsigned)w0 - 1) & 7) : -1;
        if ((spins & 0xff) == 0xff) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (t_start == 0) t_start = now;
            else if (now - t_start > SPIN_TIMEOUT_TICKS) return -1;
        }
    }
}

// ---- partial rss of one panel, data in memory (LDS or global) ---------------------
template <typename T, int VEC, bool NT = false>
__device__ __forceinline__ double panel_rss(const T* __restrict__ xp, const T* __restrict__ yp,
                                            const double* __restrict__ u, int K) {
    constexpr int RP = 64 * VEC;
    // 16 columns per step: 16 independent 64*VEC*sizeof(T)-byte reads in flight per wave
    // (only ~8 waves run per CU, so the depth has to come from each wave), two FMA chains.
    constexpr int UN = 16;
    double a0[VEC], a1[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) { a0[v] = (double)yp[v]; a1[v] = 0.0; }
    int j = 0;
    for (; j + UN <= K; j += UN) {
        T x[UN][VEC];
#pragma unroll
        for (int q = 0; q < UN; ++q)
#pragma unroll
            for (int v = 0; v < VEC; ++v)
                x[q][v] = NT ? __builtin_nontemporal_load(&xp[(size_t)(j + q) * RP + v])
                             : xp[(size_t)(j + q) * RP + v];
#pragma unroll
        for (int q = 0; q < UN; q += 2) {
            const double u0 = u[j + q], u1 = u[j + q + 1];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                a0[v] = fma(-(double)x[q][v], u0, a0[v]);
                a1[v] = fma(-(double)x[q + 1][v], u1, a1[v]);
            }
        }
    }
    for (; j < K; ++j) {
        const double u0 = u[j];
#pragma unroll
        for (int v = 0; v < VEC; ++v) a0[v] = fma(-(double)xp[(size_t)j * RP + v], u0, a0[v]);
    }
    double s = 0.0;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        const double r = a0[v] + a1[v];
        s = fma(r, r, s);
    }
    return s;
}

// ---- u_j as an FMA operand without an LDS read per column ---------------------------------
// The residual pass needs u_j, the same value in all 64 lanes, for every column j.  Read from
// LDS at a lane-uniform address that is one ds_read per column and wave (2 LDS cycles per
// double): with 5 waves x 32 columns per CU the pass was bound by the LDS pipe (~320 cycles),
// not by its 32 FMAs (128).  Instead every wave loads u ONCE, lane l <- u[16 r + (l & 15)] in
// register r (one conflict-free ds_read_b64 per 16 columns), and the FMA takes its operand
// through DPP: v_fmac_f64_dpp ... row_newbcast:n hands every lane the value held by lane n of
// its own 16-lane row (the only DPP control 64-bit operations have on gfx950).  No LDS read and
// no extra VALU instruction per column; operands and operation order are unchanged, so the
// bits of the chain are unchanged (acc + u * (-x) == fma(-x, u, acc)).
// `FIRST`: the statement opens with the two wait states a DPP read needs after a VALU write of
// its source (hipcc does not see into the asm, cdna guide 5.7 item 2; u normally comes straight
// from the LDS read, but a compiler copy right in front of the statement must be safe too).
template <int N, bool FIRST>
__device__ __forceinline__ void fmac_rowbcast_neg(double& acc, double u_rows, double x) {
    static_assert(N >= 0 && N < 16, "lane within a row");
    if constexpr (FIRST)
        asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
            : "+v"(acc) : "v"(u_rows), "v"(x), "n"(N));
    else
        asm("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
            : "+v"(acc) : "v"(u_rows), "v"(x), "n"(N));
}

// A whole block of 16 columns in ONE statement (one row per lane: four accumulators, column j into
// accumulator j mod 4, as everywhere): the two wait states in front of the first DPP read and the
// sixteen reads are issued together, so nothing -- no register copy, reload or re-materialised
// value of the allocator's, no compiler-inserted s_nop -- can come between them (round-2 advisor
// finding: statement by statement only the FIRST of a block carried its own s_nop).  Same
// instructions, operands and order as sixteen fmac_rowbcast_neg statements.
__device__ __forceinline__ void fmac16_rowbcast_neg(double (&a)[4], double u_rows, const double (&x)[16]) {
#define BMC_F(A, X, N) "v_fmac_f64_dpp %" #A ", %4, -%" #X " row_newbcast:" #N " row_mask:0xf bank_mask:0xf\n\t"
    asm("s_nop 1\n\t"
        BMC_F(0, 5, 0) BMC_F(1, 6, 1) BMC_F(2, 7, 2) BMC_F(3, 8, 3)
        BMC_F(0, 9, 4) BMC_F(1, 10, 5) BMC_F(2, 11, 6) BMC_F(3, 12, 7)
        BMC_F(0, 13, 8) BMC_F(1, 14, 9) BMC_F(2, 15, 10) BMC_F(3, 16, 11)
        BMC_F(0, 17, 12) BMC_F(1, 18, 13) BMC_F(2, 19, 14) BMC_F(3, 20, 15)
        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])
        : "v"(u_rows), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]),
          "v"(x[7]), "v"(x[8]), "v"(x[9]), "v"(x[10]), "v"(x[11]), "v"(x[12]), "v"(x[13]), "v"(x[14]),
          "v"(x[15]));
#undef BMC_F
}

template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// A value the compiler must re-read from its register at this point: keeps f32 panel data
// as f32 VGPRs (one register per element) instead of hoisting the f32->f64 conversion out
// of the iteration loop, which would double the register footprint.
__device__ __forceinline__ double as_f64_in_loop(float x) {
    asm volatile("" : "+v"(x));
    return (double)x;
}
__device__ __forceinline__ double as_f64_in_loop(double x) { return x; }

// ---- the group's row panels, pinned on chip (or streamed) ---------------------------
// Group g owns panels g, g+G, ...; wave w of the group handles local panels w, w+nw, ...
template <typename T, int VEC, int MODE, int KMAX, int PPW>
struct PanelStore {
    static constexpr int RP = 64 * VEC;
    T xr[PPW > 0 ? PPW : 1][KMAX > 0 ? KMAX : 1][VEC];   // register mode: VEC rows per lane
    T yr[PPW > 0 ? PPW : 1][VEC];
    const T* Xg;
    const T* yg;
    T* Xs;
    T* ys;
    int K, G, g, npl, nw, wave, lane, keep;

    // share_last (register residency, one panel per wave, several chains per pass): the waves
    // past the group's last panel hold a COPY of that panel, so that its chains can be split
    // among them (gibbs_multi_kernel) instead of one SIMD carrying two whole panels
    // balanced (register residency, bundles of 8 chains on 8 waves, at most 5 panels per group,
    // PPW = 2 registers sets): set 0 holds panel wave % 4, set 1 the fifth panel (index 4) --
    // see partial_rss_reg_bal
    __device__ __forceinline__ void init(const Panels& P, int G_, int g_, T* Xs_, T* ys_,
                                         bool share_last = false, bool balanced = false) {
        keep = P.stream_keep;
        Xg = reinterpret_cast<const T*>(P.X);
        yg = reinterpret_cast<const T*>(P.y);
        Xs = Xs_;
        ys = ys_;
        K = P.k;
        G = G_;
        g = g_;
        const int NP = P.npanels;
        npl = g < NP ? (NP - g + G - 1) / G : 0;
        const int tid = threadIdx.x;
        lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        nw = blockDim.x >> 6;
        if constexpr (MODE == MODE_REG) {
#pragma unroll
            for (int i = 0; i < PPW; ++i) {
                int q = wave + i * nw;
                if (share_last && PPW == 1 && npl > 0 && q >= npl) q = npl - 1;
                if (balanced) q = i == 0 ? (wave & 3) : 4;
                const bool have = q < npl;
                const int64_t p = g + (int64_t)q * G;
#pragma unroll
                for (int j = 0; j < KMAX; ++j)
#pragma unroll
                    for (int v = 0; v < VEC; ++v)
                        xr[i][j][v] = (have && j < K) ? Xg[(p * K + j) * RP + lane * VEC + v] : (T)0;
#pragma unroll
                for (int v = 0; v < VEC; ++v) yr[i][v] = have ? yg[p * RP + lane * VEC + v] : (T)0;
            }
        } else if constexpr (MODE == MODE_LDS) {
            constexpr int EPV = 16 / (int)sizeof(T);
            const int vec_per_panel = K * RP / EPV;
            for (int q = 0; q < npl; ++q) {
                const int64_t p = g + (int64_t)q * G;
                const uint4* src = reinterpret_cast<const uint4*>(Xg + p * (int64_t)K * RP);
                uint4* dst = reinterpret_cast<uint4*>(Xs + (size_t)q * K * RP);
                for (int e = tid; e < vec_per_panel; e += blockDim.x) dst[e] = src[e];
                for (int e = tid; e < RP; e += blockDim.x) ys[q * RP + e] = yg[p * RP + e];
            }
        }
    }

    // register residency, CPP chains per pass: u is [CPP][kpad] in LDS; per chain the operation
    // order is that of partial_rss below (four FMA chains per row, then (a0+a1)+(a2+a3)).
    // Left to itself hipcc issues all CPP*KMAX LDS reads of u before the first FMA and spills;
    // here the address of block b's reads is made to depend (through empty asm statements that
    // emit no code) on the accumulators of block b-DEPTH, so at most DEPTH blocks of UB values of
    // u are live at a time.
    // chains [c_lo, c_hi) only (wave-uniform): the other entries of s are left as they are
    template <int CPP>
    __device__ __forceinline__ void partial_rss_reg_multi(const double* u, int kpad, double (&s)[CPP],
                                                          int c_lo = 0, int c_hi = CPP) const {
        static_assert(MODE == MODE_REG, "register residency only");
        typedef const __attribute__((address_space(3))) double lds_cd;
#ifndef BMC_REGMULTI_LDS
        // From 16 columns on: u through DPP row broadcasts, as in partial_rss -- per chain one
        // ds_read_b64 per 16 columns (lane l <- u[c][16 r + (l & 15)]) instead of one LDS broadcast
        // read per (chain, column).  With 4 chains x 64 columns (C4) that was 256 reads per wave
        // and pass, 4096 cycles of the CU's LDS pipe beside ~3000 cycles of FMAs and conversions.
        // Per chain the operations and their order are those of partial_rss (acc + u (-x) ==
        // fma(-x, u, acc)), so a chain's bits do not depend on how many chains share the pass.
        if constexpr (KMAX >= 16) {
            lds_cd* ub = (lds_cd*)u;
            constexpr int NU = (KMAX + 15) / 16;
            // the u of several chains first (CH chains x NU registers, at most 16 doubles = 32 VGPRs:
            // all 8 chains at 32 columns), so that ONE LDS latency is paid per CH chains instead
            // of one per chain (stamps, 8 chains at C2: the pass took 2650 cycles for 256 FMAs)
            constexpr int CH = (16 / NU) >= CPP ? CPP : (16 / NU);
            static_assert(CPP % CH == 0, "chains per chunk");
            static_for<CPP / CH>([&](auto hc) {
                constexpr int c0 = decltype(hc)::value * CH;
                double urow[CH][NU];
#pragma unroll
                for (int c = 0; c < CH; ++c)
#pragma unroll
                    for (int r = 0; r < NU; ++r) urow[c][r] = ub[(c0 + c) * kpad + r * 16 + (lane & 15)];
                // (pinned where they are loaded: the DPP statements are opaque to hipcc, which
                // otherwise re-reads u from LDS in front of every one of them)
#pragma unroll
                for (int c = 0; c < CH; ++c)
#pragma unroll
                    for (int r = 0; r < NU; ++r) asm volatile("" : "+v"(urow[c][r]));
                auto one_chain = [&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    double acc[PPW][VEC][4];
#pragma unroll
                    for (int i = 0; i < PPW; ++i)
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            acc[i][v][0] = (double)yr[i][v];
                            acc[i][v][1] = acc[i][v][2] = acc[i][v][3] = 0.0;
                        }
                    // (tried in round 3: u_j of the first 4 / 8 columns of every block of 16 from an LDS
                    // broadcast read + plain v_fma_f64 instead of the DPP form, the LDS pipe being idle
                    // in this pass -- 64 chains at C2 2.34 -> 2.53 / 2.77 us per iteration: rejected)
#ifndef BMC_NO_BLOCK_ASM
                    if constexpr (PPW * VEC == 1 && KMAX % 16 == 0 && sizeof(T) == 8) {
                        static_for<NU>([&](auto rc) {      // (one statement per block of 16 columns)
                            constexpr int r = decltype(rc)::value;
                            double xb[16];
#pragma unroll
                            for (int q = 0; q < 16; ++q) xb[q] = (double)xr[0][16 * r + q][0];
                            fmac16_rowbcast_neg(acc[0][0], urow[c][r], xb);
                        });
                    } else
#endif
                    static_for<KMAX>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
#pragma unroll
                        for (int i = 0; i < PPW; ++i)
#pragma unroll
                            for (int v = 0; v < VEC; ++v)
                                fmac_rowbcast_neg<j % 16, (j % 16 == 0)>(acc[i][v][j % 4], urow[c][j / 16],
                                                                       as_f64_in_loop(xr[i][j][v]));
                    });
                    double t = 0.0;
#pragma unroll
                    for (int i = 0; i < PPW; ++i)
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            const double r = (acc[i][v][0] + acc[i][v][1]) + (acc[i][v][2] + acc[i][v][3]);
                            t = fma(r, r, t);
                        }
                    s[c0 + c] = t;
                };
                // a wave that computes every chain (the usual case) runs the chains back to back
                // with no test between them: hipcc then overlaps the closing additions of one
                // chain with the first FMAs of the next
                if (c_lo == 0 && c_hi == CPP) {
                    static_for<CH>([&](auto cc) { one_chain(cc); });
                } else {
                    static_for<CH>([&](auto cc) {
                        constexpr int c = decltype(cc)::value;
                        if (c0 + c >= c_lo && c0 + c < c_hi) one_chain(cc);
                    });
                }
            });
            return;
        }
#endif
        // blocks of UB values of u, DEPTH blocks in flight (fewer where the panel already
        // fills 128 VGPRs)
        // (Stamps: the pass is LDS-latency bound, one block of u at a time -- the volatile asm
        // statements below are ordered against each other, so DEPTH bounds the reads in flight
        // from above but hipcc does not use it.  Data-ordered (non-volatile) tokens let it run
        // ahead again and spill; left as is: this mode already beats separate launches.)
        constexpr int UB = 4;
        constexpr int DEPTH = (KMAX * VEC * (int)sizeof(T) >= 256) ? BMC_REGMULTI_DEPTH_BIG : 4;
        lds_cd* ub = (lds_cd*)u;
        int tok[DEPTH];   // tok[d]: "the FMAs of the block d+1 back have been issued" (no code)
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) tok[d] = 0;
#pragma unroll
        for (int c = 0; c < CPP; ++c) {
            if (c < c_lo || c >= c_hi) continue;
            double acc[PPW][VEC][4];
#pragma unroll
            for (int i = 0; i < PPW; ++i)
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    acc[i][v][0] = (double)yr[i][v];
                    acc[i][v][1] = acc[i][v][2] = acc[i][v][3] = 0.0;
                }
#pragma unroll
            for (int jb = 0; jb < KMAX; jb += UB) {
                int off = c * kpad + jb;
                asm volatile("" : "+v"(off) : "v"(tok[DEPTH - 1]));
                lds_cd* uc = ub + off;
                const double u0 = uc[0], u1 = uc[1], u2 = uc[2], u3 = uc[3];
#pragma unroll
                for (int i = 0; i < PPW; ++i)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        acc[i][v][0] = fma(-as_f64_in_loop(xr[i][jb][v]), u0, acc[i][v][0]);
                        acc[i][v][1] = fma(-as_f64_in_loop(xr[i][jb + 1][v]), u1, acc[i][v][1]);
                        acc[i][v][2] = fma(-as_f64_in_loop(xr[i][jb + 2][v]), u2, acc[i][v][2]);
                        acc[i][v][3] = fma(-as_f64_in_loop(xr[i][jb + 3][v]), u3, acc[i][v][3]);
                    }
#pragma unroll
                for (int d = DEPTH - 1; d > 0; --d) tok[d] = tok[d - 1];
                // an output the compiler believes is made from this block's accumulators
                int made;
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    asm volatile("" : "=v"(made) : "v"(acc[PPW - 1][v][0]), "v"(acc[PPW - 1][v][1]),
                                 "v"(acc[PPW - 1][v][2]), "v"(acc[PPW - 1][v][3]));
                }
                tok[0] = made;
            }
            double t = 0.0;
#pragma unroll
            for (int i = 0; i < PPW; ++i)
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const double r = (acc[i][v][0] + acc[i][v][1]) + (acc[i][v][2] + acc[i][v][3]);
                    t = fma(r, r, t);
                }
            s[c] = t;
        }
    }

    // Balanced form of the pass for a bundle of 8 chains on 8 waves (stamps, 64 chains at C2: a
    // wave alone issues one v_fmac_f64_dpp per ~6.5 cycles, so the wave that owned a whole panel
    // x 8 chains -- 256 FMAs -- set the length of the pass while its SIMD-mate idled after 64).
    // Waves w and w + 4 both hold panel w % 4 and take 4 of its 8 chains each; the fifth panel is
    // held by every wave, one chain each: 4 + 1 (panel, chain) units per wave, 10 per SIMD.
    // sA[cc] = lane partial of chain 4 * (wave / 4) + cc on panel wave % 4; sB = of chain `wave`
    // on panel 4.  Per (chain, panel) the operations are those of partial_rss.
    __device__ __forceinline__ void partial_rss_reg_bal(const double* u, int kpad, double (&sA)[4],
                                                        double& sB) const {
        static_assert(MODE == MODE_REG && PPW == 2 && VEC == 1 && KMAX >= 16, "balanced bundles");
        typedef const __attribute__((address_space(3))) double lds_cd;
        lds_cd* ub = (lds_cd*)u;
        constexpr int NU = (KMAX + 15) / 16;
        const int half = wave >> 2;
        double urow[5][NU];
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            const int chain = c < 4 ? 4 * half + c : wave;
#pragma unroll
            for (int r = 0; r < NU; ++r) urow[c][r] = ub[chain * kpad + r * 16 + (lane & 15)];
        }
#pragma unroll
        for (int c = 0; c < 5; ++c)
#pragma unroll
            for (int r = 0; r < NU; ++r) asm volatile("" : "+v"(urow[c][r]));
        auto one = [&](auto cc) -> double {
            constexpr int c = decltype(cc)::value;
            constexpr int i = c < 4 ? 0 : 1;            // register set: own panel, fifth panel
            double acc[4] = {(double)yr[i][0], 0.0, 0.0, 0.0};
            if constexpr (sizeof(T) == 8) {
                static_for<NU>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    double xb[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) xb[q] = (double)xr[i][16 * r + q][0];
                    fmac16_rowbcast_neg(acc, urow[c][r], xb);
                });
            } else {
                static_for<KMAX>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    fmac_rowbcast_neg<j % 16, (j % 16 == 0)>(acc[j % 4], urow[c][j / 16],
                                                           as_f64_in_loop(xr[i][j][0]));
                });
            }
            const double r = (acc[0] + acc[1]) + (acc[2] + acc[3]);
            return fma(r, r, 0.0);
        };
        static_for<4>([&](auto cc) { sA[decltype(cc)::value] = one(cc); });
        sB = one(std::integral_constant<int, 4>{});
    }

    // this lane's share of sum (y - X u)^2 over the wave's panels; u_lds zero-padded to 64
    __device__ __forceinline__ double partial_rss(const double* __restrict__ u_lds) const {
        double s = 0.0;
        if constexpr (MODE == MODE_REG) {
            double acc[PPW][VEC][4];
            // u through DPP row broadcasts (fmac_rowbcast_neg above) from 16 columns on: register
            // r of this wave holds u[16 r .. 16 r + 15], repeated in its four rows of 16 lanes.
            // (8 columns: the few LDS reads are not the bound, and the DPP form costs half a
            // cycle per FMA more -- measured 3 % slower for notebook-sized single-workgroup
            // chains.)  The in-place DPP form needs its four accumulators initialised by register
            // copies first; with several rows per lane (PPW * VEC > 1: 4 copies per row) the
            // first FMA of each chain keeps the three-operand form with u from an LDS broadcast
            // read instead, which takes its start value (y or 0) as an operand.  One row per
            // lane: all columns through DPP (C2: 1.135 vs 1.165 us per iteration).
            constexpr bool USE_DPP = KMAX >= 16;
            constexpr int JD = !USE_DPP ? KMAX : (PPW * VEC > 1 ? 4 : 0);   // first DPP column
            if constexpr (JD == 0) {
#pragma unroll
                for (int i = 0; i < PPW; ++i)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        acc[i][v][0] = (double)yr[i][v];
                        acc[i][v][1] = acc[i][v][2] = acc[i][v][3] = 0.0;
                    }
            }
#pragma unroll
            for (int j = 0; j < JD; j += 4) {
                const double u0 = u_lds[j], u1 = u_lds[j + 1], u2 = u_lds[j + 2], u3 = u_lds[j + 3];
#pragma unroll
                for (int i = 0; i < PPW; ++i)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        acc[i][v][0] = fma(-as_f64_in_loop(xr[i][j][v]), u0,
                                           j == 0 ? (double)yr[i][v] : acc[i][v][0]);
                        acc[i][v][1] = fma(-as_f64_in_loop(xr[i][j + 1][v]), u1, j == 0 ? 0.0 : acc[i][v][1]);
                        acc[i][v][2] = fma(-as_f64_in_loop(xr[i][j + 2][v]), u2, j == 0 ? 0.0 : acc[i][v][2]);
                        acc[i][v][3] = fma(-as_f64_in_loop(xr[i][j + 3][v]), u3, j == 0 ? 0.0 : acc[i][v][3]);
                    }
            }
            if constexpr (USE_DPP) {
                constexpr int NU = (KMAX + 15) / 16;
                double urow[NU];
#pragma unroll
                for (int r = 0; r < NU; ++r) urow[r] = u_lds[r * 16 + (lane & 15)];
#ifndef BMC_NO_BLOCK_ASM
                if constexpr (JD == 0 && PPW * VEC == 1 && KMAX % 16 == 0 && sizeof(T) == 8) {
                    static_for<NU>([&](auto rc) {
                        constexpr int r = decltype(rc)::value;
                        double xb[16];
#pragma unroll
                        for (int q = 0; q < 16; ++q) xb[q] = (double)xr[0][16 * r + q][0];
                        fmac16_rowbcast_neg(acc[0][0], urow[r], xb);
                    });
                } else
#endif
                static_for<KMAX - JD>([&](auto jc) {
                    constexpr int j = decltype(jc)::value + JD;
#pragma unroll
                    for (int i = 0; i < PPW; ++i)
#pragma unroll
                        for (int v = 0; v < VEC; ++v)
                            fmac_rowbcast_neg<j % 16, (j == JD || j % 16 == 0)>(
                                acc[i][v][j % 4], urow[j / 16], as_f64_in_loop(xr[i][j][v]));
                });
            }
#pragma unroll
            for (int i = 0; i < PPW; ++i)
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const double r = (acc[i][v][0] + acc[i][v][1]) + (acc[i][v][2] + acc[i][v][3]);
                    s = fma(r, r, s);
                }
        } else {
            for (int q = wave; q < npl; q += nw) {
                if constexpr (MODE == MODE_LDS) {
                    s += panel_rss<T, VEC>(Xs + (size_t)q * K * RP + lane * VEC,
                                           ys + q * RP + lane * VEC, u_lds, K);
                } else {
                    const int64_t p = g + (int64_t)q * G;
                    if (q < keep)
                        s += panel_rss<T, VEC>(Xg + p * (int64_t)K * RP + lane * VEC,
                                               yg + p * RP + lane * VEC, u_lds, K);
                    else   // streamed once per iteration: do not displace the kept panels in L2
                        s += panel_rss<T, VEC, true>(Xg + p * (int64_t)K * RP + lane * VEC,
                                                     yg + p * RP + lane * VEC, u_lds, K);
                }
            }
        }
        return s;
    }
};

// ---- all-reduce of the lane partials over the chain's groups -----------------------------
// Every wave calls it (it contains the group barrier).  In wave 0 the return value is the
// chain-wide sum (identical bits in every group); `ok` is false when the spin expired.
// Other waves get an unspecified value.  Order of summation is fixed: lane by lane over the
// waves in a fixed tree, a DPP butterfly over the lanes, groups in index order.
// SINGLE = the chain lives in ONE workgroup (G == 1): nothing to exchange, the group total is
// the chain total.  A template parameter, not a run-time test, so that the multi-group code is
// byte-for-byte what it was (a run-time `if (G == 1)` cost the C2 path 4 %).
// sum_wave_slots (several chains per pass, group_allreduce_multi): `red` holds 8 per-wave slots
// per chain; the slots of waves that do not exist stay 0 (zeroed by the caller before the
// loop).  Lane w < 8 reads slot w and three DPP steps add them in a fixed tree,
// ((r0+r1)+(r2+r3)) + ((r7+r6)+(r5+r4)): no trip count, one LDS read, 4 VGPRs.
__device__ __forceinline__ double sum_wave_slots(const double* red, int lane) {
    double v = lane < 8 ? red[lane] : 0.0;
    v += dpp_mov_f64<0xB1>(v);          // quad_perm [1,0,3,2]
    v += dpp_mov_f64<0x4E>(v);          // quad_perm [2,3,0,1]
    v += dpp_mov_f64<0x141>(v);         // row_half_mirror
    return readlane_f64(v, 0);
}

template <bool LOCAL, int STRIDE = GRAN_PAIR_STRIDE>
__device__ __forceinline__ void publish_pair(gu64* gp, int g, int lane, unsigned epoch, double s) {
    if (lane < 2) {   // lane 0 the high word, lane 1 the low word: one store instruction
        const unsigned w = lane == 0 ? (unsigned)__double2hiint(s) : (unsigned)__double2loint(s);
        granule_put<LOCAL>(gp + gran_at<STRIDE>(2 * g + lane), epoch, w);
    }
}

// Publish this group's total, gather the chain's G totals and sum them in a fixed order.
// G <= 32: one level (granule pairs 0..G-1).  G > 32 (the chain spans XCDs): two levels.
// Team j = the groups g with g mod 8 = j (the hardware deals workgroups to the 8 XCDs round
// robin, so a team normally shares an XCD and `local` then means "every team is XCD-local"):
// the team's <= 32 totals are exchanged in pairs 32j .. 32j+31 and summed; the team's first
// member publishes the team total in pair 256+j at agent scope; every group gathers those 8
// pairs.  Per step every group reads one granule per lane instead of 2G/64 registers of them,
// and only 8 stores per step cross XCDs.  The order of summation depends on G only, never
// on where the groups really run.
struct NoIdleWork {
    __device__ __forceinline__ void operator()() const {}
};


// `idle` runs after this group's total is published and before the polling starts: work placed
// there is hidden by the store -> polled-load latency the group pays anyway.
// RELAY (passes that serve several chains): only the first member of each team polls the 8 team
// totals at agent scope; it then hands the chain total to its team through the team's XCD-local
// area (pair 264 + team), which the other members poll.  One more local hop, but 8 agent-scope
// pollers per chain instead of G: with 8 chains per pass the agent-scope polls of 8 x 196 waves
// made the second level 3.5x slower than it is for one chain (stamps, dev_stamps_multi.py).
// TEAMS / LOCAL: -1 = decided at run time (G > 32; the `local` argument), 0 / 1 = known to the
// caller at compile time.  The register-resident loop kernel is instantiated for chains of at
// most 32 groups and runs one of two copies of its loop, chosen once after the placement check:
// with both run-time tests and the dead two-level code out of the loop an iteration at C2 is
// 3 % shorter (1.090 -> 1.059 us, same-box A/B).
template <bool RELAY = false, typename F = NoIdleWork, int TEAMS = -1, int LOCAL = -1>
__device__ __forceinline__ double exchange_sum(double s, gu64* gp, int G, int g, int lane,
                                               unsigned epoch, bool local, bool& ok STAMP_PARAMS,
                                               F idle = F()) {
    // gp is opaque from here on (an offset of unknown value, so that it stays a global
    // pointer): the per-lane granule addresses are then computed where they are used instead
    // of being kept in VGPRs across the whole iteration loop
    size_t opaque0 = 0;
    asm volatile("" : "+s"(opaque0));
    gp += opaque0;
    const bool teams = TEAMS < 0 ? G > 32 : TEAMS != 0;
    if constexpr (LOCAL >= 0) local = LOCAL != 0;
    const int team = teams ? (g & 7) : 0, rank = teams ? (g >> 3) : g;
    const int members = teams ? ((G - team + 7) >> 3) : G;
    gu64* gp1 = gp + gran_at(64 * team);
    if (local) publish_pair<true>(gp1, rank, lane, epoch, s);
    else publish_pair<false>(gp1, rank, lane, epoch, s);
    idle();
    // (tried in round 3, for the passes whose eight leaders per CU poll at once: a fixed s_sleep
    // of 128 / 256 / 384 cycles before the first poll -- 64 chains at C2 2.333 / 2.346 / 2.382 against
    // 2.319 us per iteration without; gpurun_out/r3_ab5.log)
    gu64 x;
    ok = granule_gather1(gp1, 2 * members, epoch, lane, x STAMP_ARGS);
    GSTAMP(5);
    // (the expired spin leaves at once: as `ok ? sum : 0` the flag travelled through the whole
    // serial path as set / test / branch pairs)
#ifndef BMC_NO_EXPECT
    if (__builtin_expect(!ok, 0)) return 0.0;
    double tot = granule_sum1(x, lane);
    if (teams) {
#else
    double tot = ok ? granule_sum1(x, lane) : 0.0;
    if (teams && ok) {
#endif
        // second level: 8 team-total pairs, then 8 relay pairs, GRAN_L2_STRIDE words (512 bytes)
        // apart: each pair is written by one XCD and polled by all (bmc_launch.h)
        gu64* gp2 = gp + (size_t)256 * GRAN_PAIR_STRIDE;
        gu64* gp3 = gp2 + (size_t)(8 + team) * GRAN_L2_STRIDE;
        if (rank == 0) publish_pair<false, GRAN_L2_STRIDE>(gp2, team, lane, epoch, tot);
        if (!RELAY || rank == 0) {
            ok = granule_gather1<BMC_POLL_DEPTH2, GRAN_L2_STRIDE>(gp2, 16, epoch, lane, x STAMP_ARGS);
            if (__builtin_expect(!ok, 0)) return 0.0;
            tot = granule_sum1(x, lane);
            if (RELAY) {
                if (local) publish_pair<true, GRAN_L2_STRIDE>(gp3, 0, lane, epoch, tot);
                else publish_pair<false, GRAN_L2_STRIDE>(gp3, 0, lane, epoch, tot);
            }
        } else {
            ok = granule_gather1<2, GRAN_L2_STRIDE>(gp3, 2, epoch, lane, x STAMP_ARGS);
            if (__builtin_expect(!ok, 0)) return 0.0;
            tot = granule_sum1(x, lane);   // the relayed double itself (+ zeros)
        }
    }
    GSTAMP(8);
    return tot;
}

// LANEWISE (register residency with one row per lane: C2, notebook-sized chains, N = 100 000 x
// 32): the waves of a group do NOT reduce their lanes first.  Every wave leaves its 64 lane
// partials in its row of `red` ([8 waves][64 lanes], rows of absent waves stay 0: zeroed by the
// caller before the loop), and after the barrier wave 0 adds the 8 rows lane by lane in a fixed
// tree, ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), and runs ONE wave-level sum -- instead of a
// wave_sum per wave in front of the barrier (two of them interleaved on the SIMD that two of a
// CU's five waves share) and a second, 8-slot one behind it.  Same-box A/B (14 rounds): C2
// 1.112 -> 1.085 us, N = 629 0.599 -> 0.570, N = 100 000 x 32 2.25 -> 2.02; but C4 (two rows
// per lane) 3.26 -> 3.41 and C5 (streamed) 15.8 -> 16.1, which therefore keep the other form:
// a wave_sum per wave, 8 slots, three DPP steps in wave 0.
// PREWRITTEN (LANEWISE only): the caller has already left its lane partials in its row of `red`
// (right behind the residual pass, ahead of its other work in front of the barrier).
// `idle`: work of the leader placed between publishing the group total and the first poll.
template <bool SINGLE = false, bool LANEWISE = false, int TEAMS = -1, int LOCAL = -1, int ROLE = -1,
          bool PREWRITTEN = false, typename F = NoIdleWork>
__device__ __forceinline__ double group_allreduce(double s, double* red, gu64* gp, int G, int g,
                                                  int wave, int nw, int lane, unsigned epoch,
                                                  bool local, bool& ok STAMP_PARAMS, F idle = F()) {
    if constexpr (LANEWISE) {
        if constexpr (!PREWRITTEN) red[wave * 64 + lane] = s;
    } else if constexpr (!PREWRITTEN) {
        s = wave_sum(s);
        if (lane == 0) red[wave] = s;
    }
    GSTAMP(3);
    __syncthreads();
    ok = true;
    if (ROLE < 0 ? wave != 0 : ROLE != 0) return 0.0;   // (ROLE: the caller knows its wave's role)
    if constexpr (LANEWISE) {
        const double r0 = red[lane], r1 = red[64 + lane], r2 = red[128 + lane], r3 = red[192 + lane];
        const double r4 = red[256 + lane], r5 = red[320 + lane], r6 = red[384 + lane], r7 = red[448 + lane];
        s = wave_sum(((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)));
    } else {
        s = sum_wave_slots(red, lane);
    }
    GSTAMP(4);
    if constexpr (SINGLE) return s;
    return exchange_sum<false, F, TEAMS, LOCAL>(s, gp, G, g, lane, epoch, local, ok STAMP_ARGS, idle);
}

// ---- several chains per pass (streaming / LDS residency) ---------------------------------
// One read of a panel column feeds CPP chains' accumulators; per chain the operation order
// is exactly that of panel_rss, so a chain's bits do not depend on how many chains share
// the pass.
template <typename T, int VEC, int CPP, bool NT = false>
__device__ __forceinline__ void panel_rss_multi(const T* __restrict__ xp, const T* __restrict__ yp,
                                                const double* __restrict__ u, int kpad, int K,
                                                double (&s)[CPP]) {
    constexpr int RP = 64 * VEC;
    // reads in flight per wave, bounded so that the CPP*VEC*2 accumulators and the staged
    // columns fit the 256-VGPR budget without spilling (fewer reads matter less here: with
    // many chains per pass the loop is FMA-bound, not latency-bound)
    // (one row per lane affords 16 reads in flight even with 8 chains: C5 x 8 chains 26.2 -> 23.1 us)
    constexpr int UNB = (VEC == 1 ? 128 : 64) / (CPP * VEC);
    constexpr int UN = UNB >= 16 ? 16 : UNB >= 4 ? UNB : 4;
    // (Eight chains per pass: hipcc parks the leaders' per-iteration state in scratch around the
    // pass -- never inside these column loops -- see spill_whitelist.txt.  Scheduling fences that
    // bound the u values in flight remove the scratch and cost 4 %: measured, rejected.)
    double a0[CPP][VEC], a1[CPP][VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        const double yv = (double)yp[v];
#pragma unroll
        for (int c = 0; c < CPP; ++c) { a0[c][v] = yv; a1[c][v] = 0.0; }
    }
    // Whole blocks of 16 columns take u through DPP row broadcasts (see fmac_rowbcast_neg): one
    // ds_read_b64 per chain and block loads u[c][j .. j+15] spread over the lanes of every row,
    // instead of one LDS broadcast read per (chain, column) -- with 8 chains those reads, 2 LDS
    // cycles per double and wave, were what bound the pass (C5 x 8 chains: 2048 per panel and
    // wave = 13.6 us of LDS pipe per iteration and CU beside 16 us of memory time).  Operands
    // and per-chain operation order are those of the column loop below, bit for bit.
    const int lane = threadIdx.x & 63;
    int j = 0;
    for (; j + 16 <= K; j += 16) {
        double urow[CPP];
#pragma unroll
        for (int c = 0; c < CPP; ++c) urow[c] = u[c * kpad + j + (lane & 15)];
        // (the values are pinned where they are loaded: the DPP statements are opaque to hipcc,
        // which otherwise re-reads u from LDS in front of every one of them and sinks each
        // column's global load, with a full vmcnt(0) wait, next to its first use)
#pragma unroll
        for (int c = 0; c < CPP; ++c) asm volatile("" : "+v"(urow[c]));
        static_for<16 / UN>([&](auto bc) {
            constexpr int B = decltype(bc)::value;
            T x[UN][VEC];
#pragma unroll
            for (int q = 0; q < UN; ++q)
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                    x[q][v] = NT ? __builtin_nontemporal_load(&xp[(size_t)(j + B * UN + q) * RP + v])
                                 : xp[(size_t)(j + B * UN + q) * RP + v];
#pragma unroll
            for (int q = 0; q < UN; ++q)
#pragma unroll
                for (int v = 0; v < VEC; ++v) asm volatile("" : "+v"(x[q][v]));
            static_for<UN>([&](auto qc) {
                constexpr int Q = decltype(qc)::value;
                constexpr int NL = B * UN + Q;          // column within the block = lane of its row
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const double xd = (double)x[Q][v];
#pragma unroll
                    for (int c = 0; c < CPP; ++c)
                        fmac_rowbcast_neg<NL, (NL == 0)>((Q & 1) ? a1[c][v] : a0[c][v],
                                                                   urow[c], xd);
                }
            });
        });
    }
    for (; j < K; ++j)
#pragma unroll
        for (int c = 0; c < CPP; ++c) {
            const double u0 = u[c * kpad + j];
            // (columns past the last whole block of 16 all go to the first chain of sums, as in
            // the single-chain panel_rss, whose blocks are 16 columns too)
#pragma unroll
            for (int v = 0; v < VEC; ++v) a0[c][v] = fma(-(double)xp[(size_t)j * RP + v], u0, a0[c][v]);
        }
#pragma unroll
    for (int c = 0; c < CPP; ++c) {
        double t = 0.0;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const double r = a0[c][v] + a1[c][v];
            t = fma(r, r, t);
        }
        s[c] += t;
    }
}

// All-reduce of CPP lane partials: wave c < CPP leads chain c (sums the waves' partials of
// its chain, publishes and gathers on that chain's granules).  Every wave calls it.
template <int CPP, typename F = NoIdleWork>
__device__ __forceinline__ double group_allreduce_multi(const double (&s)[CPP], double* red,
                                                        gu64* gp_chain0, size_t chain_stride,
                                                        int G, int g, int wave, int nw, int lane,
                                                        unsigned epoch, bool local, bool& ok STAMP_PARAMS,
                                                        F idle = F()) {
#pragma unroll
    for (int c = 0; c < CPP; ++c) {
        const double t = wave_sum(s[c]);
        if (lane == 0) red[c * 8 + wave] = t;
    }
    GSTAMP(3);
    __syncthreads();
    ok = true;
    if (wave >= CPP) return 0.0;
    const double t = sum_wave_slots(red + wave * 8, lane);
    GSTAMP(4);
    return exchange_sum<(CPP >= 4)>(t, gp_chain0 + (size_t)wave * chain_stride, G, g, lane, epoch,
                                    local, ok STAMP_ARGS, idle);
}

// Balanced bundles (partial_rss_reg_bal): rows are written by the caller; this is the leader's
// half of group_allreduce_multi_lanewise.
template <int CPP, int TEAMS = -1, typename F = NoIdleWork>
__device__ __forceinline__ double group_allreduce_multi_prewritten(double* red, gu64* gp_chain0,
                                                                   size_t chain_stride, int G, int g,
                                                                   int wave, int lane, unsigned epoch,
                                                                   bool local, bool& ok STAMP_PARAMS,
                                                                   F idle = F()) {
    GSTAMP(3);
    __syncthreads();
    ok = true;
    if (wave >= CPP) return 0.0;
    const double* r = red + (size_t)wave * 512 + lane;
    const double r0 = r[0], r1 = r[64], r2 = r[128], r3 = r[192];
    const double r4 = r[256], r5 = r[320], r6 = r[384], r7 = r[448];
    const double t = wave_sum(((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)));
    GSTAMP(4);
    return exchange_sum<(CPP >= 4), F, TEAMS>(t, gp_chain0 + (size_t)wave * chain_stride, G, g, lane,
                                             epoch, local, ok STAMP_ARGS, idle);
}

// The same for register residency with one row per lane, in the lane-wise form of
// group_allreduce<LANEWISE>: every wave leaves its 64 lane partials of every chain in LDS
// ([chain][wave][lane], rows of absent waves stay 0), and leader c adds the 8 rows of its chain
// lane by lane in the fixed tree ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) and runs ONE wave-level sum
// -- instead of CPP wave sums in every wave in front of the barrier.  Operation for operation
// the reduction of the single-chain kernel, so a chain's bits do not depend on whether it
// shares its pass.
// `row` = the local panel the wave computed (its own index, or the shared last panel's), and
// [c_lo, c_hi) the chains it computed it for: rows are indexed by PANEL, as in the single-chain
// kernel, whichever wave did the work.
template <int CPP, int TEAMS = -1, typename F = NoIdleWork>
__device__ __forceinline__ double group_allreduce_multi_lanewise(const double (&s)[CPP], double* red,
                                                                 int row, int c_lo, int c_hi,
                                                                 gu64* gp_chain0, size_t chain_stride,
                                                                 int G, int g, int wave, int lane,
                                                                 unsigned epoch, bool local,
                                                                 bool& ok STAMP_PARAMS, F idle = F()) {
#pragma unroll
    for (int c = 0; c < CPP; ++c)
        if (c >= c_lo && c < c_hi) red[(c * 8 + row) * 64 + lane] = s[c];
    GSTAMP(3);
    __syncthreads();
    ok = true;
    if (wave >= CPP) return 0.0;
    const double* r = red + (size_t)wave * 512 + lane;
    const double r0 = r[0], r1 = r[64], r2 = r[128], r3 = r[192];
    const double r4 = r[256], r5 = r[320], r6 = r[384], r7 = r[448];
    const double t = wave_sum(((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)));
    GSTAMP(4);
    return exchange_sum<(CPP >= 4), F, TEAMS>(t, gp_chain0 + (size_t)wave * chain_stride, G, g, lane,
                                             epoch, local, ok STAMP_ARGS, idle);
}

}  // namespace bmc
