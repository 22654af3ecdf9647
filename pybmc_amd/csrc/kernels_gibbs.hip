// The persistent Gibbs loop (reference pybmc/inference_utils.py:39-54) for gfx950.
//
// One launch runs ALL iterations of up to 8 independent chains.  A chain is
// served by G workgroups ("groups"); group g owns row panels g, g+G, g+2G, ...
// of the rotated design matrix and keeps them on chip for the whole run:
//   MODE_REG     in VGPRs  (K <= KMAX <= 64, PPW panels per wave)  -- no memory
//                traffic at all inside the loop except the variates,
//   MODE_LDS     pinned in LDS (any K <= 256 while the group's panels fit 160 KiB),
//   MODE_STREAM  re-read from HBM/L2 every iteration (anything larger).
//
// Per iteration t.  sigma2_{t-1} = sp/g with sp = scale_post and g the Gamma variate of
// the previous iteration (:50-52); it is carried as the pair (sp, g) so that no
// division sits on the serial path:
//   wave 0      D_j = lam_j g + sp,  r_j = rsqrt(D_j)   (d_j = 1/(lam_j/s2 + 1) = sp r_j^2)
//               u_j = r_j^2 (c1_j sp + c2_j g) + sqrt(sp) r_j xi_tj
//               (the beta | sigma2 draw of :41-45 in the basis of bmc_set_prior)
//   all waves   partial rss over the wave's rows: sum (y - Xrot u)^2        (:48-51),
//               combined per group through LDS in wave order
//   wave 0      publishes the group partial as two 8-byte {epoch, 32 data bits}
//               granules, gathers the G partials of its chain (relaxed polling: the data
//               is the flag, cdna guide G16 form R2), sums them in group order,
//               sp = (nu0 s20 + rss)/2, floor sigma2 >= 1e-6                     (:50-52)
//   last wave of group 0 records u_t and sigma_t = sqrt(sp/g) off the serial path.
// Every group of a chain computes s2 and u redundantly from the same bits, so no
// broadcast step exists: one all-gather hop per iteration is the only
// inter-workgroup traffic.  Granule slots alternate by iteration parity; a group
// can be at most one iteration ahead of the slowest one, so two parities suffice.
//
// XCD-aware placement.  The grid is 8 slots x G: slot = blockIdx.x % 8 is the label
// of the blocks that (as observed, never guaranteed) share an XCD; chain c lives in
// slot c.  At start every group publishes the XCC id it really runs on
// (HW_REG_XCC_ID); if all G groups of a chain report the same XCD the chain's
// granules are exchanged through that XCD's L2 (workgroup-scope stores that stay
// in L2 + L1-bypassing loads), otherwise through the placement-independent
// agent-scope path.  The decision is data every group reads identically, so a
// wrong placement guess costs speed, never correctness.
// All spins are bounded (wall clock); on expiry the chain's status word is set and
// every group leaves the loop.
#include "bmc_dev.h"
#include "bmc_launch.h"

namespace bmc {

constexpr int MAX_KCH = 4;        // K <= 256 columns (64 per lane-chunk)
constexpr int MAX_GRAN_REG = 8;   // 2*G <= 512 granules -> G <= 256
constexpr unsigned long long SPIN_TIMEOUT_TICKS = 400000000ull;  // 4 s of s_memrealtime (100 MHz)

enum { MODE_REG = 0, MODE_LDS = 1, MODE_STREAM = 2 };

struct LdsPlan {
    size_t u, red, ctl, y, x, total;
};

__host__ __device__ inline LdsPlan lds_plan(int K, int elem, int RP, int ppg, bool lds_resident) {
    LdsPlan L;
    const size_t kp = (size_t)((K + 63) & ~63) * sizeof(double);  // zero-padded to 64 (MODE_REG reads KMAX)
    size_t o = 0;
    L.u = o;   o += kp;
    L.red = o; o += 16 * sizeof(double);
    L.ctl = o; o += 4 * sizeof(double);
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
#else
#define STAMP(i) do {} while (0)
#endif

// ---- granule exchange ----------------------------------------------------------
// LOCAL = the chain's groups were verified to share one XCD: the store stays in that
// XCD's L2 (workgroup scope: global_store sc0) and the L1-bypassing agent-scope load
// (global_load sc1) is served by the same L2.  Otherwise the store is agent scope
// (sc1, write-through) and visible to every XCD.
template <bool LOCAL>
__device__ __forceinline__ void granule_put(gu64* g, unsigned epoch, unsigned value) {
    const gu64 w = ((gu64)epoch << 32) | (gu64)value;
    if constexpr (LOCAL)
        __hip_atomic_store(g, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else
        __hip_atomic_store(g, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Gather n2 granules of `epoch`; returns false when the bounded spin expired.
__device__ __forceinline__ bool granule_gather(const gu64* gp, int n2, unsigned epoch, int lane,
                                               gu64 (&x)[MAX_GRAN_REG]) {
    unsigned long long t_start = 0;
    for (unsigned spins = 0;; ++spins) {
        bool ok = true;
#pragma unroll
        for (int r = 0; r < MAX_GRAN_REG; ++r) {
            x[r] = 0;
            if (r * 64 < n2) {
                const int idx = r * 64 + lane;
                if (idx < n2) {
                    x[r] = granule_load(gp + idx);
                    ok = ok && ((unsigned)(x[r] >> 32) == epoch);
                }
            }
        }
        if (__all(ok)) return true;
        if ((spins & 0xff) == 0xff) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (t_start == 0) t_start = now;
            else if (now - t_start > SPIN_TIMEOUT_TICKS) return false;
        }
    }
}

// even lane 2g' holds the high word of group g', odd lane the low word -> sum over g'
__device__ __forceinline__ double granule_sum(const gu64 (&x)[MAX_GRAN_REG], int n2, int lane) {
    double part = 0.0;
#pragma unroll
    for (int r = 0; r < MAX_GRAN_REG; ++r) {
        if (r * 64 < n2) {
            const int w = (int)(unsigned)x[r];
            const int other = __builtin_amdgcn_update_dpp(0, w, 0xB1, 0xf, 0xf, true);
            const double d = __hiloint2double(w, other);
            part += ((lane & 1) == 0 && r * 64 + lane < n2) ? d : 0.0;
        }
    }
    return wave_sum(part);
}

// ---- partial rss of one panel, data in memory (LDS or global) ---------------------
template <typename T, int VEC>
__device__ __forceinline__ double panel_rss(const T* __restrict__ xp, const T* __restrict__ yp,
                                            const double* __restrict__ u, int K) {
    constexpr int RP = 64 * VEC;
    double a0[VEC], a1[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) { a0[v] = (double)yp[v]; a1[v] = 0.0; }
    int j = 0;
#pragma unroll 4
    for (; j + 1 < K; j += 2) {
        const double u0 = u[j], u1 = u[j + 1];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            a0[v] = fma(-(double)xp[(size_t)j * RP + v], u0, a0[v]);
            a1[v] = fma(-(double)xp[(size_t)(j + 1) * RP + v], u1, a1[v]);
        }
    }
    if (j < K) {
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

template <typename T, int VEC, int MODE, int KMAX, int PPW>
__global__ __launch_bounds__(MODE == MODE_REG ? 512 : 1024) void gibbs_loop_kernel(GibbsArgs a) {
    constexpr int RP = 64 * VEC;
    constexpr bool LDSRES = MODE == MODE_LDS;
    static_assert(MODE != MODE_REG || VEC == 1, "register mode keeps one row per lane");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int K = a.P.k;
    const int C = a.n_chains, G = a.G;
    // nslot = 8: slot label, blocks b and b+8 share an XCD (observed); nslot = C otherwise
    const int chain = blockIdx.x % a.nslot;
    const int g = blockIdx.x / a.nslot;
    if (chain >= C) return;                // unused slot: the whole workgroup leaves
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = blockDim.x >> 6;
    const int64_t T_it = a.iters;

    const LdsPlan L = lds_plan(K, (int)sizeof(T), RP, a.panels_per_group, LDSRES);
    double* u_lds = reinterpret_cast<double*>(smem + L.u);
    double* red = reinterpret_cast<double*>(smem + L.red);  // per-wave partials (group publish)
    double* ctl = reinterpret_cast<double*>(smem + L.ctl);  // [0] sp, [1] abort, [2] local, [3] g
    T* ys = reinterpret_cast<T*>(smem + L.y);
    T* Xs = reinterpret_cast<T*>(smem + L.x);

    const int NP = a.P.npanels;
    const int npl = g < NP ? (NP - g + G - 1) / G : 0;  // panels owned by this group
    const T* Xg = reinterpret_cast<const T*>(a.P.X);
    const T* yg = reinterpret_cast<const T*>(a.P.y);

    const int kpad = (K + 63) & ~63;
    for (int j = tid; j < kpad; j += blockDim.x) u_lds[j] = 0.0;
    if (tid == 0) { ctl[0] = a.sigma2_init; ctl[1] = 0.0; ctl[2] = 0.0; ctl[3] = 1.0; }

    // ---- pin the group's panels on chip -------------------------------------------
    T xr[PPW > 0 ? PPW : 1][KMAX > 0 ? KMAX : 1];
    T yr[PPW > 0 ? PPW : 1];
    if constexpr (MODE == MODE_REG) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int q = wave + i * nw;
            const bool have = q < npl;
            const int64_t p = g + (int64_t)q * G;
#pragma unroll
            for (int j = 0; j < KMAX; ++j)
                xr[i][j] = (have && j < K) ? Xg[(p * K + j) * RP + lane] : (T)0;
            yr[i] = have ? yg[p * RP + lane] : (T)0;
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

    // ---- where do this chain's groups really run? -------------------------------------
    gu64* gr = a.gran + (size_t)chain * (2 * a.gran_stride + a.gran_stride);
    gu64* xcc_words = gr + 2 * a.gran_stride;  // [G] one word per group: 1 + XCC id
    if (wave == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf;  // HW_REG_XCC_ID[3:0]
        if (lane == 0)
            __hip_atomic_store(xcc_words + g, (gu64)(xcc + 1), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        bool same = true, expired = false;
        unsigned long long t_start = 0;
        for (unsigned spins = 0;; ++spins) {
            bool ok = true;
            same = true;
            for (int b = 0; b < G; b += 64) {
                const int idx = b + lane;
                gu64 w = xcc + 1;
                if (idx < G) w = __hip_atomic_load(xcc_words + idx, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
                ok = ok && (w != 0);
                same = same && (w == (gu64)(xcc + 1));
            }
            if (__all(ok)) break;
            if ((spins & 0xff) == 0xff) {
                const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                if (t_start == 0) t_start = now;
                else if (now - t_start > SPIN_TIMEOUT_TICKS) { expired = true; break; }
            }
        }
        const bool all_same = __all(same);
        if (lane == 0) {
            if (expired) { ctl[1] = 1.0; a.status[chain] = 1; }
            ctl[2] = (!expired && all_same && !a.force_agent_scope) ? 1.0 : 0.0;
        }
    }
    __syncthreads();
    const bool local = ctl[2] != 0.0;
    if (g == 0 && tid == 0) a.placement[chain] = local ? 1 : 0;

    const double* xi = a.xi + (int64_t)chain * T_it * K;
    const double* gam = a.gam + (int64_t)chain * T_it;
    double* uout = a.uout + (int64_t)chain * T_it * (K + 1);
    const bool recorder = (g == 0) && (wave == nw - 1);
    const int n2 = 2 * G;

    // sigma2 = sp_eff / g_eff; starts at the OLS value (inference_utils.py:37)
    double sp_eff = a.sigma2_init, g_eff = 1.0, sq_sp = sqrt(a.sigma2_init);
    double xi_next[MAX_KCH], lam_r[MAX_KCH], c1_r[MAX_KCH], c2_r[MAX_KCH];
    double gam_next = 0.0;
    if (wave == 0) {
#pragma unroll
        for (int ch = 0; ch < MAX_KCH; ++ch) {
            const int j = ch * 64 + lane;
            xi_next[ch] = (j < K && T_it > 0) ? xi[j] : 0.0;
            lam_r[ch] = j < K ? a.lam[j] : 0.0;
            c1_r[ch] = j < K ? a.c1[j] : 0.0;
            c2_r[ch] = j < K ? a.c2[j] : 0.0;
        }
        if (T_it > 0) gam_next = gam[0];
    }
    // Exchange participants are the G groups: the waves of a group combine through LDS
    // first.  (Measured: letting every wave publish its own partial removes a barrier
    // but makes the gather 1.7x longer at 160 participants -- a net loss.)

#ifdef BMC_STAMPS
    const bool stamping = a.dbg != nullptr && blockIdx.x == 0 && wave == 0;
    unsigned long long acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = 0;
    if (stamping) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last_)::"memory");
#endif
    for (int64_t t = 0; t < T_it; ++t) {
        const unsigned epoch = (unsigned)(t + 1);
        STAMP(7);
        if (wave == 0) {
#pragma unroll
            for (int ch = 0; ch < MAX_KCH; ++ch) {
                const int j = ch * 64 + lane;
                if (ch * 64 < K && j < K) {
                    const double D = fma(lam_r[ch], g_eff, sp_eff);
                    const double r = rsqrt(D);
                    const double m = fma(c2_r[ch], g_eff, c1_r[ch] * sp_eff);
                    u_lds[j] = fma(r * r, m, (sq_sp * r) * xi_next[ch]);
                }
            }
        }
        STAMP(0);
        __syncthreads();  // B1: u (and the previous s2 / abort word) visible to all waves
        if (ctl[1] != 0.0) break;
        STAMP(1);

        const double gam_t = gam_next;
        if (wave == 0 && t + 1 < T_it) {  // prefetch next iteration's variates
#pragma unroll
            for (int ch = 0; ch < MAX_KCH; ++ch) {
                const int j = ch * 64 + lane;
                if (ch * 64 < K && j < K) xi_next[ch] = xi[(t + 1) * K + j];
            }
            gam_next = gam[t + 1];
        }
        double u_rec[MAX_KCH], sp_rec = 0.0, g_rec = 1.0;
        if (recorder) {  // copy now (wave 0 rewrites u_lds after its gather); store later
#pragma unroll
            for (int ch = 0; ch < MAX_KCH; ++ch) {
                const int j = ch * 64 + lane;
                u_rec[ch] = (ch * 64 < K && j < K) ? u_lds[j] : 0.0;
            }
            sp_rec = ctl[0];
            g_rec = ctl[3];
        }

        // ---- partial rss over this group's panels ---------------------------------
        double s = 0.0;
        if constexpr (MODE == MODE_REG) {
            double acc[PPW][4];
#pragma unroll
            for (int i = 0; i < PPW; ++i) {
                acc[i][0] = (double)yr[i];
                acc[i][1] = acc[i][2] = acc[i][3] = 0.0;
            }
#pragma unroll
            for (int j = 0; j < KMAX; j += 4) {
                // four broadcast reads of u (zero beyond K), shared by the wave's panels
                const double u0 = u_lds[j], u1 = u_lds[j + 1], u2 = u_lds[j + 2], u3 = u_lds[j + 3];
#pragma unroll
                for (int i = 0; i < PPW; ++i) {
                    acc[i][0] = fma(-(double)xr[i][j], u0, acc[i][0]);
                    acc[i][1] = fma(-(double)xr[i][j + 1], u1, acc[i][1]);
                    acc[i][2] = fma(-(double)xr[i][j + 2], u2, acc[i][2]);
                    acc[i][3] = fma(-(double)xr[i][j + 3], u3, acc[i][3]);
                }
            }
#pragma unroll
            for (int i = 0; i < PPW; ++i) {
                const double r = (acc[i][0] + acc[i][1]) + (acc[i][2] + acc[i][3]);
                s = fma(r, r, s);
            }
        } else {
            for (int q = wave; q < npl; q += nw) {
                if constexpr (MODE == MODE_LDS) {
                    s += panel_rss<T, VEC>(Xs + (size_t)q * K * RP + lane * VEC,
                                           ys + q * RP + lane * VEC, u_lds, K);
                } else {
                    const int64_t p = g + (int64_t)q * G;
                    s += panel_rss<T, VEC>(Xg + p * (int64_t)K * RP + lane * VEC,
                                           yg + p * RP + lane * VEC, u_lds, K);
                }
            }
        }
        STAMP(2);
        s = wave_sum(s);
        gu64* gp = gr + (size_t)(t & 1) * a.gran_stride;
        if (lane == 0) red[wave] = s;
        __syncthreads();  // B2: group-level combine in fixed wave order, wave 0 publishes
        if (wave == 0) {
            s = red[0];
            for (int w = 1; w < nw; ++w) s += red[w];
            if (lane == 0) {
                if (local) {
                    granule_put<true>(gp + 2 * g, epoch, (unsigned)__double2hiint(s));
                    granule_put<true>(gp + 2 * g + 1, epoch, (unsigned)__double2loint(s));
                } else {
                    granule_put<false>(gp + 2 * g, epoch, (unsigned)__double2hiint(s));
                    granule_put<false>(gp + 2 * g + 1, epoch, (unsigned)__double2loint(s));
                }
            }
        }
        STAMP(3);
        if (recorder) {
            // row t = [u_t, .]; sigma of the PREVIOUS row (its sp, g were final at B1)
#pragma unroll
            for (int ch = 0; ch < MAX_KCH; ++ch) {
                const int j = ch * 64 + lane;
                if (ch * 64 < K && j < K) uout[t * (K + 1) + j] = u_rec[ch];
            }
            if (lane == 0 && t > 0) uout[(t - 1) * (K + 1) + K] = sqrt(sp_rec / g_rec);
        }
        STAMP(4);

        if (wave == 0) {
            gu64 x[MAX_GRAN_REG];
            const bool got = granule_gather(gp, n2, epoch, lane, x);
            STAMP(5);
            if (!got) {
                if (lane == 0) { ctl[1] = 1.0; a.status[chain] = 1; }
            } else {
                const double rss = granule_sum(x, n2, lane);
                // sigma2 | beta = scale_post / g_t, floored at 1e-6            (:50-52)
                const double scale_post = (a.nu0_s20 + rss) * 0.5;
                const bool floor_hit = scale_post < 1e-6 * gam_t;
                sp_eff = floor_hit ? 1e-6 : scale_post;
                g_eff = floor_hit ? 1.0 : gam_t;
                sq_sp = sqrt(sp_eff);
                if (lane == 0) { ctl[0] = sp_eff; ctl[3] = g_eff; }
            }
            STAMP(6);
        }
    }
#ifdef BMC_STAMPS
    if (stamping && lane == 0)
        for (int i = 0; i < 8; ++i) a.dbg[i] = (long long)acc_[i];
#endif
    __syncthreads();
    if (recorder && lane == 0 && T_it > 0 && ctl[1] == 0.0)
        uout[(T_it - 1) * (K + 1) + K] = sqrt(ctl[0] / ctl[3]);
}

size_t gibbs_lds_bytes(const GibbsArgs& a) {
    return lds_plan(a.P.k, a.P.f32 ? 4 : 8, 64 * a.P.vec, a.panels_per_group, a.mode == MODE_LDS).total;
}

template <typename T, int VEC, int MODE, int KMAX, int PPW>
static hipError_t gibbs_launch_one(const GibbsArgs& a, hipStream_t s) {
    const size_t lds = gibbs_lds_bytes(a);
    hipError_t e = hipFuncSetAttribute((const void*)gibbs_loop_kernel<T, VEC, MODE, KMAX, PPW>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((gibbs_loop_kernel<T, VEC, MODE, KMAX, PPW>), dim3(a.nslot * a.G),
                       dim3(64 * a.waves), lds, s, a);
    return hipGetLastError();
}

template <typename T, int KMAX>
static hipError_t gibbs_launch_reg(const GibbsArgs& a, hipStream_t s) {
    switch (a.reg_ppw) {
        // panel data is held as f64: PPW * KMAX * 2 <= 128 VGPRs keeps the kernel spill-free
        case 1: return gibbs_launch_one<T, 1, MODE_REG, KMAX, 1>(a, s);
        case 2:
            if constexpr (KMAX <= 32) return gibbs_launch_one<T, 1, MODE_REG, KMAX, 2>(a, s);
            break;
        case 4:
            if constexpr (KMAX <= 16) return gibbs_launch_one<T, 1, MODE_REG, KMAX, 4>(a, s);
            break;
    }
    return hipErrorInvalidValue;
}

template <typename T>
static hipError_t gibbs_launch_t(const GibbsArgs& a, hipStream_t s) {
    if (a.mode == MODE_REG) {
        if (a.P.vec != 1 || a.waves > 8) return hipErrorInvalidValue;
        if (a.P.k <= 8) return gibbs_launch_reg<T, 8>(a, s);
        if (a.P.k <= 16) return gibbs_launch_reg<T, 16>(a, s);
        if (a.P.k <= 32) return gibbs_launch_reg<T, 32>(a, s);
        if (a.P.k <= 64) return gibbs_launch_reg<T, 64>(a, s);
        return hipErrorInvalidValue;
    }
#define BMC_MEM(V)                                                                   \
    (a.mode == MODE_LDS ? gibbs_launch_one<T, V, MODE_LDS, 0, 0>(a, s)               \
                        : gibbs_launch_one<T, V, MODE_STREAM, 0, 0>(a, s))
    switch (a.P.vec) {
        case 1: return BMC_MEM(1);
        case 2: return BMC_MEM(2);
        case 4:
            if constexpr (sizeof(T) == 4) return BMC_MEM(4);
            break;
    }
#undef BMC_MEM
    return hipErrorInvalidValue;
}

int gibbs_reg_capacity(int k, int f32, int ppw) {
    // data VGPRs per lane (held as f64 for both storage types): ppw * kmax * 2 <= 128
    (void)f32;
    if (k > 64) return 0;
    const int kmax = k <= 8 ? 8 : k <= 16 ? 16 : k <= 32 ? 32 : 64;
    return ppw * kmax * 2 <= 128 ? 1 : 0;
}

hipError_t launch_gibbs(const GibbsArgs& a, hipStream_t s) {
    if (a.P.k > 64 * MAX_KCH || a.G > 32 * MAX_GRAN_REG || a.G < 1 || a.waves < 1 ||
        a.waves > 16 || a.n_chains < 1 || a.n_chains > a.nslot || a.nslot > 256)
        return hipErrorInvalidValue;
    return a.P.f32 ? gibbs_launch_t<float>(a, s) : gibbs_launch_t<double>(a, s);
}

}  // namespace bmc
