// The persistent Gibbs loop (reference pybmc/inference_utils.py:39-54) for gfx950.
//
// One launch runs ALL iterations of up to C independent chains.  A chain is
// served by G workgroups ("groups"); group g owns row panels g, g+G, g+2G, ...
// of the rotated design matrix and keeps them pinned in LDS for the whole run
// when they fit (RESIDENT), otherwise streams them from HBM/L2 every iteration.
//
// Per iteration t (sigma2 = s2 from iteration t-1):
//   wave 0      u_j = d_j (c1_j + c2_j/s2) + sqrt(d_j) xi_tj,  d_j = 1/(lam_j/s2 + 1)
//               (the beta | sigma2 draw of :41-45 in the basis of bmc_set_prior)
//   all waves   partial rss over the group's rows: sum (y - Xrot u)^2       (:48-51)
//   wave 0      publishes the group partial as two 8-byte {epoch, 32 data bits}
//               granules, gathers the G partials of its chain (relaxed agent-scope
//               polling: the data is the flag, cdna guide G16 form R2), sums them in
//               group order, draws s2 = max(1/(g_t/scale_post), 1e-6)             (:50-52)
// Every group of a chain computes s2 and u redundantly from the same bits, so no
// broadcast step exists: one all-gather hop per iteration is the only
// inter-workgroup traffic.  Granule slots alternate by iteration parity; a group
// can be at most one iteration ahead of the slowest one, so two parities suffice.
// All spins are bounded (wall clock); on expiry the chain's status word is set and
// every group leaves the loop.
#include "bmc_dev.h"
#include "bmc_launch.h"

namespace bmc {

constexpr int MAX_KCH = 4;        // K <= 256 columns (64 per lane-chunk)
constexpr int MAX_GRAN_REG = 8;   // 2*G <= 512 granules -> G <= 256
constexpr unsigned long long SPIN_TIMEOUT_TICKS = 400000000ull;  // 4 s of s_memrealtime (100 MHz)

struct LdsPlan {
    size_t u, lam, c1, c2, red, ctl, y, x, total;
};

__host__ __device__ inline LdsPlan lds_plan(int K, int elem, int RP, int ppg, bool resident) {
    LdsPlan L;
    const size_t kp = (size_t)((K + 1) & ~1) * sizeof(double);
    size_t o = 0;
    L.u = o;   o += kp;
    L.lam = o; o += kp;
    L.c1 = o;  o += kp;
    L.c2 = o;  o += kp;
    L.red = o; o += 16 * sizeof(double);
    L.ctl = o; o += 4 * sizeof(double);
    L.y = o;
    if (resident) o += (size_t)ppg * RP * elem;
    o = (o + 15) & ~(size_t)15;
    L.x = o;
    if (resident) o += (size_t)ppg * K * RP * elem;
    L.total = o;
    return L;
}

template <typename T, int VEC>
__device__ __forceinline__ double panel_rss(const T* __restrict__ xp, const T* __restrict__ yp,
                                            const double* __restrict__ u, int K) {
    constexpr int RP = 64 * VEC;
    double acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = (double)yp[v];
#pragma unroll 8
    for (int j = 0; j < K; ++j) {
        const double uj = u[j];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = fma(-(double)xp[(size_t)j * RP + v], uj, acc[v]);
    }
    double s = 0.0;
#pragma unroll
    for (int v = 0; v < VEC; ++v) s = fma(acc[v], acc[v], s);
    return s;
}

template <typename T, int VEC, bool RESIDENT>
__global__ __launch_bounds__(1024) void gibbs_loop_kernel(GibbsArgs a) {
    constexpr int RP = 64 * VEC;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int K = a.P.k;
    const int C = a.n_chains, G = a.G;
    const int chain = blockIdx.x % C;  // blocks b and b+8 share an XCD: C = 8 puts a chain on one XCD
    const int g = blockIdx.x / C;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = blockDim.x >> 6;
    const int64_t T_it = a.iters;

    const LdsPlan L = lds_plan(K, (int)sizeof(T), RP, a.panels_per_group, RESIDENT);
    double* u_lds = reinterpret_cast<double*>(smem + L.u);
    double* lamS = reinterpret_cast<double*>(smem + L.lam);
    double* c1S = reinterpret_cast<double*>(smem + L.c1);
    double* c2S = reinterpret_cast<double*>(smem + L.c2);
    double* red = reinterpret_cast<double*>(smem + L.red);
    double* ctl = reinterpret_cast<double*>(smem + L.ctl);  // [0] s2, [1] abort (0/1)
    T* ys = reinterpret_cast<T*>(smem + L.y);
    T* Xs = reinterpret_cast<T*>(smem + L.x);

    const int NP = a.P.npanels;
    const int npl = g < NP ? (NP - g + G - 1) / G : 0;  // panels owned by this group
    const T* Xg = reinterpret_cast<const T*>(a.P.X);
    const T* yg = reinterpret_cast<const T*>(a.P.y);

    for (int j = tid; j < K; j += blockDim.x) {
        lamS[j] = a.lam[j];
        c1S[j] = a.c1[j];
        c2S[j] = a.c2[j];
    }
    if (tid == 0) { ctl[0] = a.sigma2_init; ctl[1] = 0.0; }
    if constexpr (RESIDENT) {
        // pin this group's panels: panel p is one contiguous block of K*RP elements
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
    __syncthreads();

    const double* xi = a.xi + (int64_t)chain * T_it * K;
    const double* gam = a.gam + (int64_t)chain * T_it;
    double* uout = a.uout + (int64_t)chain * T_it * (K + 1);
    gu64* gr = a.gran + (size_t)chain * 2 * a.gran_stride;
    const bool recorder = (g == 0) && (wave == nw - 1);
    const int n2 = 2 * G;

    double s2 = a.sigma2_init, inv_s2 = 1.0 / s2;
    double xi_next[MAX_KCH];
    double gam_next = 0.0;
    if (wave == 0) {
#pragma unroll
        for (int ch = 0; ch < MAX_KCH; ++ch) {
            const int j = ch * 64 + lane;
            xi_next[ch] = (j < K && T_it > 0) ? xi[j] : 0.0;
        }
        if (T_it > 0) gam_next = gam[0];
    }

    for (int64_t t = 0; t < T_it; ++t) {
        const unsigned epoch = (unsigned)(t + 1);
        if (wave == 0) {
#pragma unroll
            for (int ch = 0; ch < MAX_KCH; ++ch) {
                const int j = ch * 64 + lane;
                if (ch * 64 < K && j < K) {
                    const double q = fma(lamS[j], inv_s2, 1.0);
                    const double rs = rsqrt(q);  // sqrt(d_j)
                    const double d = rs * rs;
                    u_lds[j] = fma(d, fma(c2S[j], inv_s2, c1S[j]), rs * xi_next[ch]);
                }
            }
        }
        __syncthreads();  // B1: u (and the previous s2 / abort word) visible to all waves
        if (ctl[1] != 0.0) break;

        const double gam_t = gam_next;
        if (wave == 0 && t + 1 < T_it) {  // prefetch next iteration's variates
#pragma unroll
            for (int ch = 0; ch < MAX_KCH; ++ch) {
                const int j = ch * 64 + lane;
                if (ch * 64 < K && j < K) xi_next[ch] = xi[(t + 1) * K + j];
            }
            gam_next = gam[t + 1];
        }
        if (recorder) {
            for (int j = lane; j < K; j += 64) uout[t * (K + 1) + j] = u_lds[j];
            if (lane == 0 && t > 0) uout[(t - 1) * (K + 1) + K] = sqrt(ctl[0]);
        }

        // ---- partial rss over this group's panels ---------------------------------
        double s = 0.0;
        for (int q = wave; q < npl; q += nw) {
            if constexpr (RESIDENT) {
                s += panel_rss<T, VEC>(Xs + (size_t)q * K * RP + lane * VEC,
                                       ys + q * RP + lane * VEC, u_lds, K);
            } else {
                const int64_t p = g + (int64_t)q * G;
                s += panel_rss<T, VEC>(Xg + p * (int64_t)K * RP + lane * VEC,
                                       yg + p * RP + lane * VEC, u_lds, K);
            }
        }
        s = wave_sum(s);
        if (lane == 0) red[wave] = s;
        __syncthreads();  // B2

        if (wave == 0) {
            double tot = red[0];
            for (int w = 1; w < nw; ++w) tot += red[w];
            gu64* gp = gr + (size_t)(t & 1) * a.gran_stride;
            if (lane == 0) {
                granule_store(gp + 2 * g, epoch, (unsigned)__double2hiint(tot));
                granule_store(gp + 2 * g + 1, epoch, (unsigned)__double2loint(tot));
            }
            // ---- gather the chain's G partials -------------------------------------
            gu64 x[MAX_GRAN_REG];
            unsigned long long t_start = 0;
            bool expired = false;
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
                if (__all(ok)) break;
                if ((spins & 0xff) == 0xff) {
                    const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                    if (t_start == 0) t_start = now;
                    else if (now - t_start > SPIN_TIMEOUT_TICKS) { expired = true; break; }
                }
            }
            if (expired) {
                if (lane == 0) { ctl[1] = 1.0; a.status[chain] = 1; }
            } else {
                // even lane 2g' holds the high word of group g', odd lane the low word
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
                const double rss = wave_sum(part);
                // sigma2 | beta: 1 / Gamma(shape, 1/scale_post) = scale_post / g_t  (:50-52)
                const double scale_post = (a.nu0_s20 + rss) * 0.5;
                const double s2_raw = scale_post / gam_t;
                const double inv_raw = gam_t / scale_post;
                const bool floor_hit = s2_raw < 1e-6;
                s2 = floor_hit ? 1e-6 : s2_raw;
                inv_s2 = floor_hit ? (1.0 / 1e-6) : inv_raw;
                if (lane == 0) ctl[0] = s2;
            }
        }
    }
    __syncthreads();
    if (recorder && lane == 0 && T_it > 0 && ctl[1] == 0.0)
        uout[(T_it - 1) * (K + 1) + K] = sqrt(ctl[0]);
}

size_t gibbs_lds_bytes(const GibbsArgs& a) {
    return lds_plan(a.P.k, a.P.f32 ? 4 : 8, 64 * a.P.vec, a.panels_per_group, a.resident != 0).total;
}

template <typename T, int VEC, bool RES>
static hipError_t gibbs_launch_one(const GibbsArgs& a, hipStream_t s) {
    const size_t lds = gibbs_lds_bytes(a);
    hipError_t e = hipFuncSetAttribute((const void*)gibbs_loop_kernel<T, VEC, RES>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((gibbs_loop_kernel<T, VEC, RES>), dim3(a.n_chains * a.G),
                       dim3(64 * a.waves), lds, s, a);
    return hipGetLastError();
}

template <typename T, int VEC>
static hipError_t gibbs_launch_res(const GibbsArgs& a, hipStream_t s) {
    return a.resident ? gibbs_launch_one<T, VEC, true>(a, s) : gibbs_launch_one<T, VEC, false>(a, s);
}

hipError_t launch_gibbs(const GibbsArgs& a, hipStream_t s) {
    if (a.P.k > 64 * MAX_KCH || a.G > 32 * MAX_GRAN_REG || a.G < 1 || a.waves < 1 || a.waves > 16)
        return hipErrorInvalidValue;
    if (a.P.f32) {
        switch (a.P.vec) {
            case 1: return gibbs_launch_res<float, 1>(a, s);
            case 2: return gibbs_launch_res<float, 2>(a, s);
            case 4: return gibbs_launch_res<float, 4>(a, s);
        }
    } else {
        switch (a.P.vec) {
            case 1: return gibbs_launch_res<double, 1>(a, s);
            case 2: return gibbs_launch_res<double, 2>(a, s);
        }
    }
    return hipErrorInvalidValue;
}

}  // namespace bmc
