// Posterior-predictive kernels (reference pybmc/sampling_utils.py:40-84 and :18-37).
//
//   predict_weights   Wt[s][m] = sum_i theta[s][i] Vt_hat[i][m] + 1/Km,  sig[s] = theta[s][k]   (:60-67)
//   predict_gemm      R[p][s]  = sum_m preds[p][m] Wt[s][m] + z[p][s] sig[s]                    (:70-77)
//                     v_mfma_f64_16x16x4_f64, noise fused into the epilogue
//   predict_orderstat per point p: sort the S draws (bitonic, LDS), interpolate the requested
//                     order statistics like numpy's linear method (:80-82), count coverage
//                     hits sorted[lo] <= truth <= sorted[hi]                                   (:24-34)
//
// Device layout: R is [M][S_pad] (draws of one point contiguous), i.e. the reference's
// (S, M) array rndm_m in Fortran order -- the order statistics read whole rows.
#include "bmc_dev.h"
#include "bmc_launch.h"

namespace bmc {

using f64x4 = __attribute__((ext_vector_type(4))) double;

// ------------------------------------------------------------------ weights
__global__ __launch_bounds__(256) void predict_weights_kernel(
    const double* __restrict__ theta, const double* __restrict__ Vt, int32_t S, int32_t k,
    int32_t Km, int32_t S_pad, int32_t Km_pad, double* __restrict__ Wt,
    double* __restrict__ sig) {
    const int64_t total = (int64_t)S_pad * Km_pad;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const double w0 = 1.0 / (double)Km;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int32_t s = (int32_t)(e / Km_pad), m = (int32_t)(e - (int64_t)s * Km_pad);
        double v = 0.0;
        if (s < S && m < Km) {
            const double* th = theta + (int64_t)s * (k + 1);
            for (int i = 0; i < k; ++i) v = fma(th[i], Vt[(size_t)i * Km + m], v);
            v += w0;
        }
        Wt[e] = v;
        if (m == 0) sig[s] = s < S ? theta[(int64_t)s * (k + 1) + k] : 0.0;
    }
}

// --------------------------------------------------------------------- GEMM
// Workgroup tile 64 points x 64 draws, 4 waves; wave w owns points [16w, 16w+16) x 64 draws
// (4 MFMA tiles).  K-loop in slabs of 32 models staged through LDS with a row stride of
// 34 doubles: a 32-lane half reads 16 rows x 2 consecutive k = banks {4r+2k, 4r+2k+1}, all
// distinct -> conflict-free ds_read_b64.  MFMA maps (f64 form, cdna guide section 3):
// A[i = l&15][k = l>>4], B[k = l>>4][j = l&15], D: col = l&15, row = (l>>4) + 4*reg.
constexpr int PG_KT = 32, PG_LD = 34;

__device__ __forceinline__ void pg_noise(uint64_t e, uint32_t k0, uint32_t k1, double& z0,
                                         double& z1) {
    const u32x4 r = philox4x32_10(u32x4{(uint32_t)e, (uint32_t)(e >> 32), STREAM_PRED_NORMAL, 0u},
                                  k0, k1);
    const double u1 = u53_open0(r.x, r.y), u2 = u53_open0(r.z, r.w);
    const double rad = sqrt(-2.0 * log(u1));
    double sn, cs;
    sincospi(2.0 * u2, &sn, &cs);
    z0 = rad * cs;
    z1 = rad * sn;
}

__global__ __launch_bounds__(256) void predict_gemm_kernel(
    const double* __restrict__ preds, int64_t M, int32_t Km, const double* __restrict__ Wt,
    const double* __restrict__ sig, int32_t S, int32_t S_pad, int32_t Km_pad, uint64_t seed,
    const double* __restrict__ noise_replay, double* __restrict__ R) {
    __shared__ double As[64 * PG_LD];
    __shared__ double Bs[64 * PG_LD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t p0 = (int64_t)blockIdx.y * 64;
    const int32_t s0 = blockIdx.x * 64;
    const int cl = lane & 15, kq = lane >> 4;
    f64x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f64x4{0.0, 0.0, 0.0, 0.0};

    for (int m0 = 0; m0 < Km_pad; m0 += PG_KT) {
        for (int e = tid; e < 64 * PG_KT; e += 256) {
            const int r = e / PG_KT, c = e % PG_KT;
            const int64_t p = p0 + r;
            const int m = m0 + c;
            As[r * PG_LD + c] = (p < M && m < Km) ? preds[p * Km + m] : 0.0;
            Bs[r * PG_LD + c] = (m < Km_pad) ? Wt[(int64_t)(s0 + r) * Km_pad + m] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < PG_KT / 4; ++kk) {
            const int kc = kk * 4 + kq;
            const double a = As[(16 * wave + cl) * PG_LD + kc];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const double b = Bs[(16 * t + cl) * PG_LD + kc];
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int32_t s = s0 + 16 * t + cl;
        const double sg = sig[s];
#pragma unroll
        for (int h = 0; h < 2; ++h) {  // registers (2h, 2h+1) = points (pe, pe+4): one Box-Muller pair
            const int64_t pe = p0 + 16 * wave + kq + 8 * h;
            double z0 = 0.0, z1 = 0.0;
            if (noise_replay == nullptr) {
                if (s < S && pe < M) pg_noise((uint64_t)pe * (uint64_t)S + (uint64_t)s, k0, k1, z0, z1);
            } else if (s < S) {
                if (pe < M) z0 = noise_replay[(int64_t)s * M + pe];
                if (pe + 4 < M) z1 = noise_replay[(int64_t)s * M + pe + 4];
            }
            if (s < S) {
                if (pe < M) R[pe * S_pad + s] = fma(z0, sg, acc[t][2 * h]);
                if (pe + 4 < M) R[(pe + 4) * S_pad + s] = fma(z1, sg, acc[t][2 * h + 1]);
            }
        }
    }
}

// ----------------------------------------------------------- order statistics
// One workgroup sorts the S draws of one point in LDS (bitonic network over the next power of
// two, padded with +inf) and reads off what is asked.  numpy's linear interpolation:
//   lerp(a, b, t) = t >= 0.5 ? b - (b - a) * (1 - t) : a + (b - a) * t
template <int NSORT>
__global__ __launch_bounds__(1024) void predict_orderstat_kernel(
    const double* __restrict__ R, int32_t S, int32_t S_pad, int64_t M,
    const int32_t* __restrict__ q_index, const double* __restrict__ q_gamma, int32_t n_q,
    const double* __restrict__ truth, const int32_t* __restrict__ cov_lo,
    const int32_t* __restrict__ cov_hi, int32_t n_cov, double* __restrict__ bands,
    unsigned long long* __restrict__ hits) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* buf = reinterpret_cast<double*>(smem_raw);
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int64_t p = blockIdx.x; p < M; p += gridDim.x) {
        const double* row = R + p * S_pad;
        for (int i = tid; i < NSORT; i += nt) buf[i] = i < S ? row[i] : __builtin_inf();
        __syncthreads();
        for (int k = 2; k <= NSORT; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < NSORT / 2; t += nt) {
                    const int pos = 2 * t - (t & (j - 1));
                    const double a = buf[pos], b = buf[pos + j];
                    const bool up = (pos & k) == 0;
                    if ((a > b) == up) {
                        buf[pos] = b;
                        buf[pos + j] = a;
                    }
                }
                __syncthreads();
            }
        }
        if (tid < n_q) {
            const int lo = q_index[tid];
            const int hi = lo + 1 < S ? lo + 1 : S - 1;
            const double a = buf[lo], b = buf[hi], t = q_gamma[tid];
            const double diff = b - a;
            bands[(int64_t)tid * M + p] = t >= 0.5 ? b - diff * (1.0 - t) : a + diff * t;
        }
        if (truth != nullptr && tid >= 64 && tid - 64 < n_cov) {
            const int c = tid - 64;
            const double y = truth[p];
            if (buf[cov_lo[c]] <= y && y <= buf[cov_hi[c]]) atomicAdd(&hits[c], 1ull);
        }
        __syncthreads();
    }
}

hipError_t launch_predict(const PredictArgs& a, hipStream_t s) {
    {
        const int64_t total = (int64_t)a.S_pad * a.Km_pad;
        int64_t blocks = (total + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(predict_weights_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a.theta,
                           a.Vt, a.S, a.k, a.Km, a.S_pad, a.Km_pad, a.Wt, a.sig);
    }
    {
        dim3 grid(a.S_pad / 64, (unsigned)((a.M + 63) / 64));
        hipLaunchKernelGGL(predict_gemm_kernel, grid, dim3(256), 0, s, a.preds, a.M, a.Km, a.Wt,
                           a.sig, a.S, a.S_pad, a.Km_pad, a.seed, a.noise_replay, a.R);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (a.n_q > 0 || a.n_cov > 0) {
        int nsort = 64;
        while (nsort < a.S) nsort <<= 1;
        int64_t blocks = a.M < 2048 ? a.M : 2048;
        if (blocks < 1) blocks = 1;
        const size_t lds = (size_t)nsort * sizeof(double);
#define BMC_OS(NS)                                                                            \
    do {                                                                                      \
        e = hipFuncSetAttribute((const void*)predict_orderstat_kernel<NS>,                    \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);        \
        if (e != hipSuccess) return e;                                                        \
        hipLaunchKernelGGL((predict_orderstat_kernel<NS>), dim3((unsigned)blocks),            \
                           dim3(NS / 2 < 1024 ? (NS / 2 < 128 ? 128 : NS / 2) : 1024), lds, s, \
                           (const double*)a.R, a.S, a.S_pad, a.M, a.q_index, a.q_gamma, a.n_q, \
                           a.truth, a.cov_lo, a.cov_hi, a.n_cov, a.bands, a.hits);            \
    } while (0)
        switch (nsort) {
            case 64: BMC_OS(64); break;
            case 128: BMC_OS(128); break;
            case 256: BMC_OS(256); break;
            case 512: BMC_OS(512); break;
            case 1024: BMC_OS(1024); break;
            case 2048: BMC_OS(2048); break;
            case 4096: BMC_OS(4096); break;
            case 8192: BMC_OS(8192); break;
            case 16384: BMC_OS(16384); break;
            default: return hipErrorInvalidValue;
        }
#undef BMC_OS
    }
    return hipGetLastError();
}

}  // namespace bmc
