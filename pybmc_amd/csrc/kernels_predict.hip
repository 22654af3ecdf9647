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
// Wt[s][m] = sum_i theta[s][i] Vt[i][m] + 1/Km.  One workgroup = 64 model columns x 16 draws:
// lane = column (the Vt row segment is one coalesced 512-byte read per i), each of the 4 waves
// keeps 4 draws, whose theta[s][i] are wave-uniform (scalar loads).  Per output the order of
// operations is fixed: four chains over i mod 4, then ((v0 + v1) + (v2 + v3)) + 1/Km.
// (Round 2 computed one output per thread, 256 dependent loads each: 0.49 ms at C5 for 1.3 GFLOP.)
constexpr int PW_S = 16;
__global__ __launch_bounds__(256) void predict_weights_kernel(
    const double* __restrict__ theta, const double* __restrict__ Vt, int32_t S, int32_t k,
    int32_t Km, int32_t S_pad, int32_t Km_pad, double* __restrict__ Wt,
    double* __restrict__ sig) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = blockIdx.x * 64 + lane;
    const int sb = blockIdx.y * PW_S + wave * 4;      // this wave's 4 draws
    const bool mok = m < Km;
    const int mc = mok ? m : Km - 1;                  // clamped: loads stay in bounds
    double v[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) v[r][c] = 0.0;
    const double* th[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) th[r] = theta + (int64_t)(sb + r < S ? sb + r : S - 1) * (k + 1);
    int i = 0;
    for (; i + 3 < k; i += 4) {
        double x[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) x[c] = Vt[(size_t)(i + c) * Km + mc];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) v[r][c] = fma(th[r][i + c], x[c], v[r][c]);
    }
    for (; i < k; ++i) {
        const double x = Vt[(size_t)i * Km + mc];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r][0] = fma(th[r][i], x, v[r][0]);
    }
    const double w0 = 1.0 / (double)Km;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int sd = sb + r;
        if (sd < S_pad) {
            if (m < Km_pad)
                Wt[(int64_t)sd * Km_pad + m] =
                    (sd < S && mok) ? ((v[r][0] + v[r][1]) + (v[r][2] + v[r][3])) + w0 : 0.0;
            if (m == 0) sig[sd] = sd < S ? theta[(int64_t)sd * (k + 1) + k] : 0.0;
        }
    }
}

// --------------------------------------------------------------------- GEMM
// Workgroup tile 64 points x 64 draws, 4 waves; wave w owns points [16w, 16w+16) x 64 draws
// (4 MFMA tiles).  K-loop in slabs of 16 models, double-buffered in LDS: the next slab's global
// reads are issued before the current slab's MFMAs and written to the other buffer after
// them, so there is one barrier per slab and the load latency sits behind the MFMAs.
// LDS rows have a stride of 18 doubles: a 32-lane half of a ds_read_b64 reads 16 rows x 2
// consecutive k = dword banks 36 r + 2 k (+0, +1) mod 64, all distinct -> conflict-free.
// MFMA maps (f64 form, cdna guide section 3): A[i = l&15][k = l>>4], B[k = l>>4][j = l&15],
// D: col = l&15, row = (l>>4) + 4*reg.
//
// What bounds it (round 3, scripts/micro/mfma_valu_overlap.hip): on gfx950 a saturated
// v_mfma_f64_16x16x4_f64 stream (64 cycles per instruction and SIMD: 72-77 TF of the datasheet's
// 78.6 in a 1 ms burst, reached with two MFMA waves per SIMD) does NOT overlap with vector-ALU work of
// other waves on the same SIMD -- f64, f32 and integer alike: 2 MFMA + 2 VALU waves per SIMD take
// the SUM of their times, to the percent.  The kernel's time is therefore
// (MFMA instructions x 64 + VALU instructions x ~4) cycles per SIMD, and the lever is the VALU
// instruction count: round 2 executed 3 830 vector instructions per wave beside its 272 MFMAs
// (accumulators copied VGPR <-> AGPR around every slab, 64-bit index products and bounds tests in
// every fetch, un-fused polynomial arithmetic in the Box-Muller transform).  Now: operands
// padded to whole tiles (no bounds tests anywhere in the kernel), per-thread base pointers
// advanced by a constant, MFMAs in VGPR form (-amdgpu-mfma-vgpr-form), the last slab trimmed to
// the k-steps that exist, fused polynomials (bmc_math.h).
constexpr int PG_KT = 16, PG_LD = 18, PG_TM = 64;
#ifndef BMC_PG_ST
#define BMC_PG_ST 16
#endif
constexpr unsigned PG_ST = BMC_PG_ST;   // super-tile edge, in tiles (predict_gemm_kernel)

__device__ __forceinline__ void pg_noise(uint64_t e, uint32_t k0, uint32_t k1, double& z0,
                                         double& z1) {
    const u32x4 r = philox4x32_10(u32x4{(uint32_t)e, (uint32_t)(e >> 32), STREAM_PRED_NORMAL, 0u},
                                  k0, k1);
    box_muller_pair(u53_open0(r.x, r.y), u53_open0(r.z, r.w), z0, z1);   // bmc_math.h
}

// preds [M][Km] -> P [M_pad][Km_pad], zero in the padding (rows of whole tiles, columns of whole
// k-steps): the GEMM then reads and writes without a single bounds test
__global__ __launch_bounds__(256) void predict_pad_kernel(const double* __restrict__ preds, int64_t M,
                                                          int32_t Km, int64_t M_pad, int32_t Km_pad,
                                                          double* __restrict__ P) {
    const int64_t total = M_pad * Km_pad;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t p = e / Km_pad;
        const int32_t m = (int32_t)(e - p * Km_pad);
        P[e] = (p < M && m < Km) ? preds[p * Km + m] : 0.0;
    }
}

__global__ __launch_bounds__(256) void predict_gemm_kernel(
    const double* __restrict__ P, const double* __restrict__ Wt, const double* __restrict__ sig,
    int64_t M, int32_t S, int32_t S_pad, int32_t Km_pad, uint64_t seed,
    const double* __restrict__ noise_replay, double* __restrict__ R, uint32_t ntx, uint32_t nty,
    uint32_t nsx) {
    __shared__ double As[2 * PG_TM * PG_LD];
    __shared__ double Bs[2 * PG_TM * PG_LD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware tile order.  The hardware deals workgroups to the 8 XCDs round robin, each with
    // its own 4 MB L2: in plain row-major order the tiles that share a slab of preds or of the
    // weights run on 8 different XCDs and every L2 streams both matrices (18 GB fetched per launch
    // at C5 for 0.12 GB of operands).  Here XCD x = b mod 8 works through super-tiles x, x + 8, ...
    // of PG_ST x PG_ST tiles, one after the other: the ~128 workgroups an XCD runs at a time share
    // the 2 x PG_ST slabs of their super-tile (4 MB).  A permutation of the tiles only: results
    // unchanged.  Same-box A/B at C5: row-major 8.84 ms, super-tiles of 4 / 8 / 16 tiles 8.69 / 8.69 /
    // 8.60 ms.
    const unsigned b = blockIdx.x;
    const unsigned within = (b >> 3) % (PG_ST * PG_ST);
    const unsigned sidx = ((b >> 3) / (PG_ST * PG_ST)) * 8 + (b & 7);
    const unsigned sx = sidx % nsx, sy = sidx / nsx;
    const unsigned tx = sx * PG_ST + within % PG_ST, ty = sy * PG_ST + within / PG_ST;
    if (tx >= ntx || ty >= nty) return;   // (past the edge of the tile grid, or a padding super-tile)
    const int64_t p0 = (int64_t)ty * PG_TM;
    const int32_t s0 = (int32_t)tx * PG_TM;
    const int cl = lane & 15, kq = lane >> 4;
    f64x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f64x4{0.0, 0.0, 0.0, 0.0};

    // staging: element e = tid + 256 q of a 64 x 16 slab -> row sr + 16 q, column sc.  One base
    // pointer per operand and thread, rows 16 Km_pad elements apart, slabs 16 elements apart.
    // (A slab's columns past Km_pad belong to the next row or to the 16 doubles of slack behind
    // the buffers: staged, never multiplied -- the last slab runs only the k-steps that exist.)
    const int sr = tid >> 4, sc = tid & 15;
    // (a wave-uniform base per operand, advanced by the slab, plus four 32-bit byte offsets per
    // thread that never change: the loads take the scalar-base form and the loop computes no
    // address at all -- written as p[q * rstep + m0] hipcc redid the 64-bit products every slab)
    const char* abase = reinterpret_cast<const char*>(P + p0 * Km_pad);
    const char* bbase = reinterpret_cast<const char*>(Wt + (int64_t)s0 * Km_pad);
    uint32_t voff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) voff[q] = (uint32_t)(((sr + 16 * q) * Km_pad + sc) * 8);
    double ra[4], rb[4];
    auto fetch = [&](int m0) {
        const char* ab = abase + (size_t)m0 * 8;
        const char* bb = bbase + (size_t)m0 * 8;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            ra[q] = *reinterpret_cast<const double*>(ab + voff[q]);
            rb[q] = *reinterpret_cast<const double*>(bb + voff[q]);
        }
    };
    double* as_w = As + sr * PG_LD + sc;
    double* bs_w = Bs + sr * PG_LD + sc;
    auto stash = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            as_w[(buf * PG_TM + 16 * q) * PG_LD] = ra[q];
            bs_w[(buf * PG_TM + 16 * q) * PG_LD] = rb[q];
        }
    };
    const double* a_r = As + (16 * wave + cl) * PG_LD + kq;
    const double* b_r = Bs + cl * PG_LD + kq;

    const int nslab = (Km_pad + PG_KT - 1) / PG_KT;
    const int last_nk = (Km_pad - PG_KT * (nslab - 1)) / 4;   // k-steps of the last slab, 1 .. 4
    fetch(0);
    stash(0);
    __syncthreads();
    for (int sl = 0; sl + 1 < nslab; ++sl) {
        const int buf = sl & 1;
        fetch((sl + 1) * PG_KT);
        const double* Ab = a_r + buf * PG_TM * PG_LD;
        const double* Bb = b_r + buf * PG_TM * PG_LD;
#pragma unroll
        for (int kk = 0; kk < PG_KT / 4; ++kk) {
            const double a = Ab[4 * kk];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const double bv = Bb[16 * t * PG_LD + 4 * kk];
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc[t], 0, 0, 0);
            }
        }
        stash(buf ^ 1);
        __syncthreads();
    }
    {
        const int buf = (nslab - 1) & 1;
        const double* Ab = a_r + buf * PG_TM * PG_LD;
        const double* Bb = b_r + buf * PG_TM * PG_LD;
        for (int kk = 0; kk < last_nk; ++kk) {
            const double a = Ab[4 * kk];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const double bv = Bb[16 * t * PG_LD + 4 * kk];
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc[t], 0, 0, 0);
            }
        }
    }

    // epilogue: R[pe][s] = acc + z sigma_s for this lane's 16 entries -- 8 Box-Muller pairs, the
    // pair of accumulator registers (2h, 2h+1) = points (pe, pe + 4).  No bounds tests: R has
    // whole tiles of rows and S_pad columns, sig is zero past S, and a counter past the true
    // range only costs a wasted variate in the padding.
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const int64_t pe0 = p0 + 16 * wave + kq;
    const int32_t sl0 = s0 + cl;
    double* r0 = R + pe0 * S_pad + sl0;
    const uint64_t e0 = (uint64_t)pe0 * (uint64_t)S + (uint64_t)sl0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const double sg = sig[sl0 + 16 * t];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            double z0 = 0.0, z1 = 0.0;
            if (noise_replay == nullptr) {
                pg_noise(e0 + (uint64_t)(8 * h) * (uint64_t)S + (uint64_t)(16 * t), k0, k1, z0, z1);
            } else {
                const int64_t pe = pe0 + 8 * h;
                const int32_t sd = sl0 + 16 * t;
                if (sd < S) {
                    if (pe < M) z0 = noise_replay[(int64_t)sd * M + pe];
                    if (pe + 4 < M) z1 = noise_replay[(int64_t)sd * M + pe + 4];
                }
            }
            double* rp = r0 + (int64_t)(8 * h) * S_pad + 16 * t;
            // (non-temporal: the 4 GB of draws are written once and read back by another kernel
            // long after they have left every cache; kept out of L2 they do not displace the
            // operand slabs of the super-tile -- same-box A/B 7.16 -> 7.05 ms)
            __builtin_nontemporal_store(fma(z0, sg, acc[t][2 * h]), rp);
            __builtin_nontemporal_store(fma(z1, sg, acc[t][2 * h + 1]), rp + (int64_t)4 * S_pad);
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
    unsigned long long* __restrict__ hits, const int32_t* __restrict__ point_list,
    const int32_t* __restrict__ point_count) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* buf = reinterpret_cast<double*>(smem_raw);
    const int tid = threadIdx.x, nt = blockDim.x;
    // all points, or (second pass of the selection kernel below) the points it handed back
    const int64_t n_points = point_list ? (int64_t)*point_count : M;
    unsigned long long my_hits = 0;   // thread 64+c counts interval c over this workgroup's points
    for (int64_t pi = blockIdx.x; pi < n_points; pi += gridDim.x) {
        const int64_t p = point_list ? (int64_t)point_list[pi] : pi;
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
            if (buf[cov_lo[c]] <= y && y <= buf[cov_hi[c]]) my_hits += 1;
        }
        __syncthreads();
    }
    // one global atomic per (workgroup, interval): per-point atomics on these few addresses
    // serialise at the memory side (4 ms for 50 000 points x 21 intervals)
    if (my_hits) atomicAdd(&hits[tid - 64], my_hits);
}

// ----------------------------------------------------------- order statistics by selection
// Only a few dozen ranks of each point's S draws are asked for (percentile neighbours and
// coverage bounds), so a full sort is wasted work.  One 512-thread workgroup per point keeps the
// draws in registers, histograms them over SEL_BINS equal-width bins of [min, max] (LDS atomics),
// finds the bin of every requested rank from the prefix sums, collects the members of just
// those bins (<= SEL_CAP each) and ranks them by counting inside one wave.  The values returned
// are elements of the row, so the result is bit-identical to the sort.  A point whose requested
// bin holds more than SEL_CAP draws (heavy ties, far outliers) is handed to the sort kernel
// through `fail_points`.
// (bins: same-box A/B of the C5 order statistics with 2048 / 4096 / 8192 bins: 1.88 / 1.62 / 2.59 ms --
// half the draws per requested bin halve the rank counting; 8192 bins leave one workgroup per CU)
#ifndef BMC_SEL_BINS
#define BMC_SEL_BINS 4096
#endif
constexpr int SEL_BINS = BMC_SEL_BINS, SEL_CAP = 64, SEL_THREADS = 512;
constexpr int SEL_BPT = SEL_BINS / SEL_THREADS;   // consecutive bins per thread in the prefix sums

__device__ __forceinline__ int sel_bin(double x, double mn, double scale) {
    const double t = (x - mn) * scale;
    int b = (int)t;
    return b < 0 ? 0 : (b > SEL_BINS - 1 ? SEL_BINS - 1 : b);
}

// (4 waves per SIMD = two workgroups per CU = 128 VGPRs, enough for up to 24 draws per thread;
// 32 draws per thread -- 12289..16384 draws -- take the 256-VGPR budget instead of spilling)
template <int VPT>
__global__ __launch_bounds__(SEL_THREADS, VPT > 24 ? 2 : 4) void predict_select_kernel(
    const double* __restrict__ R, int32_t S, int32_t S_pad, int64_t M,
    const int32_t* __restrict__ q_index, const double* __restrict__ q_gamma, int32_t n_q,
    const double* __restrict__ truth, const int32_t* __restrict__ cov_lo,
    const int32_t* __restrict__ cov_hi, int32_t n_cov, double* __restrict__ bands,
    unsigned long long* __restrict__ hits, int32_t* __restrict__ fail_points,
    int32_t* __restrict__ fail_count) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int n_t = 2 * n_q + 2 * n_cov;                       // requested ranks
    unsigned* hist = reinterpret_cast<unsigned*>(smem_raw);    // [SEL_BINS] counts
    unsigned* slot = hist + SEL_BINS;                          // [SEL_BINS] bin -> list, or ~0u
    double* red = reinterpret_cast<double*>(slot + SEL_BINS);  // [16] min / max per wave
    unsigned* wsum = reinterpret_cast<unsigned*>(red + 16);    // [8] wave totals of the scan
    unsigned* flag = wsum + 8;                                 // [0] overflow
    unsigned* pbase = flag + 8;                                // [SEL_THREADS] exclusive prefix per thread
    unsigned* cnt = pbase + SEL_THREADS;                       // [n_t] members per list
    int* tbin = reinterpret_cast<int*>(cnt + n_t);             // [n_t] bin of rank t
    int* tk = tbin + n_t;                                      // [n_t] rank inside that bin
    int* trank = tk + n_t;                                     // [n_t] the requested ranks
    double* list = reinterpret_cast<double*>(
        smem_raw + (((char*)(trank + n_t) - smem_raw + 15) & ~(size_t)15));   // [n_t][SEL_CAP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    if (tid < 2 * n_q) {
        const int lo = q_index[tid >> 1];
        trank[tid] = (tid & 1) ? (lo + 1 < S ? lo + 1 : S - 1) : lo;
    } else if (tid < n_t) {
        const int c = (tid - 2 * n_q) >> 1;
        trank[tid] = ((tid - 2 * n_q) & 1) ? cov_hi[c] : cov_lo[c];
    }

    unsigned long long my_hits = 0;   // thread 64+c counts interval c over this workgroup's points
    const int S_arg = S;
    for (int64_t p = blockIdx.x; p < M; p += gridDim.x) {
        // S is opaque inside the loop: otherwise hipcc keeps VPT clamped offsets and VPT
        // "index < S" masks in registers across points and spills the draws themselves
        int S = S_arg;
        asm volatile("" : "+s"(S));
        const double* row = R + p * S_pad;
        double v[VPT];
        double mn = __builtin_inf(), mx = -__builtin_inf();
        // unconditional (clamped) loads, all in flight before the first use
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int idx = tid + i * SEL_THREADS;
            v[i] = row[idx < S ? idx : S - 1];
        }
#pragma unroll
        for (int i = 0; i < VPT; ++i)
            if (tid + i * SEL_THREADS < S) { mn = fmin(mn, v[i]); mx = fmax(mx, v[i]); }
        for (int b = tid; b < SEL_BINS; b += SEL_THREADS) { hist[b] = 0; slot[b] = ~0u; }
        if (tid < n_t) cnt[tid] = 0;
        if (tid == 0) flag[0] = 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mn = fmin(mn, __shfl_xor(mn, o));
            mx = fmax(mx, __shfl_xor(mx, o));
        }
        if (lane == 0) { red[wave] = mn; red[8 + wave] = mx; }
        __syncthreads();
        mn = red[0]; mx = red[8];
#pragma unroll
        for (int w = 1; w < SEL_THREADS / 64; ++w) { mn = fmin(mn, red[w]); mx = fmax(mx, red[8 + w]); }
        const double scale = (double)SEL_BINS / (mx - mn);
        // degenerate rows: every draw equal -> every rank is that value; a range too small to
        // scale (or non-finite draws) goes to the sort
        const bool flat = mx == mn;
        const bool unusable = !flat && !(scale > 0.0 && scale < 1.7e308);
        const bool usable = !flat && !unusable;
        if (usable) {
#pragma unroll
            for (int i = 0; i < VPT; ++i)
                if (tid + i * SEL_THREADS < S) atomicAdd(&hist[sel_bin(v[i], mn, scale)], 1u);
        }
        __syncthreads();
        if (usable) {
            // exclusive prefix sums over the bins, SEL_BPT consecutive bins per thread
            unsigned tot = 0;
#pragma unroll
            for (int q = 0; q < SEL_BPT; q += 4) {
                const uint4 h4 = *reinterpret_cast<const uint4*>(hist + SEL_BPT * tid + q);
                tot += (h4.x + h4.y) + (h4.z + h4.w);
            }
            unsigned inc = tot;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned up = __shfl_up(inc, o);
                if (lane >= o) inc += up;
            }
            if (lane == 63) wsum[wave] = inc;
            __syncthreads();
            unsigned base = inc - tot;
            for (int w = 0; w < wave; ++w) base += wsum[w];
            // Thread t < n_t finds the bin of requested rank t: a binary search over the threads'
            // exclusive prefixes (left in LDS), then the few bins of that thread.  (Every thread
            // testing every rank against its own 4 bins, as before, was n_t iterations in all 8
            // waves: a quarter of the kernel's vector instructions.)
            pbase[tid] = base;
            __syncthreads();
            if (tid < n_t) {
                const unsigned r = (unsigned)trank[tid];
                int lo = 0, hi = SEL_THREADS - 1;      // last thread whose prefix is <= r
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (pbase[mid] <= r) lo = mid; else hi = mid - 1;
                }
                unsigned b0 = pbase[lo];
                int j = 0;
                while (j < SEL_BPT - 1 && r >= b0 + hist[SEL_BPT * lo + j]) { b0 += hist[SEL_BPT * lo + j]; ++j; }
                tbin[tid] = SEL_BPT * lo + j;
                tk[tid] = (int)(r - b0);
                atomicMin(&slot[SEL_BPT * lo + j], (unsigned)tid);
            }
            __syncthreads();
            // members of the requested bins -> their lists (slot words read in one batch)
            // (bins are recomputed, not carried in registers from the histogram pass: the
            // opaque copy of `scale` keeps hipcc from caching VPT bin numbers and spilling)
            double scale2 = scale;
            asm volatile("" : "+v"(scale2));
#pragma unroll
            for (int i0 = 0; i0 < VPT; i0 += 4) {
                unsigned sl[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) sl[i] = slot[sel_bin(v[i0 + i], mn, scale2)];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (sl[i] != ~0u && tid + (i0 + i) * SEL_THREADS < S) {
                        const unsigned pos = atomicAdd(&cnt[sl[i]], 1u);
                        if (pos < SEL_CAP) list[sl[i] * SEL_CAP + pos] = v[i0 + i];
                        else flag[0] = 1;
                    }
            }
            __syncthreads();
            if (flag[0] == 0) {
                // one wave per list: every lane ranks its member by counting (members come from
                // the other lanes' registers) and stores it at its rank -> the list is sorted
                for (int t = wave; t < n_t; t += SEL_THREADS / 64) {
                    if (slot[tbin[t]] != (unsigned)t) continue;   // not the owner of its bin
                    const int n = (int)cnt[t];
                    double* li = list + t * SEL_CAP;
                    const double e = lane < n ? li[lane] : 0.0;
                    int c = 0;
                    for (int j = 0; j < n; ++j) {
                        const double vj = __hiloint2double(
                            __builtin_amdgcn_readlane(__double2hiint(e), j),
                            __builtin_amdgcn_readlane(__double2loint(e), j));
                        c += (vj < e || (vj == e && j < lane)) ? 1 : 0;
                    }
                    if (lane < n) li[c] = e;   // every lane has read its member: in-place is safe
                }
            }
            __syncthreads();
        }
        const bool failed = unusable || (usable && flag[0] != 0);
        if (failed) {
            if (tid == 0) fail_points[atomicAdd(fail_count, 1)] = (int32_t)p;
        } else {
            // value of requested rank t: sorted list of its bin at its rank inside the bin
            if (tid < n_q) {
                double a = mn, b = mn;
                if (usable) {
                    a = list[slot[tbin[2 * tid]] * SEL_CAP + tk[2 * tid]];
                    b = list[slot[tbin[2 * tid + 1]] * SEL_CAP + tk[2 * tid + 1]];
                }
                const double t = q_gamma[tid], diff = b - a;
                bands[(int64_t)tid * M + p] = t >= 0.5 ? b - diff * (1.0 - t) : a + diff * t;
            }
            if (truth != nullptr && tid >= 64 && tid - 64 < n_cov) {
                const int t0 = 2 * n_q + 2 * (tid - 64);
                double a = mn, b = mn;
                if (usable) {
                    a = list[slot[tbin[t0]] * SEL_CAP + tk[t0]];
                    b = list[slot[tbin[t0 + 1]] * SEL_CAP + tk[t0 + 1]];
                }
                const double y = truth[p];
                if (a <= y && y <= b) my_hits += 1;
            }
        }
        __syncthreads();
    }
    if (my_hits) atomicAdd(&hits[tid - 64], my_hits);
}

// ----------------------------------------------------------- draws in the reference's layout
// R is [M][S_pad] (a point's draws contiguous: what the order statistics read).  The reference
// returns rndm_m as a C-ordered (S, M) array (sampling_utils.py:77); this transposes on the
// device, through LDS in 64 x 64 tiles (row stride 65 doubles: conflict-free both ways), so that
// the copy back is one contiguous block in the caller's layout.
__global__ __launch_bounds__(256) void transpose_draws_kernel(const double* __restrict__ R, int64_t M,
                                                              int32_t S, int32_t S_pad,
                                                              double* __restrict__ out) {
    __shared__ double tile[64][65];
    const int64_t p0 = (int64_t)blockIdx.x * 64;
    const int32_t s0 = (int32_t)blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll 4
    for (int r = ty; r < 64; r += 4) {
        const int64_t p = p0 + r;
        const int32_t sd = s0 + tx;
        tile[r][tx] = (p < M && sd < S) ? R[p * S_pad + sd] : 0.0;
    }
    __syncthreads();
#pragma unroll 4
    for (int r = ty; r < 64; r += 4) {
        const int32_t sd = s0 + r;
        const int64_t p = p0 + tx;
        if (sd < S && p < M) out[(int64_t)sd * M + p] = tile[tx][r];
    }
}

hipError_t launch_transpose_draws(const double* R, int64_t M, int32_t S, int32_t S_pad, double* out,
                                  hipStream_t s) {
    const uint64_t gx = (uint64_t)((M + 63) / 64), gy = (uint64_t)((S + 63) / 64);
    if (gx == 0 || gy == 0) return hipSuccess;
    if (gx > 0x7fffffffull || gy > 65535ull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(transpose_draws_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, s, R, M,
                       S, S_pad, out);
    return hipGetLastError();
}

hipError_t launch_predict(const PredictArgs& a, hipStream_t s) {
    {
        const dim3 grid((unsigned)((a.Km_pad + 63) / 64), (unsigned)((a.S_pad + PW_S - 1) / PW_S));
        hipLaunchKernelGGL(predict_weights_kernel, grid, dim3(256), 0, s, a.theta, a.Vt, a.S, a.k, a.Km,
                           a.S_pad, a.Km_pad, a.Wt, a.sig);
    }
    {
        const int64_t total = a.M_pad * a.Km_pad;
        int64_t blocks = (total + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(predict_pad_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a.preds, a.M,
                           a.Km, a.M_pad, a.Km_pad, a.P);
    }
    {
        // tiles of 64 points x 64 draws, dealt out in super-tiles of PG_ST x PG_ST per XCD
        const uint32_t ntx = (uint32_t)(a.S_pad / PG_TM), nty = (uint32_t)(a.M_pad / PG_TM);
        const uint32_t nsx = (ntx + PG_ST - 1) / PG_ST, nsy = (nty + PG_ST - 1) / PG_ST;
        const uint64_t nsuper8 = ((uint64_t)nsx * nsy + 7) / 8 * 8;
        const uint64_t nblocks = nsuper8 * PG_ST * PG_ST;
        if (nblocks > 0x7fffffffull) return hipErrorInvalidValue;
        hipLaunchKernelGGL(predict_gemm_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, (const double*)a.P,
                           (const double*)a.Wt, (const double*)a.sig, a.M, a.S, a.S_pad, a.Km_pad, a.seed,
                           a.noise_replay, a.R, ntx, nty, nsx);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (a.ev_mid && (e = hipEventRecord(a.ev_mid, s)) != hipSuccess) return e;
    if (a.n_q > 0 || a.n_cov > 0) {
        int nsort = 64;
        while (nsort < a.S) nsort <<= 1;
        // selection when there are many draws and few requested ranks, otherwise the sort
        const int n_t = 2 * a.n_q + 2 * a.n_cov;
        const bool select = a.S >= 2048 && a.S <= 32 * SEL_THREADS && n_t <= 128 && a.fail_points;
        const int32_t* plist = nullptr;
        const int32_t* pcount = nullptr;
        int64_t blocks = a.M < 2048 ? a.M : 2048;
        if (blocks < 1) blocks = 1;
        if (select) {
            const size_t lds = (size_t)SEL_BINS * 8 + 16 * 8 + 16 * 4 + SEL_THREADS * 4 + (size_t)n_t * 16 + 16 +
                               (size_t)n_t * SEL_CAP * 8;
#define BMC_SEL(V)                                                                             \
    do {                                                                                       \
        e = hipFuncSetAttribute((const void*)predict_select_kernel<V>,                          \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);         \
        if (e != hipSuccess) return e;                                                         \
        hipLaunchKernelGGL((predict_select_kernel<V>), dim3((unsigned)blocks),                 \
                           dim3(SEL_THREADS), lds, s, (const double*)a.R, a.S, a.S_pad, a.M,   \
                           a.q_index, a.q_gamma, a.n_q, a.truth, a.cov_lo, a.cov_hi, a.n_cov,  \
                           a.bands, a.hits, a.fail_points, a.fail_count);                      \
    } while (0)
            const int vpt = (a.S + SEL_THREADS - 1) / SEL_THREADS;
            // draws per thread in steps of 4 (10 000 draws: 20, not 24 -- the slots past the row
            // cost every per-draw step of the kernel)
            if (vpt <= 8) BMC_SEL(8);
            else if (vpt <= 12) BMC_SEL(12);
            else if (vpt <= 16) BMC_SEL(16);
            else if (vpt <= 20) BMC_SEL(20);
            else if (vpt <= 24) BMC_SEL(24);
            else if (vpt <= 28) BMC_SEL(28);
            else BMC_SEL(32);
#undef BMC_SEL
            e = hipGetLastError();
            if (e != hipSuccess) return e;
            plist = a.fail_points;   // second pass: whatever the selection handed back
            pcount = a.fail_count;
            blocks = blocks < 256 ? blocks : 256;
        }
        const size_t lds = (size_t)nsort * sizeof(double);
#define BMC_OS(NS)                                                                            \
    do {                                                                                      \
        e = hipFuncSetAttribute((const void*)predict_orderstat_kernel<NS>,                    \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);        \
        if (e != hipSuccess) return e;                                                        \
        hipLaunchKernelGGL((predict_orderstat_kernel<NS>), dim3((unsigned)blocks),            \
                           dim3(NS / 2 < 1024 ? (NS / 2 < 128 ? 128 : NS / 2) : 1024), lds, s, \
                           (const double*)a.R, a.S, a.S_pad, a.M, a.q_index, a.q_gamma, a.n_q, \
                           a.truth, a.cov_lo, a.cov_hi, a.n_cov, a.bands, a.hits, plist,       \
                           pcount);                                                            \
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
