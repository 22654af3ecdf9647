// One-off and auxiliary gfx950 kernels of the Gibbs core:
//   panelize      user layout -> row panels                       (HBM-bound copy)
//   gram          [X y]'[X y] with v_mfma_f64_16x16x4_f64          (reference inference_utils.py:25,43)
//   rotate        Xrot = X W                                       (DESIGN.md "rotated draw")
//   residual_rss  rss = sum (y - X b)^2, streaming, two-stage sum  (reference inference_utils.py:48-51)
//   unrotate      beta_t = W u_t                                   (reference inference_utils.py:54)
//   rng_fill      Philox4x32-10 -> N(0,1) and Gamma(a,1)           (reference inference_utils.py:45,52)
#include "bmc_dev.h"
#include "bmc_launch.h"

namespace bmc {

using f64x4 = __attribute__((ext_vector_type(4))) double;

// =========================================================================
// panelize
// =========================================================================
template <typename T>
__global__ __launch_bounds__(256) void panelize_kernel(
    const T* __restrict__ Xs, const T* __restrict__ ys, int64_t n, int32_t k, int64_t ldx,
    int col_major, int32_t RP, int32_t npanels, T* __restrict__ Xp, T* __restrict__ yp) {
    const int64_t total = (int64_t)npanels * k * RP;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t p = e / ((int64_t)k * RP);
        const int64_t rem = e - p * (int64_t)k * RP;
        const int32_t j = (int32_t)(rem / RP);
        const int32_t r = (int32_t)(rem - (int64_t)j * RP);
        const int64_t row = p * RP + r;
        T v = (T)0;
        if (row < n) v = col_major ? Xs[row + (int64_t)j * ldx] : Xs[row * ldx + j];
        Xp[e] = v;
    }
    const int64_t ytotal = (int64_t)npanels * RP;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < ytotal; e += stride)
        yp[e] = e < n ? ys[e] : (T)0;
}

hipError_t launch_panelize(const void* Xsrc, const void* ysrc, int64_t n, int32_t k,
                           int64_t ldx, int col_major, int f32, int32_t vec, void* Xp,
                           void* yp, int32_t npanels, hipStream_t s) {
    const int32_t RP = 64 * vec;
    const int64_t total = (int64_t)npanels * k * RP;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    if (f32)
        hipLaunchKernelGGL(panelize_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s,
                           (const float*)Xsrc, (const float*)ysrc, n, k, ldx, col_major, RP,
                           npanels, (float*)Xp, (float*)yp);
    else
        hipLaunchKernelGGL(panelize_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, s,
                           (const double*)Xsrc, (const double*)ysrc, n, k, ldx, col_major, RP,
                           npanels, (double*)Xp, (double*)yp);
    return hipGetLastError();
}

// =========================================================================
// Gram with f64 MFMA
// =========================================================================
// G = A'A, A = [X | y | 0-pad] (Ka_pad columns), as the 16x16 tiles (ti, tj >= ti) of the upper
// triangle ("tile pairs"); the reduce kernel mirrors.  One workgroup of 8 waves owns a chunk of
// SR-row sub-panels and ALL tile pairs, so every row of the matrix is read from memory once:
//   * the sub-panel is staged into LDS as f64 [Ka_pad][SR + 2] (column-major; the row stride of
//     SR + 2 doubles makes the MFMA operand reads bank-conflict free: a 32-lane half covers 16
//     columns x 2 rows = banks {4c, 4c+1} and {4c+2, 4c+3}), double-buffered: the next sub-panel
//     travels global -> registers while the MFMAs of the current one run and is written to the
//     other buffer behind them -- one barrier per sub-panel;
//   * wave (ts, rs) of the TS x RS active waves accumulates tile pairs ts*NTW .. ts*NTW+NTW-1
//     (NTW accumulators of 8 registers, AGPRs) over the k-steps [rs*KPW, (rs+1)*KPW) of every
//     sub-panel; the loop body is branch-free (slots past the last pair recompute pair 0 and are
//     never written), so hipcc keeps several LDS reads in flight under the MFMAs.  (The first
//     version guarded every MFMA with a wave-uniform branch: ds_read, ds_read, s_waitcnt
//     lgkmcnt(0), v_mfma -- fully serialised, 27 % MFMA busy at C5 -- and re-staged every row
//     once per group of 64 tile pairs.)
//   * RS > 1 (few tile pairs: the row dimension is split instead): the RS partial accumulators
//     of a tile are added in rs order through LDS after the last sub-panel.
// Wide problems (C5: Ka_pad = 272, 153 pairs): SR = 32 (2 x 74 KB of LDS), NTW = 20, TS = 8.
// MFMA operand maps (cdna guide section 3, f64 form): lane l supplies A[i = l&15][k = l>>4] and
// B[k = l>>4][j = l&15]; D: col = l&15, row = (l>>4) + 4*reg.  For G tile (ti,tj):
// A[i][k] = a[n0+k][16ti+i], B[k][j] = a[n0+k][16tj+j].
// The chunk's tiles go to partial[chunk][pair][16 x 16] (only the pairs, compact).
constexpr int GRAM_THREADS = 512;
constexpr int GRAM_NVMAX = 9;   // two-row pieces per thread: Ka_pad * SR / 2 / 512 <= 272 * 16 / 512

template <typename T> struct Pair2 { T a, b; };

// pair id -> (ti, tj), tj >= ti: row-by-row enumeration of the upper triangle
__host__ __device__ inline void gram_pair(int id, int ntile, int& ti, int& tj) {
    int a = 0, rowlen = ntile;
    while (id >= rowlen) { id -= rowlen; ++a; --rowlen; }
    ti = a;
    tj = a + id;
}

template <typename T, int NTW>
__global__ __launch_bounds__(GRAM_THREADS) void gram_mfma_kernel(
    const T* __restrict__ X, const T* __restrict__ y, int32_t K, int32_t RP, int32_t nsub,
    int32_t subs_per_chunk, int32_t Ka_pad, int32_t ntile, int32_t npairs, int32_t SR, int32_t TS,
    int32_t RS, double* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int chunk = blockIdx.x;
    const int LDR = SR + 2;
    const int bufbytes = Ka_pad * LDR * 8;
    const bool active = wave < TS * RS;
    const int ts = active ? wave % TS : 0, rs = active ? wave / TS : 0;
    const int KPW = (SR >> 2) / RS;         // k-steps of 4 rows per wave and sub-panel (even)
    constexpr int GB = NTW > 10 ? 2 : NTW % 4 == 0 ? 4 : (NTW % 3 == 0 ? 3 : 2);   // tiles per operand group
                                                  // (20 accumulators: 160 registers, so small groups;
                                                  // groups of 4 there: same-box A/B 136.6 vs 137.9 us)
    constexpr int NG = NTW / GB;

    // byte offsets of the wave's tiles' operand columns (wave-uniform: SGPRs)
    int aoff[NTW], boff[NTW];
#pragma unroll
    for (int q = 0; q < NTW; ++q) {
        int id = ts * NTW + q, ti, tj;
        if (id >= npairs) id = 0;
        gram_pair(id, ntile, ti, tj);
        aoff[q] = __builtin_amdgcn_readfirstlane(16 * ti * LDR * 8);
        boff[q] = __builtin_amdgcn_readfirstlane(16 * tj * LDR * 8);
    }
    f64x4 acc[NTW];
#pragma unroll
    for (int q = 0; q < NTW; ++q) acc[q] = f64x4{0.0, 0.0, 0.0, 0.0};

    const int s_begin = chunk * subs_per_chunk;
    int s_end = s_begin + subs_per_chunk;
    if (s_end > nsub) s_end = nsub;
    const int kq = lane >> 4, cl = lane & 15;
    const int hsh = SR == 64 ? 5 : 4;       // SR / 2 two-row pieces per column
    const int npiece = Ka_pad << hsh;

    // (branch-free: pieces past the matrix -- zero-pad columns, or past the last piece -- read a
    // valid address of y and are zeroed, so the loads are plain instructions that hipcc can
    // leave in flight under the MFMAs)
    Pair2<T> stage[GRAM_NVMAX];
    auto fetch = [&](int s) {
        const int64_t row0 = (int64_t)s * SR;
        const int64_t p = row0 / RP;
        const int32_t r0 = (int32_t)(row0 - p * RP);
        const T* xrow = X + p * K * RP + r0;
        const T* yrow = y + row0;
        int t0 = tid;
        asm volatile("" : "+v"(t0));   // (piece offsets and masks recomputed here, not kept live)
#pragma unroll
        for (int i = 0; i < GRAM_NVMAX; ++i) {
            const int e = t0 + i * GRAM_THREADS;
            const int a = e >> hsh, r = (e & ((1 << hsh) - 1)) * 2;
            const T* src = a < K ? xrow + (int64_t)a * RP + r : yrow + r;
            Pair2<T> v = *reinterpret_cast<const Pair2<T>*>(src);
            if (a > K) v = Pair2<T>{(T)0, (T)0};
            stage[i] = v;
        }
    };
    auto put = [&](int buf) {
        double* tile = reinterpret_cast<double*>(smem_raw + (size_t)buf * bufbytes);
        int t0 = tid;
        asm volatile("" : "+v"(t0));
#pragma unroll
        for (int i = 0; i < GRAM_NVMAX; ++i) {
            const int e = t0 + i * GRAM_THREADS;
            if (e < npiece) {
                double* d = &tile[(e >> hsh) * LDR + (e & ((1 << hsh) - 1)) * 2];
                d[0] = (double)stage[i].a;
                d[1] = (double)stage[i].b;
            }
        }
    };
    if (s_begin < s_end) {
        fetch(s_begin);
        put(0);
    }
    __syncthreads();
    const int lane_off = (cl * LDR + kq + 4 * rs * KPW) * 8;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem_raw;
    int buf = 0;
    for (int s = s_begin; s < s_end; ++s) {
        if (s + 1 < s_end) fetch(s + 1);   // in flight under the MFMAs below
        if (active) {
            // Operands of the next group of GB tiles are read from LDS while the MFMAs of the
            // current group run (two register sets, alternating).  Two k-steps per trip (KPW is
            // even), so the set parity is the same at the top of every trip.  The very last
            // prefetch of a sub-panel reads past the wave's k-range (inside the LDS allocation,
            // which is padded for it) and is never used.
            // (LDS byte addresses as 32-bit integers: a generic pointer through the opaque asm
            // below would turn the reads into flat loads with full vmcnt/lgkmcnt waits)
            typedef const __attribute__((address_space(3))) double lds_cd;
            const unsigned base = lds0 + (unsigned)buf * (unsigned)bufbytes + (unsigned)lane_off;
            double oa[2][GB], ob[2][GB];
            auto load_group = [&](int set, int g, unsigned b) {
#pragma unroll
                for (int t = 0; t < GB; ++t) {
                    oa[set][t] = *reinterpret_cast<lds_cd*>(b + (unsigned)aoff[g * GB + t]);
                    ob[set][t] = *reinterpret_cast<lds_cd*>(b + (unsigned)boff[g * GB + t]);
                }
            };
            load_group(0, 0, base);
            for (int i = 0; i < KPW; i += 2) {
                // (running vector addresses of the two k-steps and of the next trip's first: kept
                // opaque, or hipcc folds the k offsets into 2*NTW more scalar offsets and spills)
                unsigned bk[3] = {base + i * 32, base + i * 32 + 32, base + i * 32 + 64};
                asm volatile("" : "+v"(bk[0]), "+v"(bk[1]), "+v"(bk[2]));
#pragma unroll
                for (int m = 0; m < 2 * NG; ++m) {
                    const int mn = m + 1;
                    load_group(mn & 1, mn % NG, bk[mn / NG]);
#pragma unroll
                    for (int t = 0; t < GB; ++t) {
                        const int q = (m % NG) * GB + t;
                        acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(oa[m & 1][t], ob[m & 1][t], acc[q], 0, 0, 0);
                    }
                    // nothing crosses: reads of group m+1, then the MFMAs of group m (left to
                    // itself hipcc hoists the reads of the whole trip and runs out of registers)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (s + 1 < s_end) put(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    // the RS row-parts of a tile, added in rs order (both LDS buffers are free now)
    if (RS > 1) {
        double* xch = reinterpret_cast<double*>(smem_raw);
        for (int r = 1; r < RS; ++r) {
            if (active && rs == r) {
#pragma unroll
                for (int q = 0; q < NTW; ++q)
#pragma unroll
                    for (int i = 0; i < 4; ++i) xch[((ts * NTW + q) * 4 + i) * 64 + lane] = acc[q][i];
            }
            __syncthreads();
            if (active && rs == 0) {
#pragma unroll
                for (int q = 0; q < NTW; ++q)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[q][i] += xch[((ts * NTW + q) * 4 + i) * 64 + lane];
            }
            __syncthreads();
        }
    }
    if (active && rs == 0) {
        double* out = partial + (size_t)chunk * npairs * 256;
#pragma unroll
        for (int q = 0; q < NTW; ++q) {
            const int id = ts * NTW + q;
            if (id < npairs) {
#pragma unroll
                for (int i = 0; i < 4; ++i) out[(size_t)id * 256 + (kq + 4 * i) * 16 + cl] = acc[q][i];
            }
        }
    }
}

// Sum the chunk partials (bit-reproducible: fixed order), mirror to the lower triangle.  A
// workgroup of 4 waves owns 16 consecutive elements of a tile; the chunks are cut into 16 runs:
// thread (wave w, lane 4 e + q) adds run 4 w + q of element e in chunk order (128 contiguous
// bytes per run and load instruction), the four runs of a wave meet through two shuffles,
// (q0 + q1) + (q2 + q3), the four waves through LDS, (w0 + w1) + (w2 + w3): dependent chains of
// nchunk / 16 additions, 16 x the loads in flight of one thread per element.
__global__ __launch_bounds__(256) void gram_reduce_kernel(const double* __restrict__ partial,
                                                          int32_t nchunk, int32_t npairs,
                                                          int32_t ntile, int32_t Ka,
                                                          double* __restrict__ out) {
    __shared__ double wsum[4][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = blockIdx.x * 16 + (lane >> 2), part = wave * 4 + (lane & 3);
    const bool in = e < npairs * 256;
    int ti = 0, tj = 0;
    gram_pair(in ? e >> 8 : 0, ntile, ti, tj);
    const int r = (e >> 4) & 15, c = e & 15;
    const int i = 16 * ti + r, j = 16 * tj + c;
    const bool live = in && i < Ka && j < Ka && (ti != tj || r <= c);
    const int per = (nchunk + 15) / 16;
    const int c0 = part * per, c1 = c0 + per < nchunk ? c0 + per : nchunk;
    double s = 0.0;
    if (live)
        for (int ch = c0; ch < c1; ++ch) s += partial[((size_t)ch * npairs) * 256 + e];
    const double s01 = s + __shfl_xor(s, 1);       // lanes 0,1 -> q0 + q1; lanes 2,3 -> q2 + q3
    const double lo = __shfl(s01, (lane & 60), 64), hi = __shfl(s01, (lane & 60) + 2, 64);
    if ((lane & 3) == 0) wsum[wave][lane >> 2] = lo + hi;
    __syncthreads();
    if (wave == 0 && (lane & 3) == 0 && live) {
        const int q = lane >> 2;
        const double t = (wsum[0][q] + wsum[1][q]) + (wsum[2][q] + wsum[3][q]);
        out[(size_t)i * Ka + j] = t;
        out[(size_t)j * Ka + i] = t;
    }
}

struct GramGeo {
    int Ka_pad, ntile, npairs, SR, nsub, spc, nchunk, NTW, TS, RS;
    size_t lds;
};

static GramGeo gram_geometry(const Panels& P) {
    GramGeo g;
    g.Ka_pad = ((P.k + 1 + 15) / 16) * 16;
    g.ntile = g.Ka_pad / 16;
    g.npairs = g.ntile * (g.ntile + 1) / 2;
    // 64-row sub-panels while two of them fit the 160 KiB of LDS, else 32 rows
    g.SR = (size_t)2 * g.Ka_pad * 66 * 8 <= (size_t)160 * 1024 ? 64 : 32;
    g.nsub = P.npanels * P.vec * (64 / g.SR);
    // accumulators per wave / tile sets / row parts: the 8 waves as busy as they can be
    const int opts[4] = {20, 10, 8, 6};
    double best = -1.0;
    g.NTW = 20; g.TS = 8; g.RS = 1;
    for (int o = 0; o < 4; ++o) {
        const int ntw = opts[o], ts = (g.npairs + ntw - 1) / ntw;
        if (ts > 8) continue;
        int rs = 1;
        while (rs * 2 * ts <= 8 && rs * 2 <= g.SR / 8) rs *= 2;   // (k-steps per wave stay even)
        // the cross-wave sum of RS > 1 needs ts * ntw tiles of 2 KiB in LDS
        while (rs > 1 && (size_t)ts * ntw * 2048 > (size_t)160 * 1024) rs /= 2;
        const double eff = (double)g.npairs / (ts * ntw) * (ts * rs) / 8.0;
        if (eff > best + 1e-9) { best = eff; g.NTW = ntw; g.TS = ts; g.RS = rs; }
    }
    g.lds = (size_t)2 * g.Ka_pad * (g.SR + 2) * 8 + 64;   // (+ pad: the kernel's last prefetch)
    if (g.RS > 1 && (size_t)g.TS * g.NTW * 2048 > g.lds) g.lds = (size_t)g.TS * g.NTW * 2048;
    // sub-panels per chunk: the workgroups run in rounds of one per CU; take the chunking whose
    // rounds x sub-panels-per-chunk is least (few, full rounds), with enough chunks to fill the chip
    long best_cost = -1;
    g.spc = 1;
    for (int c = 1; c <= 256; ++c) {
        const long chunks = (g.nsub + c - 1) / c;
        const long rounds = (chunks + 255) / 256;
        const long cost = rounds * c;
        if (best_cost < 0 || cost <= best_cost) { best_cost = cost; g.spc = c; }   // ties: fewer slabs
        if (chunks <= 256) break;   // one round already: larger chunks only idle CUs
    }
    g.nchunk = (g.nsub + g.spc - 1) / g.spc;
    return g;
}

size_t gram_scratch_bytes(const Panels& P) {
    const GramGeo g = gram_geometry(P);
    return (size_t)g.nchunk * g.npairs * 256 * sizeof(double);
}

template <typename T, int NTW>
static hipError_t launch_gram_t(const Panels& P, const GramGeo& g, void* scratch, hipStream_t s) {
    hipError_t e = hipFuncSetAttribute((const void*)gram_mfma_kernel<T, NTW>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((gram_mfma_kernel<T, NTW>), dim3(g.nchunk), dim3(GRAM_THREADS), g.lds, s,
                       (const T*)P.X, (const T*)P.y, P.k, 64 * P.vec, g.nsub, g.spc, g.Ka_pad, g.ntile,
                       g.npairs, g.SR, g.TS, g.RS, (double*)scratch);
    return hipGetLastError();
}

template <typename T>
static hipError_t launch_gram_k(const Panels& P, const GramGeo& g, void* scratch, hipStream_t s) {
    switch (g.NTW) {
        case 6: return launch_gram_t<T, 6>(P, g, scratch, s);
        case 8: return launch_gram_t<T, 8>(P, g, scratch, s);
        case 10: return launch_gram_t<T, 10>(P, g, scratch, s);
        default: return launch_gram_t<T, 20>(P, g, scratch, s);
    }
}

hipError_t launch_gram(const Panels& P, void* scratch, double* gram_out, hipStream_t s) {
    const GramGeo g = gram_geometry(P);
    if (g.Ka_pad > 272 || (size_t)g.Ka_pad * (g.SR / 2) > (size_t)GRAM_NVMAX * GRAM_THREADS)
        return hipErrorInvalidValue;
    hipError_t e = P.f32 ? launch_gram_k<float>(P, g, scratch, s) : launch_gram_k<double>(P, g, scratch, s);
    if (e != hipSuccess) return e;
    const int Ka = P.k + 1;
    hipLaunchKernelGGL(gram_reduce_kernel, dim3(g.npairs * 16), dim3(256), 0, s,
                       (const double*)scratch, g.nchunk, g.npairs, g.ntile, Ka, gram_out);
    return hipGetLastError();
}

// =========================================================================
// rotate: Xrot = X W
// =========================================================================
// One wave produces JT = 8 output columns of one panel: it streams the panel's K input columns
// once (coalesced, VEC rows per lane) and keeps 8 * VEC accumulators; W[i][j] is wave-uniform
// (scalar loads).  (Round 3 tried JT = 32 for wide outputs -- a quarter of the L2 re-reads of the
// panel: C5 0.51 -> 0.68 ms, a quarter of the waves and each waiting on four scalar loads per
// input column; not kept.  One-off per problem, beside 38 ms of host K x K algebra at K = 256.)
template <typename T, int VEC, int ROT_JT>
__global__ __launch_bounds__(256) void rotate_kernel(const T* __restrict__ X,
                                                     const double* __restrict__ W, int32_t K,
                                                     int32_t KO, T* __restrict__ Xrot,
                                                     int32_t npanels, uint32_t ncg) {
    // W is [K][KO] row-major; the output panels have KO columns
    constexpr int RP = 64 * VEC;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD-aware order: the ncg workgroups that produce the column groups of ONE panel all read
    // that panel.  The hardware deals workgroups to the 8 XCDs round robin, so they are made 8
    // apart in launch order (block b: XCD b mod 8 takes panels b mod 8, 8 + b mod 8, ...; its
    // consecutive workgroups are the column groups of one panel): the panel comes from memory
    // once, not once per XCD (C5: 836 MB fetched for a 103 MB matrix before).
    const unsigned b = blockIdx.x;
    const int64_t p = (int64_t)((b >> 3) / ncg) * 8 + (b & 7);
    const int j0 = (int)(((b >> 3) % ncg) * 4 + wave) * ROT_JT;
    if (p >= npanels || j0 >= KO) return;
    const T* xp = X + p * (int64_t)K * RP + lane * VEC;
    double acc[ROT_JT][VEC];
#pragma unroll
    for (int q = 0; q < ROT_JT; ++q)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[q][v] = 0.0;
    for (int i = 0; i < K; ++i) {
        double x[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) x[v] = (double)xp[(int64_t)i * RP + v];
#pragma unroll
        for (int q = 0; q < ROT_JT; ++q) {
            const double w = (j0 + q < KO) ? W[(size_t)i * KO + j0 + q] : 0.0;
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[q][v] = fma(x[v], w, acc[q][v]);
        }
    }
    T* op = Xrot + p * (int64_t)KO * RP + lane * VEC;
#pragma unroll
    for (int q = 0; q < ROT_JT; ++q)
        if (j0 + q < KO)
#pragma unroll
            for (int v = 0; v < VEC; ++v) op[(int64_t)(j0 + q) * RP + v] = (T)acc[q][v];
}

template <typename T>
static hipError_t rotate_dispatch(const Panels& P, const double* W, int32_t ko, void* Xrot,
                                  hipStream_t s) {
    constexpr int jt = 8;
    const uint32_t ncg = (uint32_t)((ko + 4 * jt - 1) / (4 * jt));   // column groups per panel
    const dim3 grid((unsigned)(((P.npanels + 7) / 8) * 8) * ncg);
#define BMC_ROT(V, J)                                                                       \
    hipLaunchKernelGGL((rotate_kernel<T, V, J>), grid, dim3(256), 0, s, (const T*)P.X, W, P.k, \
                       ko, (T*)Xrot, P.npanels, ncg)
    switch (P.vec) {
        case 1: BMC_ROT(1, jt); break;
        case 2: BMC_ROT(2, jt); break;
        case 4: BMC_ROT(4, jt); break;
        default: return hipErrorInvalidValue;
    }
#undef BMC_ROT
    return hipGetLastError();
}

hipError_t launch_rotate(const Panels& P, const double* W, int32_t ko, void* Xrot, hipStream_t s) {
    return P.f32 ? rotate_dispatch<float>(P, W, ko, Xrot, s)
                 : rotate_dispatch<double>(P, W, ko, Xrot, s);
}

// =========================================================================
// orthogonalize helpers (reference bmc.py:106-116 and the layout of inference_utils.py:164)
// =========================================================================
// centre: mu_n = mean_j F[n][j]; Fc = F - mu (panels); yc = truth - mu.  One lane per row.
template <int VEC>
__global__ __launch_bounds__(256) void centre_kernel(const double* __restrict__ F, int64_t n,
                                                     int32_t km, int64_t ldf,
                                                     const double* __restrict__ truth,
                                                     int32_t npanels, double* __restrict__ Fc,
                                                     double* __restrict__ yc,
                                                     double* __restrict__ mu) {
    constexpr int RP = 64 * VEC;
    const int64_t total = (int64_t)npanels * RP;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < total; row += stride) {
        const int64_t p = row / RP;
        const int32_t r = (int32_t)(row - p * RP);
        double m = 0.0;
        if (row < n) {
            const double* fr = F + row * ldf;
            for (int j = 0; j < km; ++j) m += fr[j];
            m /= (double)km;
            for (int j = 0; j < km; ++j) Fc[(p * km + j) * RP + r] = fr[j] - m;
            yc[row] = truth[row] - m;
            mu[row] = m;
        } else {
            for (int j = 0; j < km; ++j) Fc[(p * km + j) * RP + r] = 0.0;
            yc[row] = 0.0;
        }
    }
}

hipError_t launch_centre(const double* F, int64_t n, int32_t km, int64_t ldf, const double* truth,
                         int32_t vec, int32_t npanels, double* Fc, double* yc, double* mu,
                         hipStream_t s) {
    int64_t blocks = ((int64_t)npanels * 64 * vec + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (vec == 1)
        hipLaunchKernelGGL(centre_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, F, n, km, ldf,
                           truth, npanels, Fc, yc, mu);
    else if (vec == 2)
        hipLaunchKernelGGL(centre_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, s, F, n, km, ldf,
                           truth, npanels, Fc, yc, mu);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// panels [NP][K][RP] -> column-major [K][n] (what a Fortran-ordered (n, K) numpy array is)
__global__ __launch_bounds__(256) void unpanelize_kernel(const double* __restrict__ Xp, int64_t n,
                                                         int32_t k, int32_t RP,
                                                         double* __restrict__ out) {
    const int64_t total = n * k;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int32_t j = (int32_t)(e / n);
        const int64_t row = e - (int64_t)j * n;
        const int64_t p = row / RP;
        out[e] = Xp[(p * k + j) * RP + (row - p * RP)];
    }
}

hipError_t launch_unpanelize(const double* Xp, int64_t n, int32_t k, int32_t vec, double* out,
                             hipStream_t s) {
    int64_t blocks = (n * k + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(unpanelize_kernel, dim3((unsigned)blocks), dim3(256), 0, s, Xp, n, k,
                       64 * vec, out);
    return hipGetLastError();
}

// =========================================================================
// residual_rss (stand-alone streaming form)
// =========================================================================
// Wave-slot w walks panels w, w+S, ...; lane holds VEC rows; per column one
// coalesced 64*VEC*sizeof(T)-byte read of X shared by NB coefficient vectors.
// Two-stage sum inside ONE launch: every workgroup leaves a partial, draws a ticket,
// and the workgroup that arrives last adds the partials in index order (so the
// result is bit-reproducible whatever the arrival order).  Hand-off without fences
// (cdna guide Guideline 16, write-through form): 8-byte agent-scope (sc1) stores of
// the partial -> the storing wave's vmcnt(0) -> agent-scope ticket add; the last
// arriver reads every partial with agent-scope (sc1) loads.  The ticket word is
// zero on entry and the last arriver re-zeroes it.
template <typename T, int VEC, int NB, bool NT>
__device__ __forceinline__ void rss_panel_columns(const T* __restrict__ xp, const double* cf, int K,
                                                  double (&acc)[NB][VEC]) {
    constexpr int RP = 64 * VEC;
    // columns (= coalesced reads of 64 * VEC * sizeof(T) bytes) in flight per wave: 8 KiB worth,
    // i.e. 8 of the 1 KiB reads, 16 of the 512-byte ones (C4: f32, two rows per lane; C5: f64,
    // one row per lane) -- with only ~6 waves per CU the depth has to come from each wave
#ifndef BMC_RSS_INFLIGHT_BYTES
#define BMC_RSS_INFLIGHT_BYTES 8192
#endif
    constexpr int UN_ = BMC_RSS_INFLIGHT_BYTES / (RP * (int)sizeof(T));
    constexpr int UN = UN_ < 8 ? 8 : UN_ > 32 ? 32 : UN_;
#pragma unroll UN
    for (int j = 0; j < K; ++j) {
        T xv[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v)
            xv[v] = NT ? __builtin_nontemporal_load(&xp[(int64_t)j * RP + v]) : xp[(int64_t)j * RP + v];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const double c = cf[j * NB + b];
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[b][v] = fma(-(double)xv[v], c, acc[b][v]);
        }
    }
}

template <typename T, int VEC, int NB>
__global__ __launch_bounds__(256) void residual_rss_kernel(
    const T* __restrict__ X, const T* __restrict__ y, int32_t K, int32_t npanels,
    const double* __restrict__ coef, int32_t nb, double* __restrict__ partial,
    unsigned* __restrict__ ticket_word, double* __restrict__ rss, int32_t keep_panels) {
    constexpr int RP = 64 * VEC;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* cf = reinterpret_cast<double*>(smem_raw);         // [K][NB]
    double* red = cf + (size_t)K * NB;                         // [4][NB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int e = tid; e < K * NB; e += 256) {
        const int j = e / NB, b = e % NB;
        cf[e] = b < nb ? coef[(size_t)b * K + j] : 0.0;
    }
    __syncthreads();
    double s[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) s[b] = 0.0;
    const int slots = gridDim.x * 4;
    for (int p = blockIdx.x * 4 + wave; p < npanels; p += slots) {
        const T* xp = X + (int64_t)p * K * RP + lane * VEC;
        // a matrix larger than the Infinity Cache: panels past the first ~190 MB are read
        // non-temporally, so that repeated passes keep finding the first ones cached
        const bool nt = p >= keep_panels;
        double acc[NB][VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const double yv = (double)y[(int64_t)p * RP + lane * VEC + v];
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[b][v] = yv;
        }
        if (nt) rss_panel_columns<T, VEC, NB, true>(xp, cf, K, acc);
        else rss_panel_columns<T, VEC, NB, false>(xp, cf, K, acc);
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int v = 0; v < VEC; ++v) s[b] = fma(acc[b][v], acc[b][v], s[b]);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const double t = wave_sum(s[b]);
        if (lane == 0) red[wave * NB + b] = t;
    }
    __syncthreads();
    if (wave != 0) return;
    if (lane < NB) {
        const double t = ((red[lane] + red[NB + lane]) + red[2 * NB + lane]) + red[3 * NB + lane];
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(partial) +
                               (size_t)blockIdx.x * NB + lane,
                           (unsigned long long)__double_as_longlong(t), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // Up to 256 workgroups draw one ticket.  More than that (C4: 391) finish almost together, and
    // hundreds of atomic adds on ONE word are served one after the other (~30 cycles each,
    // scripts/micro/atomic_allreduce.hip) -- a tail of microseconds behind the last byte read
    // (C4: 11.9 -> 9.7 us per pass with two levels; with 196 or 40 workgroups the second hop costs
    // 0.2-0.3 us more than it saves).  Two levels: workgroup b draws from word 1 + b mod 16 (the
    // words 512 bytes apart), the last of each draws from word 0, the last of those sums.  Every
    // word is left zero by the workgroup that completed it.
    if (gridDim.x <= RSS_FLAT_TICKET_MAX) {
        unsigned ticket = 0;
        if (lane == 0)
            ticket = __hip_atomic_fetch_add(ticket_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ticket = __builtin_amdgcn_readfirstlane(ticket);
        if (ticket != gridDim.x - 1) return;
    } else {
        const unsigned sub = blockIdx.x & (RSS_TICKETS - 1);
        const unsigned n_sub = (gridDim.x - sub + RSS_TICKETS - 1) / RSS_TICKETS;
        unsigned* tw = ticket_word + (size_t)(1 + sub) * RSS_TICKET_STRIDE;
        unsigned ticket = 0;
        if (lane == 0)
            ticket = __hip_atomic_fetch_add(tw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ticket = __builtin_amdgcn_readfirstlane(ticket);
        if (ticket != n_sub - 1) return;
        unsigned top = 0;
        if (lane == 0) {
            __hip_atomic_store(tw, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            top = __hip_atomic_fetch_add(ticket_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        top = __builtin_amdgcn_readfirstlane(top);
        if (top != RSS_TICKETS - 1) return;
    }
    const int ngroups = gridDim.x;   // <= 1024: at most 16 partials per lane
    const unsigned long long* pw = reinterpret_cast<const unsigned long long*>(partial);
    for (int b = 0; b < nb; ++b) {
        // all loads of a coefficient vector in flight together (one memory round trip),
        // then a fixed-order sum
        unsigned long long w[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int gi = lane + 64 * q;
            w[q] = gi < ngroups ? __hip_atomic_load(pw + (size_t)gi * NB + b, __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT)
                                : 0ull;
        }
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += __longlong_as_double((long long)w[q]);
        t = wave_sum(t);
        if (lane == 0) rss[b] = t;
    }
    if (lane == 0)
        __hip_atomic_store(ticket_word, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int32_t rss_groups(const Panels& P) {
    int32_t g = (P.npanels + 3) / 4;
    // 4 workgroups per CU: measured best all-round on MI355X (52 MB: 12.1 us/pass,
    // 520 MB: 95 us/pass = 5.5 TB/s); more groups only add ramp-up and tail
    if (g > 1024) g = 1024;
    if (g < 1) g = 1;
    return g;
}

template <typename T, int VEC>
static hipError_t rss_dispatch(const Panels& P, const double* coef, int32_t nb, double* partial,
                               unsigned* ticket_word, double* rss_out, hipStream_t s) {
    const int32_t G = rss_groups(P);
    const double panel_bytes = (double)(P.k + 1) * 64.0 * P.vec * (P.f32 ? 4 : 8);
    const int32_t keep = panel_bytes * P.npanels > 190e6 ? (int32_t)(190e6 / panel_bytes) : P.npanels;
#define BMC_RSS(NBV)                                                                         \
    do {                                                                                     \
        const size_t lds = ((size_t)P.k * NBV + 4 * NBV) * sizeof(double);                   \
        hipError_t e = hipFuncSetAttribute((const void*)residual_rss_kernel<T, VEC, NBV>,    \
                                           hipFuncAttributeMaxDynamicSharedMemorySize,      \
                                           (int)lds);                                        \
        if (e != hipSuccess) return e;                                                       \
        hipLaunchKernelGGL((residual_rss_kernel<T, VEC, NBV>), dim3(G), dim3(256), lds, s,   \
                           (const T*)P.X, (const T*)P.y, P.k, P.npanels, coef, nb, partial,  \
                           ticket_word, rss_out, keep);                                      \
    } while (0)
    if (nb <= 1) BMC_RSS(1);
    else if (nb <= 2) BMC_RSS(2);
    else if (nb <= 4) BMC_RSS(4);
    else if (nb <= 8) BMC_RSS(8);
    else return hipErrorInvalidValue;
#undef BMC_RSS
    return hipGetLastError();
}

hipError_t launch_residual_rss(const Panels& P, const double* coef, int32_t nb, double* partial,
                               unsigned* ticket_word, double* rss_out, hipStream_t s) {
    if (P.f32) {
        switch (P.vec) {
            case 1: return rss_dispatch<float, 1>(P, coef, nb, partial, ticket_word, rss_out, s);
            case 2: return rss_dispatch<float, 2>(P, coef, nb, partial, ticket_word, rss_out, s);
            case 4: return rss_dispatch<float, 4>(P, coef, nb, partial, ticket_word, rss_out, s);
        }
    } else {
        switch (P.vec) {
            case 1: return rss_dispatch<double, 1>(P, coef, nb, partial, ticket_word, rss_out, s);
            case 2: return rss_dispatch<double, 2>(P, coef, nb, partial, ticket_word, rss_out, s);
        }
    }
    return hipErrorInvalidValue;
}

// =========================================================================
// unrotate: beta_t = W u_t
// =========================================================================
// WT is W transposed (WT[i*K + j] = W[j][i]) so that consecutive threads (j)
// read consecutive words.  Fixed i order -> bit-reproducible.
__global__ __launch_bounds__(256) void unrotate_kernel(const double* __restrict__ u,
                                                       const double* __restrict__ WT,
                                                       int32_t K, int64_t rows,
                                                       double* __restrict__ out) {
    const int64_t total = rows * (K + 1);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t r = e / (K + 1);
        const int32_t j = (int32_t)(e - r * (K + 1));
        const double* ur = u + r * (K + 1);
        double v;
        if (j == K) {
            v = ur[K];
        } else {
            v = 0.0;
            for (int i = 0; i < K; ++i) v = fma(WT[(size_t)i * K + j], ur[i], v);
        }
        out[e] = v;
    }
}

hipError_t launch_unrotate(const double* uout, const double* WT, int32_t k, int64_t rows,
                           double* samples, hipStream_t s) {
    const int64_t total = rows * (k + 1);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(unrotate_kernel, dim3((unsigned)blocks), dim3(256), 0, s, uout, WT, k,
                       rows, samples);
    return hipGetLastError();
}

// =========================================================================
// variates
// =========================================================================
// Element e of a chain's normal stream is half of Box-Muller pair e/2, generated
// from Philox counter (pair_lo, pair_hi, STREAM_NORMAL, 0) under the chain's key,
// so a chain's variates depend on (seed, e) only -- not on the launch geometry,
// the chain's index or the number of GPUs.
__device__ __forceinline__ void box_muller(u32x4 r, double& z0, double& z1) {
    box_muller_pair(u53_open0(r.x, r.y), u53_open0(r.z, r.w), z0, z1);   // bmc_math.h
}

__global__ __launch_bounds__(256) void normal_fill_kernel(const uint64_t* __restrict__ seeds,
                                                          int64_t per_chain,
                                                          double* __restrict__ out) {
    const int c = blockIdx.y;
    const uint64_t seed = seeds[c];
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const int64_t npairs = (per_chain + 1) / 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    double* o = out + (int64_t)c * per_chain;
    for (int64_t pr = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pr < npairs; pr += stride) {
        const u32x4 r = philox4x32_10(u32x4{(uint32_t)pr, (uint32_t)((uint64_t)pr >> 32),
                                            STREAM_NORMAL, 0u}, k0, k1);
        double z0, z1;
        box_muller(r, z0, z1);
        o[2 * pr] = z0;
        if (2 * pr + 1 < per_chain) o[2 * pr + 1] = z1;
    }
}

// Gamma(a, 1), Marsaglia & Tsang (2000).  Attempt m of element t uses Philox
// counters (t_lo, t_hi, STREAM_GAMMA, 2m) and (.., 2m+1).  For a < 1 the usual
// boost Gamma(a) = Gamma(a+1) * U^(1/a) is applied.
__device__ inline double gamma_mt(double a, uint64_t t, uint32_t k0, uint32_t k1) {
    const bool boost = a < 1.0;
    const double aa = boost ? a + 1.0 : a;
    const double d = aa - 1.0 / 3.0;
    const double c = 1.0 / sqrt(9.0 * d);
    double res = d;
    for (uint32_t m = 0; m < 64; ++m) {
        const u32x4 r0 = philox4x32_10(u32x4{(uint32_t)t, (uint32_t)(t >> 32), STREAM_GAMMA, 2 * m},
                                       k0, k1);
        const u32x4 r1 = philox4x32_10(
            u32x4{(uint32_t)t, (uint32_t)(t >> 32), STREAM_GAMMA, 2 * m + 1}, k0, k1);
        double x, unused;
        box_muller(r0, x, unused);
        const double u = u53_open0(r1.x, r1.y);
        double v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        const double x2 = x * x;
        if (u < 1.0 - 0.0331 * x2 * x2 || log(u) < 0.5 * x2 + d * (1.0 - v + log(v))) {
            res = d * v;
            if (boost) res *= pow(u53_open0(r1.z, r1.w), 1.0 / a);
            break;
        }
    }
    return res;
}

__global__ __launch_bounds__(256) void gamma_fill_kernel(const uint64_t* __restrict__ seeds,
                                                         double shape, int64_t per_chain,
                                                         double* __restrict__ out) {
    const int c = blockIdx.y;
    const uint64_t seed = seeds[c];
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < per_chain; t += stride)
        out[(int64_t)c * per_chain + t] = gamma_mt(shape, (uint64_t)t, k0, k1);
}

hipError_t launch_rng_fill(const uint64_t* seeds_dev, int32_t n_chains, int64_t per_chain_normals,
                           double* normals, double shape, int64_t per_chain_gammas,
                           double* gammas, hipStream_t s) {
    if (per_chain_normals > 0) {
        int64_t blocks = ((per_chain_normals + 1) / 2 + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(normal_fill_kernel, dim3((unsigned)blocks, n_chains), dim3(256), 0, s,
                           seeds_dev, per_chain_normals, normals);
    }
    if (per_chain_gammas > 0) {
        int64_t blocks = (per_chain_gammas + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(gamma_fill_kernel, dim3((unsigned)blocks, n_chains), dim3(256), 0, s,
                           seeds_dev, shape, per_chain_gammas, gammas);
    }
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void uniform_fill_kernel(uint32_t k0, uint32_t k1, int64_t n,
                                                           double* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; 2 * i < n; i += stride) {
        const u32x4 r = philox4x32_10(u32x4{(uint32_t)i, (uint32_t)((uint64_t)i >> 32),
                                            STREAM_UNIFORM, 0u}, k0, k1);
        out[2 * i] = u53_open0(r.x, r.y);
        if (2 * i + 1 < n) out[2 * i + 1] = u53_open0(r.z, r.w);
    }
}

hipError_t launch_uniform_fill(uint64_t seed, int64_t n, double* out, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    int64_t blocks = ((n + 1) / 2 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(uniform_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s,
                       (uint32_t)seed, (uint32_t)(seed >> 32), n, out);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void philox_raw_kernel(uint32_t k0, uint32_t k1,
                                                         uint32_t stream, int64_t n4,
                                                         uint32_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const u32x4 r = philox4x32_10(u32x4{(uint32_t)i, (uint32_t)((uint64_t)i >> 32), stream, 0u},
                                  k0, k1);
    out[4 * i + 0] = r.x;
    out[4 * i + 1] = r.y;
    out[4 * i + 2] = r.z;
    out[4 * i + 3] = r.w;
}

hipError_t launch_philox_raw(uint64_t seed, uint32_t stream, int64_t nblocks4, uint32_t* out,
                             hipStream_t s) {
    hipLaunchKernelGGL(philox_raw_kernel, dim3((unsigned)((nblocks4 + 255) / 256)), dim3(256), 0,
                       s, (uint32_t)seed, (uint32_t)(seed >> 32), stream, nblocks4, out);
    return hipGetLastError();
}

}  // namespace bmc
