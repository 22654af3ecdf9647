// Host-callable launchers of the gfx950 kernels (one per kernel family).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bmc {

// Storage description of the panelised problem (see bmc_dev.h "panel layout").
// u64 words between the granule pairs of consecutive groups in a (chain, parity) slot
#ifndef BMC_GRAN_PAIR_STRIDE
#define BMC_GRAN_PAIR_STRIDE 8
#endif
constexpr int GRAN_PAIR_STRIDE = BMC_GRAN_PAIR_STRIDE;
// u64 words of one (chain, parity) exchange slot: G <= 32 groups use pairs 0..G-1; larger
// chains use 8 team areas of 32 pairs, 8 team-total pairs and 8 relay pairs (bmc_loop.h,
// exchange_sum)
// (the 8 team-total pairs and the 8 relay pairs of the second level are written by one XCD each
// and polled by every group of the chain at agent scope: GRAN_L2_STRIDE words = 512 bytes
// apart.  Same-box A/B, us per iteration at 64 B / 128 B / 256 B / 512 B / 1 KiB per pair:
// N = 100 000 x 32 2.005 / 1.988 / 1.945 / 1.881 / 1.919, C4 3.291 / 3.241 / 3.156 / 3.131 / 3.164,
// N = 30 000 x 64 2.112 / 2.069 / 2.038 / 1.986 / 2.073 -- the polls of ~200 groups spread over
// the memory channels instead of landing on one or two)
#ifndef BMC_GRAN_L2_STRIDE
#define BMC_GRAN_L2_STRIDE 64
#endif
constexpr int GRAN_L2_STRIDE = BMC_GRAN_L2_STRIDE;
inline int gran_slot_words(int G) {
    return G <= 32 ? ((2 * G + 31) / 32) * 16 * GRAN_PAIR_STRIDE
                   : 256 * GRAN_PAIR_STRIDE + 16 * GRAN_L2_STRIDE;
}

struct Panels {
    const void* X;   // [NP][K][RP] of T
    const void* y;   // [NP*RP] of T
    int64_t n;       // true row count
    int32_t k;
    int32_t vec;     // rows per lane (RP = 64*vec)
    int32_t npanels;
    int32_t f32;     // 1: T = float, 0: T = double
    int32_t stream_keep;  // streaming loop: a group's first stream_keep panels are read with
                          // ordinary loads (they stay in the XCD's L2 from one iteration to
                          // the next), the others with non-temporal loads
};

// ---- set-up ----------------------------------------------------------------
// user layout (row/col-major, f32/f64, host-uploaded or device) -> panels
hipError_t launch_panelize(const void* Xsrc, const void* ysrc, int64_t n, int32_t k,
                           int64_t ldx, int col_major, int f32, int32_t vec,
                           void* Xp, void* yp, int32_t npanels, hipStream_t s);

// Augmented Gram [X y]'[X y] with v_mfma_f64_16x16x4_f64.  gram_out is
// (k+1)x(k+1) row-major f64.  scratch: >= gram_scratch_bytes().
size_t gram_scratch_bytes(const Panels& P);
hipError_t launch_gram(const Panels& P, void* scratch, double* gram_out, hipStream_t s);

// Xrot = X * W (W is P.k x ko row-major, device); output panels have ko columns, same
// rows per lane and storage type.
hipError_t launch_rotate(const Panels& P, const double* W, int32_t ko, void* Xrot, hipStream_t s);

// orthogonalize helpers: centring into panels, panels -> column-major
hipError_t launch_centre(const double* F, int64_t n, int32_t km, int64_t ldf, const double* truth,
                         int32_t vec, int32_t npanels, double* Fc, double* yc, double* mu,
                         hipStream_t s);
hipError_t launch_unpanelize(const double* Xp, int64_t n, int32_t k, int32_t vec, double* out,
                             hipStream_t s);

// rss[b] = sum_i (y_i - sum_j X_ij coef[b][j])^2, b < nb (nb <= 8), one launch.
// partial: >= rss_groups(P) * 8 doubles of scratch; ticket_word: RSS_TICKET_BYTES of zeroed u32 that
// the kernel leaves zero again.
constexpr int RSS_TICKETS = 16, RSS_TICKET_STRIDE = 128;   // u32 words: 512 bytes between ticket words
constexpr unsigned RSS_FLAT_TICKET_MAX = 256;   // more workgroups than this draw tickets in two levels
constexpr size_t RSS_TICKET_BYTES = (size_t)(RSS_TICKETS + 1) * RSS_TICKET_STRIDE * 4;
int32_t rss_groups(const Panels& P);
hipError_t launch_residual_rss(const Panels& P, const double* coef, int32_t nb,
                               double* partial, unsigned* ticket_word, double* rss_out,
                               hipStream_t s);

// samples[c][t][0..k) = W u[c][t][0..k);  samples[c][t][k] = u[c][t][k]
hipError_t launch_unrotate(const double* uout, const double* W, int32_t k, int64_t rows,
                           double* samples, hipStream_t s);

// ---- variates ---------------------------------------------------------------
// normals[c][e], e < per_chain_normals; gammas[c][t], t < per_chain_gammas
hipError_t launch_rng_fill(const uint64_t* seeds_dev, int32_t n_chains,
                           int64_t per_chain_normals, double* normals, double shape,
                           int64_t per_chain_gammas, double* gammas, hipStream_t s);
hipError_t launch_philox_raw(uint64_t seed, uint32_t stream, int64_t nblocks4,
                             uint32_t* out, hipStream_t s);

// ---- posterior predictive --------------------------------------------------------
struct PredictArgs {
    const double* preds;   // [M][Km]
    int64_t M;
    int32_t Km, k, S;      // models, kept components, draws
    int32_t S_pad, Km_pad; // multiples of 64 and 4
    int64_t M_pad;         // points rounded up to whole 64-point tiles
    double* P;             // [M_pad][Km_pad] + 16 doubles of slack: preds zero-padded (scratch)
    const double* theta;   // [S][k+1] selected posterior rows
    const double* Vt;      // [k][Km]
    double* Wt;            // [S_pad][Km_pad] + 16 doubles of slack, scratch
    double* sig;           // [S_pad] scratch
    uint64_t seed;
    const double* noise_replay;  // [S][M] or NULL (device generator)
    double* R;             // [M_pad][S_pad]
    const int32_t* q_index;
    const double* q_gamma;
    int32_t n_q;
    const double* truth;   // [M] or NULL
    const int32_t* cov_lo;
    const int32_t* cov_hi;
    int32_t n_cov;
    double* bands;         // [n_q][M]
    unsigned long long* hits;  // [n_cov], zeroed
    int32_t* fail_points;  // [M] points the selection kernel hands to the sort kernel (or NULL)
    int32_t* fail_count;   // [1], zeroed
    hipEvent_t ev_mid = nullptr;  // recorded between the GEMM and the order statistics (or NULL)
};
hipError_t launch_predict(const PredictArgs& a, hipStream_t s);
// out[s][p] = R[p][s]: the draws as the reference's C-ordered (n_draws, n_points) array
hipError_t launch_transpose_draws(const double* R, int64_t M, int32_t S, int32_t S_pad, double* out,
                                  hipStream_t s);

// ---- the persistent Gibbs loop ------------------------------------------------
struct GibbsArgs {
    Panels P;               // ROTATED panels
    const double* lam;      // [k]
    const double* c1;       // [k]  W' P b0
    const double* c2;       // [k]  W' X'y
    double nu0_s20;         // nu0 * sigma20
    double sigma2_init;
    const double* xi;       // [C][T][k]
    const double* gam;      // [C][T]
    double* uout;           // [C][T][k+1]
    unsigned long long* gran;  // [C][3][gran_stride]: 2 parities of granules + XCC words, zeroed
    int32_t gran_stride;    // u64 words per (chain, parity), >= 2*G, multiple of 32
    int32_t* status;        // [C] 0 ok, 1 timeout
    int64_t iters;
    int32_t n_chains;       // chains in THIS launch
    int32_t G;              // workgroups per chain
    int32_t waves;          // waves per workgroup
    int32_t mode;           // 0 registers, 1 LDS, 2 streaming
    int32_t reg_ppw;        // panels per wave (register mode)
    int32_t nslot;          // grid = nslot x G; 8 = one slot per XCD, else = n_chains
    int32_t force_agent_scope;  // 1: never use the XCD-local exchange
    int32_t chains_per_pass;  // > 1: gibbs_multi_kernel, bundles of that many chains
    uint32_t epoch0 = 0;      // nonce of this launch: exchange tags are epoch0 + t + 1, placement
                              // words carry it in their high half (host: never lets a tag be 0)
    int32_t bundle_bal = 0;   // 1: bundles of 8 in the balanced two-panels-per-wave layout (<= 5 panels
                              // per group, 8 waves; PanelStore::partial_rss_reg_bal)
    int32_t bundle_slots = 0; // gibbs_multi_kernel: 0 = ONE bundle, grid = G (a chain over the whole
                              // chip); > 0 = grid = bundle_slots x G, bundle b = blockIdx % slots
                              // (one bundle per XCD: chains b * cpp .. b * cpp + cpp - 1), n_chains
                              // = bundles * chains_per_pass, unused slots leave at once
    int32_t* placement;     // [C] out: 1 = chain verified on one XCD (L2-local exchange)
    int32_t panels_per_group;  // max panels a group owns
    long long* dbg;         // diagnostic builds only (-DBMC_STAMPS); NULL otherwise
    int32_t* query_regs;    // host pointer; when set launch_gibbs launches nothing and reports the
                            // VGPR count of the packed (<= 128 VGPR) variant of the kernel it
                            // would have launched, 0 if that shape has none
    int32_t pack;           // 1: launch the packed variant (two chains per XCD)
    int32_t one_wave = 0;   // 1: gibbs_wave_kernel, one wave per chain (gibbs_wave_capacity() > 0)
    int32_t* query_occupancy = nullptr;  // host pointer; when set launch_gibbs launches nothing and
                            // reports how many workgroups of the kernel / block size / LDS bytes it
                            // would have launched ONE CU admits (hipOccupancyMaxActiveBlocksPer
                            // Multiprocessor): the persistent kernels need every group resident
};
// rss from sufficient statistics (bmc_tuning.rss_mode = 1): one wave per chain, K <= 64
struct GramArgs {
    int32_t k;
    const double* lam;      // [k]
    const double* c1;       // [k]
    const double* c2;       // [k]
    const double* Gt;       // [k][k]  X~'X~ (symmetric)
    const double* u0;       // [k]     centre of the expansion (least-squares point)
    const double* g0;       // [k]     X~'(y - X~ u0)
    double rss0;            // rss(u0), from one residual pass
    double nu0_s20, sigma2_init;
    const double* xi;       // [C][T][k]
    const double* gam;      // [C][T]
    double* uout;           // [C][T][k+1]
    int64_t iters;
    int32_t n_chains;
};
hipError_t launch_gibbs_gram(const GramArgs& a, hipStream_t s);
// register-resident FMAs per iteration of the one-wave-per-chain kernel for k columns and
// npanels panels of 64 rows PER WAVE (f64 or f32 storage, one row per lane), 0 = no such kernel;
// a chain runs in 1, 2 or 4 such waves (GibbsArgs.waves)
int gibbs_wave_capacity(int k, int npanels);

struct SimplexArgs {
    Panels P;               // UN-rotated panels (the simplex sampler proposes beta itself)
    const double* Vt;       // [k][Km]  Vt_hat
    int32_t Km;
    int32_t vt_in_lds;      // Vt_hat kept in LDS (k*Km <= 4096 doubles)
    const double* step;     // [k]  S_hat * stepsize
    double nu0_s20;
    double rss_init;        // sum y^2 (beta = 0)
    const double* xi;       // [burn+iters][k]
    const double* unif;     // [n_unif]
    int64_t n_unif;
    const double* gam;      // [burn+iters]
    double* out;            // [iters][k+1]
    unsigned long long* gran;
    int32_t gran_stride;
    int32_t* status;        // 0 ok, 1 timeout, 2 uniforms exhausted
    int32_t* placement;
    long long* counters;    // [2] accepted (sampling phase), uniforms consumed
    int64_t iters, burn;
    int32_t G, waves, mode, reg_ppw, nslot, force_agent_scope, panels_per_group;
    uint32_t epoch0 = 0;    // as in GibbsArgs
    int32_t* query_occupancy = nullptr;  // as in GibbsArgs
    int32_t one_wave = 0;   // 1: simplex_wave_kernel (gibbs_wave_capacity() > 0, Km <= 64)
};
size_t simplex_lds_bytes(const SimplexArgs& a);
hipError_t launch_simplex(const SimplexArgs& a, hipStream_t s);
// uniforms in (0,1]: out[i] = u53(philox(counter = (i, STREAM_UNIFORM), key = seed))
hipError_t launch_uniform_fill(uint64_t seed, int64_t n, double* out, hipStream_t s);

size_t gibbs_lds_bytes(const GibbsArgs& a);
int gibbs_reg_capacity(int k, int f32, int rows_per_lane);  // 1 if that many rows of k columns fit in VGPRs
int gibbs_reg_multi_cap(int k, bool f32, int vec);  // most chains per pass in register residency
hipError_t launch_gibbs(const GibbsArgs& a, hipStream_t s);

}  // namespace bmc
