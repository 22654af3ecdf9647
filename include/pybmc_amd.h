/*
 * pybmc_amd.h -- C ABI of the MI355X-native Gibbs-sampling core for Bayesian
 * model combination (libpybmc_amd.so, gfx950 only).
 *
 * The reference (sudhanvalalit/pybmc) has no FFI: its boundary is three Python
 * call signatures.  Each entry point below names the reference interface it
 * replaces (file:line in the reference checkout).  INTEGRATION.md shows the
 * ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every HOST buffer is caller-owned, is read
 *     or written only during the call and is never retained;
 *   - every function returns a bmc_status (0 = OK); no exception, abort or
 *     longjmp crosses the ABI; bmc_last_error() gives the text;
 *   - a bmc_ctx belongs to one GPU and is driven by one host thread at a time;
 *     distinct contexts are independent (no global mutable state);
 *   - there is NO CPU fallback: without a usable gfx950 device bmc_create fails.
 */
#ifndef PYBMC_AMD_H
#define PYBMC_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PYBMC_AMD_ABI_VERSION 4

typedef struct bmc_ctx bmc_ctx;

typedef enum {
    BMC_OK = 0,
    BMC_EINVAL = 1,    /* bad argument            -> Python ValueError            */
    BMC_ESINGULAR = 2, /* singular C0 or X'X      -> numpy.linalg.LinAlgError     */
    BMC_EHIP = 3,      /* HIP runtime error       -> RuntimeError                 */
    BMC_ENOMEM = 4,    /* allocation failed       -> MemoryError                  */
    BMC_ETIMEOUT = 5,  /* bounded device spin expired (persistent kernel)         */
    BMC_ESTATE = 6     /* call order violated (e.g. run before set_prior)         */
} bmc_status;

enum { BMC_F64 = 0, BMC_F32 = 1 };           /* storage type of X and y        */
enum { BMC_ROW_MAJOR = 0, BMC_COL_MAJOR = 1 }; /* 1 = what U_hat is (F-order) */
enum { BMC_RNG_DEVICE = 0, BMC_RNG_REPLAY = 1 };

/* Launch geometry knobs (0 = let the library choose). */
typedef struct {
    int32_t groups_per_chain; /* workgroups that share one chain's rows        */
    int32_t waves_per_group;  /* 1..8 (workgroup = 64 * waves threads); 1 with no other knob set
                                 also asks for the one-wave-per-chain kernel wherever it exists
                                 (<= 1024 rows, rows/64 x columns <= 128; left to itself the
                                 library also runs such chains in 2, 4 or 8 waves, up to ~8000
                                 rows), > 1 keeps a small chain in the workgroup form */
    int32_t residency;        /* 0 auto, 1 registers, 2 LDS, 3 stream from HBM */
    int32_t panels_per_wave;  /* register residency: 1, 2 or 4 (1, asked for explicitly, also keeps
                                 bundles of 8 chains in the one-panel-per-wave layout instead of
                                 the balanced two-panel one; results are bit-identical) */
    int32_t force_agent_scope;/* 1 = never use the XCD-local (L2) exchange     */
    int32_t chains_per_pass;  /* chains served by one read of X (streamed / LDS-pinned panels) or by
                                 one set of register-resident panels (a chain over the whole chip; or,
                                 one-XCD shapes such as N = 10000 x 32 with 16 chains or more, a BUNDLE
                                 per XCD: 16 .. 64 chains per launch, each bit-identical to its solo
                                 run).  0 auto (up to 8; bundles from 4 chains per XCD on), 1 off,
                                 2/4/8 cap (2 also asks for bundles of 2)                            */
    int32_t rss_mode;         /* 0 (default): rss = sum (y - X beta)^2 by a pass over the data
                                 every iteration, as inference_utils.py:48-51 does.
                                 1 (opt-in, K <= 64): the same number from sufficient statistics,
                                 rss(u) = rss(u0) - 2 d'g0 + d'G d with d = u - u0, G = X~'X~,
                                 g0 = X~'(y - X~ u0) and u0 the least-squares point; no pass
                                 over the data inside the loop, one wave per chain          */
    int32_t cu_limit;         /* 0: the device's CU count.  > 0: plan as if only this many CUs could
                                 hold workgroups of a persistent launch (a CU-masked queue, a
                                 partition, a GPU shared with another process).  The persistent
                                 kernels need all groups of a launch resident at once; geometry and
                                 the residency check (BMC_EINVAL instead of a spin that times out)
                                 follow this number.  The environment variable
                                 PYBMC_AMD_CU_LIMIT, read by bmc_create, is the default for every
                                 context of the process (several ranks on one GPU: set each rank's
                                 share once, e.g. 128 for two)                                 */
} bmc_tuning;

/* Filled by bmc_gibbs_run*.  Times are HIP-event times on the context's stream. */
typedef struct {
    double loop_ms;           /* the persistent Gibbs kernel(s) only           */
    double rng_ms;            /* variate generation kernels (device RNG mode)  */
    double post_ms;           /* un-rotation of the draws into beta            */
    double total_ms;          /* first launch -> last kernel done              */
    int64_t iterations;       /* per chain                                     */
    int32_t n_chains;
    int32_t launches;         /* persistent-kernel launches (chain batches)    */
    int32_t groups_per_chain;
    int32_t waves_per_group;
    int32_t chains_per_pass;
    int32_t residency;        /* 1 registers, 2 LDS, 3 streamed, 4 none (rss_mode 1) */
    int32_t xcd_local_chains; /* chains whose groups were verified on one XCD  */
    int64_t bytes_per_pass;   /* algorithmic: (N*K + N) * sizeof(storage)      */
    int64_t passes;           /* X passes executed in total                    */
} bmc_stats;

/* ---- lifetime --------------------------------------------------------- */
int bmc_abi_version(void);
int bmc_create(int device_id, bmc_ctx** out);
void bmc_destroy(bmc_ctx* ctx);
const char* bmc_last_error(const bmc_ctx* ctx);     /* valid until next call  */
/* Use an existing HIP stream (e.g. torch's current stream); NULL = own stream. */
int bmc_set_stream(bmc_ctx* ctx, void* hip_stream);
int bmc_set_tuning(bmc_ctx* ctx, const bmc_tuning* t);

/* ---- problem: y (n,), X (n,k) ----------------------------------------------
 * Replaces the (y, X) arguments of gibbs_sampler, pybmc/inference_utils.py:4,
 * as passed by BayesianModelCombination.train, pybmc/bmc.py:188-193.
 * Element (i,j) of X is at X[i*ldx + j] (row-major) or X[i + j*ldx] (col-major,
 * the layout U_hat has after inference_utils.py:164).  On return the device
 * holds X, y, and the augmented Gram [X y]'[X y] (inference_utils.py:25 and the
 * loop-invariant X'y of :43), computed with f64 MFMA.
 * bmc_set_problem_device takes DEVICE pointers (same layouts). */
int bmc_set_problem(bmc_ctx* ctx, const void* X, int64_t n, int32_t k, int64_t ldx,
                    int layout, const void* y, int dtype);
int bmc_set_problem_device(bmc_ctx* ctx, const void* dX, int64_t n, int32_t k,
                           int64_t ldx, int layout, const void* dy, int dtype);

/* ---- orthogonalize on the device -----------------------------------------------------
 * Replaces the numerical part of BayesianModelCombination.orthogonalize,
 * pybmc/bmc.py:106-122, and USVt_hat_extraction, pybmc/inference_utils.py:147-168:
 *   mu = row mean of F over the models, y_c = truth - mu, Fc = F - mu          (:106-116)
 *   SVD of Fc through its Gram (f64 MFMA): Fc'Fc = V S^2 V', U_hat = Fc V_k S_k^-1   (:119,:164)
 * F is [n][n_models] with row stride ldf (host).  Outputs (any may be NULL): mean_out [n],
 * yc_out [n], U_hat_out [k][n] (= an (n,k) array in Fortran order, the layout of :164),
 * S_out [k], Vt_out [k][n_models] (rows of V'; Vt_hat of :166 is Vt_out[i] / S_out[i]).
 * Each right singular vector is signed so that its largest entry is positive (LAPACK's signs
 * are arbitrary; the model weights do not depend on them).  On return the context holds the
 * problem (y = y_c, X = U_hat) exactly as after bmc_set_problem, without a host round trip.
 * BMC_ESINGULAR when k reaches the null space of Fc (rows sum to zero: rank <= n_models-1). */
int bmc_orthogonalize(bmc_ctx* ctx, const double* F, int64_t n, int32_t n_models, int64_t ldf,
                      const double* truth, int32_t k, double* mean_out, double* yc_out,
                      double* U_hat_out, double* S_out, double* Vt_out);

/* ---- prior: prior_info = [b0 (k,), C0 (k,k row-major), nu0, sigma20] --------
 * Replaces inference_utils.py:21-37: P = inv(C0) (:22), inv(X'X) and the OLS
 * start value sigma2_0 = max(mean r^2, 1e-6) (:26-37).  Also builds the
 * per-problem basis used by the device loop (DESIGN.md "rotated draw"):
 *   B = P + 1e-6 I = L L',  L^-1 X'X L^-T = Q diag(lam) Q',  W = L^-T Q,
 * so that inv(X'X/s2 + P + 1e-6 I) = W diag(1/(lam/s2 + 1)) W'  (:41).
 * BMC_ESINGULAR when C0 or X'X is singular (numpy raises LinAlgError there). */
int bmc_set_prior(bmc_ctx* ctx, const double* b0, const double* C0, double nu0,
                  double sigma20);

/* ---- introspection used by the parity tests -------------------------------
 * gram: (k+1)x(k+1) row-major [X y]'[X y].  basis: W (k,k row-major), lam (k,),
 * sigma2_init.  moments: mean (k,), cov (k,k) of beta | sigma2 (:41-44). */
int bmc_get_gram(bmc_ctx* ctx, double* gram_out);
int bmc_get_basis(bmc_ctx* ctx, double* W_out, double* lam_out, double* sigma2_init);
int bmc_conditional_moments(bmc_ctx* ctx, double sigma2, double* mean_out,
                            double* cov_out);

/* ---- residual reduction: rss[b] = sum_i (y_i - sum_j X_ij beta[b][j])^2 -------
 * Replaces inference_utils.py:48-51 as a stand-alone streaming kernel over the
 * un-rotated X (nb coefficient vectors share one pass over X). */
int bmc_residual_rss(bmc_ctx* ctx, const double* beta, int32_t nb, double* rss_out);
/* Same kernel launched `reps` times back to back on data already in HBM; returns
 * the HIP-event time per launch.  Roofline measurement only. */
int bmc_residual_rss_bench(bmc_ctx* ctx, int32_t nb, int32_t reps, double* ms_per_launch);
/* The augmented Gram [X y]'[X y] of the resident problem (f64 MFMA kernel + its reduction,
 * inference_utils.py:25 and the X'y of :43) launched `reps` times back to back; HIP-event time
 * per launch pair.  Roofline measurement only. */
int bmc_gram_bench(bmc_ctx* ctx, int32_t reps, double* ms_per_launch);

/* ---- the Gibbs loop ---------------------------------------------------------
 * Replaces the loop of gibbs_sampler, pybmc/inference_utils.py:39-56, for
 * n_chains independent chains.  samples_out is [n_chains][iters][k+1] f64,
 * row t = [beta_t (k), sigma_t = sqrt(sigma2_t)]  (:54).
 * rng_mode BMC_RNG_DEVICE: seeds[n_chains]; variates from the on-device Philox
 *   generator (xi, g must be NULL).
 * rng_mode BMC_RNG_REPLAY: xi [n_chains][iters][k] standard-normal innovations
 *   in the basis of bmc_get_basis, g [n_chains][iters] Gamma((nu0+n)/2, 1)
 *   variates (seeds may be NULL).  Used to replay the reference's chain.
 * bmc_gibbs_run_device is the device-RNG form writing to a caller-owned DEVICE
 * buffer (no device->host copy of the samples); it returns after the stream has
 * drained so that the status words and the event times in `stats` are final. */
int bmc_gibbs_run(bmc_ctx* ctx, int32_t n_chains, int64_t iters, const uint64_t* seeds,
                  int rng_mode, const double* xi, const double* g, double* samples_out,
                  bmc_stats* stats);
int bmc_gibbs_run_device(bmc_ctx* ctx, int32_t n_chains, int64_t iters,
                         const uint64_t* seeds, void* d_samples_out, bmc_stats* stats);

/* ---- simplex-constrained sampler ---------------------------------------------------
 * Replaces gibbs_sampler_simplex, pybmc/inference_utils.py:59-144 (dispatched from
 * pybmc/bmc.py:173-186): random-walk Metropolis on beta with the model weights
 * beta Vt_hat + 1/n_models kept non-negative, Gibbs step for sigma2.  Uses the problem
 * of bmc_set_problem (no prior call needed).  samples_out is [iters][k+1] rows
 * [beta, sigma]; *accepted_out counts acceptances in the sampling phase (:135).
 * rng_mode BMC_RNG_DEVICE: variates from the Philox generator keyed by seed.
 * rng_mode BMC_RNG_REPLAY: xi [burn+iters][k] standard-normal proposal innovations
 *   (proposal = current + S_hat*stepsize*xi), unif [n_unif] uniforms consumed ONLY by
 *   proposals inside the simplex (:110,:132), g [burn+iters] Gamma((nu0+n)/2,1) variates.
 * The reference's argument checks (:91-94) are the caller's (Python) job: burn >= 0,
 * stepsize > 0 are re-checked here and give BMC_EINVAL. */
int bmc_simplex_run(bmc_ctx* ctx, const double* Vt_hat, int32_t n_models, const double* S_hat,
                    int64_t iters, int64_t burn, double stepsize, double nu0, double sigma20,
                    int rng_mode, uint64_t seed, const double* xi, const double* unif,
                    int64_t n_unif, const double* g, double* samples_out,
                    int64_t* accepted_out, int64_t* unif_used_out, bmc_stats* stats);

/* ---- posterior predictive -----------------------------------------------------
 * Replaces rndm_m_random_calculator, pybmc/sampling_utils.py:40-84 (callers
 * pybmc/bmc.py:227,323,367), and the interval test of coverage, :24-34.
 *   theta   [n_draws][k+1]   the posterior rows the caller selected (:57; the reference
 *                            draws 10000 of them without replacement), last column sigma
 *   weights = theta[:, :k] Vt_hat + 1/n_models                               (:60-67)
 *   rndm_m[s][p] = weights[s] . preds[p] + z[s][p] sigma_s                   (:70-77)
 * rng_mode BMC_RNG_DEVICE: z from the Philox generator keyed by seed (noise = NULL);
 * BMC_RNG_REPLAY: noise is [n_draws][n_points] row-major standard normals.
 * Order statistics: for each of n_q requests, numpy's linear interpolation between
 * sorted[q_index] and sorted[q_index+1] with weight q_gamma (:80-82); bands_out is
 * [n_q][n_points].  Coverage (optional, truth != NULL): hits[c] counts the points with
 * sorted[cov_lo[c]] <= truth <= sorted[cov_hi[c]]; n_q, n_cov <= 64; n_draws <= 16384.
 * rndm_m_out (optional) is [n_points][n_draws]: the reference's (n_draws, n_points)
 * array in Fortran order (bmc_predict_draws returns it C-ordered). */
int bmc_predict(bmc_ctx* ctx, const double* preds, int64_t n_points, int32_t n_models,
                const double* theta, int32_t n_draws, int32_t k, const double* Vt_hat,
                int rng_mode, uint64_t seed, const double* noise,
                const int32_t* q_index, const double* q_gamma, int32_t n_q,
                const double* truth, const int32_t* cov_lo, const int32_t* cov_hi,
                int32_t n_cov, double* rndm_m_out, double* bands_out, int64_t* cov_hits_out);

/* The draws of the LAST bmc_predict on this context, in the layout the caller wants (they stay
 * on the device until the next bmc_predict or bmc_destroy, so a caller that asked bmc_predict for
 * bands / coverage only can still fetch them afterwards):
 *   BMC_DRAWS_BY_POINT  out[n_points][n_draws]   the device layout
 *   BMC_DRAWS_BY_DRAW   out[n_draws][n_points]   a C-ordered (n_draws, n_points) array: exactly
 *                       what rndm_m_random_calculator returns, pybmc/sampling_utils.py:77
 *                       (transposed on the device, one contiguous copy back).
 * BMC_ESTATE when no bmc_predict has run on this context. */
enum { BMC_DRAWS_BY_POINT = 0, BMC_DRAWS_BY_DRAW = 1 };
int bmc_predict_draws(bmc_ctx* ctx, double* out, int layout);

/* HIP-event times of the LAST bmc_predict on this context (any pointer may be NULL):
 * h2d_ms the host->device copies of its inputs, gemm_ms the weight + MFMA GEMM(+noise)
 * kernels, orderstat_ms the selection / sort kernels, device_ms first copy -> last kernel. */
int bmc_predict_timing(bmc_ctx* ctx, double* h2d_ms, double* gemm_ms, double* orderstat_ms,
                       double* device_ms);

/* ---- pooling the chains of several GPUs (a capability the reference lacks) ----
 * One process per GPU; rank r samples its block of chains with no communication, then ONE
 * all-gather over RCCL (xGMI) pools the per-rank blocks.  RCCL (librccl.so.1) is loaded on
 * first use, so a single-GPU caller needs no RCCL at all.
 *   bmc_comm_unique_id   rank 0 creates the 128-byte RCCL id; the caller hands it to the other
 *                        ranks by whatever transport it has (file, socket, MPI, env)
 *   bmc_comm_init        collective over all ranks; binds the communicator to ctx's device
 *   bmc_allgather        d_recv[r * count .. (r+1) * count) = rank r's d_send[0 .. count), f64;
 *                        DEVICE pointers; d_send may alias its own slot of d_recv (in place).
 *                        Runs on the context's stream, after every sampler launch queued there,
 *                        and returns when the pooled block is complete.
 *   bmc_comm_destroy     also done by bmc_destroy */
#define BMC_COMM_ID_BYTES 128
int bmc_comm_unique_id(char id_out[BMC_COMM_ID_BYTES]);
int bmc_comm_init(bmc_ctx* ctx, int32_t world, int32_t rank, const char id[BMC_COMM_ID_BYTES]);
int bmc_allgather(bmc_ctx* ctx, const void* d_send, void* d_recv, int64_t count_per_rank);
int bmc_comm_destroy(bmc_ctx* ctx);

/* ---- on-device variates (exposed so the generator itself can be tested) ----
 * normals_out [count_normal] ~ N(0,1); gammas_out [count_gamma] ~ Gamma(shape,1). */
int bmc_rng_fill(bmc_ctx* ctx, uint64_t seed, int64_t count_normal, double* normals_out,
                 double shape, int64_t count_gamma, double* gammas_out);
/* Raw Philox4x32-10 blocks: out[4*i..4*i+3] = philox(counter = (i_lo, i_hi, stream_id, 0),
 * key = seed) for i < nblocks4.  Integer output, checked bit-for-bit by the tests. */
int bmc_philox_raw(bmc_ctx* ctx, uint64_t seed, uint32_t stream_id, int64_t nblocks4,
                   uint32_t* out);

#ifdef __cplusplus
}
#endif
#endif /* PYBMC_AMD_H */
