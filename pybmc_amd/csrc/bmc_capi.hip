// C ABI of libpybmc_amd.so (see include/pybmc_amd.h).  Host orchestration only:
// every O(N) step is a gfx950 kernel; the host does the one-off K x K algebra.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>   // types only: the library itself is loaded on first use (bmc_comm_*)

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pybmc_amd.h"
#include "bmc_launch.h"
#include "host_linalg.hpp"

using namespace bmc;

namespace {

constexpr size_t LDS_LIMIT = 160 * 1024;

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

}  // namespace

struct bmc_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    bmc_tuning tune{};
    int n_cu = 256;
    int env_cu_limit = 0;   // PYBMC_AMD_CU_LIMIT at bmc_create: the default of bmc_tuning.cu_limit
    uint64_t nonce_state = 0x9E3779B97F4A7C15ull;   // per-launch exchange nonces (launch_nonce)

    // problem
    bool have_problem = false, have_prior = false;
    int64_t n = 0;
    int32_t k = 0, vec = 1, npanels = 0, f32 = 0;
    DevBuf Xraw, Yp, Xrot;
    std::vector<double> gram;

    // prior / basis (host copies)
    std::vector<double> W, lam, c1, c2, b0;
    double nu0 = 0, s20 = 0, sigma2_init = 0;
    DevBuf dW, dWT, dLam, dC1, dC2;
    // sufficient statistics in the rotated basis (rss_mode 1; host part made by bmc_set_prior
    // when k <= 64, device part on first use)
    std::vector<double> Gt, u0, g0;
    bool have_gram_dev = false;
    double rss0 = 0;
    DevBuf dGt, dU0, dG0;

    // scratch
    DevBuf gramScratch, gramOut, rssPartial, rssOut, coef, stage, ticket;
    // run buffers
    DevBuf xi, gam, uout, samples, gran, status, seeds, dbg, placement;
    // predictive buffers
    DevBuf pPreds, pPad, pTheta, pVt, pWt, pSig, pR, pRT, pNoise, pAux, pBands;
    int64_t pM = 0;                        // last bmc_predict: points, draws, padded draws
    int32_t pS = 0, pS_pad = 0;
    DevBuf sVt, sStep, sUnif, sOut, sCnt;
    DevBuf oFc, oMu, oW, oOut;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    double predict_ms[4] = {0, 0, 0, 0};   // last bmc_predict: h2d, gemm, order statistics, device
    // pinned staging for results that go back to pageable host memory (copy_to_host)
    void* hstage[2] = {nullptr, nullptr};
    hipEvent_t hev[2] = {nullptr, nullptr};
    // pooling over GPUs (bmc_comm_*): RCCL communicator bound to this context's device
    ncclComm_t comm = nullptr;
    int32_t comm_world = 0, comm_rank = 0;
};

namespace {

// Base of a persistent launch's exchange tags (GibbsArgs.epoch0): different for every launch
// (splitmix64 of a per-context counter), so that words an earlier launch left behind -- in the
// exchange buffer the launch re-zeroes, or in a cache that still holds a line of it -- can never
// be taken for this launch's.  Tags are epoch0 + 1 .. epoch0 + n_tags; none may be 0 (the
// zeroed state), so the base keeps them below 2^32 when the run is short enough to allow it.
uint32_t launch_nonce(bmc_ctx* c, uint64_t n_tags) {
    uint64_t z = (c->nonce_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const uint64_t room = 0xffffffffull - (n_tags + 1);   // largest base with no wrap
    if (n_tags + 2 >= 0xffffffffull) return 0;
    return (uint32_t)(1 + z % room);
}

int fail(bmc_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}

#define HIPCHK(ctx, expr)                                                              \
    do {                                                                               \
        hipError_t e__ = (expr);                                                       \
        if (e__ != hipSuccess)                                                         \
            return fail(ctx, e__ == hipErrorOutOfMemory ? BMC_ENOMEM : BMC_EHIP,       \
                        std::string(#expr) + ": " + hipGetErrorString(e__));           \
    } while (0)

int ensure(bmc_ctx* c, DevBuf& b, size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (b.cap >= bytes) return BMC_OK;
    if (b.p) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    HIPCHK(c, hipMalloc(&b.p, bytes));
    b.cap = bytes;
    return BMC_OK;
}

void release(DevBuf& b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

// Device -> caller-owned (pageable) host memory.  A plain hipMemcpy into fresh numpy memory
// pins the destination pages first, which costs far more than the transfer for the sizes that
// matter here (13.2 MB of samples at C2: 29 ms, against 0.3 ms of DMA -- bench.py extra.e2e).
// Instead: DMA into two pinned staging blocks in turn and copy out of one with the CPU while the
// next is in flight; large results are copied out by several host threads (one thread moves
// ~8 GB/s into untouched pages, the 4 GB of C5's rndm_m would take 0.5 s): 8 threads from 8 MB on.
constexpr size_t HSTAGE_BYTES = (size_t)32 << 20;
void host_copy(char* dst, const char* src, size_t bytes) {
    const size_t MT_MIN = (size_t)8 << 20;
    // (C5's 4 GB of draws, whole bmc_predict call: 2 threads 0.184 s, 4 0.119, 8 0.097, 16 0.093)
    unsigned nt = bytes >= MT_MIN ? 8 : 1;
    const unsigned hw = std::thread::hardware_concurrency();
    if (hw && nt > hw) nt = hw;
    if (nt <= 1) {
        std::memcpy(dst, src, bytes);
        return;
    }
    std::vector<std::thread> th;
    const size_t per = ((bytes / nt) + 4095) & ~(size_t)4095;
    for (unsigned i = 0; i < nt; ++i) {
        const size_t off = (size_t)i * per;
        if (off >= bytes) break;
        const size_t n = bytes - off < per ? bytes - off : per;
        th.emplace_back([=] { std::memcpy(dst + off, src + off, n); });
    }
    for (auto& t : th) t.join();
}

// rows of `row_bytes` taken every `src_pitch` bytes on the device, written densely to `dst`
// (src_pitch == row_bytes: one contiguous block of row_bytes * rows).  Blocks until the data is
// in `dst`.
int copy_to_host(bmc_ctx* c, void* dst, const void* src_dev, size_t row_bytes, size_t src_pitch,
                 size_t rows) {
    if (row_bytes == 0 || rows == 0) return BMC_OK;
    for (int i = 0; i < 2; ++i) {
        if (!c->hstage[i] && hipHostMalloc(&c->hstage[i], HSTAGE_BYTES, hipHostMallocDefault) != hipSuccess) {
            c->hstage[i] = nullptr;
            (void)hipGetLastError();
        }
        if (!c->hev[i] && hipEventCreateWithFlags(&c->hev[i], hipEventDisableTiming) != hipSuccess)
            c->hev[i] = nullptr;
    }
    const bool dense = src_pitch == row_bytes;
    const bool staged = c->hstage[0] && c->hstage[1] && c->hev[0] && c->hev[1] &&
                        (dense || row_bytes <= HSTAGE_BYTES);
    if (!staged) {   // (no pinned memory to be had: the plain route)
        HIPCHK(c, hipMemcpy2DAsync(dst, row_bytes, src_dev, src_pitch, row_bytes, rows,
                                   hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return BMC_OK;
    }
    // pieces: dense -> byte ranges of at most one staging block; pitched -> whole rows
    const size_t total = dense ? row_bytes * rows : rows;          // bytes, or rows
    const size_t per = dense ? HSTAGE_BYTES : HSTAGE_BYTES / row_bytes;
    const size_t unit = dense ? 1 : row_bytes;                     // host bytes per unit of `total`
    char* out = (char*)dst;
    size_t issued = 0;
    size_t p_n[2] = {0, 0}, p_off[2] = {0, 0};
    bool pending[2] = {false, false};
    int next = 0;   // slot of the next transfer; the other slot holds the older pending one
    while (issued < total || pending[0] || pending[1]) {
        while (issued < total && !pending[next]) {
            const size_t n = total - issued < per ? total - issued : per;
            if (dense)
                HIPCHK(c, hipMemcpyAsync(c->hstage[next], (const char*)src_dev + issued, n,
                                         hipMemcpyDeviceToHost, c->stream));
            else
                HIPCHK(c, hipMemcpy2DAsync(c->hstage[next], row_bytes,
                                           (const char*)src_dev + issued * src_pitch, src_pitch,
                                           row_bytes, n, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipEventRecord(c->hev[next], c->stream));
            pending[next] = true;
            p_n[next] = n;
            p_off[next] = issued * unit;
            issued += n;
            next ^= 1;
        }
        // drain the older pending block (`next` if both are pending, else the one that is)
        const int o = pending[next] ? next : next ^ 1;
        HIPCHK(c, hipEventSynchronize(c->hev[o]));
        host_copy(out + p_off[o], (const char*)c->hstage[o], p_n[o] * unit);
        pending[o] = false;
        next = o;
    }
    return BMC_OK;
}

Panels panels_of(const bmc_ctx* c, const void* X) {
    Panels P;
    P.X = X;
    P.y = c->Yp.p;
    P.n = c->n;
    P.k = c->k;
    P.vec = c->vec;
    P.npanels = c->npanels;
    P.f32 = c->f32;
    P.stream_keep = 1 << 30;
    return P;
}

// rows per lane: one row per lane whenever the whole matrix can stay in the chip's VGPRs
// (256 CUs x 8 waves x ppw panels of 64 rows); otherwise wide (16-byte) reads once there
// are enough panels to occupy the chip, narrower panels for small N.
int choose_vec(int64_t n, int32_t k, int f32) {
    // register residency: the narrowest panel that lets every panel have its own wave (more
    // waves = shorter serial FMA phase); two rows per lane (wider reads for the streaming
    // kernels that share the layout) once one row per lane would need two panels per wave
    const int64_t waves_chip = 256 * 8;   // sized for the full chip; geometry re-checks the fit
    if (gibbs_reg_capacity(k, f32, 1) && (n + 63) / 64 <= waves_chip) return 1;
    if (gibbs_reg_capacity(k, f32, 2) && (n + 127) / 128 <= waves_chip) return 2;
    for (int ppw : {2, 4})
        if (gibbs_reg_capacity(k, f32, ppw) && (n + 63) / 64 <= waves_chip * ppw) return 1;
    int vec = f32 ? 4 : 2;
    while (vec > 1 && (n + 64 * vec - 1) / (64 * vec) < 1024) vec >>= 1;
    return vec;
}

int finish_problem(bmc_ctx* c);

int set_problem_common(bmc_ctx* c, const void* dX, const void* dy, int64_t n, int32_t k,
                       int64_t ldx, int layout, int dtype) {
    c->have_problem = c->have_prior = false;
    c->n = n;
    c->k = k;
    c->f32 = dtype == BMC_F32;
    c->vec = choose_vec(n, k, c->f32);
    const int RP = 64 * c->vec;
    c->npanels = (int32_t)((n + RP - 1) / RP);
    const size_t es = c->f32 ? 4 : 8;
    int rc;
    if ((rc = ensure(c, c->Xraw, (size_t)c->npanels * k * RP * es))) return rc;
    if ((rc = ensure(c, c->Xrot, (size_t)c->npanels * k * RP * es))) return rc;
    if ((rc = ensure(c, c->Yp, (size_t)c->npanels * RP * es))) return rc;
    HIPCHK(c, launch_panelize(dX, dy, n, k, ldx, layout == BMC_COL_MAJOR, c->f32, c->vec,
                              c->Xraw.p, c->Yp.p, c->npanels, c->stream));
    return finish_problem(c);
}

// Gram of the panelised problem -> host copy; marks the problem as set.
int finish_problem(bmc_ctx* c) {
    const int32_t k = c->k;
    int rc;
    const Panels P = panels_of(c, c->Xraw.p);
    if ((rc = ensure(c, c->gramScratch, gram_scratch_bytes(P)))) return rc;
    const size_t gsz = (size_t)(k + 1) * (k + 1);
    if ((rc = ensure(c, c->gramOut, gsz * 8))) return rc;
    HIPCHK(c, launch_gram(P, c->gramScratch.p, (double*)c->gramOut.p, c->stream));
    c->gram.assign(gsz, 0.0);
    HIPCHK(c, hipMemcpyAsync(c->gram.data(), c->gramOut.p, gsz * 8, hipMemcpyDeviceToHost,
                             c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_problem = true;
    return BMC_OK;
}

int check_problem_args(bmc_ctx* c, const void* X, int64_t n, int32_t k, int64_t ldx, int layout,
                       const void* y, int dtype) {
    if (!c) return BMC_EINVAL;
    if (!X || !y) return fail(c, BMC_EINVAL, "X and y must not be NULL");
    if (n < 1 || k < 1) return fail(c, BMC_EINVAL, "need n >= 1 and k >= 1");
    if (k > 256) return fail(c, BMC_EINVAL, "k > 256 columns is not supported");
    if (dtype != BMC_F64 && dtype != BMC_F32) return fail(c, BMC_EINVAL, "dtype must be 0 or 1");
    if (layout != BMC_ROW_MAJOR && layout != BMC_COL_MAJOR)
        return fail(c, BMC_EINVAL, "layout must be 0 (row-major) or 1 (col-major)");
    if (layout == BMC_ROW_MAJOR ? ldx < k : ldx < n)
        return fail(c, BMC_EINVAL, "leading dimension too small");
    return BMC_OK;
}

int ensure_ticket(bmc_ctx* c) {
    if (c->ticket.p) return BMC_OK;
    int rc = ensure(c, c->ticket, bmc::RSS_TICKET_BYTES);
    if (rc) return rc;
    HIPCHK(c, hipMemsetAsync(c->ticket.p, 0, bmc::RSS_TICKET_BYTES, c->stream));  // the kernel keeps it zero
    return BMC_OK;
}

int rss_on_raw(bmc_ctx* c, const double* coef_host, int32_t nb, double* out_host) {
    const Panels P = panels_of(c, c->Xraw.p);
    int rc;
    if ((rc = ensure_ticket(c))) return rc;
    if ((rc = ensure(c, c->rssPartial, (size_t)rss_groups(P) * 8 * sizeof(double)))) return rc;
    // all coefficient vectors up in one copy, one launch per 8 of them back to back on the
    // stream (the kernel re-zeroes its ticket), all results down in one copy, one sync
    const int32_t nb8 = (nb + 7) / 8 * 8;
    if ((rc = ensure(c, c->rssOut, (size_t)nb8 * sizeof(double)))) return rc;
    if ((rc = ensure(c, c->coef, (size_t)nb8 * c->k * sizeof(double)))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->coef.p, coef_host, (size_t)nb * c->k * sizeof(double),
                             hipMemcpyHostToDevice, c->stream));
    for (int32_t b0 = 0; b0 < nb; b0 += 8) {
        const int32_t m = nb - b0 < 8 ? nb - b0 : 8;
        HIPCHK(c, launch_residual_rss(P, (const double*)c->coef.p + (size_t)b0 * c->k, m,
                                      (double*)c->rssPartial.p, (unsigned*)c->ticket.p,
                                      (double*)c->rssOut.p + b0, c->stream));
    }
    HIPCHK(c, hipMemcpyAsync(out_host, c->rssOut.p, (size_t)nb * sizeof(double), hipMemcpyDeviceToHost,
                             c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BMC_OK;
}

struct Geometry {
    int chains_per_launch, G, waves, ppg, mode, ppw, nslot;
    int one_wave;   // the chain runs in ONE wave (gibbs_wave_kernel)
};

constexpr int RES_AUTO = 0, RES_REG = 1, RES_STREAM = 3;  // 2 = LDS
// one-wave chains: register-resident FMAs per iteration (rows per lane x columns, padded) up to
// which one wave beats the workgroup form.  Same-box, us per iteration, wave / workgroup:
// 12 x 4 (N = 629, K = 3) 0.351 / 0.629, 16 x 4 0.394 / 0.636, 12 x 8 0.469 / 0.633,
// 2 x 32 0.521 / 0.545; 16 x 8 0.675 / 0.636, 8 x 16 0.657 / 0.562, 4 x 32 0.739 / 0.550
// (gpurun_out/r3_wave_ab3.log)
constexpr int ONE_WAVE_MAX_FMAS = 96;
// Chip shape from the device properties (MI355X in SPX mode: 256 CUs = 8 XCDs x 32; a
// partitioned device exposes fewer CUs, and the co-residency bound must follow it).
struct Chip {
    int groups_max;   // one resident workgroup per CU
    int xcds;         // slots: blocks b and b + xcds share an XCD (observed round-robin)
    int cu_per_xcd;
};
Chip chip_of(const bmc_ctx* c) {
    Chip ch;
    ch.groups_max = c->n_cu > 0 ? c->n_cu : 256;
    // bmc_tuning.cu_limit: fewer CUs can hold this context's persistent workgroups than the
    // device reports (CU-masked queue, a GPU shared with another process)
    const int cu_limit = c->tune.cu_limit > 0 ? c->tune.cu_limit : c->env_cu_limit;
    if (cu_limit > 0 && cu_limit < ch.groups_max) ch.groups_max = cu_limit;
    if (ch.groups_max > 256) ch.groups_max = 256;   // the gather holds 2 x 256 granules
    ch.xcds = ch.groups_max >= 64 ? ch.groups_max / 32 : 1;
    ch.cu_per_xcd = ch.groups_max / ch.xcds;
    return ch;
}

// Pick the launch geometry.  Preference order: row panels in VGPRs with each chain on
// one XCD (8 slots x <= 32 groups), then panels pinned in LDS, then streaming.
Geometry choose_geometry(const bmc_ctx* c, int n_chains, bool allow_one_wave = false,
                         int max_waves = 1) {
    const Chip chip = chip_of(c);
    const int MAX_GROUPS_PER_LAUNCH = chip.groups_max, XCD_COUNT = chip.xcds,
              CU_PER_XCD = chip.cu_per_xcd;
    const int RP = 64 * c->vec;
    const size_t es = c->f32 ? 4 : 8;
    const size_t panel_bytes = (size_t)(c->k + 1) * RP * es;
    // u slices for up to 8 chains per pass + partial sums + control words + alignment slack
    const size_t fixed = (size_t)((c->k + 63) / 64 * 64) * 8 * 8 + (512 + 8) * 8 + 64;
    const int NP = c->npanels;
    const bmc_tuning& tu = c->tune;
    auto lds_fits = [&](int G) {
        const int ppg = (NP + G - 1) / G;
        return fixed + (size_t)ppg * panel_bytes <= LDS_LIMIT;
    };
    Geometry g{};
    g.ppw = 1;
    // ---- register residency: G <= 32 groups of <= 8 waves, 1/2/4 panels per wave ----
    // ---- small problem: ONE workgroup holds the whole chain in registers -------------------
    // No inter-workgroup exchange, no co-residency requirement (measured 0.85 us/iteration at
    // N = 629 against 1.4 with ten single-wave groups), and every chain is an independent
    // workgroup, so hundreds of chains run side by side in one launch.
    // Smaller still (the reference's data set, 629 x 3): the chain in ONE wave, rows and columns
    // in its registers, no hand-over of any kind inside an iteration (gibbs_wave_kernel; measured
    // at N = 629, K = 3: see DESIGN.md 4.1).  waves_per_group = 1 asks for it, > 1 or an
    // explicit panels_per_wave keep the workgroup form.
    if (allow_one_wave && (tu.residency == RES_AUTO || tu.residency == RES_REG) && c->vec == 1 &&
        tu.groups_per_chain <= 1 && tu.waves_per_group <= 1 && tu.panels_per_wave <= 0) {
        // 1, 2, 4 (one per SIMD) or 8 waves: the fewest that keep a wave's FMAs per iteration
        // within the measured crossover
        int nw = 0, fmas = 0;
        for (int w : {1, 2, 4, 8}) {
            if (w > 1 && (w > max_waves || tu.waves_per_group == 1)) break;
            const int f = bmc::gibbs_wave_capacity(c->k, (int)((NP + w - 1) / w));
            if (w == 8 && f > 64) break;   // (8 waves: 256 registers each, shapes up to 64 FMAs)
            if (f > 0 && (f <= ONE_WAVE_MAX_FMAS || tu.waves_per_group == 1)) { nw = w; fmas = f; break; }
        }
        if (nw > 0 && fmas > 0) {
            g.mode = 0;
            g.ppw = (int)((NP + nw - 1) / nw);
            g.G = 1;
            g.waves = nw;
            g.ppg = (int)NP;
            g.chains_per_launch = n_chains < 2048 ? n_chains : 2048;
            g.nslot = g.chains_per_launch;
            g.one_wave = 1;
            return g;
        }
    }
    if ((tu.residency == RES_AUTO || tu.residency == RES_REG) && c->vec == 1 &&
        tu.groups_per_chain <= 1) {
        for (int want : {4, 8}) {
            for (int ppw : {1, 2, 4}) {
                if (tu.panels_per_wave > 0 && tu.panels_per_wave != ppw) continue;
                if (!gibbs_reg_capacity(c->k, c->f32, ppw)) continue;
                const int waves = (NP + ppw - 1) / ppw;
                if (waves > want || (tu.waves_per_group > 0 && waves > tu.waves_per_group)) continue;
                g.mode = 0;
                g.ppw = ppw;
                g.G = 1;
                g.waves = tu.waves_per_group > 0 ? tu.waves_per_group : waves;
                g.ppg = NP;
                g.chains_per_launch = n_chains < 2048 ? n_chains : 2048;
                g.nslot = g.chains_per_launch;
                return g;
            }
        }
    }
    if ((tu.residency == RES_AUTO || tu.residency == RES_REG) && c->vec <= 2) {
        // Prefer the fewest panels per wave that keep the chain on ONE XCD (32 groups x 8 waves):
        // its exchange is a single hop through that XCD's L2 (~0.4 us) where a chain spread over
        // the chip pays two levels (~1.3 us), which outweighs one or three more panels per wave
        // (~0.2 us each at K = 32); N = 30000, K = 32: 2.1 -> 1.5 us per iteration, and 8 chains
        // then run side by side, one per XCD.
        int first_ppw = 1;
        if (tu.panels_per_wave <= 0 && tu.groups_per_chain <= 0 && c->vec == 1)
            for (int ppw : {1, 2, 4})
                if (gibbs_reg_capacity(c->k, c->f32, ppw) && (int64_t)CU_PER_XCD * 8 * ppw >= NP) {
                    first_ppw = ppw;
                    break;
                }
        for (int ppw : {1, 2, 4}) {
            if (ppw < first_ppw) continue;
            if (tu.panels_per_wave > 0 && tu.panels_per_wave != ppw) continue;
            if (c->vec == 2 && ppw != 1) continue;
            if (!gibbs_reg_capacity(c->k, c->f32, ppw * c->vec)) continue;
            // one XCD (32 CUs) per chain while the panels fit there (measured: 32 groups x 5
            // waves beats 20 x 8 at C2); otherwise the whole chip serves one chain at a time
            int G = tu.groups_per_chain;
            if (G <= 0) {
                G = NP < CU_PER_XCD ? NP : CU_PER_XCD;
                if ((int64_t)G * 8 * ppw < NP) {
                    G = (int)((NP + 8 * ppw - 1) / (8 * ppw));
                    if (G > MAX_GROUPS_PER_LAUNCH) continue;
                }
            }
            if (G > MAX_GROUPS_PER_LAUNCH) continue;
            // a chain over several XCDs: whole teams (groups g mod 8), so that the kernel can
            // put team j on XCD j whichever XCD the launch starts on
            if (tu.groups_per_chain <= 0 && G > CU_PER_XCD && XCD_COUNT == 8 &&
                ((G + 7) & ~7) <= MAX_GROUPS_PER_LAUNCH)
                G = (G + 7) & ~7;
            const int ppg_reg = (NP + G - 1) / G;
            int waves = tu.waves_per_group > 0 ? tu.waves_per_group : (ppg_reg + ppw - 1) / ppw;
            if (waves > 8 || (int64_t)G * waves * ppw < NP) continue;
            g.mode = 0;
            g.ppw = ppw;
            g.G = G;
            g.waves = waves;
            if (G <= CU_PER_XCD) {
                // one chain per XCD (run_common widens this when more chains fit side by side)
                g.nslot = XCD_COUNT;
                g.chains_per_launch = n_chains < XCD_COUNT ? n_chains : XCD_COUNT;
            } else {
                g.chains_per_launch = MAX_GROUPS_PER_LAUNCH / G;
                if (g.chains_per_launch > n_chains) g.chains_per_launch = n_chains;
                g.nslot = g.chains_per_launch;
            }
            g.ppg = (NP + G - 1) / G;
            return g;
        }
    }
    // ---- LDS residency or streaming ------------------------------------------------------
    const int t_waves = tu.waves_per_group;
    int cpl = n_chains < XCD_COUNT ? n_chains : XCD_COUNT;
    int G;
    if (tu.groups_per_chain > 0) {
        G = tu.groups_per_chain;
        if (G > MAX_GROUPS_PER_LAUNCH) G = MAX_GROUPS_PER_LAUNCH;
        if (cpl > MAX_GROUPS_PER_LAUNCH / G) cpl = MAX_GROUPS_PER_LAUNCH / G;
        if (cpl < 1) cpl = 1;
    } else {
        // fewer chains per launch until the panels fit in LDS (or one chain is left)
        while (cpl > 1 && !lds_fits(MAX_GROUPS_PER_LAUNCH / cpl)) --cpl;
        const int gmax = MAX_GROUPS_PER_LAUNCH / cpl;
        const int want_waves = t_waves > 0 ? t_waves : 4;
        G = (NP + want_waves - 1) / want_waves;
        if (G > gmax) G = gmax;
        if (G < 1) G = 1;
        if (!lds_fits(G) && lds_fits(gmax))
            while (!lds_fits(G)) ++G;
        if (G > CU_PER_XCD && XCD_COUNT == 8 && ((G + 7) & ~7) <= gmax) G = (G + 7) & ~7;   // whole teams
    }
    if (G > NP) G = NP;
    g.G = G;
    g.chains_per_launch = cpl;
    g.ppg = (NP + G - 1) / G;
    const bool want_stream = tu.residency == RES_STREAM;
    g.mode = (!want_stream && lds_fits(G)) ? 1 : 2;
    int waves = t_waves > 0 ? t_waves : (g.ppg < 4 ? g.ppg : (g.mode == 1 ? (g.ppg < 8 ? g.ppg : 8) : 8));
    if (waves > 8) waves = 8;   // 512-thread workgroups: 256 VGPRs per lane, no spills
    if (waves < 1) waves = 1;
    g.waves = waves;
    // one slot per XCD while a chain's groups fit one XCD's CUs; otherwise any placement
    g.nslot = (G <= CU_PER_XCD) ? XCD_COUNT : cpl;
    return g;
}

// rss_mode 1: upload G, u0, g0 and take rss(u0) from ONE residual pass over the rotated panels
int gram_device_setup(bmc_ctx* c) {
    if (c->have_gram_dev) return BMC_OK;
    const int k = c->k;
    if (k > 64 || c->Gt.empty())
        return fail(c, BMC_EINVAL, "rss_mode 1 (sufficient statistics) supports at most 64 columns");
    int rc;
    if ((rc = ensure(c, c->dGt, (size_t)k * k * 8)) || (rc = ensure(c, c->dU0, (size_t)k * 8)) ||
        (rc = ensure(c, c->dG0, (size_t)k * 8)))
        return rc;
    HIPCHK(c, hipMemcpyAsync(c->dGt.p, c->Gt.data(), (size_t)k * k * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->dU0.p, c->u0.data(), (size_t)k * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->dG0.p, c->g0.data(), (size_t)k * 8, hipMemcpyHostToDevice, c->stream));
    const Panels P = panels_of(c, c->Xrot.p);
    if ((rc = ensure_ticket(c))) return rc;
    if ((rc = ensure(c, c->rssPartial, (size_t)rss_groups(P) * 8 * sizeof(double)))) return rc;
    if ((rc = ensure(c, c->rssOut, 8 * sizeof(double)))) return rc;
    HIPCHK(c, launch_residual_rss(P, (const double*)c->dU0.p, 1, (double*)c->rssPartial.p,
                                  (unsigned*)c->ticket.p, (double*)c->rssOut.p, c->stream));
    HIPCHK(c, hipMemcpyAsync(&c->rss0, c->rssOut.p, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_gram_dev = true;
    return BMC_OK;
}

// The workgroups of a persistent launch wait for each other inside the kernel, so all of them
// must be resident at once.  A plain launch checks nothing (an over-subscribed grid would spin
// until the bounded spins expire, 4 s): ask the runtime how many workgroups of exactly this
// kernel, block size and LDS footprint one CU admits and compare with what the launch keeps
// resident.  `resident` = workgroups that stay in the loop (unused slots leave at once).
template <typename Args, typename Launch>
int check_residency(bmc_ctx* c, Args a, int resident, Launch launch, const char* what) {
    int32_t per_cu = 0;
    a.query_occupancy = &per_cu;
    const hipError_t e = launch(a, c->stream);
    if (e != hipSuccess)
        return fail(c, e == hipErrorInvalidValue ? BMC_EINVAL : BMC_EHIP,
                    std::string(what) + ": no kernel for this geometry (" + hipGetErrorString(e) + ")");
    const long cap = (long)per_cu * chip_of(c).groups_max;
    if ((long)resident > cap)
        return fail(c, BMC_EINVAL,
                    std::string(what) + ": the launch needs " + std::to_string(resident) +
                        " co-resident workgroups but the device admits " + std::to_string(cap) + " (" +
                        std::to_string(per_cu) + " per CU x " + std::to_string(chip_of(c).groups_max) +
                        " CUs); use fewer groups_per_chain / waves_per_group or another residency");
    return BMC_OK;
}

// an explicit geometry request that the (possibly limited) chip cannot hold is an error, not
// something to clamp silently
int check_tuning_fits(bmc_ctx* c) {
    const Chip chip = chip_of(c);
    if (c->tune.groups_per_chain > chip.groups_max)
        return fail(c, BMC_EINVAL,
                    "groups_per_chain = " + std::to_string(c->tune.groups_per_chain) + " exceeds the " +
                        std::to_string(chip.groups_max) +
                        " workgroups that can be resident at once (one per CU; bmc_tuning.cu_limit)");
    return BMC_OK;
}

int run_common(bmc_ctx* c, int32_t n_chains, int64_t iters, const uint64_t* seeds, int rng_mode,
               const double* xi, const double* g, double* samples_host, void* samples_dev,
               bmc_stats* stats) {
    if (!c) return BMC_EINVAL;
    if (!c->have_problem || !c->have_prior)
        return fail(c, BMC_ESTATE, "bmc_set_problem and bmc_set_prior must be called first");
    if (n_chains < 1 || iters < 0) return fail(c, BMC_EINVAL, "need n_chains >= 1, iters >= 0");
    if (iters >= 0xffffffffll) return fail(c, BMC_EINVAL, "iters must be < 2^32 - 1");
    if (rng_mode == BMC_RNG_DEVICE) {
        if (!seeds) return fail(c, BMC_EINVAL, "seeds required in device RNG mode");
        if (xi || g) return fail(c, BMC_EINVAL, "xi/g must be NULL in device RNG mode");
    } else if (rng_mode == BMC_RNG_REPLAY) {
        if (!xi || !g) return fail(c, BMC_EINVAL, "xi and g required in replay mode");
    } else {
        return fail(c, BMC_EINVAL, "rng_mode must be 0 or 1");
    }
    const int K = c->k;
    const size_t T = (size_t)iters, C = (size_t)n_chains;
    int rc;
    if ((rc = ensure(c, c->xi, C * T * K * 8))) return rc;
    if ((rc = ensure(c, c->gam, C * T * 8))) return rc;
    if ((rc = ensure(c, c->uout, C * T * (K + 1) * 8))) return rc;
    double* d_samples = (double*)samples_dev;
    if (!d_samples) {
        if ((rc = ensure(c, c->samples, C * T * (K + 1) * 8))) return rc;
        d_samples = (double*)c->samples.p;
    }
    if ((rc = check_tuning_fits(c))) return rc;
    Geometry geo = choose_geometry(c, n_chains, true, 8);
    // One-XCD register residency with more than 8 chains.
    // (a) 16 chains or more: the register-resident panels of an XCD's 32 groups serve a BUNDLE of
    //     2 / 4 / 8 chains per pass (gibbs_multi_kernel, one bundle per XCD: 16 .. 64 chains in one
    //     launch); every chain bit-identical to its solo run.
    // (b) 9 .. 15 chains left: chains c and c + 8 share XCD c % 8, two workgroups per CU side by
    //     side.  That needs 4 waves per SIMD (two 5-wave groups must fit whatever SIMDs their
    //     waves land on), i.e. the kernel variant held to 128 VGPRs, which exists for light
    //     shapes only.  (Three or four per XCD are not used: measured, the launch then stalls.)
    const int xcds = chip_of(c).xcds;
    const bool one_xcd_reg = geo.mode == 0 && geo.G > 1 && geo.nslot == xcds && xcds > 1 &&
                             geo.G <= chip_of(c).cu_per_xcd && c->tune.groups_per_chain <= 0;
    const bool xcd_bundles = one_xcd_reg && geo.ppw == 1 && c->vec == 1 && geo.G <= 32 &&
                             c->tune.chains_per_pass != 1 &&
                             bmc::gibbs_reg_multi_cap(c->k, c->f32 != 0, c->vec) >= 2;
    int pack_ok = 0;
    if (one_xcd_reg && n_chains > geo.nslot && 2 * geo.waves <= 16) {
        int32_t regs = 0;
        GibbsArgs q{};
        q.P = panels_of(c, c->Xrot.p);
        q.G = geo.G; q.waves = geo.waves; q.mode = geo.mode; q.reg_ppw = geo.ppw;
        q.nslot = geo.nslot; q.n_chains = 1; q.chains_per_pass = 1; q.panels_per_group = geo.ppg;
        q.query_regs = &regs;
        int32_t per_cu = 0;
        GibbsArgs qo = q;
        qo.query_regs = nullptr;
        qo.pack = 1;
        qo.query_occupancy = &per_cu;
        if (launch_gibbs(q, c->stream) == hipSuccess && regs > 0 && regs <= 128 &&
            launch_gibbs(qo, c->stream) == hipSuccess && per_cu >= 2)
            pack_ok = 1;
    }
    // the most chains one launch can hold (sizes the exchange words)
    int max_per_launch = geo.chains_per_launch;
    if (pack_ok) max_per_launch = 2 * geo.nslot;
    if (xcd_bundles) max_per_launch = 8 * xcds;
    if (max_per_launch < 8) max_per_launch = 8;
    const int gran_stride = bmc::gran_slot_words(geo.G);
    if ((rc = ensure(c, c->gran, (size_t)max_per_launch * 3 * gran_stride * 8))) return rc;
    if ((rc = ensure(c, c->status, C * sizeof(int32_t)))) return rc;
    if ((rc = ensure(c, c->placement, C * sizeof(int32_t)))) return rc;
    HIPCHK(c, hipMemsetAsync(c->placement.p, 0, C * sizeof(int32_t), c->stream));
    if ((rc = ensure(c, c->seeds, C * sizeof(uint64_t)))) return rc;

    HIPCHK(c, hipMemsetAsync(c->status.p, 0, C * sizeof(int32_t), c->stream));
    HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    if (iters > 0) {
        if (rng_mode == BMC_RNG_DEVICE) {
            HIPCHK(c, hipMemcpyAsync(c->seeds.p, seeds, C * sizeof(uint64_t),
                                     hipMemcpyHostToDevice, c->stream));
            const double shape = (c->nu0 + (double)c->n) / 2.0;  // inference_utils.py:50
            HIPCHK(c, launch_rng_fill((const uint64_t*)c->seeds.p, n_chains, (int64_t)T * K,
                                      (double*)c->xi.p, shape, (int64_t)T, (double*)c->gam.p,
                                      c->stream));
        } else {
            HIPCHK(c, hipMemcpyAsync(c->xi.p, xi, C * T * K * 8, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(c->gam.p, g, C * T * 8, hipMemcpyHostToDevice, c->stream));
        }
    }
    HIPCHK(c, hipEventRecord(c->ev[1], c->stream));

    GibbsArgs a;
    a.P = panels_of(c, c->Xrot.p);
    a.lam = (const double*)c->dLam.p;
    a.c1 = (const double*)c->dC1.p;
    a.c2 = (const double*)c->dC2.p;
    a.nu0_s20 = c->nu0 * c->s20;
    a.sigma2_init = c->sigma2_init;
    a.gran = (unsigned long long*)c->gran.p;
    a.gran_stride = gran_stride;
    a.iters = iters;
    a.G = geo.G;
    a.waves = geo.waves;
    a.mode = geo.mode;
    a.reg_ppw = geo.ppw;
    a.nslot = geo.nslot;
    a.force_agent_scope = c->tune.force_agent_scope;
    a.panels_per_group = geo.ppg;
    a.one_wave = geo.one_wave;
    if (geo.mode == 2) {
        // A matrix larger than the 256 MiB Infinity Cache, swept once per iteration, would evict
        // itself before it is read again.  Each group then reads its first panels normally, about
        // 190 MB in all, which stay cached from one iteration to the next, and the rest with
        // non-temporal loads that do not displace them (410 MB: 72 -> 62.6 us per iteration).
        const double total = (double)c->npanels * (double)(K + 1) * 64.0 * c->vec * (c->f32 ? 4 : 8);
        const double budget = 190e6;
        if (total > budget) a.P.stream_keep = (int32_t)((double)geo.ppg * budget / total);
    }
    a.dbg = nullptr;
    a.query_regs = nullptr;
    a.pack = 0;
#ifdef BMC_STAMPS
    if ((rc = ensure(c, c->dbg, 12 * sizeof(long long)))) return rc;
    HIPCHK(c, hipMemsetAsync(c->dbg.p, 0, 12 * sizeof(long long), c->stream));
    a.dbg = (long long*)c->dbg.p;
#endif
    // chains per pass: when the panels are NOT register-resident one read of X can serve up to
    // 8 chains (one leader wave per chain); 0 = automatic, 1 = off
    int cpp_max = 1, waves_multi = geo.waves;
    // (register residency: only the whole-chip form, one chain bundle per launch, one panel per
    // wave; the one-XCD-per-chain form already runs 8 chains side by side)
    const bool reg_multi_ok = geo.mode == 0 && geo.nslot < 8 && geo.G > 1 && geo.ppw == 1;
    if ((geo.mode != 0 || reg_multi_ok) && n_chains > 1 && c->tune.chains_per_pass != 1) {
        // every chain of a pass needs a leader wave: widen the workgroup if the panels alone
        // would ask for fewer waves (the extra waves own no panel, they only lead a chain)
        int want = n_chains >= 8 ? 8 : n_chains >= 4 ? 4 : 2;
        if (c->tune.chains_per_pass > 1 && c->tune.chains_per_pass < want)
            want = c->tune.chains_per_pass >= 4 ? 4 : 2;
        if (a.waves < want && c->tune.waves_per_group <= 0) a.waves = want;
        waves_multi = a.waves;
        cpp_max = a.waves >= 8 ? 8 : a.waves >= 4 ? 4 : a.waves >= 2 ? 2 : 1;
        if (cpp_max > want) cpp_max = want;
        if (reg_multi_ok) {
            const int cap = bmc::gibbs_reg_multi_cap(K, a.P.f32 != 0, a.P.vec);
            if (cpp_max > cap) cpp_max = cap < 2 ? 1 : cap;
        }
    }
    const int waves_single = geo.waves;
    int launches = 0, cpp_used = 1, waves_used = 0;
    const bool gram_mode = c->tune.rss_mode == 1;
    if (gram_mode && iters > 0) {
        if ((rc = gram_device_setup(c))) return rc;
        GramArgs ga;
        ga.k = K;
        ga.lam = a.lam; ga.c1 = a.c1; ga.c2 = a.c2;
        ga.Gt = (const double*)c->dGt.p; ga.u0 = (const double*)c->dU0.p; ga.g0 = (const double*)c->dG0.p;
        ga.rss0 = c->rss0;
        ga.nu0_s20 = a.nu0_s20; ga.sigma2_init = a.sigma2_init;
        ga.xi = (const double*)c->xi.p; ga.gam = (const double*)c->gam.p; ga.uout = (double*)c->uout.p;
        ga.iters = iters;
        ga.n_chains = n_chains;
        HIPCHK(c, launch_gibbs_gram(ga, c->stream));
        launches = 1;
    }
    int64_t passes = 0;
    for (int c0 = 0; !gram_mode && iters > 0 && c0 < n_chains;) {
        const int left = n_chains - c0;
        int cpp = 1, m, resident;
        a.bundle_slots = 0;
        a.bundle_bal = 0;
        a.pack = 0;
        a.nslot = geo.nslot;
        // bundles pay from 4 chains per XCD on (measured at C2, us per iteration for all chains:
        // 16 chains 1.37 as bundles of 2 against 1.05 packed two per XCD; 32 chains 1.81 as
        // bundles of 4 against 2 x 1.05; 64 chains 2.33 as bundles of 8); with fewer they are
        // used when asked for (chains_per_pass = 2) or when the shape has no packed variant
        if (xcd_bundles && left >= 2 * xcds &&
            (left >= 4 * xcds || !pack_ok || c->tune.chains_per_pass > 1)) {
            // one bundle per XCD: as many chains per bundle as keep all XCDs busy
            int cap = bmc::gibbs_reg_multi_cap(K, a.P.f32 != 0, a.P.vec);
            if (c->tune.chains_per_pass > 1 && c->tune.chains_per_pass < cap) cap = c->tune.chains_per_pass;
            cpp = 2;
            while (cpp * 2 <= cap && cpp * 2 * xcds <= left) cpp *= 2;
            // 40 .. 63 chains: bundles of 8 on 5 .. 7 XCDs in one launch (2.3 us per iteration at C2)
            // rather than bundles of 4 on all 8 (1.8 us for 32 of them) plus a second launch
            if (cap >= 8 && cpp == 4 && left >= 5 * 8) cpp = 8;
            const int bundles = left / cpp < xcds ? left / cpp : xcds;
            m = bundles * cpp;
            a.bundle_slots = xcds;
            a.waves = waves_single > cpp ? waves_single : cpp;   // a leader wave per chain
            // bundles of 8 on 8 waves, at most 5 panels per group, two panels of K columns in a
            // wave's registers: the balanced layout (4 chains of panel w % 4 + 1 chain of the
            // fifth panel per wave instead of 8 chains of one panel on waves 0 .. 3)
#ifndef BMC_NO_BAL
            // (panels_per_wave = 1 asked for explicitly keeps the one-panel layout: the A/B knob)
            a.bundle_bal = (cpp == 8 && a.waves == 8 && geo.ppg <= 5 && K > 8 &&
                            c->tune.panels_per_wave != 1 &&
                            bmc::gibbs_reg_capacity(K, a.P.f32 != 0, 2)) ? 1 : 0;
#endif
            resident = bundles * a.G;
            passes += (int64_t)bundles * iters;
        } else if (cpp_max > 1 && left >= 2) {
            while (cpp * 2 <= left && cpp * 2 <= cpp_max) cpp *= 2;
            m = cpp;
            a.waves = waves_multi;
            resident = a.G;
            passes += iters;
        } else {
            a.waves = waves_single;
            if (pack_ok && left > geo.nslot) {
                a.pack = 1;
                a.nslot = 2 * geo.nslot;
            }
            const int cap = a.pack ? a.nslot : geo.chains_per_launch;
            m = left < cap ? left : cap;
            resident = m * a.G;
            passes += (int64_t)m * iters;
        }
        a.n_chains = m;
        a.chains_per_pass = cpp;
        a.epoch0 = launch_nonce(c, (uint64_t)iters);
        if (cpp > cpp_used) cpp_used = cpp;
        if (a.waves > waves_used) waves_used = a.waves;
        a.xi = (const double*)c->xi.p + (size_t)c0 * T * K;
        a.gam = (const double*)c->gam.p + (size_t)c0 * T;
        a.uout = (double*)c->uout.p + (size_t)c0 * T * (K + 1);
        a.status = (int32_t*)c->status.p + c0;
        a.placement = (int32_t*)c->placement.p + c0;
        HIPCHK(c, hipMemsetAsync(c->gran.p, 0, (size_t)m * 3 * gran_stride * 8, c->stream));
        if (gibbs_lds_bytes(a) > LDS_LIMIT) return fail(c, BMC_EINVAL, "LDS plan exceeds 160 KiB");
        if (a.G > 1 || cpp > 1)   // (a single-workgroup chain waits for nobody)
            if ((rc = check_residency(c, a, resident, launch_gibbs, "persistent Gibbs kernel")))
                return rc;
        HIPCHK(c, launch_gibbs(a, c->stream));
        ++launches;
        c0 += m;
    }
    HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
    if (iters > 0)
        HIPCHK(c, launch_unrotate((const double*)c->uout.p, (const double*)c->dWT.p, K,
                                  (int64_t)(C * T), d_samples, c->stream));
    HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
    std::vector<int32_t> st(C, 0), place(C, 0);
    HIPCHK(c, hipMemcpyAsync(st.data(), c->status.p, C * sizeof(int32_t), hipMemcpyDeviceToHost,
                             c->stream));
    HIPCHK(c, hipMemcpyAsync(place.data(), c->placement.p, C * sizeof(int32_t),
                             hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (samples_host && iters > 0)
        if ((rc = copy_to_host(c, samples_host, d_samples, C * T * (K + 1) * 8, C * T * (K + 1) * 8, 1)))
            return rc;
    if (stats) {
        float ms = 0;
        std::memset(stats, 0, sizeof(*stats));
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1])); stats->rng_ms = ms;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev[1], c->ev[2])); stats->loop_ms = ms;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev[2], c->ev[3])); stats->post_ms = ms;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[3])); stats->total_ms = ms;
        stats->iterations = iters;
        stats->n_chains = n_chains;
        stats->launches = launches;
        stats->groups_per_chain = geo.G;
        stats->waves_per_group = waves_used ? waves_used : geo.waves;   // widened for leader waves
        stats->chains_per_pass = cpp_used;
        stats->residency = gram_mode ? 4 : geo.mode + 1;
        stats->xcd_local_chains = 0;
        for (size_t i = 0; i < C; ++i) stats->xcd_local_chains += place[i] ? 1 : 0;
        stats->bytes_per_pass = ((int64_t)c->n * K + c->n) * (c->f32 ? 4 : 8);
        // a pass that serves several chains counts once
        stats->passes = gram_mode ? 0 : passes;
        if (gram_mode) { stats->groups_per_chain = 1; stats->waves_per_group = 1; }
    }
    for (size_t i = 0; i < C; ++i)
        if (st[i] != 0)
            return fail(c, BMC_ETIMEOUT, "persistent Gibbs kernel: bounded spin expired (chain " +
                                             std::to_string(i) + ")");
    return BMC_OK;
}

}  // namespace

extern "C" {

int bmc_abi_version(void) { return PYBMC_AMD_ABI_VERSION; }

int bmc_create(int device_id, bmc_ctx** out) {
    if (!out) return BMC_EINVAL;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return BMC_EHIP;
    if (device_id < 0 || device_id >= count) return BMC_EINVAL;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return BMC_EHIP;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return BMC_EHIP;  // MI355X only
    bmc_ctx* c = new (std::nothrow) bmc_ctx();
    if (!c) return BMC_ENOMEM;
    c->device = device_id;
    c->n_cu = prop.multiProcessorCount;
    c->nonce_state ^= (uint64_t)(uintptr_t)c * 0xD6E8FEB86659FD93ull;   // contexts differ
    // Several processes on one GPU cannot see each other's persistent launches: each is told its
    // share once, in the environment (e.g. 2 ranks per GPU: PYBMC_AMD_CU_LIMIT=128), and every
    // context it creates plans for that many CUs unless bmc_tuning.cu_limit says otherwise.
    if (const char* e = std::getenv("PYBMC_AMD_CU_LIMIT")) {
        const long v = std::strtol(e, nullptr, 10);
        if (v > 0 && v < 100000) c->env_cu_limit = (int)v;
    }
    if (hipSetDevice(device_id) != hipSuccess ||
        hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return BMC_EHIP;
    }
    c->own_stream = true;
    for (auto& e : c->ev)
        if (hipEventCreate(&e) != hipSuccess) {
            bmc_destroy(c);
            return BMC_EHIP;
        }
    *out = c;
    return BMC_OK;
}

void bmc_destroy(bmc_ctx* c) {
    if (!c) return;
    (void)bmc_comm_destroy(c);
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (DevBuf* b : {&c->Xraw, &c->Yp, &c->Xrot, &c->dW, &c->dWT, &c->dLam, &c->dC1, &c->dC2,
                      &c->gramScratch, &c->gramOut, &c->rssPartial, &c->rssOut, &c->coef, &c->ticket,
                      &c->stage, &c->xi, &c->gam, &c->uout, &c->samples, &c->gran, &c->status,
                      &c->seeds, &c->dbg, &c->placement, &c->pPreds, &c->pTheta, &c->pVt,
                      &c->pPad, &c->pWt, &c->pSig, &c->pR, &c->pRT, &c->pNoise, &c->pAux, &c->pBands, &c->sVt,
                      &c->sStep, &c->sUnif, &c->sOut, &c->sCnt, &c->oFc, &c->oMu, &c->oW, &c->oOut})
        release(*b);
    for (auto& e : c->ev)
        if (e) (void)hipEventDestroy(e);
    for (int i = 0; i < 2; ++i) {
        if (c->hev[i]) (void)hipEventDestroy(c->hev[i]);
        if (c->hstage[i]) (void)hipHostFree(c->hstage[i]);
    }
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* bmc_last_error(const bmc_ctx* c) { return c ? c->err.c_str() : "null context"; }

int bmc_set_stream(bmc_ctx* c, void* hip_stream) {
    if (!c) return BMC_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    if (hip_stream) {
        c->stream = (hipStream_t)hip_stream;
        c->own_stream = false;
    } else {
        HIPCHK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    return BMC_OK;
}

int bmc_set_tuning(bmc_ctx* c, const bmc_tuning* t) {
    if (!c) return BMC_EINVAL;
    if (!t) {
        c->tune = bmc_tuning{};
        return BMC_OK;
    }
    if (t->groups_per_chain < 0 || t->groups_per_chain > 256 || t->waves_per_group < 0 ||
        t->waves_per_group > 8 || t->residency < 0 || t->residency > 3 ||
        (t->panels_per_wave != 0 && t->panels_per_wave != 1 && t->panels_per_wave != 2 &&
         t->panels_per_wave != 4) ||
        (t->chains_per_pass != 0 && t->chains_per_pass != 1 && t->chains_per_pass != 2 &&
         t->chains_per_pass != 4 && t->chains_per_pass != 8) ||
        (t->rss_mode != 0 && t->rss_mode != 1) || t->cu_limit < 0)
        return fail(c, BMC_EINVAL, "tuning out of range (groups 0..256, waves 0..8, residency 0..3, "
                                   "panels_per_wave 0/1/2/4, chains_per_pass 0/1/2/4/8, rss_mode 0/1, "
                                   "cu_limit >= 0)");
    c->tune = *t;
    return BMC_OK;
}

int bmc_set_problem(bmc_ctx* c, const void* X, int64_t n, int32_t k, int64_t ldx, int layout,
                    const void* y, int dtype) {
    int rc = check_problem_args(c, X, n, k, ldx, layout, y, dtype);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t es = dtype == BMC_F32 ? 4 : 8;
    const size_t xbytes = (size_t)(layout == BMC_COL_MAJOR ? (size_t)ldx * k : (size_t)ldx * n) * es;
    const size_t ybytes = (size_t)n * es;
    const size_t yoff = (xbytes + 255) & ~(size_t)255;
    if ((rc = ensure(c, c->stage, yoff + ybytes))) return rc;
    // the last column/row of a strided host matrix may be shorter than ldx
    const size_t xcopy = layout == BMC_COL_MAJOR ? ((size_t)ldx * (k - 1) + n) * es
                                                 : ((size_t)ldx * (n - 1) + k) * es;
    HIPCHK(c, hipMemcpyAsync(c->stage.p, X, xcopy, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync((char*)c->stage.p + yoff, y, ybytes, hipMemcpyHostToDevice, c->stream));
    rc = set_problem_common(c, c->stage.p, (char*)c->stage.p + yoff, n, k, ldx, layout, dtype);
    return rc;
}

int bmc_set_problem_device(bmc_ctx* c, const void* dX, int64_t n, int32_t k, int64_t ldx,
                           int layout, const void* dy, int dtype) {
    int rc = check_problem_args(c, dX, n, k, ldx, layout, dy, dtype);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    return set_problem_common(c, dX, dy, n, k, ldx, layout, dtype);
}

int bmc_set_prior(bmc_ctx* c, const double* b0, const double* C0, double nu0, double sigma20) {
    if (!c) return BMC_EINVAL;
    if (!c->have_problem) return fail(c, BMC_ESTATE, "bmc_set_problem must be called first");
    if (!b0 || !C0) return fail(c, BMC_EINVAL, "b0 and C0 must not be NULL");
    HIPCHK(c, hipSetDevice(c->device));
    c->have_prior = false;
    const int k = c->k, ka = k + 1;
    // P = inv(C0)                                          (inference_utils.py:22)
    bmc_la::Mat P(C0, C0 + (size_t)k * k);
    if (!bmc_la::invert(P, k)) return fail(c, BMC_ESINGULAR, "Singular matrix (b_mean_cov)");
    // OLS start value                                      (inference_utils.py:26-37)
    bmc_la::Mat A((size_t)k * k);
    std::vector<double> xty(k);
    for (int i = 0; i < k; ++i) {
        for (int j = 0; j < k; ++j) A[(size_t)i * k + j] = c->gram[(size_t)i * ka + j];
        xty[i] = c->gram[(size_t)i * ka + k];
    }
    std::vector<double> bols;
    if (!bmc_la::solve(A, xty, k, bols)) return fail(c, BMC_ESINGULAR, "Singular matrix (X'X)");
    double rss0 = 0.0;
    int rc = rss_on_raw(c, bols.data(), 1, &rss0);
    if (rc) return rc;
    double s2 = rss0 / (double)c->n;
    if (!(s2 >= 1e-6)) s2 = s2 != s2 ? s2 : 1e-6;  // max(s2, 1e-6); NaN propagates
    c->sigma2_init = s2;
    // basis: B = P + 1e-6 I = L L',  L^-1 A L^-T = Q diag(lam) Q',  W = L^-T Q
    bmc_la::Mat B((size_t)k * k);
    for (int i = 0; i < k; ++i)
        for (int j = 0; j < k; ++j)
            B[(size_t)i * k + j] = 0.5 * (P[(size_t)i * k + j] + P[(size_t)j * k + i]) +
                                   (i == j ? 1e-6 : 0.0);
    bmc_la::Mat L, Li;
    if (!bmc_la::cholesky(B, k, L))
        return fail(c, BMC_EINVAL, "prior precision inv(b_mean_cov) + 1e-6 I is not positive definite");
    bmc_la::lower_inverse(L, k, Li);
    bmc_la::Mat tmp((size_t)k * k, 0.0), M((size_t)k * k, 0.0);
    for (int i = 0; i < k; ++i)          // tmp = Li * A
        for (int m = 0; m <= i; ++m) {
            const double l = Li[(size_t)i * k + m];
            if (l == 0.0) continue;
            for (int j = 0; j < k; ++j) tmp[(size_t)i * k + j] += l * A[(size_t)m * k + j];
        }
    for (int i = 0; i < k; ++i)          // M = tmp * Li'
        for (int j = 0; j < k; ++j) {
            long double s = 0.0L;
            for (int m = 0; m <= j; ++m) s += (long double)tmp[(size_t)i * k + m] * Li[(size_t)j * k + m];
            M[(size_t)i * k + j] = (double)s;
        }
    for (int i = 0; i < k; ++i)
        for (int j = i + 1; j < k; ++j) {
            const double v = 0.5 * (M[(size_t)i * k + j] + M[(size_t)j * k + i]);
            M[(size_t)i * k + j] = M[(size_t)j * k + i] = v;
        }
    bmc_la::Mat Q;
    bmc_la::sym_eigh(M, k, c->lam, Q);
    c->W.assign((size_t)k * k, 0.0);
    for (int i = 0; i < k; ++i)          // W = Li' Q
        for (int j = 0; j < k; ++j) {
            long double s = 0.0L;
            for (int m = i; m < k; ++m) s += (long double)Li[(size_t)m * k + i] * Q[(size_t)m * k + j];
            c->W[(size_t)i * k + j] = (double)s;
        }
    std::vector<double> Pb0(k, 0.0);
    for (int i = 0; i < k; ++i) {
        long double s = 0.0L;
        for (int j = 0; j < k; ++j) s += (long double)P[(size_t)i * k + j] * b0[j];
        Pb0[i] = (double)s;
    }
    c->c1.assign(k, 0.0);
    c->c2.assign(k, 0.0);
    for (int j = 0; j < k; ++j) {
        long double s1 = 0.0L, s2l = 0.0L;
        for (int i = 0; i < k; ++i) {
            s1 += (long double)c->W[(size_t)i * k + j] * Pb0[i];
            s2l += (long double)c->W[(size_t)i * k + j] * xty[i];
        }
        c->c1[j] = (double)s1;
        c->c2[j] = (double)s2l;
    }
    c->b0.assign(b0, b0 + k);
    c->nu0 = nu0;
    c->s20 = sigma20;
    // rss_mode 1 (k <= 64): G = W'AW (= diag(lam) up to rounding), the least-squares point u0 in
    // the rotated basis and g0 = X~'(y - X~ u0) = c2 - G u0, all in extended precision
    c->Gt.clear();
    c->have_gram_dev = false;
    if (k <= 64) {
        std::vector<long double> AW((size_t)k * k, 0.0L);
        for (int i = 0; i < k; ++i)
            for (int m = 0; m < k; ++m) {
                const long double aim = A[(size_t)i * k + m];
                for (int j = 0; j < k; ++j) AW[(size_t)i * k + j] += aim * c->W[(size_t)m * k + j];
            }
        c->Gt.assign((size_t)k * k, 0.0);
        std::vector<long double> Gl((size_t)k * k, 0.0L);
        for (int i = 0; i < k; ++i)
            for (int j = 0; j < k; ++j) {
                long double sum = 0.0L;
                for (int m = 0; m < k; ++m) sum += (long double)c->W[(size_t)m * k + i] * AW[(size_t)m * k + j];
                Gl[(size_t)i * k + j] = sum;
            }
        for (int i = 0; i < k; ++i)
            for (int j = 0; j < k; ++j)
                c->Gt[(size_t)i * k + j] = (double)(0.5L * (Gl[(size_t)i * k + j] + Gl[(size_t)j * k + i]));
        double gmax = 0.0;
        for (int j = 0; j < k; ++j) gmax = std::max(gmax, c->Gt[(size_t)j * k + j]);
        c->u0.assign(k, 0.0);
        for (int j = 0; j < k; ++j) {
            const double gj = c->Gt[(size_t)j * k + j];
            c->u0[j] = gj > 1e-14 * gmax ? c->c2[j] / gj : 0.0;
        }
        c->g0.assign(k, 0.0);
        for (int i = 0; i < k; ++i) {
            long double sum = c->c2[i];
            for (int j = 0; j < k; ++j) sum -= (long double)c->Gt[(size_t)i * k + j] * c->u0[j];
            c->g0[i] = (double)sum;
        }
    }
    std::vector<double> WT((size_t)k * k);
    for (int i = 0; i < k; ++i)
        for (int j = 0; j < k; ++j) WT[(size_t)i * k + j] = c->W[(size_t)j * k + i];
    const size_t kk = (size_t)k * k * 8;
    if ((rc = ensure(c, c->dW, kk)) || (rc = ensure(c, c->dWT, kk)) ||
        (rc = ensure(c, c->dLam, (size_t)k * 8)) || (rc = ensure(c, c->dC1, (size_t)k * 8)) ||
        (rc = ensure(c, c->dC2, (size_t)k * 8)))
        return rc;
    HIPCHK(c, hipMemcpyAsync(c->dW.p, c->W.data(), kk, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->dWT.p, WT.data(), kk, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->dLam.p, c->lam.data(), (size_t)k * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->dC1.p, c->c1.data(), (size_t)k * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->dC2.p, c->c2.data(), (size_t)k * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, launch_rotate(panels_of(c, c->Xraw.p), (const double*)c->dW.p, c->k, c->Xrot.p,
                            c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // WT (stack vector) must outlive the copy
    c->have_prior = true;
    return BMC_OK;
}

int bmc_orthogonalize(bmc_ctx* c, const double* F, int64_t n, int32_t km, int64_t ldf,
                      const double* truth, int32_t k, double* mean_out, double* yc_out,
                      double* U_hat_out, double* S_out, double* Vt_out) {
    if (!c) return BMC_EINVAL;
    if (!F || !truth) return fail(c, BMC_EINVAL, "F and truth must not be NULL");
    if (n < 1 || km < 1 || k < 1 || k > km || ldf < km)
        return fail(c, BMC_EINVAL, "need n >= 1, 1 <= components_kept <= n_models, ldf >= n_models");
    if (km > 255) return fail(c, BMC_EINVAL, "more than 255 models is not supported");
    if (k > n) return fail(c, BMC_EINVAL, "components_kept exceeds the number of rows");
    HIPCHK(c, hipSetDevice(c->device));
    c->have_problem = c->have_prior = false;
    // panels of the centred matrix use the row-per-lane choice of the FINAL (n x k) problem
    const int vec = choose_vec(n, k, 0);
    const int RP = 64 * vec;
    const int32_t npanels = (int32_t)((n + RP - 1) / RP);
    int rc;
    const size_t fbytes = (size_t)((size_t)ldf * (n - 1) + km) * 8;
    const size_t toff = (fbytes + 255) & ~(size_t)255;
    if ((rc = ensure(c, c->stage, toff + (size_t)n * 8)) ||
        (rc = ensure(c, c->oFc, (size_t)npanels * km * RP * 8)) ||
        (rc = ensure(c, c->Yp, (size_t)npanels * RP * 8)) ||
        (rc = ensure(c, c->oMu, (size_t)npanels * RP * 8)))
        return rc;
    HIPCHK(c, hipMemcpyAsync(c->stage.p, F, fbytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync((char*)c->stage.p + toff, truth, (size_t)n * 8, hipMemcpyHostToDevice,
                             c->stream));
    HIPCHK(c, launch_centre((const double*)c->stage.p, n, km, ldf,
                            (const double*)((char*)c->stage.p + toff), vec, npanels,
                            (double*)c->oFc.p, (double*)c->Yp.p, (double*)c->oMu.p, c->stream));
    // Gram of the centred matrix: Fc'Fc = V S^2 V'  (the SVD of bmc.py:119 through its Gram)
    Panels P;
    P.X = c->oFc.p; P.y = c->Yp.p; P.n = n; P.k = km; P.vec = vec; P.npanels = npanels; P.f32 = 0;
    const size_t gsz = (size_t)(km + 1) * (km + 1);
    if ((rc = ensure(c, c->gramScratch, gram_scratch_bytes(P))) ||
        (rc = ensure(c, c->gramOut, gsz * 8)))
        return rc;
    HIPCHK(c, launch_gram(P, c->gramScratch.p, (double*)c->gramOut.p, c->stream));
    std::vector<double> ga(gsz);
    HIPCHK(c, hipMemcpyAsync(ga.data(), c->gramOut.p, gsz * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    bmc_la::Mat Gm((size_t)km * km), Q;
    for (int i = 0; i < km; ++i)
        for (int j = 0; j < km; ++j) Gm[(size_t)i * km + j] = ga[(size_t)i * (km + 1) + j];
    std::vector<double> ev;
    bmc_la::sym_eigh(Gm, km, ev, Q);
    std::vector<int> order(km);
    for (int i = 0; i < km; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return ev[a] > ev[b]; });
    const double s0 = std::sqrt(ev[order[0]] > 0 ? ev[order[0]] : 0.0);
    const double sk = std::sqrt(ev[order[k - 1]] > 0 ? ev[order[k - 1]] : 0.0);
    // rows of the centred matrix sum to zero -> rank <= n_models - 1 (bmc.py:114-119); the
    // Gram route also loses accuracy like (s0/sk)^2, so refuse ill-conditioned requests
    if (!(sk > s0 * 1e-6) || !(s0 > 0))
        return fail(c, BMC_ESINGULAR,
                    "components_kept reaches the (numerical) null space of the centred model matrix");
    std::vector<double> W((size_t)km * k), Vt((size_t)k * km), S(k);
    for (int q = 0; q < k; ++q) {
        const int col = order[q];
        S[q] = std::sqrt(ev[col]);
        // sign convention: the entry of largest magnitude of each right singular vector is > 0
        int big = 0;
        for (int i = 1; i < km; ++i)
            if (std::fabs(Q[(size_t)i * km + col]) > std::fabs(Q[(size_t)big * km + col])) big = i;
        const double sg = Q[(size_t)big * km + col] < 0 ? -1.0 : 1.0;
        for (int i = 0; i < km; ++i) {
            const double v = sg * Q[(size_t)i * km + col];
            Vt[(size_t)q * km + i] = v;
            W[(size_t)i * k + q] = v / S[q];          // U_hat = Fc V S^-1
        }
    }
    // the sampler's problem: X = U_hat (n x k panels), y = centred truth
    c->n = n;
    c->k = k;
    c->f32 = 0;
    c->vec = vec;
    c->npanels = npanels;
    if ((rc = ensure(c, c->Xraw, (size_t)npanels * k * RP * 8)) ||
        (rc = ensure(c, c->Xrot, (size_t)npanels * k * RP * 8)) ||
        (rc = ensure(c, c->oW, W.size() * 8)))
        return rc;
    HIPCHK(c, hipMemcpyAsync(c->oW.p, W.data(), W.size() * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, launch_rotate(P, (const double*)c->oW.p, k, c->Xraw.p, c->stream));
    if (U_hat_out) {
        if ((rc = ensure(c, c->oOut, (size_t)n * k * 8))) return rc;
        HIPCHK(c, launch_unpanelize((const double*)c->Xraw.p, n, k, vec, (double*)c->oOut.p, c->stream));
        HIPCHK(c, hipMemcpyAsync(U_hat_out, c->oOut.p, (size_t)n * k * 8, hipMemcpyDeviceToHost,
                                 c->stream));
    }
    if (mean_out)
        HIPCHK(c, hipMemcpyAsync(mean_out, c->oMu.p, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    if (yc_out)
        HIPCHK(c, hipMemcpyAsync(yc_out, c->Yp.p, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // W (host vector) must outlive its copy
    if (S_out) std::memcpy(S_out, S.data(), (size_t)k * 8);
    if (Vt_out) std::memcpy(Vt_out, Vt.data(), (size_t)k * km * 8);
    return finish_problem(c);
}

int bmc_get_gram(bmc_ctx* c, double* out) {
    if (!c || !out) return BMC_EINVAL;
    if (!c->have_problem) return fail(c, BMC_ESTATE, "no problem set");
    std::memcpy(out, c->gram.data(), c->gram.size() * sizeof(double));
    return BMC_OK;
}

int bmc_get_basis(bmc_ctx* c, double* W_out, double* lam_out, double* sigma2_init) {
    if (!c) return BMC_EINVAL;
    if (!c->have_prior) return fail(c, BMC_ESTATE, "no prior set");
    if (W_out) std::memcpy(W_out, c->W.data(), c->W.size() * sizeof(double));
    if (lam_out) std::memcpy(lam_out, c->lam.data(), c->lam.size() * sizeof(double));
    if (sigma2_init) *sigma2_init = c->sigma2_init;
    return BMC_OK;
}

int bmc_conditional_moments(bmc_ctx* c, double sigma2, double* mean_out, double* cov_out) {
    if (!c) return BMC_EINVAL;
    if (!c->have_prior) return fail(c, BMC_ESTATE, "no prior set");
    const int k = c->k;
    std::vector<double> d(k), m(k);
    for (int j = 0; j < k; ++j) {
        d[j] = 1.0 / (c->lam[j] / sigma2 + 1.0);
        m[j] = d[j] * (c->c1[j] + c->c2[j] / sigma2);
    }
    if (mean_out)
        for (int i = 0; i < k; ++i) {
            long double s = 0.0L;
            for (int j = 0; j < k; ++j) s += (long double)c->W[(size_t)i * k + j] * m[j];
            mean_out[i] = (double)s;
        }
    if (cov_out)
        for (int i = 0; i < k; ++i)
            for (int j = 0; j < k; ++j) {
                long double s = 0.0L;
                for (int q = 0; q < k; ++q)
                    s += (long double)c->W[(size_t)i * k + q] * d[q] * c->W[(size_t)j * k + q];
                cov_out[(size_t)i * k + j] = (double)s;
            }
    return BMC_OK;
}

int bmc_residual_rss(bmc_ctx* c, const double* beta, int32_t nb, double* rss_out) {
    if (!c) return BMC_EINVAL;
    if (!c->have_problem) return fail(c, BMC_ESTATE, "no problem set");
    if (!beta || !rss_out || nb < 1) return fail(c, BMC_EINVAL, "beta/rss_out/nb invalid");
    HIPCHK(c, hipSetDevice(c->device));
    return rss_on_raw(c, beta, nb, rss_out);
}

int bmc_residual_rss_bench(bmc_ctx* c, int32_t nb, int32_t reps, double* ms_per_launch) {
    if (!c) return BMC_EINVAL;
    if (!c->have_problem) return fail(c, BMC_ESTATE, "no problem set");
    if (nb < 1 || nb > 8 || reps < 1 || !ms_per_launch) return fail(c, BMC_EINVAL, "bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    const Panels P = panels_of(c, c->Xraw.p);
    int rc;
    if ((rc = ensure(c, c->rssPartial, (size_t)rss_groups(P) * 8 * sizeof(double)))) return rc;
    if ((rc = ensure(c, c->rssOut, 8 * sizeof(double)))) return rc;
    if ((rc = ensure(c, c->coef, (size_t)8 * c->k * sizeof(double)))) return rc;
    if ((rc = ensure_ticket(c))) return rc;
    std::vector<double> cf((size_t)nb * c->k);
    for (size_t i = 0; i < cf.size(); ++i) cf[i] = 0.01 * (double)((i * 2654435761u) % 97) - 0.5;
    HIPCHK(c, hipMemcpyAsync(c->coef.p, cf.data(), cf.size() * 8, hipMemcpyHostToDevice, c->stream));
    for (int i = 0; i < 3; ++i)
        HIPCHK(c, launch_residual_rss(P, (const double*)c->coef.p, nb, (double*)c->rssPartial.p,
                                      (unsigned*)c->ticket.p, (double*)c->rssOut.p, c->stream));
    HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    for (int i = 0; i < reps; ++i)
        HIPCHK(c, launch_residual_rss(P, (const double*)c->coef.p, nb, (double*)c->rssPartial.p,
                                      (unsigned*)c->ticket.p, (double*)c->rssOut.p, c->stream));
    HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
    *ms_per_launch = (double)ms / reps;
    return BMC_OK;
}

int bmc_gram_bench(bmc_ctx* c, int32_t reps, double* ms_per_launch) {
    if (!c) return BMC_EINVAL;
    if (!c->have_problem) return fail(c, BMC_ESTATE, "no problem set");
    if (reps < 1 || !ms_per_launch) return fail(c, BMC_EINVAL, "bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    const Panels P = panels_of(c, c->Xraw.p);
    int rc;
    if ((rc = ensure(c, c->gramScratch, gram_scratch_bytes(P)))) return rc;
    if ((rc = ensure(c, c->gramOut, (size_t)(c->k + 1) * (c->k + 1) * 8))) return rc;
    for (int i = 0; i < 2; ++i)
        HIPCHK(c, launch_gram(P, c->gramScratch.p, (double*)c->gramOut.p, c->stream));
    HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    for (int i = 0; i < reps; ++i)
        HIPCHK(c, launch_gram(P, c->gramScratch.p, (double*)c->gramOut.p, c->stream));
    HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
    *ms_per_launch = (double)ms / reps;
    return BMC_OK;
}

int bmc_gibbs_run(bmc_ctx* c, int32_t n_chains, int64_t iters, const uint64_t* seeds, int rng_mode,
                  const double* xi, const double* g, double* samples_out, bmc_stats* stats) {
    if (!c) return BMC_EINVAL;
    if (!samples_out && iters > 0) return fail(c, BMC_EINVAL, "samples_out must not be NULL");
    HIPCHK(c, hipSetDevice(c->device));
    return run_common(c, n_chains, iters, seeds, rng_mode, xi, g, samples_out, nullptr, stats);
}

int bmc_gibbs_run_device(bmc_ctx* c, int32_t n_chains, int64_t iters, const uint64_t* seeds,
                         void* d_samples_out, bmc_stats* stats) {
    if (!c) return BMC_EINVAL;
    if (!d_samples_out && iters > 0) return fail(c, BMC_EINVAL, "d_samples_out must not be NULL");
    HIPCHK(c, hipSetDevice(c->device));
    return run_common(c, n_chains, iters, seeds, BMC_RNG_DEVICE, nullptr, nullptr, nullptr,
                      d_samples_out, stats);
}

int bmc_rng_fill(bmc_ctx* c, uint64_t seed, int64_t count_normal, double* normals_out, double shape,
                 int64_t count_gamma, double* gammas_out) {
    if (!c) return BMC_EINVAL;
    if (count_normal < 0 || count_gamma < 0 || (count_normal > 0 && !normals_out) ||
        (count_gamma > 0 && (!gammas_out || !(shape > 0.0))))
        return fail(c, BMC_EINVAL, "bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if ((rc = ensure(c, c->xi, (size_t)count_normal * 8))) return rc;
    if ((rc = ensure(c, c->gam, (size_t)count_gamma * 8))) return rc;
    if ((rc = ensure(c, c->seeds, sizeof(uint64_t)))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->seeds.p, &seed, sizeof(seed), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, launch_rng_fill((const uint64_t*)c->seeds.p, 1, count_normal, (double*)c->xi.p, shape,
                              count_gamma, (double*)c->gam.p, c->stream));
    if (count_normal)
        HIPCHK(c, hipMemcpyAsync(normals_out, c->xi.p, (size_t)count_normal * 8,
                                 hipMemcpyDeviceToHost, c->stream));
    if (count_gamma)
        HIPCHK(c, hipMemcpyAsync(gammas_out, c->gam.p, (size_t)count_gamma * 8,
                                 hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BMC_OK;
}

int bmc_simplex_run(bmc_ctx* c, const double* Vt_hat, int32_t Km, const double* S_hat,
                    int64_t iters, int64_t burn, double stepsize, double nu0, double sigma20,
                    int rng_mode, uint64_t seed, const double* xi, const double* unif,
                    int64_t n_unif, const double* g, double* samples_out, int64_t* accepted_out,
                    int64_t* unif_used_out, bmc_stats* stats) {
    if (!c) return BMC_EINVAL;
    if (!c->have_problem) return fail(c, BMC_ESTATE, "bmc_set_problem must be called first");
    if (!Vt_hat || !S_hat || Km < 1) return fail(c, BMC_EINVAL, "Vt_hat/S_hat/n_models invalid");
    if (burn < 0) return fail(c, BMC_EINVAL, "Burn-in iterations must be non-negative.");
    if (!(stepsize > 0)) return fail(c, BMC_EINVAL, "Stepsize must be positive.");
    if (iters < 0 || burn + iters >= 0xffffffffll) return fail(c, BMC_EINVAL, "bad iteration count");
    if (iters > 0 && !samples_out) return fail(c, BMC_EINVAL, "samples_out must not be NULL");
    if (rng_mode == BMC_RNG_REPLAY) {
        if (!xi || !g || (n_unif > 0 && !unif) || n_unif < 0)
            return fail(c, BMC_EINVAL, "xi, g and unif required in replay mode");
    } else if (rng_mode == BMC_RNG_DEVICE) {
        if (xi || g || unif) return fail(c, BMC_EINVAL, "xi/g/unif must be NULL in device RNG mode");
    } else {
        return fail(c, BMC_EINVAL, "rng_mode must be 0 or 1");
    }
    HIPCHK(c, hipSetDevice(c->device));
    const int K = c->k;
    const size_t Tt = (size_t)(burn + iters);
    int rc;
    // -log_likelihood at beta = 0 (inference_utils.py:83-85) through the residual kernel
    std::vector<double> zero(K, 0.0);
    double rss0 = 0.0;
    if ((rc = rss_on_raw(c, zero.data(), 1, &rss0))) return rc;
    if ((rc = ensure(c, c->xi, Tt * K * 8)) || (rc = ensure(c, c->gam, Tt * 8)) ||
        (rc = ensure(c, c->sVt, (size_t)K * Km * 8)) || (rc = ensure(c, c->sStep, (size_t)K * 8)) ||
        (rc = ensure(c, c->sOut, (size_t)iters * (K + 1) * 8)) || (rc = ensure(c, c->sCnt, 64)) ||
        (rc = ensure(c, c->status, 16)) || (rc = ensure(c, c->placement, 16)) ||
        (rc = ensure(c, c->seeds, 16)))
        return rc;
    if (rng_mode == BMC_RNG_DEVICE) n_unif = (int64_t)Tt;
    if ((rc = ensure(c, c->sUnif, (size_t)(n_unif > 0 ? n_unif : 1) * 8))) return rc;
    std::vector<double> step(K);
    for (int j = 0; j < K; ++j) step[j] = std::sqrt(S_hat[j] * S_hat[j] * stepsize * stepsize);  // :80
    if ((rc = check_tuning_fits(c))) return rc;
    const Geometry geo = choose_geometry(c, 1, Km <= 64, 4);   // (a model per lane)
    const int gran_stride = bmc::gran_slot_words(geo.G);
    if ((rc = ensure(c, c->gran, (size_t)3 * gran_stride * 8))) return rc;
    HIPCHK(c, hipMemsetAsync(c->gran.p, 0, (size_t)3 * gran_stride * 8, c->stream));
    HIPCHK(c, hipMemsetAsync(c->status.p, 0, 16, c->stream));
    HIPCHK(c, hipMemsetAsync(c->placement.p, 0, 16, c->stream));
    HIPCHK(c, hipMemsetAsync(c->sCnt.p, 0, 64, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->sVt.p, Vt_hat, (size_t)K * Km * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->sStep.p, step.data(), (size_t)K * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    if (Tt > 0) {
        if (rng_mode == BMC_RNG_DEVICE) {
            HIPCHK(c, hipMemcpyAsync(c->seeds.p, &seed, 8, hipMemcpyHostToDevice, c->stream));
            const double shape = (nu0 + (double)c->n) / 2.0;                       // :115
            HIPCHK(c, launch_rng_fill((const uint64_t*)c->seeds.p, 1, (int64_t)Tt * K,
                                      (double*)c->xi.p, shape, (int64_t)Tt, (double*)c->gam.p,
                                      c->stream));
            HIPCHK(c, launch_uniform_fill(seed, n_unif, (double*)c->sUnif.p, c->stream));
        } else {
            HIPCHK(c, hipMemcpyAsync(c->xi.p, xi, Tt * K * 8, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(c->gam.p, g, Tt * 8, hipMemcpyHostToDevice, c->stream));
            if (n_unif > 0)
                HIPCHK(c, hipMemcpyAsync(c->sUnif.p, unif, (size_t)n_unif * 8, hipMemcpyHostToDevice,
                                         c->stream));
        }
    }
    HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    SimplexArgs a;
    a.P = panels_of(c, c->Xraw.p);
    a.Vt = (const double*)c->sVt.p;
    a.Km = Km;
    a.vt_in_lds = (size_t)K * Km <= 4096;
    a.step = (const double*)c->sStep.p;
    a.nu0_s20 = nu0 * sigma20;
    a.rss_init = rss0;
    a.xi = (const double*)c->xi.p;
    a.unif = (const double*)c->sUnif.p;
    a.n_unif = n_unif;
    a.gam = (const double*)c->gam.p;
    a.out = (double*)c->sOut.p;
    a.gran = (unsigned long long*)c->gran.p;
    a.gran_stride = gran_stride;
    a.status = (int32_t*)c->status.p;
    a.placement = (int32_t*)c->placement.p;
    a.counters = (long long*)c->sCnt.p;
    a.iters = iters;
    a.burn = burn;
    a.G = geo.G;
    a.waves = geo.waves;
    a.mode = geo.mode;
    a.reg_ppw = geo.ppw;
    a.nslot = geo.nslot;
    a.force_agent_scope = c->tune.force_agent_scope;
    a.panels_per_group = geo.ppg;
    a.one_wave = geo.one_wave;
    a.epoch0 = launch_nonce(c, (uint64_t)Tt);
    if (a.vt_in_lds && simplex_lds_bytes(a) > LDS_LIMIT) a.vt_in_lds = 0;
    if (simplex_lds_bytes(a) > LDS_LIMIT) return fail(c, BMC_EINVAL, "LDS plan exceeds 160 KiB");
    if (Tt > 0) {
        if (a.G > 1 &&
            (rc = check_residency(c, a, a.G, launch_simplex, "persistent simplex kernel")))
            return rc;
        HIPCHK(c, launch_simplex(a, c->stream));
    }
    HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
    int32_t st = 0, place = 0;
    long long cnt[2] = {0, 0};
    HIPCHK(c, hipMemcpyAsync(&st, c->status.p, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&place, c->placement.p, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(cnt, c->sCnt.p, 16, hipMemcpyDeviceToHost, c->stream));
    if (iters > 0)
        HIPCHK(c, hipMemcpyAsync(samples_out, c->sOut.p, (size_t)iters * (K + 1) * 8,
                                 hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (accepted_out) *accepted_out = cnt[0];
    if (unif_used_out) *unif_used_out = cnt[1];
    if (stats) {
        float ms = 0;
        std::memset(stats, 0, sizeof(*stats));
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1])); stats->rng_ms = ms;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev[1], c->ev[2])); stats->loop_ms = ms;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[2])); stats->total_ms = ms;
        stats->iterations = burn + iters;
        stats->n_chains = 1;
        stats->launches = Tt > 0 ? 1 : 0;
        stats->groups_per_chain = geo.G;
        stats->waves_per_group = geo.waves;
        stats->chains_per_pass = 1;
        stats->residency = geo.mode + 1;
        stats->xcd_local_chains = place ? 1 : 0;
        stats->bytes_per_pass = ((int64_t)c->n * K + c->n) * (c->f32 ? 4 : 8);
        stats->passes = cnt[1];
    }
    if (st == 1) return fail(c, BMC_ETIMEOUT, "persistent simplex kernel: bounded spin expired");
    if (st == 2) return fail(c, BMC_EINVAL, "replay: fewer uniforms supplied than proposals inside the simplex");
    return BMC_OK;
}

int bmc_predict(bmc_ctx* c, const double* preds, int64_t M, int32_t Km, const double* theta,
                int32_t S, int32_t k, const double* Vt_hat, int rng_mode, uint64_t seed,
                const double* noise, const int32_t* q_index, const double* q_gamma, int32_t n_q,
                const double* truth, const int32_t* cov_lo, const int32_t* cov_hi, int32_t n_cov,
                double* rndm_m_out, double* bands_out, int64_t* cov_hits_out) {
    if (!c) return BMC_EINVAL;
    if (!preds || !theta || !Vt_hat) return fail(c, BMC_EINVAL, "preds/theta/Vt_hat must not be NULL");
    if (M < 1 || Km < 1 || k < 1 || S < 1) return fail(c, BMC_EINVAL, "empty predictive problem");
    if (S > 16384) return fail(c, BMC_EINVAL, "n_draws > 16384 is not supported");
    if (n_q < 0 || n_q > 64 || n_cov < 0 || n_cov > 64)
        return fail(c, BMC_EINVAL, "n_q and n_cov must be in 0..64");
    if (n_q > 0 && (!q_index || !q_gamma || !bands_out))
        return fail(c, BMC_EINVAL, "order statistics requested without index/gamma/output");
    if (n_cov > 0 && (!truth || !cov_lo || !cov_hi || !cov_hits_out))
        return fail(c, BMC_EINVAL, "coverage requested without truth/bounds/output");
    if (rng_mode == BMC_RNG_REPLAY ? !noise : noise != nullptr)
        return fail(c, BMC_EINVAL, "noise must be given exactly in replay mode");
    if (rng_mode != BMC_RNG_REPLAY && rng_mode != BMC_RNG_DEVICE)
        return fail(c, BMC_EINVAL, "rng_mode must be 0 or 1");
    for (int i = 0; i < n_q; ++i)
        if (q_index[i] < 0 || q_index[i] >= S) return fail(c, BMC_EINVAL, "q_index out of range");
    for (int i = 0; i < n_cov; ++i)
        if (cov_lo[i] < 0 || cov_lo[i] >= S || cov_hi[i] < 0 || cov_hi[i] >= S)
            return fail(c, BMC_EINVAL, "coverage index out of range");
    HIPCHK(c, hipSetDevice(c->device));
    c->pM = 0;   // (no draws to fetch until this call has produced them)
    PredictArgs a;
    a.M = M; a.Km = Km; a.k = k; a.S = S;
    a.S_pad = (S + 63) / 64 * 64;
    a.Km_pad = (Km + 3) / 4 * 4;
    a.M_pad = (M + 63) / 64 * 64;
    a.seed = seed;
    a.n_q = n_q; a.n_cov = n_cov;
    const size_t szP = (size_t)M * Km * 8, szT = (size_t)S * (k + 1) * 8, szV = (size_t)k * Km * 8;
    int rc;
    if ((rc = ensure(c, c->pPreds, szP)) || (rc = ensure(c, c->pTheta, szT)) ||
        (rc = ensure(c, c->pVt, szV)) ||
        (rc = ensure(c, c->pWt, ((size_t)a.S_pad * a.Km_pad + 16) * 8)) ||
        (rc = ensure(c, c->pPad, ((size_t)a.M_pad * a.Km_pad + 16) * 8)) ||
        (rc = ensure(c, c->pSig, (size_t)a.S_pad * 8)) ||
        (rc = ensure(c, c->pR, (size_t)a.M_pad * a.S_pad * 8)) ||
        (rc = ensure(c, c->pBands, (size_t)(n_q > 0 ? n_q : 1) * M * 8)))
        return rc;
    // aux block: q_index[64] i32 | cov_lo[64] | cov_hi[64] | q_gamma[64] f64 | hits[64] u64 |
    //            fail_count (16 B) | truth[M] f64 | fail_points[M] i32
    const size_t offQ = 0, offLo = 256, offHi = 512, offG = 768, offH = 768 + 512, offFC = offH + 512;
    const size_t offT = offFC + 16, offFP = offT + (size_t)M * 8;
    if ((rc = ensure(c, c->pAux, offFP + (size_t)M * 4))) return rc;
    char* aux = (char*)c->pAux.p;
    HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    HIPCHK(c, hipMemsetAsync(aux, 0, offT, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->pPreds.p, preds, szP, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->pTheta.p, theta, szT, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->pVt.p, Vt_hat, szV, hipMemcpyHostToDevice, c->stream));
    if (n_q) {
        HIPCHK(c, hipMemcpyAsync(aux + offQ, q_index, n_q * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(aux + offG, q_gamma, n_q * 8, hipMemcpyHostToDevice, c->stream));
    }
    if (n_cov) {
        HIPCHK(c, hipMemcpyAsync(aux + offLo, cov_lo, n_cov * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(aux + offHi, cov_hi, n_cov * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(aux + offT, truth, (size_t)M * 8, hipMemcpyHostToDevice, c->stream));
    }
    a.noise_replay = nullptr;
    if (noise) {
        if ((rc = ensure(c, c->pNoise, (size_t)S * M * 8))) return rc;
        HIPCHK(c, hipMemcpyAsync(c->pNoise.p, noise, (size_t)S * M * 8, hipMemcpyHostToDevice, c->stream));
        a.noise_replay = (const double*)c->pNoise.p;
    }
    a.preds = (const double*)c->pPreds.p;
    a.theta = (const double*)c->pTheta.p;
    a.Vt = (const double*)c->pVt.p;
    a.Wt = (double*)c->pWt.p;
    a.P = (double*)c->pPad.p;
    a.sig = (double*)c->pSig.p;
    a.R = (double*)c->pR.p;
    a.q_index = (const int32_t*)(aux + offQ);
    a.q_gamma = (const double*)(aux + offG);
    a.truth = n_cov ? (const double*)(aux + offT) : nullptr;
    a.cov_lo = (const int32_t*)(aux + offLo);
    a.cov_hi = (const int32_t*)(aux + offHi);
    a.bands = (double*)c->pBands.p;
    a.hits = (unsigned long long*)(aux + offH);
    a.fail_count = (int32_t*)(aux + offFC);
    a.fail_points = (int32_t*)(aux + offFP);
    a.ev_mid = c->ev[2];
    HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    HIPCHK(c, launch_predict(a, c->stream));
    HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
    if (n_q)
        HIPCHK(c, hipMemcpyAsync(bands_out, c->pBands.p, (size_t)n_q * M * 8, hipMemcpyDeviceToHost,
                                 c->stream));
    if (n_cov)
        HIPCHK(c, hipMemcpyAsync(cov_hits_out, aux + offH, (size_t)n_cov * 8, hipMemcpyDeviceToHost,
                                 c->stream));
    if (rndm_m_out)
        HIPCHK(c, hipMemcpy2DAsync(rndm_m_out, (size_t)S * 8, c->pR.p, (size_t)a.S_pad * 8,
                                   (size_t)S * 8, (size_t)M, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    {
        float ms = 0;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1])); c->predict_ms[0] = ms;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev[1], c->ev[2])); c->predict_ms[1] = ms;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev[2], c->ev[3])); c->predict_ms[2] = ms;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[3])); c->predict_ms[3] = ms;
    }
    c->pM = M;
    c->pS = S;
    c->pS_pad = a.S_pad;
    return BMC_OK;
}

int bmc_predict_draws(bmc_ctx* c, double* out, int layout) {
    if (!c) return BMC_EINVAL;
    if (!out) return fail(c, BMC_EINVAL, "out must not be NULL");
    if (layout != BMC_DRAWS_BY_POINT && layout != BMC_DRAWS_BY_DRAW)
        return fail(c, BMC_EINVAL, "layout must be 0 (by point) or 1 (by draw)");
    if (c->pM < 1 || !c->pR.p) return fail(c, BMC_ESTATE, "no bmc_predict has run on this context");
    HIPCHK(c, hipSetDevice(c->device));
    const int64_t M = c->pM;
    const int32_t S = c->pS, S_pad = c->pS_pad;
    int rc;
    if (layout == BMC_DRAWS_BY_POINT)
        return copy_to_host(c, out, c->pR.p, (size_t)S * 8, (size_t)S_pad * 8, (size_t)M);
    if ((rc = ensure(c, c->pRT, (size_t)S * M * 8))) return rc;
    HIPCHK(c, launch_transpose_draws((const double*)c->pR.p, M, S, S_pad, (double*)c->pRT.p,
                                     c->stream));
    // (rows of one draw: M doubles each, dense)
    return copy_to_host(c, out, c->pRT.p, (size_t)M * 8, (size_t)M * 8, (size_t)S);
}

int bmc_predict_timing(bmc_ctx* c, double* h2d_ms, double* gemm_ms, double* orderstat_ms,
                       double* device_ms) {
    if (!c) return BMC_EINVAL;
    if (h2d_ms) *h2d_ms = c->predict_ms[0];
    if (gemm_ms) *gemm_ms = c->predict_ms[1];
    if (orderstat_ms) *orderstat_ms = c->predict_ms[2];
    if (device_ms) *device_ms = c->predict_ms[3];
    return BMC_OK;
}

// ---- pooling over GPUs: RCCL, loaded on first use --------------------------------------
extern "C++" {
namespace {
struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t,
                              hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
// One RCCL per process: the copy already in the process (torch loads its own, same soname)
// wins; otherwise the loader's search path, then the ROCm install.  BMC_RCCL_SONAME (tests
// only) replaces the candidate list, so that a failing load can be exercised.
// The table is built ONCE, by a function-local static (C++11: thread-safe, so two host threads
// that drive two contexts cannot see it half filled); the loader's error text is taken right
// after the failing dlopen / dlsym -- dlerror() clears itself when read -- and kept.
struct RcclLoad {
    Rccl r;
    std::string err;
};
RcclLoad load_rccl() {
    RcclLoad L;
    Rccl& r = L.r;
    std::vector<std::string> names;
    if (const char* forced = std::getenv("BMC_RCCL_SONAME")) names = {forced};
    else names = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const std::string& name : names)
        if (!r.h) r.h = dlopen(name.c_str(), RTLD_NOW | RTLD_NOLOAD);
    for (const std::string& name : names)
        if (!r.h) {
            (void)dlerror();
            r.h = dlopen(name.c_str(), RTLD_NOW | RTLD_GLOBAL);
            if (!r.h) {
                const char* e = dlerror();
                L.err = e ? e : (name + ": dlopen failed");
            }
        }
    if (!r.h) {
        if (L.err.empty()) L.err = "librccl.so.1 not found";
        return L;
    }
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.h, "ncclCommInitRank");
    r.AllGather = (decltype(r.AllGather))dlsym(r.h, "ncclAllGather");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.h, "ncclCommDestroy");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy || !r.GetErrorString) {
        r.h = nullptr;
        L.err = "RCCL symbols missing (ncclGetUniqueId / ncclCommInitRank / ncclAllGather / "
                "ncclCommDestroy / ncclGetErrorString)";
    }
    return L;
}
const RcclLoad& rccl_state() {
    static const RcclLoad L = load_rccl();
    return L;
}
const Rccl* rccl() {
    const RcclLoad& L = rccl_state();
    return L.r.h ? &L.r : nullptr;
}
#define RCCLCHK(ctx, R, expr)                                                          \
    do {                                                                               \
        ncclResult_t r__ = (expr);                                                     \
        if (r__ != ncclSuccess)                                                        \
            return fail(ctx, BMC_EHIP, std::string(#expr) + ": " + (R)->GetErrorString(r__)); \
    } while (0)
}  // namespace
}  // extern "C++"

int bmc_comm_unique_id(char id_out[BMC_COMM_ID_BYTES]) {
    static_assert(BMC_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    if (!id_out) return BMC_EINVAL;
    const Rccl* R = rccl();
    if (!R) return BMC_EHIP;
    ncclUniqueId id;
    if (R->GetUniqueId(&id) != ncclSuccess) return BMC_EHIP;
    std::memcpy(id_out, id.internal, BMC_COMM_ID_BYTES);
    return BMC_OK;
}

int bmc_comm_destroy(bmc_ctx* c) {
    if (!c) return BMC_EINVAL;
    if (c->comm) {
        const Rccl* R = rccl();
        (void)hipSetDevice(c->device);
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        if (R) (void)R->CommDestroy(c->comm);
        c->comm = nullptr;
        c->comm_world = c->comm_rank = 0;
    }
    return BMC_OK;
}

int bmc_comm_init(bmc_ctx* c, int32_t world, int32_t rank, const char id[BMC_COMM_ID_BYTES]) {
    if (!c) return BMC_EINVAL;
    if (!id || world < 1 || rank < 0 || rank >= world)
        return fail(c, BMC_EINVAL, "need world >= 1, 0 <= rank < world and an id");
    const Rccl* R = rccl();
    if (!R) return fail(c, BMC_EHIP, "RCCL could not be loaded: " + rccl_state().err);
    HIPCHK(c, hipSetDevice(c->device));
    bmc_comm_destroy(c);
    ncclUniqueId uid;
    std::memcpy(uid.internal, id, BMC_COMM_ID_BYTES);
    RCCLCHK(c, R, R->CommInitRank(&c->comm, world, uid, rank));
    c->comm_world = world;
    c->comm_rank = rank;
    return BMC_OK;
}

int bmc_allgather(bmc_ctx* c, const void* d_send, void* d_recv, int64_t count_per_rank) {
    if (!c) return BMC_EINVAL;
    if (!c->comm) return fail(c, BMC_ESTATE, "bmc_comm_init must be called first");
    if (!d_send || !d_recv || count_per_rank < 0) return fail(c, BMC_EINVAL, "bad arguments");
    const Rccl* R = rccl();
    if (!R) return fail(c, BMC_EHIP, "RCCL is not loaded");
    HIPCHK(c, hipSetDevice(c->device));
    if (count_per_rank > 0)
        RCCLCHK(c, R, R->AllGather(d_send, d_recv, (size_t)count_per_rank, ncclFloat64, c->comm,
                                   c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BMC_OK;
}

#ifdef BMC_STAMPS
// Diagnostic build only; not part of the public ABI.
int bmc_dev_get_stamps(bmc_ctx* c, long long* out8) {
    if (!c || !out8 || !c->dbg.p) return BMC_EINVAL;
    HIPCHK(c, hipMemcpy(out8, c->dbg.p, 12 * sizeof(long long), hipMemcpyDeviceToHost));
    return BMC_OK;
}
#endif

int bmc_philox_raw(bmc_ctx* c, uint64_t seed, uint32_t stream_id, int64_t nblocks4, uint32_t* out) {
    if (!c) return BMC_EINVAL;
    if (nblocks4 < 1 || !out) return fail(c, BMC_EINVAL, "bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if ((rc = ensure(c, c->xi, (size_t)nblocks4 * 16))) return rc;
    HIPCHK(c, launch_philox_raw(seed, stream_id, nblocks4, (uint32_t*)c->xi.p, c->stream));
    HIPCHK(c, hipMemcpyAsync(out, c->xi.p, (size_t)nblocks4 * 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BMC_OK;
}

}  // extern "C"
