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
#include "bmc_loop.h"

#ifndef BMC_LEADER_PRIO
#define BMC_LEADER_PRIO 3      // s_setprio of the leader wave of a group (0: leave it alone)
#endif

namespace bmc {

// u_j | sigma2 with sigma2 = sp / g (the beta | sigma2 draw of inference_utils.py:41-45 in the
// basis of bmc_set_prior):  D = lam g + sp, r = rsqrt(D), d_j = sp r^2,
//   u_j = r^2 (c1 sp + c2 g) + sqrt(sp) r xi.
// sqrt(sp) is taken as sp * rsqrt(sp) right here, beside rsqrt(D): two independent
// v_rsq_f64 + one Newton step each, which the in-order pipeline overlaps, instead of a full
// sqrt (rsq + two Newton steps + range scaling, ~14 dependent f64 operations) at the end of the
// previous iteration's sigma2 step, all of it on the serial path.
// 1/sqrt(x) for finite x > 0: v_rsq_f64 and one Newton step, the arithmetic of the library
// rsqrt() without its tests for 0 / inf / nan arguments (D and sp are sums of positive terms).
__device__ __forceinline__ double rsqrt_pos(double x) {
#ifdef BMC_LIB_RSQRT
    return rsqrt(x);
#else
    const double y0 = __builtin_amdgcn_rsq(x);
    const double e = fma(-x * y0, y0, 1.0);           // 1 - x y0^2
    return fma(y0 * e, fma(e, 0.375, 0.5), y0);       // y0 (1 + e/2 + 3 e^2 / 8)
#endif
}

__device__ __forceinline__ double draw_u(double lam, double c1, double c2, double xi, double sp,
                                         double g, double sq_sp_unused) {
#ifndef BMC_SQRT_SEPARATE
    (void)sq_sp_unused;
    const double D = fma(lam, g, sp);
    const double r = rsqrt_pos(D);
    const double rs = rsqrt_pos(sp);
    const double m = fma(c2, g, c1 * sp);
    return fma(r * r, m, ((sp * rs) * r) * xi);
#else
    const double D = fma(lam, g, sp);
    const double r = rsqrt(D);
    const double m = fma(c2, g, c1 * sp);
    return fma(r * r, m, (sq_sp_unused * r) * xi);
#endif
}
#ifndef BMC_SQRT_SEPARATE
#define BMC_SQRT_OF(x) 0.0     /* not used: draw_u takes sqrt(sp) as sp * rsqrt(sp) */
#else
#define BMC_SQRT_OF(x) sqrt(x)
#endif

// PACK: the same kernel held to 128 VGPRs (4 waves per SIMD), so that two 5-wave groups of
// different chains fit a CU side by side whatever SIMDs their waves land on -- used when more
// than 8 chains share a launch (two per XCD).  It exists only for shapes with one panel of at
// most 64 data VGPRs per wave (e.g. K = 32 doubles); at 136 VGPRs the unpacked kernel is 3 %
// faster per iteration, so a single chain keeps that one.
template <typename T, int VEC, int MODE, int KMAX, int PPW>
constexpr bool loop_can_pack() {
    return MODE == MODE_REG && PPW == 1 && VEC == 1 && KMAX * (int)(sizeof(T) / 4) <= 64 &&
           !(sizeof(T) == 4 && KMAX == 64);   // (f32, 64 columns spills at 128)
}

// SMALLG: the host promises G <= 32 (one-level exchange); the loop then exists in two copies,
// one per exchange scope, chosen once after the placement check (exchange_sum, TEAMS / LOCAL).
template <typename T, int VEC, int MODE, int KMAX, int PPW, bool SINGLE = false, bool PACK = false,
          bool SMALLG = false>
__global__ __attribute__((amdgpu_flat_work_group_size(1, 512), amdgpu_waves_per_eu(PACK ? 4 : 2)))
void gibbs_loop_kernel(GibbsArgs a) {
    constexpr int RP = 64 * VEC;
    // lane-chunks of 64 columns: register residency means K <= 64, so one chunk is known at
    // compile time (fewer live registers and no dead branches in the leader's serial phase)
    constexpr int KCH = (MODE == MODE_REG) ? 1 : MAX_KCH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int K = a.P.k;
    const int C = a.n_chains, G = a.G;
    // nslot = 8: slot label, blocks b and b+8 share an XCD (observed); nslot = C otherwise
    const int chain = blockIdx.x % a.nslot;
    int g = blockIdx.x / a.nslot;
    if (chain >= C) return;                // unused slot: the whole workgroup leaves
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = blockDim.x >> 6;
    const int64_t T_it = a.iters;
#ifndef BMC_NO_XCD_REMAP
    // One chain over the whole chip (G > 32, a multiple of 8): team j = the groups with g mod 8 = j
    // is meant to be the groups of XCD j.  The hardware deals the workgroups of a launch round
    // robin, but starts at an XCD that depends on the queue: renumber the groups so that group
    // g' = 8 (b / 8) + (XCC id of block b) -- team j then runs on XCD j whatever the start was
    // (measured: 2.02 us per iteration at N = 100 000 x 32 with team j on XCD j, 2.25 rotated by
    // 5 to 7; C4 3.28 vs 3.45-3.55).  Falls back to g = b when the placement is not a rotation.
    if constexpr (!SINGLE && !SMALLG) {
        if (G > 32 && (G & 7) == 0 && a.nslot == 1) {
            __shared__ int rot_c;
            if (wave == 0) {
                const int c = detect_rotation(a.gran + (size_t)2 * a.gran_stride + 256, G, g, lane, a.epoch0);
                if (lane == 0) rot_c = c;
            }
            __syncthreads();
            if (rot_c >= 0) g = (g & ~7) | ((g + rot_c) & 7);
        }
    }
#endif

    const LdsPlan L = lds_plan(K, (int)sizeof(T), RP, a.panels_per_group, MODE == MODE_LDS);
    double* u_lds = reinterpret_cast<double*>(smem + L.u);
    double* red = reinterpret_cast<double*>(smem + L.red);  // per-wave partials
    double* ctl = reinterpret_cast<double*>(smem + L.ctl);  // [0] sp, [1] abort, [2] local, [3] g

    const int kpad = (K + 63) & ~63;
    for (int j = tid; j < kpad; j += blockDim.x) u_lds[j] = 0.0;
    for (int j = tid; j < RED_DOUBLES; j += blockDim.x) red[j] = 0.0;   // rows / slots of absent waves stay 0
    if (tid == 0) { ctl[0] = a.sigma2_init; ctl[1] = 0.0; ctl[2] = 0.0; ctl[3] = 1.0; }

    PanelStore<T, VEC, MODE, KMAX, PPW> store;
    store.init(a.P, G, g, reinterpret_cast<T*>(smem + L.x), reinterpret_cast<T*>(smem + L.y));

    // ---- where do this chain's groups really run? -------------------------------------
    gu64* gr = a.gran + (size_t)chain * 3 * a.gran_stride;
    if constexpr (!SINGLE) {
        if (wave == 0) {
            const int place = detect_placement(gr + 2 * a.gran_stride, G, g, lane, a.epoch0);
            if (lane == 0) {
                if (place < 0) { ctl[1] = 1.0; a.status[chain] = 1; }
                ctl[2] = (place == 1 && !a.force_agent_scope) ? 1.0 : 0.0;
            }
        }
    }
    __syncthreads();
    const bool local = ctl[2] != 0.0;
    if (g == 0 && tid == 0) a.placement[chain] = (local || SINGLE) ? 1 : 0;

    const double* xi = a.xi + (int64_t)chain * T_it * K;
    const double* gam = a.gam + (int64_t)chain * T_it;
    double* uout = a.uout + (int64_t)chain * T_it * (K + 1);
    // the wave that records the draws (group 0): wave 1 -- on a SIMD of its own -- rather than the
    // last one, which at five waves on four SIMDs shares the leader's SIMD and would issue its
    // ~35 f64 operations of sqrt(sp / g) per iteration beside the serial chain
#ifdef BMC_REC_LAST
    const bool recorder = (g == 0) && (wave == nw - 1);
#else
    const bool recorder = (g == 0) && (wave == (nw > 1 ? 1 : 0));
#endif

    // Every load issued so far (the panels into registers) is complete from here on, and hipcc
    // knows it: otherwise it guards the first FMA of every iteration with s_waitcnt vmcnt(0),
    // which makes the recording wave wait for ITS stores of the previous iteration.
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), other counters untouched
    // sigma2 = sp_eff / g_eff; starts at the OLS value (inference_utils.py:37)
    double sp_eff = a.sigma2_init, g_eff = 1.0, sq_sp = BMC_SQRT_OF(a.sigma2_init);
    double xi_next[KCH], lam_r[KCH], c1_r[KCH], c2_r[KCH];
    double gam_next = 0.0;
    if (wave == 0) {
#pragma unroll
        for (int ch = 0; ch < KCH; ++ch) {
            const int j = ch * 64 + lane;
            xi_next[ch] = (j < K && T_it > 0) ? xi[j] : 0.0;
            lam_r[ch] = j < K ? a.lam[j] : 0.0;
            c1_r[ch] = j < K ? a.c1[j] : 0.0;
            c2_r[ch] = j < K ? a.c2[j] : 0.0;
        }
        if (T_it > 0) gam_next = gam[0];
    }

#ifdef BMC_STAMPS
    const bool stamping = a.dbg != nullptr && blockIdx.x == 0 && wave == 0;
    unsigned long long acc_[12] = {}, last_ = 0;
    if (stamping) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last_)::"memory");
#endif
    // One iteration loop, instantiated per (exchange scope, role, recording duty).  ROLE: 0 =
    // the leader wave (wave 0), 1 = a worker wave, -1 = tested at run time; REC likewise for the
    // recording duty of group 0's last wave.  The chains of at most 32 groups (SMALLG) run
    // role-specific copies -- every wave executes the same two barriers per iteration and leaves
    // the loop at the same place, so the copies stay in step -- which takes the leader's and
    // the recorder's tests out of each other's loops.
    auto run_loop = [&](auto scope_c, auto role_c, auto rec_c) {
    constexpr int LOCALK = decltype(scope_c)::value;     // -1: `local` tested where it is used
    constexpr int ROLE = decltype(role_c)::value, REC = decltype(rec_c)::value;
    const bool is_leader = ROLE < 0 ? wave == 0 : ROLE == 0;
    const bool is_rec = REC < 0 ? recorder : REC == 1;
    // The leader wave carries the serial chain of the iteration; the wave that shares its SIMD
    // (five waves on four SIMDs at C2; in group 0 that wave also records the draws) otherwise
    // takes issue slots from it whenever both are ready.  Same-box A/B, s_setprio 3 for the
    // leader: C2 0.941 -> 0.922 us per iteration, 16 chains at C2 1.056 -> 1.027.
    if (BMC_LEADER_PRIO > 0 && is_leader) __builtin_amdgcn_s_setprio(BMC_LEADER_PRIO);
    for (int64_t t = 0; t < T_it; ++t) {
        // (epoch0: the launch's nonce -- a granule an earlier launch left in some cache carries
        // another base and is never taken for this launch's; 0 is never a live tag)
        const unsigned epoch = (unsigned)(t + 1) + a.epoch0;
        STAMP(7);
        if (is_leader) {
#pragma unroll
            for (int ch = 0; ch < KCH; ++ch) {
                const int j = ch * 64 + lane;
                // register residency (one lane-chunk): every lane draws, no exec-mask region on
                // the serial path; columns K..63 have lam = c1 = c2 = xi = 0 and draw u = 0 into
                // the zero padding of u_lds
                if (MODE == MODE_REG || (ch * 64 < K && j < K))
                    u_lds[j] = draw_u(lam_r[ch], c1_r[ch], c2_r[ch], xi_next[ch], sp_eff, g_eff, sq_sp);
            }
#ifndef BMC_CTL_EARLY
            // the (sp, g) pair of the sigma2 just drawn, for the recording wave: written here,
            // behind u, rather than between the sigma2 step and the draw it feeds (both are in
            // front of barrier B1, after which the recorder reads them)
            if (lane == 0) { ctl[0] = sp_eff; ctl[3] = g_eff; }
#endif
        }
        STAMP(0);
        __syncthreads();  // B1: u (and the previous sp, g / abort word) visible to all waves
        // the abort word is read here with u but tested after the residual pass: a test here
        // would put an LDS round trip between the barrier and the first FMA of every wave
        const double abort_w = ctl[1];
        STAMP(1);

        const double gam_t = gam_next;
        // prefetch next iteration's variates.  A single-workgroup chain has nothing long after
        // the residual pass to hide the loads behind, so it issues them here; chains with an
        // exchange issue them behind the pass (below).
        auto prefetch = [&]() {
            if (is_leader && t + 1 < T_it) {
#pragma unroll
                for (int ch = 0; ch < KCH; ++ch) {
                    const int j = ch * 64 + lane;
                    if (ch * 64 < K && j < K) xi_next[ch] = xi[(t + 1) * K + j];
                }
                gam_next = gam[t + 1];
            }
        };
        if constexpr (SINGLE) prefetch();
        double u_rec[KCH], sp_rec = 0.0, g_rec = 1.0;
        if (is_rec) {  // copy now (wave 0 rewrites u_lds after its gather); store later
#pragma unroll
            for (int ch = 0; ch < KCH; ++ch) {
                const int j = ch * 64 + lane;
                u_rec[ch] = (ch * 64 < K && j < K) ? u_lds[j] : 0.0;
            }
            sp_rec = ctl[0];
            g_rec = ctl[3];
        }

        // ---- partial rss over this group's panels, then over the chain's groups ---------
        const double part = store.partial_rss(u_lds);
        constexpr bool LANEWISE = (MODE == MODE_REG && VEC == 1);
#ifndef BMC_TAIL
#define BMC_TAIL 1
#endif
        // (round 3) BMC_TAIL 1: the lane partials go to LDS at once, ahead of the abort test and
        // the leader's prefetch (stamps had 216 cycles between the end of the pass and the LDS
        // write -- on the leader's way to the group barrier, i.e. on every wave's).  2: the
        // prefetch moved behind the publication of the group total as well -- slower (C2 0.945 ->
        // 0.984 us): vmcnt counts in order, so the polls then wait for the prefetch's L2 misses.
        // (BMC_TAIL 3: the same for the wave-sum form of the group sum -- two rows per lane,
        // LDS-pinned and streamed panels: the wave's total goes to its slot before the abort test.
        // Same-box A/B: C4 3.067 -> 3.057, C5 15.76 -> 15.80, 410 MB 62.0 -> 64.2 us: not the default)
        constexpr bool EARLY = ((BMC_TAIL >= 1) && LANEWISE && !SINGLE) || ((BMC_TAIL == 3) && !SINGLE);
        constexpr bool IDLE_PREFETCH = (BMC_TAIL == 2) && EARLY;
        if constexpr (EARLY) {
            if constexpr (LANEWISE) {
                red[wave * 64 + lane] = part;
            } else {
                const double ws = wave_sum(part);
                if (lane == 0) red[wave] = ws;
            }
        }
        {   // (the empty asm ties the test to `part`, or hipcc moves it back up)
            double abort_late = abort_w;
            asm volatile("" : "+v"(abort_late) : "v"(part));
            if (__builtin_amdgcn_readfirstlane(__double2hiint(abort_late)) != 0) break;
        }
        STAMP(2);
        // behind the residual pass: issued in front of it, hipcc made wave 0's first FMA wait
        // for these loads (vmcnt is in-order and the panel registers were loaded "before" them
        // as far as the loop header can tell); the exchange hides them here
        if constexpr (!SINGLE && !IDLE_PREFETCH) prefetch();
        bool got;
        auto idle_prefetch = [&]() { if constexpr (IDLE_PREFETCH) prefetch(); };
        const double rss = group_allreduce<SINGLE, LANEWISE, (SMALLG ? 0 : -1), LOCALK, ROLE, EARLY>(
            part, red, gr + (size_t)(t & 1) * a.gran_stride, G, g, wave, nw, lane, epoch, local,
            got STAMP_ARGS, idle_prefetch);
        STAMP(8);
        if (is_rec) {
            // row t = [u_t, .]; sigma of the PREVIOUS row (its sp, g were final at B1)
#pragma unroll
            for (int ch = 0; ch < KCH; ++ch) {
                const int j = ch * 64 + lane;
                if (ch * 64 < K && j < K) uout[t * (K + 1) + j] = u_rec[ch];
            }
            if (lane == 0 && t > 0) uout[(t - 1) * (K + 1) + K] = sqrt(sp_rec / g_rec);
        }
        if (is_leader) {
            if (__builtin_expect(!got, 0)) {
                if (lane == 0) { ctl[1] = 1.0; a.status[chain] = 1; }
            } else {
                // sigma2 | beta = scale_post / g_t, floored at 1e-6            (:50-52)
                const double scale_post = (a.nu0_s20 + rss) * 0.5;
                const bool floor_hit = scale_post < 1e-6 * gam_t;
                sp_eff = floor_hit ? 1e-6 : scale_post;
                g_eff = floor_hit ? 1.0 : gam_t;
                sq_sp = BMC_SQRT_OF(sp_eff);
#ifdef BMC_CTL_EARLY
                if (lane == 0) { ctl[0] = sp_eff; ctl[3] = g_eff; }
#endif
            }
            STAMP(6);
        }
    }
#ifndef BMC_CTL_EARLY
    // (the last sigma2, for the row recorded behind the loop)
    if (is_leader && lane == 0 && ctl[1] == 0.0) { ctl[0] = sp_eff; ctl[3] = g_eff; }
#endif
    };   // run_loop
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using IR = std::integral_constant<int, -1>;
    if constexpr (SMALLG) {
        auto by_role = [&](auto scope_c) {
            if (wave == 0) {
                if (recorder) run_loop(scope_c, I0{}, I1{});   // (a group of one wave)
                else run_loop(scope_c, I0{}, I0{});
            } else {
                if (recorder) run_loop(scope_c, I1{}, I1{});
                else run_loop(scope_c, I1{}, I0{});
            }
        };
        if (local) by_role(I1{});
        else by_role(I0{});
    } else {
        run_loop(IR{}, IR{}, IR{});
    }
#ifdef BMC_STAMPS
    if (stamping && lane == 0)
        for (int i = 0; i < 12; ++i) a.dbg[i] = (long long)acc_[i];
#endif
    __syncthreads();
    if (recorder && lane == 0 && T_it > 0 && ctl[1] == 0.0)
        uout[(T_it - 1) * (K + 1) + K] = sqrt(ctl[0] / ctl[3]);
}

// ======================================================================================
// Several chains per pass.  When the panels are streamed (or LDS-pinned), one read of X
// serves CPP chains: wave c < CPP is the leader of chain c (its u, its granules, its sigma2
// state, and -- in group 0 -- its recorded draws); all waves accumulate CPP partial sums per
// panel column.  Arithmetic per chain is that of gibbs_loop_kernel, operation for operation.
// ======================================================================================
// SLOTTED: one bundle per XCD slot (bundle_slots > 0), which implies G <= 32: the one-level
// exchange is known at compile time and the two-level code leaves the loop.
// BAL: bundles of 8 chains on 8 waves with at most 5 panels per group, two register sets per
// wave (PPW = 2): the balanced layout of PanelStore::partial_rss_reg_bal.
template <typename T, int VEC, int MODE, int CPP, int KMAX = 0, int PPW = 0, bool SLOTTED = false,
          bool BAL = false>
__global__ __launch_bounds__(512) void gibbs_multi_kernel(GibbsArgs a) {
    constexpr int RP = 64 * VEC;
    // lane-chunks of 64 columns: register residency means K <= 64, so one chunk is known at
    // compile time (fewer live registers and no dead branches in the leader's serial phase)
    constexpr int KCH = (MODE == MODE_REG) ? 1 : MAX_KCH;
    // register residency, one row per lane: the lane-wise group sum of the single-chain kernel
    // (group_allreduce<LANEWISE>), so that a chain is bit-identical to its solo run
    constexpr bool LANEWISE = (MODE == MODE_REG && VEC == 1);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int K = a.P.k, G = a.G;
    // one bundle of CPP chains over the whole grid, or (bundle_slots > 0) one bundle per slot:
    // blocks b and b + slots share an XCD (observed), bundle b % slots runs chains
    // [CPP (b % slots), CPP (b % slots) + CPP) on the G groups b / slots
    const int slots = SLOTTED ? a.bundle_slots : 1;
    const int bundle = (int)(blockIdx.x % (unsigned)slots);
    int g = (int)(blockIdx.x / (unsigned)slots);
    if (bundle * CPP >= a.n_chains) return;   // unused slot: the whole workgroup leaves
    const int chain0 = bundle * CPP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = blockDim.x >> 6;
    const int64_t T_it = a.iters;
    gu64* gran0 = a.gran + (size_t)chain0 * 3 * a.gran_stride;   // this bundle's first chain
#ifndef BMC_NO_XCD_REMAP
    // team j on XCD j whichever XCD the launch starts on (see gibbs_loop_kernel)
    if (!SLOTTED && G > 32 && (G & 7) == 0) {
        __shared__ int rot_c;
        if (wave == 0) {
            const int c = detect_rotation(gran0 + (size_t)2 * a.gran_stride + 256, G, g, lane, a.epoch0);
            if (lane == 0) rot_c = c;
        }
        __syncthreads();
        if (rot_c >= 0) g = (g & ~7) | ((g + rot_c) & 7);
    }
#endif

    const LdsPlan L = lds_plan(K, (int)sizeof(T), RP, a.panels_per_group, MODE == MODE_LDS, 0, CPP,
                               LANEWISE ? CPP : 1);
    double* u_lds = reinterpret_cast<double*>(smem + L.u);   // [CPP][kpad]
    double* red = reinterpret_cast<double*>(smem + L.red);   // [CPP][8] or [CPP][8][64]
    double* ctl = reinterpret_cast<double*>(smem + L.ctl);   // [1] abort, [2] local
    const int kpad = (K + 63) & ~63;
    for (int j = tid; j < kpad * CPP; j += blockDim.x) u_lds[j] = 0.0;
    // rows / slots of absent waves stay 0
    for (int j = tid; j < RED_DOUBLES * (LANEWISE ? CPP : 1); j += blockDim.x) red[j] = 0.0;
    if (tid == 0) { ctl[1] = 0.0; ctl[2] = 0.0; }

    PanelStore<T, VEC, MODE, KMAX, PPW> store;
    store.init(a.P, G, g, reinterpret_cast<T*>(smem + L.x), reinterpret_cast<T*>(smem + L.y),
               LANEWISE && !BAL, BAL);
    // Which (panel, chains) this wave computes.  Lane-wise form (one panel per wave): wave w < npl
    // owns local panel w; the waves past the last panel hold a copy of it and split its CPP chains
    // with its owner -- C2: 5 panels on a CU's 4 SIMDs, 8 waves: 8 + 2 (panel, chain) units per
    // SIMD instead of 16 on the SIMD that carries panels 0 and 4.
    int c_lo = 0, c_hi = CPP, row = wave;
    if constexpr (LANEWISE) {
        const int npl = store.npl;
        if (npl == 0) {
            c_hi = 0;
        } else if (wave >= npl - 1) {
            const int sharers = nw - npl + 1, r = wave - (npl - 1);
            c_lo = r * CPP / sharers;
            c_hi = (r + 1) * CPP / sharers;
            row = npl - 1;
        }
    }

    const size_t chain_stride = (size_t)3 * a.gran_stride;
    if (wave == 0) {
        const int place = detect_placement(gran0 + 2 * a.gran_stride, G, g, lane, a.epoch0);
        if (lane == 0) {
            if (place < 0) { ctl[1] = 1.0; a.status[chain0] = 1; }
            ctl[2] = (place == 1 && !a.force_agent_scope) ? 1.0 : 0.0;
        }
    }
    __syncthreads();
    const bool local = ctl[2] != 0.0;
    if (g == 0 && tid < CPP) a.placement[chain0 + tid] = local ? 1 : 0;

    // (a slotted bundle of 8 chains runs 8 waves -- the host sees to it -- so every wave leads)
    const bool leader = (SLOTTED && CPP == 8) ? true : wave < CPP;
    const int chain_l = leader ? wave : 0;            // within the bundle
    const int chain = chain0 + chain_l;               // within the launch
    const double* xi = a.xi + (int64_t)chain * T_it * K;
    const double* gam = a.gam + (int64_t)chain * T_it;
    double* uout = a.uout + (int64_t)chain * T_it * (K + 1);
    double* u_mine = u_lds + (size_t)chain_l * kpad;

    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): see gibbs_loop_kernel
    double sp_eff = a.sigma2_init, g_eff = 1.0, sq_sp = BMC_SQRT_OF(a.sigma2_init);
    double xi_next[KCH], lam_r[KCH], c1_r[KCH], c2_r[KCH];
    double gam_next = 0.0;
    if (leader) {
#pragma unroll
        for (int ch = 0; ch < KCH; ++ch) {
            const int j = ch * 64 + lane;
            xi_next[ch] = (j < K && T_it > 0) ? xi[j] : 0.0;
            lam_r[ch] = j < K ? a.lam[j] : 0.0;
            c1_r[ch] = j < K ? a.c1[j] : 0.0;
            c2_r[ch] = j < K ? a.c2[j] : 0.0;
        }
        if (T_it > 0) gam_next = gam[0];
    }

#ifdef BMC_STAMPS
    const bool stamping = a.dbg != nullptr && blockIdx.x == 0 && wave == 0;
    unsigned long long acc_[12] = {}, last_ = 0;
    if (stamping) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last_)::"memory");
#endif
    for (int64_t t = 0; t < T_it; ++t) {
        const unsigned epoch = (unsigned)(t + 1) + a.epoch0;
        GSTAMP(7);
        double u_rec[KCH];
        const double sp_rec = sp_eff, g_rec = g_eff;   // sigma2 of the previous row
        if (leader) {
#pragma unroll
            for (int ch = 0; ch < KCH; ++ch) {
                const int j = ch * 64 + lane;
                u_rec[ch] = 0.0;
                if (ch * 64 < K && j < K) {
                    u_rec[ch] = draw_u(lam_r[ch], c1_r[ch], c2_r[ch], xi_next[ch], sp_eff, g_eff, sq_sp);
                    u_mine[j] = u_rec[ch];
                }
            }
        }
        GSTAMP(0);
        __syncthreads();  // B1
        GSTAMP(1);
        const double abort_w = ctl[1];   // tested after the residual pass (see gibbs_loop_kernel)
        const double gam_t = gam_next;
        double s[CPP];
#pragma unroll
        for (int c = 0; c < CPP; ++c) s[c] = 0.0;
        if constexpr (BAL) {
            // (rows of the lane-wise sum are indexed by PANEL: set 0 = panel wave % 4, chains
            // 4 (wave / 4) .. + 3; set 1 = panel 4, chain `wave`)
            double sA[4], sB;
            store.partial_rss_reg_bal(u_lds, kpad, sA, sB);
            const int npl = store.npl, pa = wave & 3, half = wave >> 2;
            if (pa < npl) {
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) red[((4 * half + cc) * 8 + pa) * 64 + lane] = sA[cc];
            }
            if (npl > 4) red[(wave * 8 + 4) * 64 + lane] = sB;
            s[CPP - 1] = sB;   // (ties the abort test below to the end of the pass)
        } else if constexpr (MODE == MODE_REG) {
            store.template partial_rss_reg_multi<CPP>(u_lds, kpad, s, c_lo, c_hi);
        }
        for (int q = store.wave; MODE != MODE_REG && q < store.npl; q += store.nw) {
            if constexpr (MODE == MODE_LDS) {
                panel_rss_multi<T, VEC, CPP>(store.Xs + (size_t)q * K * RP + lane * VEC,
                                             store.ys + q * RP + lane * VEC, u_lds, kpad, K, s);
            } else if constexpr (MODE == MODE_STREAM) {
                const int64_t p = g + (int64_t)q * G;
                if (q < store.keep)
                    panel_rss_multi<T, VEC, CPP>(store.Xg + p * (int64_t)K * RP + lane * VEC,
                                                 store.yg + p * RP + lane * VEC, u_lds, kpad, K, s);
                else
                    panel_rss_multi<T, VEC, CPP, true>(store.Xg + p * (int64_t)K * RP + lane * VEC,
                                                       store.yg + p * RP + lane * VEC, u_lds, kpad, K, s);
            }
        }
        GSTAMP(2);
        {
            double abort_late = abort_w;
            asm volatile("" : "+v"(abort_late) : "v"(s[CPP - 1]));
            if (__builtin_amdgcn_readfirstlane(__double2hiint(abort_late)) != 0) break;
        }
        // next iteration's variates, behind the residual pass (see gibbs_loop_kernel).  Loaded by
        // every wave, at clamped indices, with no branch around the loads: under `if (leader &&
        // t + 1 < T)` hipcc loads into a temporary and copies it into the loop-carried register
        // at the join, with an s_waitcnt vmcnt(0) right here -- a global-load latency on every
        // wave's way to the group barrier (stamps, 64 chains at C2: 630 cycles between the pass
        // and the barrier).  A wave that leads no chain reads the first chain's variates and
        // never uses them.
        {
            const int64_t tn = t + 1 < T_it ? t + 1 : t;
#pragma unroll
            for (int ch = 0; ch < KCH; ++ch) {
                const int j = ch * 64 + lane;
                xi_next[ch] = xi[tn * K + (j < K ? j : K - 1)];
            }
            gam_next = gam[tn];
        }
        bool got;
        // group 0's leaders record their chain (row t, and sigma of the previous row) between
        // publishing and polling: the stamps showed this work, placed after the exchange, holding
        // back group 0 -- and with it every group -- by ~1500 cycles per pass
        // (round 3, tried: the sqrt(sp / g) deferred to 64 roots at a time, as in gibbs_wave_kernel --
        // slower here, 64 chains at C2 2.110 -> 2.146 us, N = 100 000 x 8 chains 5.90 -> 6.04: this
        // slot is already hidden behind the exchange, and the captures add to every iteration)
        auto record = [&]() {
            if (g == 0) {
#pragma unroll
                for (int ch = 0; ch < KCH; ++ch) {
                    const int j = ch * 64 + lane;
                    if (ch * 64 < K && j < K) uout[t * (K + 1) + j] = u_rec[ch];
                }
                if (lane == 0 && t > 0) uout[(t - 1) * (K + 1) + K] = sqrt(sp_rec / g_rec);
            }
        };
        double rss;
        if constexpr (BAL)
            rss = group_allreduce_multi_prewritten<CPP, (SLOTTED ? 0 : -1)>(
                red, gran0 + (size_t)(t & 1) * a.gran_stride, chain_stride, G, g, wave, lane, epoch,
                local, got STAMP_ARGS, record);
        else if constexpr (LANEWISE)
            rss = group_allreduce_multi_lanewise<CPP, (SLOTTED ? 0 : -1)>(
                s, red, row, c_lo, c_hi, gran0 + (size_t)(t & 1) * a.gran_stride, chain_stride, G, g,
                wave, lane, epoch, local, got STAMP_ARGS, record);
        else
            rss = group_allreduce_multi<CPP>(s, red, gran0 + (size_t)(t & 1) * a.gran_stride,
                                             chain_stride, G, g, wave, nw, lane, epoch, local,
                                             got STAMP_ARGS, record);
        if (leader) {
            if (__builtin_expect(!got, 0)) {
                if (lane == 0) { ctl[1] = 1.0; a.status[chain] = 1; }
            } else {
                const double scale_post = (a.nu0_s20 + rss) * 0.5;
                const bool floor_hit = scale_post < 1e-6 * gam_t;
                sp_eff = floor_hit ? 1e-6 : scale_post;
                g_eff = floor_hit ? 1.0 : gam_t;
                sq_sp = BMC_SQRT_OF(sp_eff);
            }
        }
        GSTAMP(6);
    }
#ifdef BMC_STAMPS
    if (stamping && lane == 0)
        for (int i = 0; i < 12; ++i) a.dbg[i] = (long long)acc_[i];
#endif
    __syncthreads();
    if (leader && g == 0 && lane == 0 && T_it > 0 && ctl[1] == 0.0)
        uout[(T_it - 1) * (K + 1) + K] = sqrt(sp_eff / g_eff);
}

// ======================================================================================
// Opt-in (bmc_tuning.rss_mode = 1, K <= 64): the same chain with rss taken from sufficient
// statistics instead of a pass over the data.  For any centre u0,
//     |y - X~u|^2 = |y - X~u0|^2 - 2 d'X~'(y - X~u0) + d'(X~'X~)d,      d = u - u0,
// an identity, so with rss(u0) from one residual pass, g0 = X~'(y - X~u0) and G = X~'X~ fixed,
// an iteration costs K^2 FMAs and touches no data.  u0 is the least-squares point, where
// rss(u0) is smallest and both other terms vanish to first order, so nothing cancels: every
// term is >= 0 or tiny.  One wave per chain (lane j = component j, row j of G in registers),
// no barriers, no exchange; any number of chains per launch.  The draw u is computed with the
// operations of gibbs_loop_kernel, so a chain differs from the data-pass chain only through
// the rounding of rss.
// ======================================================================================
template <int KMAX>
__global__ __launch_bounds__(64) void gibbs_gram_kernel(GramArgs a) {
    __shared__ double d_lds[64];
    // 64 output rows [u_t, sigma_t] staged here and written out together (see gibbs_wave_kernel:
    // the correctly rounded sqrt(sp / g) of every row and the stores leave the loop)
    __shared__ double rows[64 * (KMAX + 1)];
    const int lane = threadIdx.x, K = a.k;
    const int chain = blockIdx.x;
    if (chain >= a.n_chains) return;
    const int64_t T_it = a.iters;
    const double* xi = a.xi + (int64_t)chain * T_it * K;
    const double* gam = a.gam + (int64_t)chain * T_it;
    double* uout = a.uout + (int64_t)chain * T_it * (K + 1);
    const bool act = lane < K;
    double grow[KMAX];   // row `lane` of G
#pragma unroll
    for (int i = 0; i < KMAX; ++i) grow[i] = (act && i < K) ? a.Gt[(size_t)lane * K + i] : 0.0;
    const double lam = act ? a.lam[lane] : 0.0, c1 = act ? a.c1[lane] : 0.0;
    const double c2 = act ? a.c2[lane] : 0.0, u0 = act ? a.u0[lane] : 0.0;
    const double g0x2 = act ? 2.0 * a.g0[lane] : 0.0;
    d_lds[lane] = 0.0;
    double sp_eff = a.sigma2_init, g_eff = 1.0, sq_sp = BMC_SQRT_OF(a.sigma2_init);
    double sp_cap = 1.0, g_cap = 1.0;   // lane i: the (sp, g) pair behind staged row i
    double xi_next = (act && T_it > 0) ? xi[lane] : 0.0;
    double gam_next = T_it > 0 ? gam[0] : 1.0;
    const int K1 = K + 1;
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): see gibbs_wave_kernel
    for (int64_t t = 0; t < T_it; ++t) {
        const int slot = (int)(t & 63);
        // u | sigma2, the operations of gibbs_loop_kernel (lanes K .. 63: all inputs 0, u = 0)
        const double u = draw_u(lam, c1, c2, xi_next, sp_eff, g_eff, sq_sp);
        const double gam_t = gam_next;
        {
            const int64_t tn = t + 1 < T_it ? t + 1 : t;
            xi_next = act ? xi[tn * K + lane] : 0.0;
            gam_next = gam[tn];
        }
        if (act) rows[slot * K1 + lane] = u;
        // d -> every lane (same wave: the LDS executes its writes and reads in order)
        const double d = u - u0;
        d_lds[lane] = d;
        double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
#pragma unroll
        for (int i = 0; i < KMAX; i += 4) {
            acc0 = fma(grow[i], d_lds[i], acc0);
            acc1 = fma(grow[i + 1], d_lds[i + 1], acc1);
            acc2 = fma(grow[i + 2], d_lds[i + 2], acc2);
            acc3 = fma(grow[i + 3], d_lds[i + 3], acc3);
        }
        const double gd = (acc0 + acc1) + (acc2 + acc3);
        const double q = wave_sum(d * (gd - g0x2));
        double rss = a.rss0 + q;
        rss = rss > 0.0 ? rss : 0.0;
        // sigma2 | beta = scale_post / g_t, floored at 1e-6            (:50-52)
        const double scale_post = (a.nu0_s20 + rss) * 0.5;
        const bool floor_hit = scale_post < 1e-6 * gam_t;
        sp_eff = floor_hit ? 1e-6 : scale_post;
        g_eff = floor_hit ? 1.0 : gam_t;
        sq_sp = BMC_SQRT_OF(sp_eff);
        const bool mine = lane == slot;
        sp_cap = mine ? sp_eff : sp_cap;
        g_cap = mine ? g_eff : g_cap;
        if (slot == 63 || t + 1 == T_it) {
            const int nrows = slot + 1;
            const double sig = sqrt(sp_cap / g_cap);
            if (lane < nrows) rows[lane * K1 + K] = sig;
            double* dst = uout + (t - slot) * K1;
            for (int idx = lane; idx < nrows * K1; idx += 64) dst[idx] = rows[idx];
        }
    }
}

// ======================================================================================
// A chain small enough for ONE wave (the reference's own data set: 629 rows, 3 kept components;
// SURVEY 8, configuration C1): lane l keeps rows l, 64 + l, 128 + l, .. of the rotated matrix in
// registers, RMAX rows x KMAX columns, and the whole iteration runs in that wave -- u goes from
// the lanes that draw it to the FMAs through v_readlane (SGPR operands), the residual sum is
// one wave_sum; no LDS, no barrier, no exchange, nothing to wait for.  The workgroup form of
// the same chain (gibbs_loop_kernel<.., SINGLE>: 5 waves, two barriers, u and the lane
// partials through LDS) spends ~1340 cycles per iteration at N = 629, K = 3, nearly all of it
// hand-over latency.  Any number of chains per launch, one wave each.
// The draw and the sigma2 step are those of gibbs_loop_kernel; per row the residual is the
// chain acc = y, acc = fma(-x_j, u_j, acc), j ascending, rows summed into two accumulators.
// ======================================================================================
template <typename T, int RMAX, int KMAX, int NWMAX = 1>
__global__ __launch_bounds__(64 * NWMAX) void gibbs_wave_kernel(GibbsArgs a) {
    constexpr bool MANY = NWMAX > 1;   // 4: 2 or 4 waves (512 registers each); 8: 8 waves (256 each)
    // 64 output rows [u_t, sigma_t] staged here and written out together (below)
    __shared__ double rows[64 * (KMAX + 1)];
    __shared__ double wsum[2][8];   // (2 .. 8 waves per chain: the waves' totals, two parities)
    const int lane = threadIdx.x & 63, K = a.P.k, NP = a.P.npanels;
    // Up to 4 waves per chain (a.waves; one per SIMD, so each still has 512 registers): wave w
    // keeps panels [w rpw, (w + 1) rpw) and EVERY wave runs the whole iteration -- the draw, the
    // sigma2 step -- on identical inputs; what they exchange is one double per wave and
    // iteration, through LDS, with one barrier.  N = 2500, K = 8: 4 x 12 x 8 registers.
    // (MANY is a template parameter: with the test at run time the one-wave chain of the
    // reference's size went from 0.351 to 0.381 us per iteration)
    const int nw = MANY ? (int)(blockDim.x >> 6) : 1;
    const int wave = MANY ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;
    const int rpw = (NP + nw - 1) / nw, p0 = wave * rpw;
    const int chain = blockIdx.x;
    if (chain >= a.n_chains) return;
    const int64_t T_it = a.iters;
    const double* xi = a.xi + (int64_t)chain * T_it * K;
    const double* gam = a.gam + (int64_t)chain * T_it;
    double* uout = a.uout + (int64_t)chain * T_it * (K + 1);
    const T* Xp = reinterpret_cast<const T*>(a.P.X);
    const T* yp = reinterpret_cast<const T*>(a.P.y);
    double x[RMAX][KMAX], y[RMAX];
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
        const bool have = r < rpw && p0 + r < NP;
#pragma unroll
        for (int j = 0; j < KMAX; ++j)
            x[r][j] = (have && j < K) ? (double)Xp[((size_t)(p0 + r) * K + j) * 64 + lane] : 0.0;
        y[r] = have ? (double)yp[(size_t)(p0 + r) * 64 + lane] : 0.0;
    }
    const bool act = lane < K;
    // (lanes K .. 63: lam = c1 = c2 = xi = 0 draw u = 0 -- no exec-mask region around the draw)
    const double lam = act ? a.lam[lane] : 0.0, c1 = act ? a.c1[lane] : 0.0;
    const double c2 = act ? a.c2[lane] : 0.0;
    if (threadIdx.x == 0) a.placement[chain] = 1;
    const bool rec = wave == 0;          // the wave that records the draws
    if constexpr (MANY) {
        if (threadIdx.x < 16) wsum[threadIdx.x >> 3][threadIdx.x & 7] = 0.0;   // absent waves stay 0
        __syncthreads();
    }
    double sp_eff = a.sigma2_init, g_eff = 1.0;
    double sp_cap = 1.0, g_cap = 1.0;   // lane i: the (sp, g) pair behind staged row i
    double xi_next = (act && T_it > 0) ? xi[lane] : 0.0;
    double gam_next = T_it > 0 ? gam[0] : 1.0;
    const int K1 = K + 1;
    // every load issued so far is complete from here on, and hipcc knows it: merged with the
    // loop's own state its wait in front of the first FMA was vmcnt(1) -- behind the prefetch
    // just issued, a memory latency per iteration (see gibbs_loop_kernel)
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    for (int64_t t = 0; t < T_it; ++t) {
        const int slot = (int)(t & 63);
        const double u = draw_u(lam, c1, c2, xi_next, sp_eff, g_eff, 0.0);
        const double gam_t = gam_next;
        {   // next iteration's variates (clamped index: no branch around the loads)
            const int64_t tn = t + 1 < T_it ? t + 1 : t;
            xi_next = act ? xi[tn * K + lane] : 0.0;
            gam_next = gam[tn];
        }
        if (act && rec) rows[slot * K1 + lane] = u;
        double uj[KMAX];
#pragma unroll
        for (int j = 0; j < KMAX; ++j) uj[j] = readlane_f64(u, j);
        // (round 3, tried and not kept: columns outside / rows inside, so that the rows' chains of K
        // dependent FMAs interleave instead of running one after the other as hipcc schedules
        // this form -- 12 x 4 registers 0.353 -> 0.349 us, but 12 x 8 0.469 -> 0.488 and the
        // 4-wave form of it 0.628 -> 0.680; four chains of squares instead of two: 0.349 -> 0.353;
        // a wave-uniform branch for the floor instead of the two selects: 0.349 -> 0.405)
        double part0 = 0.0, part1 = 0.0;
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            double acc = y[r];
#pragma unroll
            for (int j = 0; j < KMAX; ++j) acc = fma(-x[r][j], uj[j], acc);
            if (r & 1) part1 = fma(acc, acc, part1);
            else part0 = fma(acc, acc, part0);
        }
        double rss = wave_sum(part0 + part1);
        if constexpr (MANY) {
            // the waves' totals, in wave order; parity t & 1: a wave that is already in iteration
            // t + 1 writes the other set while a slower one still reads this one
            double* ws = wsum[t & 1];
            if (lane == 0) ws[wave] = rss;
            __syncthreads();
            rss = (ws[0] + ws[1]) + (ws[2] + ws[3]);
            if constexpr (NWMAX > 4) rss += (ws[4] + ws[5]) + (ws[6] + ws[7]);
        }
        // sigma2 | beta = scale_post / g_t, floored at 1e-6            (:50-52)
        const double scale_post = (a.nu0_s20 + rss) * 0.5;
        const bool floor_hit = scale_post < 1e-6 * gam_t;
        sp_eff = floor_hit ? 1e-6 : scale_post;
        g_eff = floor_hit ? 1.0 : gam_t;
        // The recorded sigma_t = sqrt(sp / g) is a correctly rounded division and square root,
        // ~35 dependent f64 operations that nothing in this wave can hide.  Lane (t mod 64)
        // keeps the pair instead, and every 64 iterations all lanes take their roots at once
        // and the 64 staged rows go out as contiguous stores -- the loop itself then has no
        // store in flight, so its waits for the prefetched variates are exact.  (First version,
        // sqrt and stores in the loop: 0.49 us per iteration at N = 629, K = 3.)
        const bool mine = lane == slot;
        sp_cap = mine ? sp_eff : sp_cap;
        g_cap = mine ? g_eff : g_cap;
        if (rec && (slot == 63 || t + 1 == T_it)) {
            const int nrows = slot + 1;
            const double sig = sqrt(sp_cap / g_cap);
            if (lane < nrows) rows[lane * K1 + K] = sig;
            double* dst = uout + (t - slot) * K1;
            for (int idx = lane; idx < nrows * K1; idx += 64) dst[idx] = rows[idx];
        }
    }
}

hipError_t launch_gibbs_gram(const GramArgs& a, hipStream_t s) {
    if (a.k < 1 || a.k > 64 || a.n_chains < 1) return hipErrorInvalidValue;
    const dim3 grid((unsigned)a.n_chains), block(64);
    if (a.k <= 8) hipLaunchKernelGGL(gibbs_gram_kernel<8>, grid, block, 0, s, a);
    else if (a.k <= 16) hipLaunchKernelGGL(gibbs_gram_kernel<16>, grid, block, 0, s, a);
    else if (a.k <= 32) hipLaunchKernelGGL(gibbs_gram_kernel<32>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(gibbs_gram_kernel<64>, grid, block, 0, s, a);
    return hipGetLastError();
}

// ======================================================================================
// Simplex-constrained sampler (reference pybmc/inference_utils.py:78-144): random-walk
// Metropolis on beta with the weights beta Vt_hat + 1/Km kept on the simplex, Gibbs step
// for sigma2.  Same machinery: the proposal's rss is the group all-reduce above.  A
// proposal outside the simplex skips the residual pass AND consumes no uniform (:102,
// :124): every group evaluates the simplex test on identical bits, so all skip together;
// the exchange epoch counts exchanges, not iterations.
// ======================================================================================
// SINGLE: the chain lives in ONE workgroup (G == 1, register residency: the reference's own
// sizes): nothing to exchange -- as a run-time test it cost the multi-group shapes 2 %.
template <typename T, int VEC, int MODE, int KMAX, int PPW, bool SINGLE = false>
__global__ __launch_bounds__(512) void simplex_loop_kernel(SimplexArgs a) {
    constexpr int RP = 64 * VEC;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int K = a.P.k, Km = a.Km, G = a.G;
    const int chain = blockIdx.x % a.nslot;
    const int g = blockIdx.x / a.nslot;
    if (chain >= 1) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = blockDim.x >> 6;
    const int64_t T_tot = a.burn + a.iters;

    const int aux_n = a.vt_in_lds ? K * Km : 0;
    const LdsPlan L = lds_plan(K, (int)sizeof(T), RP, a.panels_per_group, MODE == MODE_LDS, aux_n);
    double* u_lds = reinterpret_cast<double*>(smem + L.u);
    double* red = reinterpret_cast<double*>(smem + L.red);
    double* ctl = reinterpret_cast<double*>(smem + L.ctl);  // [1] abort, [2] local, [4] inside
    double* vt_lds = reinterpret_cast<double*>(smem + L.aux);
    const double* vt = a.vt_in_lds ? vt_lds : a.Vt;

    const int kpad = (K + 63) & ~63;
    for (int j = tid; j < kpad; j += blockDim.x) u_lds[j] = 0.0;
    for (int j = tid; j < RED_DOUBLES; j += blockDim.x) red[j] = 0.0;   // rows / slots of absent waves stay 0
    if (a.vt_in_lds)
        for (int e = tid; e < K * Km; e += blockDim.x) vt_lds[e] = a.Vt[e];
    if (tid == 0) { ctl[1] = 0.0; ctl[2] = 0.0; ctl[4] = 0.0; }

    PanelStore<T, VEC, MODE, KMAX, PPW> store;
    store.init(a.P, G, g, reinterpret_cast<T*>(smem + L.x), reinterpret_cast<T*>(smem + L.y));

    gu64* gr = a.gran;
    if constexpr (!SINGLE) {
        if (wave == 0) {
            const int place = detect_placement(gr + 2 * a.gran_stride, G, g, lane, a.epoch0);
            if (lane == 0) {
                if (place < 0) { ctl[1] = 1.0; a.status[0] = 1; }
                ctl[2] = (place == 1 && !a.force_agent_scope) ? 1.0 : 0.0;
            }
        }
    }
    __syncthreads();
    const bool local = ctl[2] != 0.0;
    if (g == 0 && tid == 0) a.placement[0] = (local || SINGLE) ? 1 : 0;

    // lane-chunks of 64 columns: one, known at compile time, in register mode (K <= 64)
    constexpr int KCH = (MODE == MODE_REG) ? 1 : MAX_KCH;
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): see gibbs_loop_kernel
    double b_cur[KCH], b_prop[KCH], step_r[KCH], xi_next[KCH];
    double rss_cur = a.rss_init;                       // -log_likelihood_current (:85)
    double s2 = a.rss_init / (double)a.P.n;            // :86
    double gam_next = 0.0, unif_next = 0.5;
    int64_t iu = 0, accepted = 0;
    unsigned nex = 0;                                   // exchanges so far
    const double w0 = 1.0 / (double)Km;
    if (wave == 0) {
#pragma unroll
        for (int ch = 0; ch < KCH; ++ch) {
            const int j = ch * 64 + lane;
            b_cur[ch] = 0.0;                            // :82
            b_prop[ch] = 0.0;
            step_r[ch] = j < K ? a.step[j] : 0.0;
            xi_next[ch] = (j < K && T_tot > 0) ? a.xi[j] : 0.0;
        }
        if (T_tot > 0) gam_next = a.gam[0];
        if (a.n_unif > 0) unif_next = a.unif[0];
    }

#if BMC_LEADER_PRIO > 0
    if (wave == 0) __builtin_amdgcn_s_setprio(BMC_LEADER_PRIO);   // (the leader, as in gibbs_loop_kernel)
#endif
    for (int64_t t = 0; t < T_tot; ++t) {
        if (wave == 0) {
            // proposal b_cur + diag(S_hat stepsize) xi  (:98,:121: mvn with a diagonal cov)
#pragma unroll
            for (int ch = 0; ch < KCH; ++ch) {
                const int j = ch * 64 + lane;
                if (ch * 64 < K && j < K) {
                    b_prop[ch] = fma(step_r[ch], xi_next[ch], b_cur[ch]);
                    u_lds[j] = b_prop[ch];
                }
            }
            __builtin_amdgcn_wave_barrier();
            // omegas = b_prop Vt_hat + 1/Km >= 0 ?                          (:99-102)
            bool neg = false;
            for (int m = lane; m < Km; m += 64) {
                double o0 = w0, o1 = 0.0, o2 = 0.0, o3 = 0.0;
                int j = 0;
                for (; j + 3 < K; j += 4) {
                    o0 = fma(u_lds[j], vt[(size_t)j * Km + m], o0);
                    o1 = fma(u_lds[j + 1], vt[(size_t)(j + 1) * Km + m], o1);
                    o2 = fma(u_lds[j + 2], vt[(size_t)(j + 2) * Km + m], o2);
                    o3 = fma(u_lds[j + 3], vt[(size_t)(j + 3) * Km + m], o3);
                }
                for (; j < K; ++j) o0 = fma(u_lds[j], vt[(size_t)j * Km + m], o0);
                neg = neg || ((o0 + o1) + (o2 + o3) < 0.0);
            }
            const bool inside = !__any(neg);
            if (lane == 0) ctl[4] = inside ? 1.0 : 0.0;
        }
        __syncthreads();  // B1
        if (ctl[1] != 0.0) break;
        const bool inside = ctl[4] != 0.0;

        const double gam_t = gam_next;
        // next iteration's variates (the FMAs of the residual pass do not wait for these loads:
        // the s_waitcnt in front of the loop, see gibbs_loop_kernel)
        if (wave == 0 && t + 1 < T_tot) {
#pragma unroll
            for (int ch = 0; ch < KCH; ++ch) {
                const int j = ch * 64 + lane;
                if (ch * 64 < K && j < K) xi_next[ch] = a.xi[(t + 1) * K + j];
            }
            gam_next = a.gam[t + 1];
        }

        if (inside) {
            const double part = store.partial_rss(u_lds);
            bool got;
#ifdef BMC_STAMPS
            bool stamping = false;
            unsigned long long acc_[12] = {}, last_ = 0;
#endif
            // (a chain in ONE workgroup has nothing to exchange: without SINGLE it published its
            // total and polled it back through L2 -- N = 629: 1.55 -> 1.32 us per step)
            const double rss_prop = group_allreduce<SINGLE, (MODE == MODE_REG && VEC == 1)>(
                part, red, gr + (size_t)(nex & 1) * a.gran_stride, G, g, wave, nw, lane,
                nex + 1 + a.epoch0, local, got STAMP_ARGS);
            ++nex;
            if (wave == 0) {
                if (!got || iu >= a.n_unif) {
                    if (lane == 0) { ctl[1] = 1.0; a.status[0] = got ? 2 : 1; }
                } else {
                    // min(1, exp((ll_prop - ll_cur) / sigma2)), ll = -rss     (:106-109)
                    const double ratio = exp((rss_cur - rss_prop) / s2);
                    const double p_acc = ratio < 1.0 ? ratio : 1.0;
                    const double uu = unif_next;
                    ++iu;
                    if (iu < a.n_unif) unif_next = a.unif[iu];
                    if (uu < p_acc) {                                          // :110-112
#pragma unroll
                        for (int ch = 0; ch < KCH; ++ch) b_cur[ch] = b_prop[ch];
                        rss_cur = rss_prop;
                        if (t >= a.burn) ++accepted;
                    }
                }
            }
        }
        if (wave == 0) {
            // sigma2 = 1 / Gamma(shape, 1/scale_post), no floor on this path      (:115-117)
            const double scale_post = (a.nu0_s20 + rss_cur) * 0.5;
            s2 = scale_post / gam_t;
            if (g == 0 && t >= a.burn) {
                double* row = a.out + (t - a.burn) * (K + 1);
#pragma unroll
                for (int ch = 0; ch < KCH; ++ch) {
                    const int j = ch * 64 + lane;
                    if (ch * 64 < K && j < K) row[j] = b_cur[ch];
                }
                if (lane == 0) row[K] = sqrt(s2);
            }
        }
    }
    if (g == 0 && tid == 0) {
        a.counters[0] = accepted;
        a.counters[1] = iu;
    }
}

size_t gibbs_lds_bytes(const GibbsArgs& a) {
    const int cpp = a.chains_per_pass > 1 ? a.chains_per_pass : 1;
    // (several chains per pass in register residency with one row per lane: lane-wise group sum)
    const bool lanewise_multi = cpp > 1 && a.mode == MODE_REG && a.P.vec == 1;
    return lds_plan(a.P.k, a.P.f32 ? 4 : 8, 64 * a.P.vec, a.panels_per_group, a.mode == MODE_LDS, 0,
                    cpp, lanewise_multi ? cpp : 1).total;
}
size_t simplex_lds_bytes(const SimplexArgs& a) {
    return lds_plan(a.P.k, a.P.f32 ? 4 : 8, 64 * a.P.vec, a.panels_per_group, a.mode == MODE_LDS,
                    a.vt_in_lds ? a.P.k * a.Km : 0).total;
}

// ---- dispatch over <T, VEC, MODE, KMAX, PPW>: one table for both kernels ----------------
struct GibbsTag {};
struct SimplexTag {};

// Launch `fn` -- or, when `occ` is set, launch nothing and report how many workgroups of this
// block size and LDS footprint one CU admits (the persistent kernels spin on each other, so the
// host checks residency before it launches: bmc_capi.hip, check_residency).
template <typename Args>
static hipError_t launch_or_query(const void* fn, dim3 grid, dim3 block, size_t lds, hipStream_t s,
                                  const Args& a, int32_t* occ) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (occ) {
        int nb = 0;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, (int)block.x, lds);
        *occ = nb;
        return e;
    }
    Args copy = a;
    void* params[] = {&copy};
    e = hipLaunchKernel(fn, grid, block, params, lds, s);
    return e != hipSuccess ? e : hipGetLastError();
}

template <typename T, int VEC, int MODE, int KMAX, int PPW>
static hipError_t launch_one(GibbsTag, const GibbsArgs& a, hipStream_t s) {
    const size_t lds = gibbs_lds_bytes(a);
    if (a.query_regs) {   // report the packed variant's VGPR count (0: none), do not launch
        *a.query_regs = 0;
        if constexpr (loop_can_pack<T, VEC, MODE, KMAX, PPW>()) {
            hipFuncAttributes at;
            const hipError_t e = hipFuncGetAttributes(
                &at, (const void*)gibbs_loop_kernel<T, VEC, MODE, KMAX, PPW, false, true, true>);
            if (e != hipSuccess) return e;
            *a.query_regs = at.numRegs;
        }
        return hipSuccess;
    }
    const dim3 grid(a.nslot * a.G), block(64 * a.waves);
    if constexpr (loop_can_pack<T, VEC, MODE, KMAX, PPW>()) {
        if (a.pack && a.G > 1)   // (two chains per XCD implies G <= 32)
            return launch_or_query(
                (const void*)gibbs_loop_kernel<T, VEC, MODE, KMAX, PPW, false, true, true>, grid, block,
                lds, s, a, a.query_occupancy);
    }
    if constexpr (MODE == MODE_REG) {
        if (a.G == 1)   // the chain fits one workgroup: no exchange code at all
            return launch_or_query((const void*)gibbs_loop_kernel<T, VEC, MODE, KMAX, PPW, true>,
                                   grid, block, lds, s, a, a.query_occupancy);
        if (a.G <= 32)  // one-level exchange, known at compile time (SMALLG)
            return launch_or_query(
                (const void*)gibbs_loop_kernel<T, VEC, MODE, KMAX, PPW, false, false, true>, grid, block,
                lds, s, a, a.query_occupancy);
    }
    return launch_or_query((const void*)gibbs_loop_kernel<T, VEC, MODE, KMAX, PPW>, grid, block,
                           lds, s, a, a.query_occupancy);
}
template <typename T, int VEC, int MODE, int KMAX, int PPW>
static hipError_t launch_one(SimplexTag, const SimplexArgs& a, hipStream_t s) {
    if constexpr (MODE == MODE_REG) {
        if (a.G == 1)
            return launch_or_query((const void*)simplex_loop_kernel<T, VEC, MODE, KMAX, PPW, true>,
                                   dim3(a.nslot * a.G), dim3(64 * a.waves), simplex_lds_bytes(a), s, a,
                                   a.query_occupancy);
    }
    return launch_or_query((const void*)simplex_loop_kernel<T, VEC, MODE, KMAX, PPW>,
                           dim3(a.nslot * a.G), dim3(64 * a.waves), simplex_lds_bytes(a), s, a,
                           a.query_occupancy);
}

template <typename Tag, typename Args, typename T, int KMAX>
static hipError_t launch_reg(const Args& a, hipStream_t s) {
    // rows per lane (PPW * VEC) * KMAX * (registers per element) <= 128 data VGPRs keeps the
    // kernel spill-free; VEC = 2 panels are supported with one panel per wave
    if (a.P.vec == 2) {
        if constexpr (KMAX * sizeof(T) <= 256)
            if (a.reg_ppw == 1) return launch_one<T, 2, MODE_REG, KMAX, 1>(Tag{}, a, s);
        return hipErrorInvalidValue;
    }
    switch (a.reg_ppw) {
        case 1: return launch_one<T, 1, MODE_REG, KMAX, 1>(Tag{}, a, s);
        case 2:
            if constexpr (KMAX * sizeof(T) <= 256) return launch_one<T, 1, MODE_REG, KMAX, 2>(Tag{}, a, s);
            break;
        case 4:
            if constexpr (KMAX * sizeof(T) <= 128) return launch_one<T, 1, MODE_REG, KMAX, 4>(Tag{}, a, s);
            break;
    }
    return hipErrorInvalidValue;
}

template <typename Tag, typename Args, typename T>
static hipError_t launch_t(const Args& a, hipStream_t s) {
    if (a.mode == MODE_REG) {
        if (a.P.vec != 1 && a.P.vec != 2) return hipErrorInvalidValue;
        if (a.P.k <= 8) return launch_reg<Tag, Args, T, 8>(a, s);
        if (a.P.k <= 16) return launch_reg<Tag, Args, T, 16>(a, s);
        if (a.P.k <= 32) return launch_reg<Tag, Args, T, 32>(a, s);
        if (a.P.k <= 64) return launch_reg<Tag, Args, T, 64>(a, s);
        return hipErrorInvalidValue;
    }
#define BMC_MEM(V)                                                                   \
    (a.mode == MODE_LDS ? launch_one<T, V, MODE_LDS, 0, 0>(Tag{}, a, s)               \
                        : launch_one<T, V, MODE_STREAM, 0, 0>(Tag{}, a, s))
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

int gibbs_reg_capacity(int k, int f32, int rows_per_lane) {
    // data VGPRs per lane: f64 panels take two registers per element, f32 panels one;
    // rows_per_lane = panels per wave x rows per lane of a panel
    if (k > 64) return 0;
    const int kmax = k <= 8 ? 8 : k <= 16 ? 16 : k <= 32 ? 32 : 64;
    return rows_per_lane * kmax * (f32 ? 1 : 2) <= 128 ? 1 : 0;
}

template <typename Args>
static bool geometry_ok(const Args& a) {
    return a.P.k <= 64 * MAX_KCH && a.G <= MAX_GROUPS && a.G >= 1 && a.waves >= 1 &&
           a.waves <= 8 && a.nslot >= 1 && a.nslot <= 2048;
}

template <typename T, int VEC, int MODE>
static hipError_t launch_multi_cpp(const GibbsArgs& a, hipStream_t s) {
    const size_t lds = gibbs_lds_bytes(a);
#define BMC_MULTI(C)                                                                        \
    return launch_or_query((const void*)gibbs_multi_kernel<T, VEC, MODE, C>, dim3(a.G),     \
                           dim3(64 * a.waves), lds, s, a, a.query_occupancy)
    switch (a.chains_per_pass) {
        case 2: BMC_MULTI(2);
        case 4: BMC_MULTI(4);
        case 8: BMC_MULTI(8);
    }
#undef BMC_MULTI
    return hipErrorInvalidValue;
}

template <typename T>
static hipError_t launch_multi_t(const GibbsArgs& a, hipStream_t s) {
#define BMC_MM(V)                                                         \
    (a.mode == MODE_LDS ? launch_multi_cpp<T, V, MODE_LDS>(a, s)           \
                        : launch_multi_cpp<T, V, MODE_STREAM>(a, s))
    switch (a.P.vec) {
        case 1: return BMC_MM(1);
        case 2: return BMC_MM(2);
        case 4:
            if constexpr (sizeof(T) == 4) return BMC_MM(4);
            break;
    }
#undef BMC_MM
    return hipErrorInvalidValue;
}

// register residency with several chains per pass: one panel per wave (PPW = 1).  The panel
// (KMAX*VEC values) plus two blocks of u plus the leaders' state must fit 256 VGPRs: the most
// chains per pass that hipcc compiles without scratch, per (columns, storage type, rows per lane)
static constexpr int reg_multi_cap(int kmax, bool f32, int vec) {
    if (vec == 1) return 8;
    if (vec == 2) return kmax * (f32 ? 4 : 8) >= 256 ? 4 : 8;   // 128 VGPRs of panel: 4 chains
    return 0;
}
int gibbs_reg_multi_cap(int k, bool f32, int vec) {
    const int kmax = k <= 8 ? 8 : k <= 16 ? 16 : k <= 32 ? 32 : k <= 64 ? 64 : 0;
    return kmax ? reg_multi_cap(kmax, f32, vec) : 0;
}

// register residency with several chains per pass: one panel per wave (PPW = 1)
template <typename T, int VEC, int KMAX>
static hipError_t launch_multi_reg_k(const GibbsArgs& a, hipStream_t s) {
    const size_t lds = gibbs_lds_bytes(a);
#define BMC_MR(C)                                                                            \
    do {                                                                                         \
        if constexpr (VEC == 1 && C == 8 && KMAX >= 16 && KMAX * sizeof(T) <= 256) {             \
            if (a.bundle_slots > 0 && a.bundle_bal)                                              \
                return launch_or_query(                                                          \
                    (const void*)gibbs_multi_kernel<T, VEC, MODE_REG, C, KMAX, 2, true, true>,     \
                    dim3(a.bundle_slots * a.G), dim3(64 * a.waves), lds, s, a, a.query_occupancy); \
        }                                                                                        \
        if (a.bundle_bal) return hipErrorInvalidValue;                                           \
        if constexpr (VEC == 1) {                                                                \
            if (a.bundle_slots > 0)                                                              \
                return launch_or_query(                                                          \
                    (const void*)gibbs_multi_kernel<T, VEC, MODE_REG, C, KMAX, 1, true>,           \
                    dim3(a.bundle_slots * a.G), dim3(64 * a.waves), lds, s, a, a.query_occupancy); \
        }                                                                                        \
        if (a.bundle_slots > 0) return hipErrorInvalidValue;                                     \
        return launch_or_query((const void*)gibbs_multi_kernel<T, VEC, MODE_REG, C, KMAX, 1>,    \
                               dim3(a.G), dim3(64 * a.waves), lds, s, a, a.query_occupancy);     \
    } while (0)
    // only the combinations that fit the 256-VGPR budget without spilling are built
    constexpr int CMAX = reg_multi_cap(KMAX, sizeof(T) == 4, VEC);
    switch (a.chains_per_pass) {
        case 2: if constexpr (CMAX >= 2) BMC_MR(2); break;
        case 4: if constexpr (CMAX >= 4) BMC_MR(4); break;
        case 8: if constexpr (CMAX >= 8) BMC_MR(8); break;
    }
#undef BMC_MR
    return hipErrorInvalidValue;
}

template <typename T>
static hipError_t launch_multi_reg(const GibbsArgs& a, hipStream_t s) {
    if (a.reg_ppw != 1) return hipErrorInvalidValue;
    const int k = a.P.k;
    if (a.P.vec == 1) {
        if (k <= 8) return launch_multi_reg_k<T, 1, 8>(a, s);
        if (k <= 16) return launch_multi_reg_k<T, 1, 16>(a, s);
        if (k <= 32) return launch_multi_reg_k<T, 1, 32>(a, s);
        if (k <= 64) return launch_multi_reg_k<T, 1, 64>(a, s);
    } else if (a.P.vec == 2) {
        if (k <= 8) return launch_multi_reg_k<T, 2, 8>(a, s);
        if (k <= 16) return launch_multi_reg_k<T, 2, 16>(a, s);
        if (k <= 32) return launch_multi_reg_k<T, 2, 32>(a, s);
        if constexpr (sizeof(T) == 4)
            if (k <= 64) return launch_multi_reg_k<T, 2, 64>(a, s);
    }
    return hipErrorInvalidValue;
}

// The simplex-constrained sampler in the same one-wave form (simplex_loop_kernel's steps: the
// proposal, the simplex test of its weights, the residual sum of an inside proposal, the
// Metropolis test, the sigma2 draw), at most 64 models: lane m keeps column m of Vt_hat.  The
// kept rows [beta_t, sigma_t] are staged 64 at a time as in gibbs_wave_kernel.
template <typename T, int RMAX, int KMAX, bool MANY = false>
__global__ __launch_bounds__(MANY ? 256 : 64) void simplex_wave_kernel(SimplexArgs a) {
    __shared__ double rows[64 * (KMAX + 1)];
    __shared__ double wsum[2][4];   // (MANY: 2 or 4 waves, as in gibbs_wave_kernel)
    const int lane = threadIdx.x & 63, K = a.P.k, NP = a.P.npanels, Km = a.Km;
    const int nw = MANY ? (int)(blockDim.x >> 6) : 1;
    const int wave = MANY ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;
    const int rpw = (NP + nw - 1) / nw, p0 = wave * rpw;
    const bool rec = wave == 0;
    unsigned nex = 0;   // residual sums exchanged so far (parity of the LDS slots)
    if (blockIdx.x != 0) return;
    const int64_t T_tot = a.burn + a.iters;
    const T* Xp = reinterpret_cast<const T*>(a.P.X);
    const T* yp = reinterpret_cast<const T*>(a.P.y);
    double x[RMAX][KMAX], y[RMAX], vtr[KMAX];
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
        const bool have = r < rpw && p0 + r < NP;
#pragma unroll
        for (int j = 0; j < KMAX; ++j)
            x[r][j] = (have && j < K) ? (double)Xp[((size_t)(p0 + r) * K + j) * 64 + lane] : 0.0;
        y[r] = have ? (double)yp[(size_t)(p0 + r) * 64 + lane] : 0.0;
    }
    if constexpr (MANY) {
        if (threadIdx.x < 8) wsum[threadIdx.x >> 2][threadIdx.x & 3] = 0.0;
        __syncthreads();
    }
    const bool act = lane < K, model = lane < Km;
#pragma unroll
    for (int j = 0; j < KMAX; ++j) vtr[j] = (j < K && model) ? a.Vt[(size_t)j * Km + lane] : 0.0;
    const double step = act ? a.step[lane] : 0.0;
    if (threadIdx.x == 0) a.placement[0] = 1;
    double b_cur = 0.0;                                 // :82
    double rss_cur = a.rss_init;                        // -log_likelihood_current (:85)
    double s2 = a.rss_init / (double)a.P.n;             // :86
    double s2_cap = 1.0;                                // lane i: sigma2 behind staged row i
    double xi_next = (act && T_tot > 0) ? a.xi[lane] : 0.0;
    double gam_next = T_tot > 0 ? a.gam[0] : 1.0;
    double unif_next = a.n_unif > 0 ? a.unif[0] : 0.5;
    int64_t iu = 0, accepted = 0;
    const double w0 = 1.0 / (double)Km;
    const int K1 = K + 1;
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): see gibbs_wave_kernel
    for (int64_t t = 0; t < T_tot; ++t) {
        // proposal b_cur + diag(S_hat stepsize) xi  (:98,:121); lanes K .. 63 stay 0
        const double b_prop = fma(step, xi_next, b_cur);
        const double gam_t = gam_next;
        {
            const int64_t tn = t + 1 < T_tot ? t + 1 : t;
            xi_next = act ? a.xi[tn * K + lane] : 0.0;
            gam_next = a.gam[tn];
        }
        double uj[KMAX];
#pragma unroll
        for (int j = 0; j < KMAX; ++j) uj[j] = readlane_f64(b_prop, j);
        // omegas = b_prop Vt_hat + 1/Km >= 0 ?                          (:99-102)
        double o[4] = {w0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = 0; j < KMAX; ++j) o[j & 3] = fma(uj[j], vtr[j], o[j & 3]);
        const bool neg = model && ((o[0] + o[1]) + (o[2] + o[3]) < 0.0);
        if (!__any(neg)) {
            double part0 = 0.0, part1 = 0.0;
#pragma unroll
            for (int r = 0; r < RMAX; ++r) {
                double acc = y[r];
#pragma unroll
                for (int j = 0; j < KMAX; ++j) acc = fma(-x[r][j], uj[j], acc);
                if (r & 1) part1 = fma(acc, acc, part1);
                else part0 = fma(acc, acc, part0);
            }
            double rss_prop = wave_sum(part0 + part1);
            if constexpr (MANY) {   // every wave takes this branch together: identical tests
                double* ws = wsum[nex & 1];
                if (lane == 0) ws[wave] = rss_prop;
                __syncthreads();
                rss_prop = (ws[0] + ws[1]) + (ws[2] + ws[3]);
                ++nex;
            }
            if (iu >= a.n_unif) {
                if (threadIdx.x == 0) a.status[0] = 2;
                break;
            }
            // min(1, exp((ll_prop - ll_cur) / sigma2)), ll = -rss     (:106-109)
            const double ratio = exp((rss_cur - rss_prop) / s2);
            const double p_acc = ratio < 1.0 ? ratio : 1.0;
            const double uu = unif_next;
            ++iu;
            if (iu < a.n_unif) unif_next = a.unif[iu];
            if (uu < p_acc) {                                          // :110-112
                b_cur = b_prop;
                rss_cur = rss_prop;
                if (t >= a.burn) ++accepted;
            }
        }
        // sigma2 = 1 / Gamma(shape, 1/scale_post), no floor on this path      (:115-117)
        s2 = ((a.nu0_s20 + rss_cur) * 0.5) / gam_t;
        if (t >= a.burn) {
            const int64_t kept = t - a.burn;
            const int slot = (int)(kept & 63);
            if (act && rec) rows[slot * K1 + lane] = b_cur;
            s2_cap = lane == slot ? s2 : s2_cap;
            if (rec && (slot == 63 || t + 1 == T_tot)) {
                const int nrows = slot + 1;
                const double sig = sqrt(s2_cap);
                if (lane < nrows) rows[lane * K1 + K] = sig;
                double* dst = a.out + (kept - slot) * K1;
                for (int idx = lane; idx < nrows * K1; idx += 64) dst[idx] = rows[idx];
            }
        }
    }
    if (threadIdx.x == 0) {
        a.counters[0] = accepted;
        a.counters[1] = iu;
    }
}

// row panels x columns a wave keeps: RMAX * KMAX <= 128 (and at most 16 panels = 1024 rows)
static constexpr int wave_kmax(int k) { return k <= 4 ? 4 : k <= 8 ? 8 : k <= 16 ? 16 : k <= 32 ? 32 : 0; }
static constexpr int wave_rmax(int np) { return np <= 2 ? 2 : np <= 4 ? 4 : np <= 8 ? 8 : np <= 12 ? 12 : np <= 16 ? 16 : 0; }
int gibbs_wave_capacity(int k, int npanels) {   // npanels: per WAVE
    const int km = wave_kmax(k), rm = wave_rmax(npanels);
    return (km > 0 && rm > 0 && km * rm <= 128) ? km * rm : 0;
}

template <typename T, int RMAX, int KM>
static const void* wave_kernel_of(GibbsTag, int waves) {
    if (waves > 4) {   // 8 waves: two per SIMD, 256 registers each -- the shapes that fit them
        if constexpr (RMAX * KM <= 64) return (const void*)gibbs_wave_kernel<T, RMAX, KM, 8>;   // (12 x 8 spills at 256)
        return nullptr;
    }
    return waves > 1 ? (const void*)gibbs_wave_kernel<T, RMAX, KM, 4> : (const void*)gibbs_wave_kernel<T, RMAX, KM, 1>;
}
template <typename T, int RMAX, int KM>
static const void* wave_kernel_of(SimplexTag, int waves) {
    if (waves > 4) return nullptr;
    return waves > 1 ? (const void*)simplex_wave_kernel<T, RMAX, KM, true>
                     : (const void*)simplex_wave_kernel<T, RMAX, KM, false>;
}

template <typename Tag, typename T, int RMAX, typename Args>
static hipError_t launch_wave_r(const Args& a, int n_blocks, hipStream_t s) {
    const dim3 grid((unsigned)n_blocks), block(64 * (a.waves > 1 ? a.waves : 1));
#define BMC_WV(KM)                                                                              \
    if constexpr (RMAX * KM <= 128) {                                                           \
        const void* fn = wave_kernel_of<T, RMAX, KM>(Tag{}, a.waves);                           \
        if (!fn) return hipErrorInvalidValue;                                                   \
        return launch_or_query(fn, grid, block, 0, s, a, a.query_occupancy);                    \
    }                                                                                           \
    break
    switch (wave_kmax(a.P.k)) {
        case 4: BMC_WV(4);
        case 8: BMC_WV(8);
        case 16: BMC_WV(16);
        case 32: BMC_WV(32);
    }
#undef BMC_WV
    return hipErrorInvalidValue;
}

template <typename Tag, typename T, typename Args>
static hipError_t launch_wave(const Args& a, int n_blocks, hipStream_t s) {
    const int nw = a.waves > 1 ? a.waves : 1;
    const int rpw = (a.P.npanels + nw - 1) / nw;
    if (a.P.vec != 1 || !gibbs_wave_capacity(a.P.k, rpw) || n_blocks < 1 ||
        (nw != 1 && nw != 2 && nw != 4 && nw != 8))
        return hipErrorInvalidValue;
    switch (wave_rmax(rpw)) {
        case 2: return launch_wave_r<Tag, T, 2>(a, n_blocks, s);
        case 4: return launch_wave_r<Tag, T, 4>(a, n_blocks, s);
        case 8: return launch_wave_r<Tag, T, 8>(a, n_blocks, s);
        case 12: return launch_wave_r<Tag, T, 12>(a, n_blocks, s);
        case 16: return launch_wave_r<Tag, T, 16>(a, n_blocks, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_gibbs(const GibbsArgs& a, hipStream_t s) {
    if (a.one_wave) {
        if (a.query_regs) return hipErrorInvalidValue;
        return a.P.f32 ? launch_wave<GibbsTag, float>(a, a.n_chains, s)
                       : launch_wave<GibbsTag, double>(a, a.n_chains, s);
    }
    if (a.chains_per_pass > 1) {
        if (a.query_regs) return hipErrorInvalidValue;
        // bundles of chains_per_pass chains (one, or one per slot); a leader wave per chain
        if (!geometry_ok(a) || a.waves < a.chains_per_pass || a.n_chains < a.chains_per_pass ||
            a.n_chains % a.chains_per_pass != 0 || a.bundle_slots < 0 ||
            a.n_chains / a.chains_per_pass > (a.bundle_slots > 0 ? a.bundle_slots : 1) ||
            (a.bundle_slots > 0 && (a.G > 32 || a.mode != MODE_REG)))
            return hipErrorInvalidValue;
        if (a.mode == MODE_REG)
            return a.P.f32 ? launch_multi_reg<float>(a, s) : launch_multi_reg<double>(a, s);
        return a.P.f32 ? launch_multi_t<float>(a, s) : launch_multi_t<double>(a, s);
    }
    if (!geometry_ok(a) || a.n_chains < 1 || a.n_chains > a.nslot) return hipErrorInvalidValue;
    return a.P.f32 ? launch_t<GibbsTag, GibbsArgs, float>(a, s)
                   : launch_t<GibbsTag, GibbsArgs, double>(a, s);
}

hipError_t launch_simplex(const SimplexArgs& a, hipStream_t s) {
    if (a.one_wave) {
        if (a.Km < 1 || a.Km > 64) return hipErrorInvalidValue;
        return a.P.f32 ? launch_wave<SimplexTag, float>(a, 1, s) : launch_wave<SimplexTag, double>(a, 1, s);
    }
    if (!geometry_ok(a) || a.Km < 1) return hipErrorInvalidValue;
    return a.P.f32 ? launch_t<SimplexTag, SimplexArgs, float>(a, s)
                   : launch_t<SimplexTag, SimplexArgs, double>(a, s);
}

}  // namespace bmc
