"""ctypes binding of libpybmc_amd.so (the C ABI declared in include/pybmc_amd.h).

There is no CPU fallback: if the shared library is missing, or no gfx950 GPU is
usable, the calls raise.  Status codes are mapped back to the exception types the
reference raises at the same places (numpy ``LinAlgError`` for singular matrices,
``ValueError`` for bad arguments).
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpybmc_amd.so")

ABI_VERSION = 4   # PYBMC_AMD_ABI_VERSION of include/pybmc_amd.h this binding was written for
BMC_OK, BMC_EINVAL, BMC_ESINGULAR, BMC_EHIP, BMC_ENOMEM, BMC_ETIMEOUT, BMC_ESTATE = range(7)
BMC_F64, BMC_F32 = 0, 1
BMC_ROW_MAJOR, BMC_COL_MAJOR = 0, 1
BMC_RNG_DEVICE, BMC_RNG_REPLAY = 0, 1


class Tuning(C.Structure):
    _fields_ = [("groups_per_chain", C.c_int32), ("waves_per_group", C.c_int32),
                ("residency", C.c_int32), ("panels_per_wave", C.c_int32),
                ("force_agent_scope", C.c_int32), ("chains_per_pass", C.c_int32),
                ("rss_mode", C.c_int32), ("cu_limit", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("loop_ms", C.c_double), ("rng_ms", C.c_double), ("post_ms", C.c_double),
                ("total_ms", C.c_double), ("iterations", C.c_int64), ("n_chains", C.c_int32),
                ("launches", C.c_int32), ("groups_per_chain", C.c_int32),
                ("waves_per_group", C.c_int32), ("chains_per_pass", C.c_int32),
                ("residency", C.c_int32), ("xcd_local_chains", C.c_int32),
                ("bytes_per_pass", C.c_int64),
                ("passes", C.c_int64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


# name -> (restype, argtypes); every symbol include/pybmc_amd.h declares
_P = C.c_void_p
_D = C.POINTER(C.c_double)
PROTOTYPES = {
    "bmc_abi_version": (C.c_int, []),
    "bmc_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "bmc_destroy": (None, [_P]),
    "bmc_last_error": (C.c_char_p, [_P]),
    "bmc_set_stream": (C.c_int, [_P, _P]),
    "bmc_set_tuning": (C.c_int, [_P, C.POINTER(Tuning)]),
    "bmc_set_problem": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int64, C.c_int, _P, C.c_int]),
    "bmc_set_problem_device": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int64, C.c_int, _P,
                                         C.c_int]),
    "bmc_orthogonalize": (C.c_int, [_P, _D, C.c_int64, C.c_int32, C.c_int64, _D, C.c_int32, _D, _D, _D,
                                    _D, _D]),
    "bmc_set_prior": (C.c_int, [_P, _D, _D, C.c_double, C.c_double]),
    "bmc_get_gram": (C.c_int, [_P, _D]),
    "bmc_get_basis": (C.c_int, [_P, _D, _D, _D]),
    "bmc_conditional_moments": (C.c_int, [_P, C.c_double, _D, _D]),
    "bmc_residual_rss": (C.c_int, [_P, _D, C.c_int32, _D]),
    "bmc_residual_rss_bench": (C.c_int, [_P, C.c_int32, C.c_int32, _D]),
    "bmc_gram_bench": (C.c_int, [_P, C.c_int32, _D]),
    "bmc_predict_timing": (C.c_int, [_P, _D, _D, _D, _D]),
    "bmc_predict_draws": (C.c_int, [_P, _D, C.c_int]),
    "bmc_comm_unique_id": (C.c_int, [C.c_char_p]),
    "bmc_comm_init": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_char_p]),
    "bmc_allgather": (C.c_int, [_P, _P, _P, C.c_int64]),
    "bmc_comm_destroy": (C.c_int, [_P]),
    "bmc_gibbs_run": (C.c_int, [_P, C.c_int32, C.c_int64, C.POINTER(C.c_uint64), C.c_int, _D, _D,
                                _D, C.POINTER(Stats)]),
    "bmc_gibbs_run_device": (C.c_int, [_P, C.c_int32, C.c_int64, C.POINTER(C.c_uint64), _P,
                                       C.POINTER(Stats)]),
    "bmc_simplex_run": (C.c_int, [_P, _D, C.c_int32, _D, C.c_int64, C.c_int64, C.c_double, C.c_double,
                                  C.c_double, C.c_int, C.c_uint64, _D, _D, C.c_int64, _D, _D,
                                  C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(Stats)]),
    "bmc_predict": (C.c_int, [_P, _D, C.c_int64, C.c_int32, _D, C.c_int32, C.c_int32, _D, C.c_int,
                              C.c_uint64, _D, C.POINTER(C.c_int32), _D, C.c_int32, _D,
                              C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32, _D, _D,
                              C.POINTER(C.c_int64)]),
    "bmc_rng_fill": (C.c_int, [_P, C.c_uint64, C.c_int64, _D, C.c_double, C.c_int64, _D]),
    "bmc_philox_raw": (C.c_int, [_P, C.c_uint64, C.c_uint32, C.c_int64, C.POINTER(C.c_uint32)]),
}

_lib = None
_lib_lock = threading.Lock()


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  torch's wheel bundles its own libamdhip64.so (same
    soname as /opt/rocm's); if torch is imported AFTER this library bound the system copy, the
    process would hold two runtimes and device pointers could not be shared (bench.py and
    pybmc_amd.chains hand torch tensors to the C ABI).  So when torch is installed but not yet
    imported, its copy is loaded first; the dynamic loader then resolves our NEEDED entry and
    torch's own to that one object.  Set PYBMC_AMD_SYSTEM_HIP=1 to skip this."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("PYBMC_AMD_SYSTEM_HIP") == "1":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def bind(path, mode=C.RTLD_GLOBAL):
    """dlopen one build of the library and bind every prototype."""
    lib = C.CDLL(path, mode=mode)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.bmc_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{path}: ABI version mismatch")
    return lib


def load_library():
    """dlopen the in-tree library and bind every prototype.  Raises if absent."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `make -C pybmc_amd/csrc` "
                "(or __graft_entry__.build()).  pybmc_amd has no CPU fallback.")
        _share_hip_runtime_with_torch()
        _lib = bind(LIB_PATH)
        return _lib


def _dptr(a):
    return a.ctypes.data_as(_D) if a is not None else None


class BmcError(RuntimeError):
    pass


class Context:
    """One bmc_ctx: one GPU, one host thread at a time."""

    def __init__(self, device=0, lib=None):
        # `lib`: another build of the same ABI (scripts/ab.py compares builds in one process)
        self._lib = lib if lib is not None else load_library()
        h = _P()
        rc = self._lib.bmc_create(int(device), C.byref(h))
        if rc != BMC_OK:
            raise BmcError(
                f"bmc_create(device={device}) failed with status {rc}: no usable gfx950 "
                "(MI355X) device.  pybmc_amd has no CPU fallback.")
        self._h = h
        self.device = int(device)
        self.n = self.k = 0
        # bumped whenever the resident (y, X) changes: lets a caller that left a problem on the
        # device (BayesianModelCombination.orthogonalize(method="device")) find out whether it
        # is still there before sampling from it
        self.problem_generation = 0
        self._last_predict = None     # (n_points, n_draws) of the last predict()
        # a bmc_ctx is driven by one host thread at a time: the functional API
        # (gibbs_sampler, rndm_m_random_calculator, ...) holds this lock for the whole
        # set_problem -> set_prior -> run sequence, so that two threads sharing the
        # per-device context serialise instead of sampling each other's problem
        self.lock = threading.RLock()

    # -- plumbing --------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.bmc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc == BMC_OK:
            return
        msg = (self._lib.bmc_last_error(self._h) or b"").decode("utf-8", "replace")
        if rc == BMC_ESINGULAR:
            raise np.linalg.LinAlgError(msg or "Singular matrix")
        if rc == BMC_EINVAL:
            raise ValueError(msg)
        if rc == BMC_ENOMEM:
            raise MemoryError(msg)
        raise BmcError(f"status {rc}: {msg}")

    def set_stream(self, hip_stream_ptr):
        self._check(self._lib.bmc_set_stream(self._h, _P(hip_stream_ptr or 0)))

    def set_tuning(self, groups_per_chain=0, waves_per_group=0, residency=0, panels_per_wave=0,
                   force_agent_scope=0, chains_per_pass=0, rss_mode=0, cu_limit=0):
        """residency: 0 auto, 1 registers, 2 LDS, 3 stream from HBM; chains_per_pass: 0 auto,
        1 off, 2/4/8 cap (chains served per pass over X, or per bundle of an XCD's resident panels); rss_mode: 0 a pass over the data every iteration (the reference's
        computation, default), 1 the same number from sufficient statistics (opt-in, K <= 64);
        cu_limit: plan persistent launches for at most this many CUs (0 = the device's)."""
        t = Tuning(groups_per_chain, waves_per_group, residency, panels_per_wave,
                   force_agent_scope, chains_per_pass, rss_mode, cu_limit)
        self._check(self._lib.bmc_set_tuning(self._h, C.byref(t)))

    # -- problem / prior ---------------------------------------------------------
    def set_problem(self, y, X, dtype=None):
        """y (n,), X (n,k).  Keeps X's own memory order when it is C- or F-contiguous."""
        X = np.asarray(X)
        y = np.asarray(y)
        if X.ndim != 2 or y.ndim != 1 or X.shape[0] != y.shape[0]:
            raise ValueError("X must be (n, k) and y (n,)")
        if dtype is None:
            dtype = np.float32 if (X.dtype == np.float32 and y.dtype == np.float32) else np.float64
        dtype = np.dtype(dtype)
        if dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise ValueError("dtype must be float32 or float64")
        if X.flags.f_contiguous and not X.flags.c_contiguous:
            Xc = np.asfortranarray(X, dtype=dtype)
            layout, ldx = BMC_COL_MAJOR, X.shape[0]
        else:
            Xc = np.ascontiguousarray(X, dtype=dtype)
            layout, ldx = BMC_ROW_MAJOR, X.shape[1]
        yc = np.ascontiguousarray(y, dtype=dtype)
        n, k = X.shape
        self.problem_generation += 1   # (before the call: a failed call leaves no problem)
        self._check(self._lib.bmc_set_problem(
            self._h, Xc.ctypes.data_as(_P), n, k, max(ldx, 1), layout, yc.ctypes.data_as(_P),
            BMC_F32 if dtype == np.float32 else BMC_F64))
        self.n, self.k = n, k

    def set_problem_device(self, x_ptr, n, k, ldx, layout, y_ptr, f32=False):
        self.problem_generation += 1
        self._check(self._lib.bmc_set_problem_device(
            self._h, _P(x_ptr), n, k, ldx, layout, _P(y_ptr), BMC_F32 if f32 else BMC_F64))
        self.n, self.k = n, k

    def orthogonalize(self, F, truth, k, want_U=True):
        """Centre + thin SVD through the Gram on the device.  Returns
        (mu, y_c, U_hat or None, S_hat, Vt_rows); the context then holds (y_c, U_hat)."""
        F = np.ascontiguousarray(F, dtype=np.float64)
        truth = np.ascontiguousarray(truth, dtype=np.float64).reshape(-1)
        n, km = F.shape
        if truth.shape[0] != n:
            raise ValueError("truth must have one value per row of F")
        mu, yc = np.empty(n), np.empty(n)
        U = np.empty((k, n)) if want_U else None
        S, Vt = np.empty(k), np.empty((k, km))
        self.problem_generation += 1
        self._check(self._lib.bmc_orthogonalize(self._h, _dptr(F), n, km, km, _dptr(truth), int(k),
                                                _dptr(mu), _dptr(yc), _dptr(U), _dptr(S), _dptr(Vt)))
        self.n, self.k = n, int(k)
        return mu, yc, (U.T if U is not None else None), S, Vt

    def set_prior(self, b0, C0, nu0, sigma20):
        b0 = np.ascontiguousarray(b0, dtype=np.float64).reshape(-1)
        C0 = np.ascontiguousarray(C0, dtype=np.float64)
        if b0.shape[0] != self.k or C0.shape != (self.k, self.k):
            raise ValueError("prior shapes do not match the design matrix")
        self._check(self._lib.bmc_set_prior(self._h, _dptr(b0), _dptr(C0), float(nu0),
                                            float(sigma20)))

    # -- introspection -------------------------------------------------------------
    def gram(self):
        out = np.empty((self.k + 1, self.k + 1))
        self._check(self._lib.bmc_get_gram(self._h, _dptr(out)))
        return out

    def basis(self):
        W = np.empty((self.k, self.k))
        lam = np.empty(self.k)
        s2 = C.c_double()
        self._check(self._lib.bmc_get_basis(self._h, _dptr(W), _dptr(lam), C.byref(s2)))
        return W, lam, s2.value

    def conditional_moments(self, sigma2):
        mean = np.empty(self.k)
        cov = np.empty((self.k, self.k))
        self._check(self._lib.bmc_conditional_moments(self._h, float(sigma2), _dptr(mean),
                                                      _dptr(cov)))
        return mean, cov

    def residual_rss(self, beta):
        beta = np.ascontiguousarray(np.atleast_2d(beta), dtype=np.float64)
        if beta.shape[1] != self.k:
            raise ValueError("beta must have k columns")
        out = np.empty(beta.shape[0])
        self._check(self._lib.bmc_residual_rss(self._h, _dptr(beta), beta.shape[0], _dptr(out)))
        return out

    def residual_rss_bench(self, nb=1, reps=20):
        ms = C.c_double()
        self._check(self._lib.bmc_residual_rss_bench(self._h, nb, reps, C.byref(ms)))
        return ms.value

    def gram_bench(self, reps=20):
        ms = C.c_double()
        self._check(self._lib.bmc_gram_bench(self._h, reps, C.byref(ms)))
        return ms.value

    def predict_timing(self):
        """HIP-event times (ms) of the last predict() on this context."""
        v = [C.c_double() for _ in range(4)]
        self._check(self._lib.bmc_predict_timing(self._h, *[C.byref(x) for x in v]))
        return dict(zip(("h2d_ms", "gemm_ms", "select_ms", "device_ms"), (x.value for x in v)))

    # -- pooling over GPUs without torch (RCCL through the C ABI) ------------------------
    @staticmethod
    def comm_unique_id():
        """128-byte RCCL id made by rank 0; hand it to the other ranks by any transport."""
        buf = C.create_string_buffer(128)
        rc = load_library().bmc_comm_unique_id(buf)
        if rc != BMC_OK:
            raise BmcError(f"bmc_comm_unique_id failed with status {rc}: RCCL could not be loaded")
        return buf.raw

    def comm_init(self, world, rank, unique_id):
        if len(unique_id) != 128:
            raise ValueError("unique_id must be the 128 bytes of comm_unique_id()")
        self._check(self._lib.bmc_comm_init(self._h, int(world), int(rank), bytes(unique_id)))

    def allgather(self, send_ptr, recv_ptr, count_per_rank):
        """f64 all-gather of DEVICE buffers on the context's stream; blocks until done."""
        self._check(self._lib.bmc_allgather(self._h, _P(send_ptr), _P(recv_ptr),
                                            int(count_per_rank)))

    def comm_destroy(self):
        self._check(self._lib.bmc_comm_destroy(self._h))

    # -- the loop --------------------------------------------------------------------
    def gibbs_run(self, n_chains, iters, seeds=None, xi=None, g=None):
        """Returns (samples [n_chains, iters, k+1], stats dict)."""
        out = np.empty((n_chains, iters, self.k + 1))
        st = Stats()
        if xi is not None or g is not None:
            xi = np.ascontiguousarray(xi, dtype=np.float64).reshape(n_chains, iters, self.k)
            g = np.ascontiguousarray(g, dtype=np.float64).reshape(n_chains, iters)
            rc = self._lib.bmc_gibbs_run(self._h, n_chains, iters, None, BMC_RNG_REPLAY,
                                         _dptr(xi), _dptr(g), _dptr(out), C.byref(st))
        else:
            sd = np.ascontiguousarray(seeds, dtype=np.uint64).reshape(n_chains)
            rc = self._lib.bmc_gibbs_run(self._h, n_chains, iters,
                                         sd.ctypes.data_as(C.POINTER(C.c_uint64)),
                                         BMC_RNG_DEVICE, None, None, _dptr(out), C.byref(st))
        self._check(rc)
        return out, st.as_dict()

    def gibbs_run_device(self, n_chains, iters, seeds, out_ptr):
        sd = np.ascontiguousarray(seeds, dtype=np.uint64).reshape(n_chains)
        st = Stats()
        self._check(self._lib.bmc_gibbs_run_device(
            self._h, n_chains, iters, sd.ctypes.data_as(C.POINTER(C.c_uint64)), _P(out_ptr),
            C.byref(st)))
        return st.as_dict()

    # -- simplex sampler --------------------------------------------------------------------
    def simplex_run(self, Vt_hat, S_hat, iters, nu0, sigma20, burn, stepsize, seed=0, xi=None,
                    unif=None, g=None, return_stats=False):
        """Returns (samples (iters, k+1), accepted) [+ (uniforms used, stats)]."""
        Vt_hat = np.ascontiguousarray(Vt_hat, dtype=np.float64)
        S_hat = np.ascontiguousarray(S_hat, dtype=np.float64).reshape(-1)
        if Vt_hat.ndim != 2 or Vt_hat.shape[0] != self.k or S_hat.shape[0] != self.k:
            raise ValueError("Vt_hat must be (k, n_models) and S_hat (k,)")
        out = np.empty((iters, self.k + 1))
        acc, used, st = C.c_int64(), C.c_int64(), Stats()
        if xi is not None:
            tt = burn + iters
            xi = np.ascontiguousarray(xi, dtype=np.float64).reshape(tt, self.k)
            g = np.ascontiguousarray(g, dtype=np.float64).reshape(tt)
            unif = np.ascontiguousarray(unif, dtype=np.float64).reshape(-1)
            rc = self._lib.bmc_simplex_run(
                self._h, _dptr(Vt_hat), Vt_hat.shape[1], _dptr(S_hat), iters, burn, stepsize, nu0,
                sigma20, BMC_RNG_REPLAY, 0, _dptr(xi), _dptr(unif), unif.shape[0], _dptr(g),
                _dptr(out), C.byref(acc), C.byref(used), C.byref(st))
        else:
            rc = self._lib.bmc_simplex_run(
                self._h, _dptr(Vt_hat), Vt_hat.shape[1], _dptr(S_hat), iters, burn, stepsize, nu0,
                sigma20, BMC_RNG_DEVICE, int(seed) & (2 ** 64 - 1), None, None, 0, None,
                _dptr(out), C.byref(acc), C.byref(used), C.byref(st))
        self._check(rc)
        if return_stats:
            return out, acc.value, used.value, st.as_dict()
        return out, acc.value

    # -- posterior predictive --------------------------------------------------------------
    def predict_draws(self, order="C"):
        """The draws of the last predict() as an (n_draws, n_points) array: order "C" is the
        reference's layout (sampling_utils.py:77; transposed on the device), order "F" a
        Fortran-ordered array of the same shape (the device layout, no transpose)."""
        if self._last_predict is None:
            raise BmcError("no predict() has run on this context")
        M, S = self._last_predict
        if order == "C":
            out = np.empty((S, M))
            self._check(self._lib.bmc_predict_draws(self._h, _dptr(out), 1))
            return out
        out = np.empty((M, S))
        self._check(self._lib.bmc_predict_draws(self._h, _dptr(out), 0))
        return out.T

    def predict(self, preds, theta, Vt_hat, seed=0, noise=None, q=(2.5, 50, 97.5), truth=None,
                cov_percentiles=None, want_draws=True, draws_order="C"):
        """preds (M, Km), theta (S, k+1) selected posterior rows, Vt_hat (k, Km).
        Returns (rndm_m (S, M) or None, bands (len(q), M), coverage or None).  rndm_m is
        C-ordered like the reference's (draws_order="C", default) or Fortran-ordered
        (draws_order="F": the device layout, a point's draws contiguous)."""
        preds = np.ascontiguousarray(preds, dtype=np.float64)
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        Vt_hat = np.ascontiguousarray(Vt_hat, dtype=np.float64)
        M, Km = preds.shape
        S, k1 = theta.shape
        if Vt_hat.shape != (k1 - 1, Km):
            raise ValueError("Vt_hat must be (k, n_models) with k = theta.shape[1] - 1")
        qi, qg = order_stat_plan(S, q)
        bands = np.empty((len(q), M))
        lo = hi = hits = None
        tr = None
        n_cov = 0
        if truth is not None:
            tr = np.ascontiguousarray(truth, dtype=np.float64).reshape(M)
            lo, hi = coverage_plan(S, cov_percentiles)
            n_cov = len(lo)
            hits = np.zeros(n_cov, dtype=np.int64)
        nz = None
        if noise is not None:
            nz = np.ascontiguousarray(noise, dtype=np.float64)
            if nz.shape != (S, M):
                raise ValueError("noise must be (n_draws, n_points)")
        i32p = C.POINTER(C.c_int32)
        self._check(self._lib.bmc_predict(
            self._h, _dptr(preds), M, Km, _dptr(theta), S, k1 - 1, _dptr(Vt_hat),
            BMC_RNG_REPLAY if nz is not None else BMC_RNG_DEVICE, int(seed) & (2 ** 64 - 1),
            _dptr(nz), qi.ctypes.data_as(i32p), _dptr(qg), len(q), _dptr(tr),
            lo.ctypes.data_as(i32p) if lo is not None else None,
            hi.ctypes.data_as(i32p) if hi is not None else None, n_cov, None,
            _dptr(bands), hits.ctypes.data_as(C.POINTER(C.c_int64)) if hits is not None else None))
        self._last_predict = (M, S)
        cov = None if hits is None else [int(h) / M * 100 for h in hits]
        return (self.predict_draws(draws_order) if want_draws else None), bands, cov

    # -- variates -----------------------------------------------------------------------
    def rng_fill(self, seed, n_normal=0, shape=1.0, n_gamma=0):
        z = np.empty(n_normal)
        g = np.empty(n_gamma)
        self._check(self._lib.bmc_rng_fill(self._h, int(seed), n_normal, _dptr(z), float(shape),
                                           n_gamma, _dptr(g)))
        return z, g

    def philox_raw(self, seed, stream_id, nblocks4):
        out = np.empty(4 * nblocks4, dtype=np.uint32)
        self._check(self._lib.bmc_philox_raw(self._h, int(seed), int(stream_id), nblocks4,
                                             out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out.reshape(nblocks4, 4)


def order_stat_plan(n, percentiles):
    """(index, weight) of numpy.percentile's default linear method for each percentile:
    virtual index (n - 1) * q in float64, as numpy does (reference sampling_utils.py:80-82
    calls np.percentile)."""
    idx, gam = [], []
    for p in percentiles:
        qq = np.true_divide(np.float64(p), 100)
        vi = (n - 1) * qq
        lo = np.floor(vi)
        g = vi - lo
        lo = int(lo)
        if lo >= n - 1:
            lo, g = n - 1, 0.0
        if lo < 0:
            lo, g = 0, 0.0
        idx.append(lo)
        gam.append(float(g))
    return np.array(idx, dtype=np.int32), np.array(gam, dtype=np.float64)


def coverage_plan(n, percentiles):
    """Index pairs of reference sampling_utils.py:30-31 (truncating int())."""
    lo = [int((0.5 - p / 200) * n) for p in percentiles]
    hi = [int((0.5 + p / 200) * n) - 1 for p in percentiles]
    hi = [h if h >= 0 else n + h for h in hi]   # python negative index
    return np.array(lo, dtype=np.int32), np.array(hi, dtype=np.int32)


_default_ctx = {}
_ctx_lock = threading.Lock()


def default_context(device=0):
    """A cached per-device context for the functional API (creation is thread-safe; use
    ``with ctx.lock:`` around a sequence of calls that belongs together)."""
    ctx = _default_ctx.get(device)
    if ctx is None:
        with _ctx_lock:
            ctx = _default_ctx.get(device)
            if ctx is None:
                ctx = _default_ctx[device] = Context(device)
    return ctx
