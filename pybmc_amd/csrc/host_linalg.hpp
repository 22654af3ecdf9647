// Small dense K x K linear algebra on the host (K <= a few hundred).
// One-off set-up work of the sampler (reference inference_utils.py:22,26 and the
// factorisation behind :41); everything O(N) runs on the GPU.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <vector>

namespace bmc_la {

using Mat = std::vector<double>;  // row-major k*k

// In-place inverse by Gauss-Jordan with partial pivoting.
// Returns false on an exactly zero pivot (numpy raises LinAlgError then).
inline bool invert(Mat& a, int k) {
    Mat inv((size_t)k * k, 0.0);
    for (int i = 0; i < k; ++i) inv[(size_t)i * k + i] = 1.0;
    for (int c = 0; c < k; ++c) {
        int p = c;
        double best = std::fabs(a[(size_t)c * k + c]);
        for (int r = c + 1; r < k; ++r) {
            double v = std::fabs(a[(size_t)r * k + c]);
            if (v > best) { best = v; p = r; }
        }
        if (best == 0.0 || !std::isfinite(best)) return false;
        if (p != c)
            for (int j = 0; j < k; ++j) {
                std::swap(a[(size_t)p * k + j], a[(size_t)c * k + j]);
                std::swap(inv[(size_t)p * k + j], inv[(size_t)c * k + j]);
            }
        const double piv = 1.0 / a[(size_t)c * k + c];
        for (int j = 0; j < k; ++j) {
            a[(size_t)c * k + j] *= piv;
            inv[(size_t)c * k + j] *= piv;
        }
        for (int r = 0; r < k; ++r) {
            if (r == c) continue;
            const double f = a[(size_t)r * k + c];
            if (f == 0.0) continue;
            for (int j = 0; j < k; ++j) {
                a[(size_t)r * k + j] -= f * a[(size_t)c * k + j];
                inv[(size_t)r * k + j] -= f * inv[(size_t)c * k + j];
            }
        }
    }
    a.swap(inv);
    return true;
}

// Solve A x = b (A destroyed), partial pivoting.  false when singular.
inline bool solve(Mat a, std::vector<double> b, int k, std::vector<double>& x) {
    for (int c = 0; c < k; ++c) {
        int p = c;
        double best = std::fabs(a[(size_t)c * k + c]);
        for (int r = c + 1; r < k; ++r) {
            double v = std::fabs(a[(size_t)r * k + c]);
            if (v > best) { best = v; p = r; }
        }
        if (best == 0.0 || !std::isfinite(best)) return false;
        if (p != c) {
            for (int j = 0; j < k; ++j) std::swap(a[(size_t)p * k + j], a[(size_t)c * k + j]);
            std::swap(b[p], b[c]);
        }
        for (int r = c + 1; r < k; ++r) {
            const double f = a[(size_t)r * k + c] / a[(size_t)c * k + c];
            if (f == 0.0) continue;
            for (int j = c; j < k; ++j) a[(size_t)r * k + j] -= f * a[(size_t)c * k + j];
            b[r] -= f * b[c];
        }
    }
    x.assign(k, 0.0);
    for (int r = k - 1; r >= 0; --r) {
        long double s = b[r];
        for (int j = r + 1; j < k; ++j) s -= (long double)a[(size_t)r * k + j] * x[j];
        x[r] = (double)(s / a[(size_t)r * k + r]);
    }
    return true;
}

// Lower Cholesky factor of a symmetric matrix (lower triangle read).
// false when not positive definite.
inline bool cholesky(const Mat& a, int k, Mat& L) {
    L.assign((size_t)k * k, 0.0);
    for (int i = 0; i < k; ++i)
        for (int j = 0; j <= i; ++j) {
            long double s = a[(size_t)i * k + j];
            for (int m = 0; m < j; ++m)
                s -= (long double)L[(size_t)i * k + m] * L[(size_t)j * k + m];
            if (i == j) {
                if (!(s > 0.0L)) return false;
                L[(size_t)i * k + i] = (double)sqrtl(s);
            } else {
                L[(size_t)i * k + j] = (double)(s / L[(size_t)j * k + j]);
            }
        }
    return true;
}

// Linv = L^-1 for lower-triangular L.
inline void lower_inverse(const Mat& L, int k, Mat& Li) {
    Li.assign((size_t)k * k, 0.0);
    for (int c = 0; c < k; ++c) {
        Li[(size_t)c * k + c] = 1.0 / L[(size_t)c * k + c];
        for (int r = c + 1; r < k; ++r) {
            long double s = 0.0L;
            for (int m = c; m < r; ++m)
                s -= (long double)L[(size_t)r * k + m] * Li[(size_t)m * k + c];
            Li[(size_t)r * k + c] = (double)(s / L[(size_t)r * k + r]);
        }
    }
}

// Cyclic Jacobi eigen-decomposition of a symmetric matrix: A = Q diag(w) Q'.
// Q columns are eigenvectors.  Accurate to a few ulp of ||A||, which is what the
// 1e-6 parity bar on posterior summaries leans on.
inline void jacobi_eigh(Mat a, int k, std::vector<double>& w, Mat& Q) {
    Q.assign((size_t)k * k, 0.0);
    for (int i = 0; i < k; ++i) Q[(size_t)i * k + i] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < k; ++i) {
            diag += a[(size_t)i * k + i] * a[(size_t)i * k + i];
            for (int j = i + 1; j < k; ++j) off += a[(size_t)i * k + j] * a[(size_t)i * k + j];
        }
        if (off <= 1e-34 * (diag + off) || off == 0.0) break;
        for (int p = 0; p < k - 1; ++p)
            for (int q = p + 1; q < k; ++q) {
                const double apq = a[(size_t)p * k + q];
                if (apq == 0.0) continue;
                const double app = a[(size_t)p * k + p], aqq = a[(size_t)q * k + q];
                if (std::fabs(apq) < 1e-300) continue;
                const double theta = (aqq - app) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) /
                                 (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int m = 0; m < k; ++m) {  // columns p,q
                    const double amp = a[(size_t)m * k + p], amq = a[(size_t)m * k + q];
                    a[(size_t)m * k + p] = c * amp - s * amq;
                    a[(size_t)m * k + q] = s * amp + c * amq;
                }
                for (int m = 0; m < k; ++m) {  // rows p,q
                    const double apm = a[(size_t)p * k + m], aqm = a[(size_t)q * k + m];
                    a[(size_t)p * k + m] = c * apm - s * aqm;
                    a[(size_t)q * k + m] = s * apm + c * aqm;
                }
                for (int m = 0; m < k; ++m) {
                    const double qmp = Q[(size_t)m * k + p], qmq = Q[(size_t)m * k + q];
                    Q[(size_t)m * k + p] = c * qmp - s * qmq;
                    Q[(size_t)m * k + q] = s * qmp + c * qmq;
                }
            }
    }
    w.resize(k);
    for (int i = 0; i < k; ++i) w[i] = a[(size_t)i * k + i];
}

}  // namespace bmc_la
