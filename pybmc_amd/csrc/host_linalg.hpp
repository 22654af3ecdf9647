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

// Symmetric eigen-decomposition by Householder tridiagonalisation followed by the implicit
// QL iteration with Wilkinson shifts (the classical tred2/tql2 scheme).  O(K^3) with a small
// constant: at K = 256 it takes 0.08 s where the cyclic Jacobi sweeps take 1 s, with a
// reconstruction error of 3e-15 relative (measured).  A = Q diag(w) Q', Q columns are
// eigenvectors.
inline bool tridiag_ql_eigh(Mat a, int n, std::vector<double>& d, Mat& Q) {
    std::vector<double> e(n, 0.0), e_scratch(n, 0.0);
    d.assign(n, 0.0);
    // ---- reduction to tridiagonal form; `a` is overwritten by the accumulated transform
    for (int i = n - 1; i >= 1; --i) {
        const int l = i - 1;
        double h = 0.0, scale = 0.0;
        if (l > 0) {
            for (int k = 0; k <= l; ++k) scale += std::fabs(a[(size_t)i * n + k]);
            if (scale == 0.0) {
                e[i] = a[(size_t)i * n + l];
            } else {
                for (int k = 0; k <= l; ++k) {
                    a[(size_t)i * n + k] /= scale;
                    h += a[(size_t)i * n + k] * a[(size_t)i * n + k];
                }
                double f = a[(size_t)i * n + l];
                const double g = f >= 0.0 ? -std::sqrt(h) : std::sqrt(h);
                e[i] = scale * g;
                h -= f * g;
                a[(size_t)i * n + l] = f - g;
                f = 0.0;
                for (int j = 0; j <= l; ++j) {
                    a[(size_t)j * n + i] = a[(size_t)i * n + j] / h;
                    double gg = 0.0;
                    for (int k = 0; k <= j; ++k) gg += a[(size_t)j * n + k] * a[(size_t)i * n + k];
                    for (int k = j + 1; k <= l; ++k) gg += a[(size_t)k * n + j] * a[(size_t)i * n + k];
                    e[j] = gg / h;
                    f += e[j] * a[(size_t)i * n + j];
                }
                const double hh = f / (h + h);
                for (int j = 0; j <= l; ++j) {
                    f = a[(size_t)i * n + j];
                    const double gg = e[j] - hh * f;
                    e[j] = gg;
                    for (int k = 0; k <= j; ++k)
                        a[(size_t)j * n + k] -= f * e[k] + gg * a[(size_t)i * n + k];
                }
            }
        } else {
            e[i] = a[(size_t)i * n + l];
        }
        d[i] = h;
    }
    d[0] = 0.0;
    e[0] = 0.0;
    for (int i = 0; i < n; ++i) {
        const int l = i - 1;
        if (d[i] != 0.0) {
            // g[j] = sum_k a[i][k] a[k][j], then a[k][j] -= g[j] a[k][i]: both as sweeps over
            // contiguous rows (per j the operations and their order are those of the textbook
            // column loops)
            std::vector<double>& g = e_scratch;
            std::fill(g.begin(), g.begin() + l + 1, 0.0);
            for (int k = 0; k <= l; ++k) {
                const double aik = a[(size_t)i * n + k];
                const double* row = &a[(size_t)k * n];
                for (int j = 0; j <= l; ++j) g[j] += aik * row[j];
            }
            for (int k = 0; k <= l; ++k) {
                const double aki = a[(size_t)k * n + i];
                double* row = &a[(size_t)k * n];
                for (int j = 0; j <= l; ++j) row[j] -= g[j] * aki;
            }
        }
        d[i] = a[(size_t)i * n + i];
        a[(size_t)i * n + i] = 1.0;
        for (int j = 0; j <= l; ++j) a[(size_t)j * n + i] = a[(size_t)i * n + j] = 0.0;
    }
    // ---- implicit QL on the tridiagonal (d, e), rotations accumulated into `a`.  A rotation
    // mixes columns i and i+1 of the transform; it is applied to ROWS of the transpose, which are
    // contiguous (same arithmetic, same results; K = 256: 80 -> 25 ms).
    {
        Mat at((size_t)n * n);
        for (int r = 0; r < n; ++r)
            for (int c = 0; c < n; ++c) at[(size_t)c * n + r] = a[(size_t)r * n + c];
        a.swap(at);
    }
    for (int i = 1; i < n; ++i) e[i - 1] = e[i];
    e[n - 1] = 0.0;
    for (int l = 0; l < n; ++l) {
        int iter = 0, m;
        do {
            for (m = l; m < n - 1; ++m) {
                const double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
                if (std::fabs(e[m]) <= 2.3e-16 * dd) break;
            }
            if (m != l) {
                if (++iter > 60) return false;
                double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
                double r = std::hypot(g, 1.0);
                g = d[m] - d[l] + e[l] / (g + (g >= 0.0 ? std::fabs(r) : -std::fabs(r)));
                double s = 1.0, c = 1.0, p = 0.0;
                int i;
                for (i = m - 1; i >= l; --i) {
                    double f = s * e[i];
                    const double b = c * e[i];
                    r = std::hypot(f, g);
                    e[i + 1] = r;
                    if (r == 0.0) {
                        d[i + 1] -= p;
                        e[m] = 0.0;
                        break;
                    }
                    s = f / r;
                    c = g / r;
                    g = d[i + 1] - p;
                    r = (d[i] - g) * s + 2.0 * c * b;
                    p = s * r;
                    d[i + 1] = g + p;
                    g = c * r - b;
                    double* __restrict__ ri = &a[(size_t)i * n];
                    double* __restrict__ rj = &a[(size_t)(i + 1) * n];
                    for (int k = 0; k < n; ++k) {
                        const double fk = rj[k];
                        rj[k] = s * ri[k] + c * fk;
                        ri[k] = c * ri[k] - s * fk;
                    }
                }
                if (r == 0.0 && i >= l) continue;
                d[l] -= p;
                e[l] = g;
                e[m] = 0.0;
            }
        } while (m != l);
    }
    Q.assign((size_t)n * n, 0.0);
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) Q[(size_t)c * n + r] = a[(size_t)r * n + c];
    return true;
}

// Dispatcher: Jacobi for tiny matrices, tridiagonal QL above (falls back to Jacobi if the
// QL iteration does not converge).
inline void sym_eigh(const Mat& a, int k, std::vector<double>& w, Mat& Q) {
    if (k > 8 && tridiag_ql_eigh(a, k, w, Q)) return;
    jacobi_eigh(a, k, w, Q);
}

}  // namespace bmc_la
