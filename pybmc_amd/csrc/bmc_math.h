// Elementary functions of the on-device variate generators, written out so that the Box-Muller
// transform costs a few dozen f64 operations instead of the general-purpose library calls
// (log, sincospi: special cases, denormals, huge arguments -- none of which can occur here).
// Plain C++ (no HIP headers): the CPU test suite compiles the same text with g++ and checks it
// against libm (tests/test_host_math.py).  Algorithms: the classic fdlibm kernels (Sun
// Microsystems, freely distributable) -- log by s = f / (2 + f) and an odd polynomial in s,
// sin / cos on [-pi/4, pi/4] by their minimax polynomials -- with the argument reductions that
// the restricted domains allow.  Each result is within 1-2 ulp of the correctly rounded value.
#pragma once
#include <stdint.h>
#include <string.h>
#include <math.h>

#if defined(__HIPCC__) 
#define BMC_HD __host__ __device__ inline
#else
#define BMC_HD inline
#endif

namespace bmc {

// fma(a, b, c) with c a compile-time constant of a polynomial.  On the GPU the constant is handed
// to v_fma_f64 as a scalar-register operand: left to itself hipcc copies every such constant into
// a vector register pair in front of a v_fmac (two extra vector instructions per coefficient --
// 32 per Box-Muller pair, and vector instructions are what the predictive GEMM's epilogue is
// made of).  Same operation, same bits; the CPU build is plain fma.
BMC_HD double fma_c(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
    return d;
#else
    return fma(a, b, c);
#endif
}

// natural logarithm of a NORMAL double 0 < x (the generators call it with x in [2^-53, 1])
BMC_HD double log_normal_arg(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t ix;
    memcpy(&ix, &x, 8);
    // x = 2^k m with m in [sqrt(1/2), sqrt(2)): shift the exponent so that the high word of
    // sqrt(1/2) lands on an exponent boundary
    uint32_t hx = (uint32_t)(ix >> 32);
    hx += 0x3ff00000u - 0x3fe6a09eu;
    const int k = (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
    ix = ((uint64_t)hx << 32) | (ix & 0xffffffffull);
    double m;
    memcpy(&m, &ix, 8);
    // (the polynomials are written with explicit fma: one instruction per coefficient on the
    // GPU -- the library is compiled with -ffp-contract=off -- and the same, exactly rounded,
    // operation in the CPU build of this text, so both produce the same bits)
    const double f = m - 1.0;
    const double hfsq = (0.5 * f) * f;
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * fma_c(w, fma_c(w, Lg6, Lg4), Lg2);
    const double t2 = z * fma_c(w, fma_c(w, fma_c(w, Lg7, Lg5), Lg3), Lg1);
    const double R = t2 + t1;
    const double dk = (double)k;
    const double lo = fma(s, hfsq + R, dk * ln2_lo);
    return fma(dk, ln2_hi, f - (hfsq - lo));
}

// sin(2 pi u) and cos(2 pi u) for 0 <= u <= 1: 4u = q + r with q the nearest integer (exact),
// the kernels on (pi/2) r in [-pi/4, pi/4], then the quadrant
BMC_HD void sincos_2pi(double u, double& sn, double& cs) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double t = 4.0 * u;
    const double q = rint(t);
    const double r = t - q;                         // exact, |r| <= 1/2
    const double x = r * 1.57079632679489661923;    // |x| <= pi/4
    const double z = x * x;
    const double v = z * x;
    const double rs = fma_c(z, fma_c(z, fma_c(z, fma_c(z, S6, S5), S4), S3), S2);
    const double ksin = fma(v, fma_c(z, rs, S1), x);
    const double w = z * z;
    const double rc = fma(w * w, fma_c(z, fma_c(z, C6, C5), C4), z * fma_c(z, fma_c(z, C3, C2), C1));
    const double hz = 0.5 * z;
    const double w1 = 1.0 - hz;
    const double kcos = w1 + fma(z, rc, (1.0 - w1) - hz);
    const int iq = (int)q & 3;
    const double a = (iq & 1) ? kcos : ksin;        // |sin| of the quadrant
    const double b = (iq & 1) ? ksin : kcos;
    sn = (iq & 2) ? -a : a;
    cs = ((iq + 1) & 2) ? -b : b;
}

// two independent N(0,1) variates from two uniforms, u1 in (0, 1] and u2 in [0, 1]  (Box-Muller)
BMC_HD void box_muller_pair(double u1, double u2, double& z0, double& z1) {
#if defined(BMC_LIBM_NORMALS) && defined(__HIPCC__)   // A/B builds: the general-purpose library calls
    const double rad = sqrt(-2.0 * log(u1));
    double sn, cs;
    sincospi(2.0 * u2, &sn, &cs);
#else
    const double rad = sqrt(-2.0 * log_normal_arg(u1));
    double sn, cs;
    sincos_2pi(u2, sn, cs);
#endif
    z0 = rad * cs;
    z1 = rad * sn;
}

}  // namespace bmc
