// CPU check of pybmc_amd/csrc/bmc_math.h (the same text the gfx950 kernels compile): worst
// error, in units in the last place, of log on (0, 1] and of sin / cos (2 pi u) against
// long-double libm.  Prints three numbers and the special values; tests/test_host_math.py reads them.
#include "../pybmc_amd/csrc/bmc_math.h"
#include <cmath>
#include <cstdio>
#include <random>

static double ulps(double got, double want) {
    if (got == want) return 0.0;
    const double u = std::fabs(std::nextafter(want, INFINITY) - want);
    return std::fabs(got - want) / u;
}

int main() {
    std::mt19937_64 g(1);
    double wl = 0.0, ws = 0.0, wc = 0.0;
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (long i = 0; i < 3000000; ++i) {
        const uint64_t r = g();
        double u = (double)((r >> 11) + 1) * (1.0 / 9007199254740992.0);   // u53 in (0, 1]
        if (i % 7 == 0) u = std::ldexp(u, -(int)(r & 31));                  // small arguments too
        if (u < 1.2e-16) u = 1.2e-16;
        wl = std::fmax(wl, ulps(bmc::log_normal_arg(u), (double)logl((long double)u)));
        const double v = (double)(g() >> 11) * (1.0 / 9007199254740992.0);
        double sn, cs;
        bmc::sincos_2pi(v, sn, cs);
        const double rs = (double)sinl(two_pi * (long double)v), rc = (double)cosl(two_pi * (long double)v);
        // near their zeros the error of sin / cos is measured in ulps of 1
        ws = std::fmax(ws, std::fabs(rs) > 1e-3 ? ulps(sn, rs) : std::fabs(sn - rs) / 2.220446049250313e-16);
        wc = std::fmax(wc, std::fabs(rc) > 1e-3 ? ulps(cs, rc) : std::fabs(cs - rc) / 2.220446049250313e-16);
    }
    std::printf("%.4f %.4f %.4f\n", wl, ws, wc);
    const double us[5] = {0.0, 0.25, 0.5, 0.75, 1.0};
    for (double u : us) {
        double sn, cs;
        bmc::sincos_2pi(u, sn, cs);
        std::printf("%.17g %.17g\n", sn, cs);
    }
    std::printf("%.17g %.17g\n", bmc::log_normal_arg(1.0), bmc::log_normal_arg(std::ldexp(1.0, -53)));
    double z0, z1;
    bmc::box_muller_pair(0.5, 0.125, z0, z1);
    std::printf("%.17g %.17g\n", z0, z1);
    return 0;
}
