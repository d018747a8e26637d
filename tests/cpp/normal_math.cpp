// Accuracy of csrc/kb_normal.h (the Box-Muller pieces the device and the host replay share) against long double, and the bits of
// a few fixed arguments (printed for tests/test_normal_math_cpu.py, which also pins them: the GPU test compares the device's bits
// with the host's through kb_noise_sample).
// g++ -O2 -std=c++17 -ffp-contract=off tests/cpp/normal_math.cpp -o normal_math
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <random>

#include "../../gokalman_amd/csrc/kb_normal.h"

static double ulp(double x) {
    x = std::fabs(x);
    return std::nextafter(x, INFINITY) - x;
}

int main() {
    std::mt19937_64 rng(12345);
    const double two53 = 1.0 / 9007199254740992.0;
    double worst_log = 0, worst_sc = 0;
    const long double pil = 3.141592653589793238462643383279502884L;
    auto check = [&](uint64_t a, uint64_t b) {
        const double u = ((double)a + 1.0) * two53, v = (double)b * two53;
        const double got = kb::neg2log(u);
        const long double want = -2.0L * logl((long double)u);
        const double e1 = want == 0 ? std::fabs(got) : (double)(fabsl((long double)got - want) / (long double)ulp((double)want));
        if (e1 > worst_log) worst_log = e1;
        double s, c;
        kb::sincos2pi(v, s, c);
        // reference: the same exact reduction (t = 2 v, nearest quarter turn), then long double sin / cos of pi r and the rotation
        const double t = v + v;
        const int q = (int)(t + t + 0.5);
        const long double r = (long double)t - 0.5L * q;
        const long double s0 = sinl(pil * r), c0 = cosl(pil * r);
        long double ws, wc;
        switch (q & 3) {
        case 0: ws = s0; wc = c0; break;
        case 1: ws = c0; wc = -s0; break;
        case 2: ws = -s0; wc = -c0; break;
        default: ws = -c0; wc = s0; break;
        }
        for (int which = 0; which < 2; which++) {
            const long double w = which ? wc : ws;
            const double g = which ? c : s;
            const double e = w == 0 ? std::fabs(g) / 1.1102230246251565e-16 : (double)(fabsl((long double)g - w) / (long double)ulp((double)w));
            if (e > worst_sc) worst_sc = e;
        }
    };
    const unsigned long long top = (1ull << 53) - 1;
    for (unsigned long long a : {0ull, 1ull, 2ull, 3ull, top, top - 1, top / 2, top / 2 + 1, (1ull << 52), (1ull << 52) - 1, (unsigned long long)(0.70710678118654752 * 9007199254740992.0)})
        for (unsigned long long b : {0ull, 1ull, top, top / 2, top / 4, top / 4 + 1, 3 * (top / 4), top / 8, (1ull << 50), (1ull << 51), (1ull << 52), 3 * (1ull << 51)})
            check(a, b);
    for (int i = 0; i < 4000000; i++) check(rng() >> 11, rng() >> 11);
    for (int i = 0; i < 200000; i++) check((rng() >> 11) >> (i % 50), (rng() >> 11) >> (i % 50));   // small u (large radii), angles near 0
    for (int i = 0; i < 200000; i++) check(top - ((rng() >> 11) >> (12 + i % 40)), top - ((rng() >> 11) >> (12 + i % 40)));   // u near 1, angles near 2 pi
    std::printf("worst_neg2log_ulp %.3f\nworst_sincos_ulp %.3f\n", worst_log, worst_sc);
    std::printf("neg2log_of_1 %.17g\n", kb::neg2log(1.0));
    double s, c;
    for (double v : {0.0, 0.125, 0.25, 0.5, 0.75, 0.3, 0.9999999999999999}) {
        kb::sincos2pi(v, s, c);
        std::printf("sincos %.17g %016llx %016llx\n", v, (unsigned long long)kb::normal_bits(s), (unsigned long long)kb::normal_bits(c));
    }
    for (double u : {1.0, 0.5, 0.25, 0.7071067811865476, 1e-3, 1.1102230246251565e-16, 0.9999999999999999}) std::printf("neg2log %.17g %016llx\n", u, (unsigned long long)kb::normal_bits(kb::neg2log(u)));
    return 0;
}
