// kb_normal.h -- the two transcendental pieces of the Box-Muller transform, written for THIS use instead of calling the math
// library: on gfx950 `log` is 94 VALU instructions, `sincospi` 64 -- together 160 of the ~200 a pair of normals costs, and the
// Monte-Carlo / chi-square kernels are VALU-issue-bound on exactly that.  Both arguments have a known range and a known form:
//   neg2log(u)    -2 ln u for u = k 2^-53, k = 1 .. 2^53, i.e. u in (0, 1]            (~40 instructions)
//   sincos2pi(v)  sin(2 pi v), cos(2 pi v) for v = k 2^-53, k = 0 .. 2^53 - 1        (~35 instructions)
// Only +, -, *, correctly rounded division and explicit fused multiply-adds are used, under `fp contract(off)`, so the host
// (kb_noise_sample, the replay the tests and the chi-square property checks rely on) and the device produce the SAME BITS.
// Accuracy (tests/test_normal_math_cpu.py, against long double): neg2log <= 1 ulp, sincos2pi <= 1 ulp of the larger of the two
// results in absolute terms (|error| <= 1.2e-16): far inside what a Monte-Carlo draw needs.
//
// Plain C++ (g++ compiles it for the CPU test); KB_HD becomes __host__ __device__ under hipcc.
#pragma once
#include <stdint.h>
#include <string.h>

#ifndef KB_HD
#ifdef __HIPCC__
#define KB_HD __host__ __device__
#else
#define KB_HD
#endif
#endif

namespace kb {

KB_HD inline uint64_t normal_bits(double x) { uint64_t b; memcpy(&b, &x, 8); return b; }
KB_HD inline double normal_from_bits(uint64_t b) { double x; memcpy(&x, &b, 8); return x; }

// -2 ln u, u in (0, 1] a normal double.  u = 2^e m, m in [sqrt(1/2), sqrt(2)); f = m - 1, s = f / (2 + f),
// ln m = f - f^2/2 + s (f^2/2 + R(s^2)) with the degree-7 polynomial of the classic fdlibm log (error of R < 2^-58.45).
KB_HD inline double neg2log(double u) {
#pragma clang fp contract(off)
    uint64_t b = normal_bits(u);
    int e = (int)(b >> 52) - 1023;
    b = (b & 0x000fffffffffffffull) | 0x3ff0000000000000ull;   // m in [1, 2)
    if (b >= 0x3ff6a09e667f3bcdull) {                            // m >= sqrt(2): halve it
        b -= 0x0010000000000000ull;
        e += 1;
    }
    const double f = normal_from_bits(b) - 1.0;                  // exact
    const double s = f / (2.0 + f);
    const double z = s * s, w = z * z;
    const double t1 = w * __builtin_fma(w, __builtin_fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01), 6.666666666666735130e-01);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)e;
    // ln u = dk ln2_hi - ((hfsq - (s (hfsq + R) + dk ln2_lo)) - f) (fdlibm's assembly order); negated here so that u = 1 gives +0
    const double nln = ((hfsq - __builtin_fma(s, hfsq + R, dk * 1.90821492927058770002e-10)) - f) - dk * 6.93147180369123816490e-01;
    return 2.0 * nln;
}

// sin(2 pi v), cos(2 pi v), v in [0, 1): t = 2 v in [0, 2) -> nearest multiple q / 2 of a quarter turn, r = t - q / 2 in
// [-1/4, 1/4] (exact), then the Taylor polynomials of sin(pi r), cos(pi r) (|pi r| <= pi / 4: truncation < 5e-17).
KB_HD inline void sincos2pi(double v, double &sn, double &cs) {
#pragma clang fp contract(off)
    const double t = v + v;
    const double qd = (double)(int)(t + t + 0.5);               // 0 .. 4
    const int q = (int)qd;
    const double r = __builtin_fma(qd, -0.5, t);                 // exact
    const double x2 = r * r;
    // sin(pi r) / r = pi - pi^3/3! r^2 + ... ; cos(pi r) = 1 - pi^2/2! r^2 + ...
    double ps = 7.952054001475513e-07;                        // pi^17/17!
    ps = __builtin_fma(ps, x2, -2.1915353447830217e-05);   // -pi^15/15!
    ps = __builtin_fma(ps, x2, 0.00046630280576761255);   // pi^13/13!
    ps = __builtin_fma(ps, x2, -0.0073704309457143504);   // -pi^11/11!
    ps = __builtin_fma(ps, x2, 0.08214588661112823);   // pi^9/9!
    ps = __builtin_fma(ps, x2, -0.5992645293207921);   // -pi^7/7!
    ps = __builtin_fma(ps, x2, 2.5501640398773455);   // pi^5/5!
    ps = __builtin_fma(ps, x2, -5.16771278004997);   // -pi^3/3!
    ps = __builtin_fma(ps, x2, 3.141592653589793);   // pi^1/1!
    const double s = ps * r;
    double pc = -1.3878952462213771e-07;   // -pi^18/18!
    pc = __builtin_fma(pc, x2, 4.303069587032947e-06);   // pi^16/16!
    pc = __builtin_fma(pc, x2, -0.0001046381049248457);   // -pi^14/14!
    pc = __builtin_fma(pc, x2, 0.0019295743094039231);   // pi^12/12!
    pc = __builtin_fma(pc, x2, -0.02580689139001406);   // -pi^10/10!
    pc = __builtin_fma(pc, x2, 0.2353306303588932);   // pi^8/8!
    pc = __builtin_fma(pc, x2, -1.3352627688545895);   // -pi^6/6!
    pc = __builtin_fma(pc, x2, 4.0587121264167685);   // pi^4/4!
    pc = __builtin_fma(pc, x2, -4.934802200544679);   // -pi^2/2!
    const double c = __builtin_fma(pc, x2, 1.0);
    // quarter turns: q = 0, 4: (s, c); 1: (c, -s); 2: (-s, -c); 3: (-c, s)
    const bool swap = (q & 1) != 0;
    const double a0 = swap ? c : s, a1 = swap ? s : c;
    sn = (q == 2 || q == 3) ? -a0 : a0;
    cs = (q == 1 || q == 2) ? -a1 : a1;
}

}  // namespace kb
