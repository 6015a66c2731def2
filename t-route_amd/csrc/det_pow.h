/*
 * det_pow.h -- bit-reproducible x**y for the fp32 Muskingum-Cunge path.
 *
 * Why: the reference kernel calls libm powf 8-9 times per segment-step
 * (R**(2/3), R**(5/3), ... MCsingleSegStime_f2py_NOLOOP.f90:169,:251-264,:328,
 * :356-362).  Every other operation it uses (+ - * / sqrt, comparisons) is
 * IEEE-exact and therefore identical on x86 and on gfx950; powf is the one
 * operation whose last bit depends on the library (glibc: 0.82 ulp, ocml:
 * different rounding).  Without the short-timestep assumption the reference
 * recurrence is numerically chaotic from a cold start (a 1e-15 relative input
 * change moves p99 of the flows by 20 % in fp64), so "close" is not testable
 * there -- only bit-equality is.  This header defines powf in terms of
 * operations that ARE exactly reproducible:
 *
 *     powf(x, y) := RN_float( exp2_d( (double)y * log2_d((double)x) ) )
 *
 * with log2_d / exp2_d built from IEEE double + - * / and fma only, in a fixed
 * order (no libm, no hardware transcendental, no tables).  Their error is
 * ~1e-15, so the float result is the correctly rounded power except with
 * probability ~1e-7 per call: at least as accurate as the libm it stands in for.
 *
 * The same source is compiled for the device (HIP) and, by the TEST oracle
 * (oracle/mc_oracle.c, "det" instantiation), for the host, which is what makes
 * GPU-vs-oracle comparisons bit-exact.  Requires -ffp-contract=off (all fused
 * operations are written explicitly) and round-to-nearest.
 *
 * Plain C99 / C++ / HIP.
 */
#ifndef TRMC_DET_POW_H
#define TRMC_DET_POW_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define TRMC_DP_FN __host__ __device__ __forceinline__
#else
#define TRMC_DP_FN static inline
#endif

TRMC_DP_FN uint64_t trmc_dp_bits(double x)
{
    uint64_t u;
    memcpy(&u, &x, sizeof u);
    return u;
}
TRMC_DP_FN double trmc_dp_from_bits(uint64_t u)
{
    double x;
    memcpy(&x, &u, sizeof x);
    return x;
}

/* log2(x) for a double that was converted from a float (so: never a double
 * subnormal).  x < 0 or NaN -> NaN ; 0 -> -inf ; +inf -> +inf. */
TRMC_DP_FN double trmc_det_log2(double x)
{
    const uint64_t b = trmc_dp_bits(x);
    if (x != x || (b >> 63)) {               /* NaN or negative (incl. -0: pow(-0,y>0)=0 handled below) */
        if (x == 0.0) return trmc_dp_from_bits(0xfff0000000000000ull); /* -0 -> -inf */
        return trmc_dp_from_bits(0x7ff8000000000000ull);
    }
    if (x == 0.0) return trmc_dp_from_bits(0xfff0000000000000ull);
    if (b == 0x7ff0000000000000ull) return x;

    int64_t e = (int64_t)((b >> 52) & 0x7ff) - 1023;
    uint64_t mant = b & 0x000fffffffffffffull;
    uint64_t mb = mant | 0x3ff0000000000000ull;             /* m in [1,2) */
    if (mant > 0x6a09e667f3bcdull) {                        /* m > sqrt(2): halve */
        mb = mant | 0x3fe0000000000000ull;
        e += 1;
    }
    const double m = trmc_dp_from_bits(mb);                 /* [sqrt(2)/2, sqrt(2)] */
    const double f = m - 1.0;                               /* exact */
    const double s = f / (2.0 + f);                         /* |s| <= 0.1716 */
    const double z = s * s;
    /* atanh series: log(1+f) = 2s (1 + z/3 + z^2/5 + ... + z^8/17), remainder < 1e-15 rel */
    double q = 1.0 / 17.0;
    q = __builtin_fma(z, q, 1.0 / 15.0);
    q = __builtin_fma(z, q, 1.0 / 13.0);
    q = __builtin_fma(z, q, 1.0 / 11.0);
    q = __builtin_fma(z, q, 1.0 / 9.0);
    q = __builtin_fma(z, q, 1.0 / 7.0);
    q = __builtin_fma(z, q, 1.0 / 5.0);
    q = __builtin_fma(z, q, 1.0 / 3.0);
    const double p = __builtin_fma(z, q, 1.0);
    const double lg = (s + s) * p;                          /* ln(m) */
    return __builtin_fma(lg, 1.4426950408889634 /* log2(e) */, (double)e);
}

/* 2**p as a double, for |p| small enough that the result is a normal double
 * (callers clamp).  NaN -> NaN. */
TRMC_DP_FN double trmc_det_exp2(double p)
{
    if (p != p) return p;
    if (p > 1000.0) return trmc_dp_from_bits(0x7ff0000000000000ull);
    if (p < -1000.0) return 0.0;
    const int64_t n = (int64_t)(p + (p >= 0.0 ? 0.5 : -0.5)); /* round half away, exact */
    const double r = p - (double)n;                           /* exact, |r| <= 0.5 */
    const double u = r * 0.6931471805599453;                  /* ln 2 */
    /* e**u, |u| <= 0.3466, Taylor to u^13/13! (remainder < 2e-17) */
    double t = 1.0 / 6227020800.0;
    t = __builtin_fma(u, t, 1.0 / 479001600.0);
    t = __builtin_fma(u, t, 1.0 / 39916800.0);
    t = __builtin_fma(u, t, 1.0 / 3628800.0);
    t = __builtin_fma(u, t, 1.0 / 362880.0);
    t = __builtin_fma(u, t, 1.0 / 40320.0);
    t = __builtin_fma(u, t, 1.0 / 5040.0);
    t = __builtin_fma(u, t, 1.0 / 720.0);
    t = __builtin_fma(u, t, 1.0 / 120.0);
    t = __builtin_fma(u, t, 1.0 / 24.0);
    t = __builtin_fma(u, t, 1.0 / 6.0);
    t = __builtin_fma(u, t, 0.5);
    t = __builtin_fma(u, t, 1.0);
    t = __builtin_fma(u, t, 1.0);                             /* in (0.70, 1.42) */
    return trmc_dp_from_bits(trmc_dp_bits(t) + ((uint64_t)n << 52)); /* * 2**n, exact */
}

/* x**y for y > 0 from the precomputed L = trmc_det_log2((double)x). */
TRMC_DP_FN float trmc_det_powf_from_log(double L, float y)
{
    return (float)trmc_det_exp2((double)y * L);
}

TRMC_DP_FN float trmc_det_powf(float x, float y)
{
    return trmc_det_powf_from_log(trmc_det_log2((double)x), y);
}

#endif /* TRMC_DET_POW_H */
