/*
 * det_pow.h -- bit-reproducible powf for the fp32 Muskingum-Cunge path:
 * a restatement of the powf of the C library the reference links against.
 *
 * Why: the reference kernel calls libm powf 8-9 times per segment-step
 * (R**(2/3), R**(5/3), ... MCsingleSegStime_f2py_NOLOOP.f90:169,:251-264,:328,
 * :356-362).  Every other operation it uses (+ - * / sqrt, comparisons) is
 * IEEE-exact and therefore identical on x86 and on gfx950; powf is the one
 * operation whose last bit depends on the library.  Without the short-timestep
 * assumption the reference recurrence is numerically chaotic from a cold
 * start, so "close" is not testable there -- only bit-equality is.
 *
 * Third-party algorithm restated here (not part of the reference tree):
 *   GNU C Library 2.35 (Ubuntu GLIBC 2.35-0ubuntu3.11, the libm the reference
 *   Fortran links in this image), powf:
 *     sysdeps/ieee754/flt-32/e_powf.c            (log2_inline, exp2_inline, powf)
 *     sysdeps/ieee754/flt-32/e_powf_log2_data.c  (16-entry {1/c, log2 c} table, degree-5 polynomial)
 *     sysdeps/ieee754/flt-32/e_exp2f_data.c      (32-entry 2**(i/32) table, degree-3 polynomial)
 *   (upstream: ARM optimized-routines, math/powf.c; published error 0.82 ulp).
 *   The x86-64 build dispatches to the variant compiled with FMA contraction
 *   (sysdeps/x86_64/fpu/multiarch/e_powf.c) on every FMA-capable CPU, so each
 *   a*b+c below is ONE fused operation, written explicitly.
 *   powf(x,y) = (float) exp2_d( (double)y * log2_d(x) ), all in IEEE double:
 *     log2_d : x = 2**k * z, z in [0x1.66p-1, 0x1.66p0); table entry i from the
 *              top 4 mantissa bits; r = z/c - 1; degree-5 polynomial in r
 *     exp2_d : p = n/32 + r; table entry n mod 32 with the exponent n div 32
 *              added into its bits; degree-3 polynomial in r
 *   Since only IEEE double operations and integer bit manipulation occur, the
 *   same source gives the same bits on the host and on the device, and
 *   tests/powf_exhaustive.c shows it equals this image's libm powf for EVERY
 *   float x at the two exponents the kernel uses.
 *
 * The tables live in caller-provided memory so the device can keep them in LDS
 * (64 doubles = 512 B): layout  tab[0..31]  = {invc_i, logc_i} i = 0..15
 *                               tab[32..63] = bits of exp2 table as doubles' storage
 *
 * Requires -ffp-contract=off (every fused operation is explicit) and
 * round-to-nearest.  Plain C99 / C++ / HIP.
 *
 * Provenance and licence of the restated third-party material: the algorithm and the table / coefficient values are
 * those of the GNU C Library 2.35 (LGPL-2.1-or-later), whose powf / pow are the ARM Optimized Routines implementations
 * (math/powf.c, math/pow.c and their data files; Copyright (c) 2017-2018 Arm Limited; SPDX MIT OR Apache-2.0 WITH
 * LLVM-exception).  Nothing here comes from the T-Route tree.
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

#define TRMC_POW_TAB_WORDS 64 /* 64-bit words */

/* glibc 2.35 __powf_log2_data.tab ({invc, logc}) followed by __exp2f_data.tab */
#define TRMC_POW_TAB_VALUES {\
    0x3ff661ec79f8f3beull, 0xbfdefec65b963019ull, 0x3ff571ed4aaf883dull, 0xbfdb0b6832d4fca4ull,\
    0x3ff49539f0f010b0ull, 0xbfd7418b0a1fb77bull, 0x3ff3c995b0b80385ull, 0xbfd39de91a6dcf7bull,\
    0x3ff30d190c8864a5ull, 0xbfd01d9bf3f2b631ull, 0x3ff25e227b0b8ea0ull, 0xbfc97c1d1b3b7af0ull,\
    0x3ff1bb4a4a1a343full, 0xbfc2f9e393af3c9full, 0x3ff12358f08ae5baull, 0xbfb960cbbf788d5cull,\
    0x3ff0953f419900a7ull, 0xbfaa6f9db6475fceull, 0x3ff0000000000000ull, 0x0000000000000000ull,\
    0x3fee608cfd9a47acull, 0x3fb338ca9f24f53dull, 0x3feca4b31f026aa0ull, 0x3fc476a9543891baull,\
    0x3feb2036576afce6ull, 0x3fce840b4ac4e4d2ull, 0x3fe9c2d163a1aa2dull, 0x3fd40645f0c6651cull,\
    0x3fe886e6037841edull, 0x3fd88e9c2c1b9ff8ull, 0x3fe767dcf5534862ull, 0x3fdce0a44eb17bccull,\
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,\
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,\
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,\
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,\
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,\
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,\
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,\
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull,\
}
static const uint64_t trmc_pow_tab_init[TRMC_POW_TAB_WORDS] = TRMC_POW_TAB_VALUES;

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
TRMC_DP_FN uint32_t trmc_sp_bits(float x)
{
    uint32_t u;
    memcpy(&u, &x, sizeof u);
    return u;
}
TRMC_DP_FN float trmc_sp_from_bits(uint32_t u)
{
    float x;
    memcpy(&x, &u, sizeof x);
    return x;
}

/* 2**(k/32) with the exponent of k added in: the double whose bits are  t + (ki << 47)  (glibc: `t += ki << 47`).  The
 * shifted term has no bits below 2**47, so the sum is an addition into the HIGH word alone; on the device that is ONE
 * 32-bit shift-add (the compiler makes a shift, a move and a 64-bit add of the plain form). */
TRMC_DP_FN double trmc_dp_exp2_scale(uint64_t t, uint64_t ki)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t hi;
    asm("v_lshl_add_u32 %0, %1, 15, %2" : "=v"(hi) : "v"((uint32_t)ki), "v"((uint32_t)(t >> 32)));
    return __hiloint2double((int)hi, (int)(uint32_t)t);
#else
    return trmc_dp_from_bits(t + (ki << 47));
#endif
}

/* log2(x) as glibc's powf forms it, plus the special values its callers rely on:
 * NaN or finite x < 0 -> NaN ; +-0 -> -inf ; +-inf -> +inf (y is a positive non-integer). */
TRMC_DP_FN double trmc_det_log2(float x, const uint64_t *tab)
{
    uint32_t ix = trmc_sp_bits(x);
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) { /* x < 0x1p-126, inf or nan */
        if ((ix << 1) == 0) return trmc_dp_from_bits(0xfff0000000000000ull);          /* +-0 */
        if ((ix << 1) == 0xff000000u) return trmc_dp_from_bits(0x7ff0000000000000ull); /* +-inf */
        if ((ix << 1) > 0xff000000u || (ix >> 31)) return trmc_dp_from_bits(0x7ff8000000000000ull);
        ix = trmc_sp_bits(x * 8388608.0f /* 0x1p23f */); /* subnormal: normalise */
        ix &= 0x7fffffffu;
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const uint32_t i = (tmp >> 19) & 15u;
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int32_t k = (int32_t)top >> 23;
    const double invc = trmc_dp_from_bits(tab[2 * i]);
    const double logc = trmc_dp_from_bits(tab[2 * i + 1]);
    const double z = (double)trmc_sp_from_bits(iz);
    const double r = __builtin_fma(z, invc, -1.0);
    const double y0 = logc + (double)k;
    const double r2 = r * r;
    double y = __builtin_fma(0x1.27616c9496e0bp-2, r, -0x1.71969a075c67ap-2);
    const double p = __builtin_fma(0x1.ec70a6ca7baddp-2, r, -0x1.7154748bef6c8p-1);
    const double r4 = r2 * r2;
    double q = __builtin_fma(0x1.71547652ab82bp0, r, y0);
    q = __builtin_fma(p, r2, q);
    y = __builtin_fma(y, r4, q);
    return y;
}

/* x**y (y > 0, not an integer) from L = trmc_det_log2(x). */
TRMC_DP_FN float trmc_det_powf_from_log(double L, float y, const uint64_t *tab)
{
    const double ylogx = (double)y * L;
    if (!(__builtin_fabs(ylogx) < 126.0)) { /* one test on the common path; the three cases of e_powf.c behind it */
        if (ylogx != ylogx) return trmc_sp_from_bits(0x7fc00000u);
        if (ylogx > 0x1.fffffffd1d571p+6) return trmc_sp_from_bits(0x7f800000u); /* overflow */
        if (ylogx <= -150.0) return 0.0f;                                        /* underflow */
    }
    double kd = ylogx + 0x1.8p+47;            /* round to a multiple of 1/32 */
    const uint64_t ki = trmc_dp_bits(kd);
    kd -= 0x1.8p+47;
    const double r = ylogx - kd;
    const double s = trmc_dp_exp2_scale(tab[32 + (ki & 31u)], ki);
    const double z = __builtin_fma(0x1.c6af84b912394p-5, r, 0x1.ebfce50fac4f3p-3);
    const double r2 = r * r;
    double w = __builtin_fma(0x1.62e42ff0c52d6p-1, r, 1.0);
    w = __builtin_fma(z, r2, w);
    w = w * s;
    return (float)w;
}

/* The same two steps for arguments whose range the caller has established, without the range tests (they only select
 * special values, so the results are the ones above):
 *   trmc_det_log2_normal          x positive, finite, not subnormal
 *   trmc_det_powf_from_log_inrange   |y * L| < 126 (neither overflow nor underflow nor NaN)
 * mc_segment.hpp uses them for the hydraulic radius of a channel whose parameters and depth lie in the ranges of
 * DevMathF::fast_ok (trmc.hip), where 2**-64 < R < 2**63 is shown. */
TRMC_DP_FN double trmc_det_log2_normal(float x, const uint64_t *tab)
{
    const uint32_t ix = trmc_sp_bits(x);
    const uint32_t tmp = ix - 0x3f330000u;
    const uint32_t i = (tmp >> 19) & 15u;
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int32_t k = (int32_t)top >> 23;
    const double invc = trmc_dp_from_bits(tab[2 * i]);
    const double logc = trmc_dp_from_bits(tab[2 * i + 1]);
    const double z = (double)trmc_sp_from_bits(iz);
    const double r = __builtin_fma(z, invc, -1.0);
    const double y0 = logc + (double)k;
    const double r2 = r * r;
    double y = __builtin_fma(0x1.27616c9496e0bp-2, r, -0x1.71969a075c67ap-2);
    const double p = __builtin_fma(0x1.ec70a6ca7baddp-2, r, -0x1.7154748bef6c8p-1);
    const double r4 = r2 * r2;
    double q = __builtin_fma(0x1.71547652ab82bp0, r, y0);
    q = __builtin_fma(p, r2, q);
    y = __builtin_fma(y, r4, q);
    return y;
}
TRMC_DP_FN float trmc_det_powf_from_log_inrange(double L, float y, const uint64_t *tab)
{
    const double ylogx = (double)y * L;
    double kd = ylogx + 0x1.8p+47;
    const uint64_t ki = trmc_dp_bits(kd);
    kd -= 0x1.8p+47;
    const double r = ylogx - kd;
    const double s = trmc_dp_exp2_scale(tab[32 + (ki & 31u)], ki);
    const double z = __builtin_fma(0x1.c6af84b912394p-5, r, 0x1.ebfce50fac4f3p-3);
    const double r2 = r * r;
    double w = __builtin_fma(0x1.62e42ff0c52d6p-1, r, 1.0);
    w = __builtin_fma(z, r2, w);
    w = w * s;
    return (float)w;
}

TRMC_DP_FN float trmc_det_powf(float x, float y, const uint64_t *tab)
{
    return trmc_det_powf_from_log(trmc_det_log2(x, tab), y, tab);
}

#endif /* TRMC_DET_POW_H */
