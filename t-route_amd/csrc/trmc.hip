// trmc.hip -- HIP kernels and C ABI of the MI355X Muskingum-Cunge engine (gfx950).
//
// Data layout in HBM (plan order = level-major, see topology.hpp):
//   params      8 SoA columns [nseg_pad] (dx bw tw twcc n ncc cs s0) + dt (scalar, or a
//               9th column when the caller's dt column is not uniform)
//   up_ptr/idx  CSR of upstream plan positions (int32)
//   level       int32 [nseg_pad]
//   qlat_tm     [nq][nseg_pad]           forcing, one coalesced row per forcing interval
//   q/v/d_tm    [nsteps+1][nseg_pad]     time-major state AND result: step t reads row
//               t-1 (own previous flow/depth, upstream previous flows) and, without the
//               short-timestep assumption, row t of upstream levels
//   out         [nseg][nsteps][3]        caller layout (mc_reach.pyx:807-813), produced
//               by an LDS-tiled transpose at the end of the route
// nseg_pad rounds nseg up to 64 so every time row starts on a 256-byte boundary.
//
// Launch structure
//   assume_short_ts = 1 : step t of every segment depends only on step t-1
//       (mc_reach.pyx:504-505, :135-136) -> one launch per timestep over all
//       routed segments.
//   assume_short_ts = 0 : (level l, step t) depends on (<l, t) and (l, t-1)
//       -> anti-diagonal wavefront d = l + t; the segments of all levels on
//       one diagonal form ONE contiguous slice of the plan order and run as
//       one launch; nlevels + nsteps - 1 launches in total.
// One thread per segment-step; all loads/stores are unit-stride except the
// upstream gather, which the intra-level ordering keeps near-monotone.
//
// No CPU fallback exists in this library.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/trmc.h"
#include "det_pow.h"
#include "det_pow64.h"
#include "mc_segment.hpp"
#include "levelpool.hpp"
#include "topology.hpp"
#include "internal.hpp"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

} // namespace
namespace trmc {
int fail_with(int code, const std::string &msg) { return fail(code, msg); }
} // namespace trmc
namespace {

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(e_ == hipErrorOutOfMemory ? TRMC_ENOMEM : TRMC_EHIP,                   \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                    \
    } while (0)

// ---------------------------------------------------------------- device math
// fp32: the bit-reproducible power of det_pow.h (IEEE double + - * / fma only), so the
// whole fp32 path is bit-comparable with the host oracle; sqrtf and / are the
// correctly rounded forms (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt).
__device__ const uint64_t d_pow_tab[TRMC_POW_TAB_WORDS] = TRMC_POW_TAB_VALUES;

// fp32: the bit-reproducible powf of det_pow.h (tables staged in LDS by the kernel), so the whole
// fp32 path is bit-comparable with the host; sqrtf and / are the correctly rounded forms (hipcc
// default -fhip-fp32-correctly-rounded-divide-sqrt).
struct DevMathF {
    const uint64_t *tab; // LDS copy of d_pow_tab
    using Log = double;
    __device__ __forceinline__ Log log_of(float x) const { return trmc_det_log2(x, tab); }
    __device__ __forceinline__ float pow_l(Log l, float, float y) const { return trmc_det_powf_from_log(l, y, tab); }
    __device__ __forceinline__ float pow(float x, float y) const { return trmc_det_powf(x, y, tab); }
    // the hydraulic radius under fast_ok's ranges: the bound is derived at fast_ok
    __device__ __forceinline__ Log log_of_r(float x, bool ok) const { return ok ? trmc_det_log2_normal(x, tab) : trmc_det_log2(x, tab); }
    __device__ __forceinline__ float pow_l_r(Log l, float, float y, bool ok) const
    {
        return ok ? trmc_det_powf_from_log_inrange(l, y, tab) : trmc_det_powf_from_log(l, y, tab);
    }
    __device__ __forceinline__ float sqrt(float x) const { return ::sqrtf(x); }
    // wave-wide AND over the active lanes: a scalar, so the branch on it is a uniform one
    __device__ __forceinline__ bool all(bool p) const { return __all(p) != 0; }
    static constexpr bool kInbank = true;

    // The four Muskingum coefficients C1..C4 = n_i / D (f90:303-312) with ONE reciprocal.
    // hipcc expands an fp32 division into  v_div_scale x2, v_rcp, the refinement
    //     y1 = fma(fma(-b, y0, 1), y0, y0);  q0 = a*y1;  q1 = fma(fma(-b, q0, a), y1, q0);
    //     q  = div_fmas(fma(-b, q1, a), y1, q1)
    // and v_div_fixup.  v_div_scale / v_div_fmas / v_div_fixup only act on operands that are zero,
    // non-finite, subnormal or more than 2**96 apart, or on a subnormal quotient (CDNA3 ISA guide,
    // V_DIV_SCALE_F32 / V_DIV_FIXUP_F32); otherwise they are the identity and the division IS the
    // refinement above.  The kernel proves once per segment-step (`coef_ok`: dt in [2**-20, 2**40] and
    // n4 = ql*dt either +0 or 2**-60 <= |n4| <= 2**60) and once per coefficient set (D <= 2**52) that
    // all four divisions are of that kind: D >= dt/2 >= 2**-21; n1 = Km*X + dt/2 in [2**-21, 2**52];
    // n2, n3 are differences of two such floats, hence +0 or of magnitude >= 2**-45; so no scaling,
    // no fix-up, every quotient normal or +0 (for which the refinement returns +0 as IEEE does).  Then
    // the same instructions are issued, minus the three helpers, and y1 once instead of four times:
    // bit-identical by construction.  Anything else takes the plain divisions.
    // The divisions of the hydraulic point (hydraulics_at: radius and composite n by the wetted perimeter, the top-width
    // term of the celerity, the reciprocal of the composite n) under the same argument.  `sane` is established once per
    // plan on the host (params_sane: bw, n, cs, twcc, ncc in [2**-14, 2**17], cs, twcc and ncc possibly 0; dt, dx, s0 as at k_of); with the in-bank and
    // over-bank depths in [2**-30, 2**17] the operands are: perimeter W = wp + wpc in [2**-14, 2**36]; area sum in
    // [2**-44, 2**51]; wp*n + wpc*ncc in [2**-28, 2**54]; the composite n in [2**-64, 2**18]; bw + 2hz in
    // [2**-14, 2**36] under 2*sqrt(1 + z*z) in [2, 2**19] -- all normal, no pair more than 2**80 apart, every quotient
    // normal: the scaling and fix-up instructions are the identity, and the refinement below IS the division.
    // The hydraulic radius under the same conditions (log_of_r / pow_l_r skip the power's special-value tests): R is
    // the mediant of the in-bank and the over-bank quotient area / wetted perimeter, so it lies between them.  In bank,
    // A/WP = (bw + h z) h / (bw + 2 h sq), sq = sqrt(1 + z*z) >= 1: since bw + 2 h sq <= 2 sq (bw + h) and
    // (bw + h z) / (bw + h) >= min(1, z), A/WP >= h min(1, z) / (2 sq) >= 2**-30 * 2**-17 / 2**15.1 > 2**-63, and
    // A/WP <= h (1 + h z / bw) <= 2**17 (1 + 2**17 2**14 2**14) < 2**63; over bank, twcc h / (twcc + 2 h) lies between
    // min(h, twcc) / 3 and h.  With one rounding of the quotient: 2**-64 < R < 2**63, a positive normal float whose
    // logarithm times 5/3 stays within +-107 -- inside the +-126 where glibc's powf takes its ordinary path.
    // The Muskingum K of an IN-BANK point (hydraulics_inbank: h <= bankfull depth, so area sum = A, perimeter = WP).
    // `sane` also bounds s0 in [2**-30, 2**10], dx in [2**-10, 2**19] and dt in [2**-20, 2**40] (params_sane).  The
    // celerity is ck = (sqrt(s0)/n) (5/3 r23 - 2/3 r53 q), r23 = R**(2/3), r53 = R**(5/3), q = 2 sq / twl.  In exact
    // arithmetic r53 q = r23 [A 2 sq / (WP twl)] and the bracket is the product of (bw + h z) / (bw + 2 h z) in [1/2, 1)
    // and 2 h sq / (bw + 2 h sq) in (0, 1): below one.  Each of the dozen roundings on the way (A, WP, twl, the two
    // quotients, the powers -- glibc's powf is within one unit in the last place --, the products) moves a term by at
    // most 2**-23 of itself, so the computed bracket is at least r23 (5/3 (1 - 2**-21) - 2/3 (1 + 2**-20)) > 0.99 r23 and at
    // most 5/3 r23 (1 + 2**-21): with r23 in [2**-43, 2**42] and sqrt(s0)/n in [2**-32, 2**19] the celerity is a positive
    // normal number in [2**-76, 2**62].  Hence max(0, ck) = ck, the guard `ck > 0` holds, and dx / ck divides a number of
    // exponent <= 18 by one of exponent >= -76: exponents less than 96 apart, numerator above 2**-103, quotient in
    // [2**-72, 2**95] -- none of the cases in which v_div_scale / v_div_fmas / v_div_fixup act (see div4), so the
    // division IS the refinement below; max(dt, K) with both operands ordinary numbers is v_max.
    bool sane;
    __device__ __forceinline__ float k_of(float dx, float ck) const { return quot(dx, ck, refined_rcp(ck)); }
    __device__ __forceinline__ float max_num(float a, float b) const { return __builtin_fmaxf(a, b); }
    // sqrt(x), correctly rounded.  hipcc expands sqrtf into: scale x by 2**32 if x < 2**-96, v_sqrt_f32 (one unit in the
    // last place), the two neighbours s-, s+ of that result with the residuals x - s- s and x - s+ s (one fma each) choosing
    // among the three, unscale, and pass zeros / infinities / NaNs through.  For an ordinary x >= 2**-96 the scaling and
    // the pass-through are the identity; what is left is issued here -- the same instructions, eight fewer.
    __device__ __forceinline__ float sqrt_r(float x, bool ok) const
    {
        if (!ok) return ::sqrtf(x);
        const float s = __builtin_amdgcn_sqrtf(x);
        const float s_dn = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
        const float r_dn = __builtin_fmaf(-s_dn, s, x), r_up = __builtin_fmaf(-s_up, s, x);
        float r = (r_dn <= 0.0f) ? s_dn : s;
        r = (r_up > 0.0f) ? s_up : r;
        return r;
    }
    __device__ __forceinline__ bool fast_ok(float h, float h_in, float h_over) const
    {
        return sane && h_in >= 0x1p-30f && h <= 0x1p17f && (h_over == 0.0f || h_over >= 0x1p-30f);
    }
    __device__ __forceinline__ static float refined_rcp(float b)
    {
        const float y0 = __builtin_amdgcn_rcpf(b);
        return __builtin_fmaf(__builtin_fmaf(-b, y0, 1.0f), y0, y0);
    }
    __device__ __forceinline__ void div2(float a1, float a2, float b, bool ok, float &q1, float &q2) const
    {
        if (ok) {
            const float y1 = refined_rcp(b);
            q1 = quot(a1, b, y1);
            q2 = quot(a2, b, y1);
        } else {
            q1 = a1 / b;
            q2 = a2 / b;
        }
    }
    __device__ __forceinline__ float div1(float a, float b, bool ok) const
    {
        if (ok) return quot(a, b, refined_rcp(b));
        return a / b;
    }
    // the remaining divisions of a secant iteration (weighting factor, K = dx / celerity, secant update): IEEE
    __device__ __forceinline__ float divx(float a, float b) const
    {
        return a / b;
    }
    __device__ __forceinline__ static float quot(float a, float b, float y1)
    {
        const float q0 = a * y1;
        const float q1 = __builtin_fmaf(__builtin_fmaf(-b, q0, a), y1, q0);
        return __builtin_fmaf(__builtin_fmaf(-b, q1, a), y1, q1);
    }
    bool coef_ok;
    __device__ __forceinline__ static float refined_quot(float a, float b, float y1)
    {
        const float q0 = a * y1;
        const float q1 = __builtin_fmaf(__builtin_fmaf(-b, q0, a), y1, q0);
        return __builtin_fmaf(__builtin_fmaf(-b, q1, a), y1, q1);
    }
    __device__ __forceinline__ void div4(float n1, float n2, float n3, float n4, float d, float &q1, float &q2,
                                         float &q3, float &q4) const
    {
        if (coef_ok && d <= 0x1p52f) {
            const float y0 = __builtin_amdgcn_rcpf(d);
            const float y1 = __builtin_fmaf(__builtin_fmaf(-d, y0, 1.0f), y0, y0);
            q1 = refined_quot(n1, d, y1);
            q2 = refined_quot(n2, d, y1);
            q3 = refined_quot(n3, d, y1);
            q4 = refined_quot(n4, d, y1);
        } else {
            q1 = n1 / d;
            q2 = n2 / d;
            q3 = n3 / d;
            q4 = n4 / d;
        }
    }
};
// The same arithmetic for the dataflow kernels, without the in-bank body: there a wavefront steps through time by itself
// and what counts is the latency of ITS step -- registers and code size -- not the instruction count of a full device
// (measured: a lone 4 096-row chain 6.6 us per step against 7.3 with the in-bank body, the general-mode CONUS day the same).
#ifndef TRMC_FLOW_INBANK
#define TRMC_FLOW_INBANK 0
#endif
struct DevMathFlow : DevMathF {
    static constexpr bool kInbank = TRMC_FLOW_INBANK != 0;
};
// fp32, TOLERANCE arithmetic (a plan created with trmc_plan_options.arithmetic = TRMC_ARITH_TOLERANCE): the power as
// exp2(y * log2(x)) on the hardware's v_log_f32 / v_exp_f32, every division as a * v_rcp_f32(b), the square root as
// v_sqrt_f32 -- each within one unit in the last place of its exact result, the power within |y log2 x| 2**-23 (some 1e-6 for
// a hydraulic radius between a millimetre and a hundred metres).  NOT bit-comparable with the reference: its results are
// stated and tested against a tolerance (include/trmc.h, tests/test_gpu_tolerance.py), and the 1 % exit of the secant
// iteration (MCsingleSegStime_f2py_NOLOOP.f90:83) turns a last-place difference into a different iteration count now and
// then.  The special values the step relies on keep their meaning: log2(0) = -inf and exp2(-inf) = 0 (a dry point's
// power is 0), a quotient by zero is inf or NaN and is discarded by the selects that guard it (mc_segment.hpp).
struct DevMathTol {
    const uint64_t *tab; // unused (no tables)
    bool sane;
    bool coef_ok;        // unused
    using Log = float;
    __device__ __forceinline__ Log log_of(float x) const { return __builtin_amdgcn_logf(x); }
    __device__ __forceinline__ float pow_l(Log l, float, float y) const { return __builtin_amdgcn_exp2f(y * l); }
    __device__ __forceinline__ float pow(float x, float y) const { return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)); }
    __device__ __forceinline__ Log log_of_r(float x, bool) const { return __builtin_amdgcn_logf(x); }
    __device__ __forceinline__ float pow_l_r(Log l, float, float y, bool) const { return __builtin_amdgcn_exp2f(y * l); }
    __device__ __forceinline__ float sqrt(float x) const { return __builtin_amdgcn_sqrtf(x); }
    __device__ __forceinline__ float sqrt_r(float x, bool) const { return __builtin_amdgcn_sqrtf(x); }
    __device__ __forceinline__ bool all(bool p) const { return __all(p) != 0; }
    static constexpr bool kInbank = true;
    // (the ranges under which the in-bank body of the hydraulic point stands for the general one are DevMathF's: its argument
    // leaves a margin of 1 % on the sign of the celerity, far more than these operations give away)
    __device__ __forceinline__ bool fast_ok(float h, float h_in, float h_over) const
    {
        return sane && h_in >= 0x1p-30f && h <= 0x1p17f && (h_over == 0.0f || h_over >= 0x1p-30f);
    }
    __device__ __forceinline__ float k_of(float dx, float ck) const { return dx * __builtin_amdgcn_rcpf(ck); }
    __device__ __forceinline__ float max_num(float a, float b) const { return __builtin_fmaxf(a, b); }
    __device__ __forceinline__ void div2(float a1, float a2, float b, bool, float &q1, float &q2) const
    {
        const float y = __builtin_amdgcn_rcpf(b);
        q1 = a1 * y;
        q2 = a2 * y;
    }
    __device__ __forceinline__ float div1(float a, float b, bool) const { return a * __builtin_amdgcn_rcpf(b); }
    __device__ __forceinline__ float divx(float a, float b) const { return a * __builtin_amdgcn_rcpf(b); }
    __device__ __forceinline__ void div4(float n1, float n2, float n3, float n4, float d, float &q1, float &q2, float &q3,
                                         float &q4) const
    {
        const float y = __builtin_amdgcn_rcpf(d);
        q1 = n1 * y;
        q2 = n2 * y;
        q3 = n3 * y;
        q4 = n4 * y;
    }
};
struct DevMathTolFlow : DevMathTol {
    static constexpr bool kInbank = TRMC_FLOW_INBANK != 0;
};
// fp64: the bit-reproducible double power of det_pow64.h (glibc 2.35 pow restated, the one the reference links when it
// is built with -fdefault-real-8: oracle/_ref/libmc_ref_qj0_f64.so), so that the fp64 path -- BASELINE configs[1] -- is
// bit-comparable with the reference too, not merely close; / and sqrt are the correctly rounded forms.
struct DevMathD {
    const uint64_t *tab; // unused
    using Log = double;
    __device__ __forceinline__ Log log_of(double x) const { return x; }
    __device__ __forceinline__ double pow_l(Log, double x, double y) const { return det_pow64(x, y); }
    __device__ __forceinline__ double pow(double x, double y) const { return det_pow64(x, y); }
    __device__ __forceinline__ Log log_of_r(double x, bool) const { return x; }
    __device__ __forceinline__ double pow_l_r(Log, double x, double y, bool) const { return det_pow64(x, y); }
    __device__ __forceinline__ double sqrt(double x) const { return ::sqrt(x); }
    __device__ __forceinline__ bool all(bool p) const { return __all(p) != 0; }
    static constexpr bool kInbank = false;
    bool coef_ok; // unused
    bool sane;    // unused
    __device__ __forceinline__ bool fast_ok(double, double, double) const { return false; }
    __device__ __forceinline__ void div2(double a1, double a2, double b, bool, double &q1, double &q2) const
    {
        q1 = a1 / b;
        q2 = a2 / b;
    }
    __device__ __forceinline__ double div1(double a, double b, bool) const { return a / b; }
    __device__ __forceinline__ double divx(double a, double b) const { return a / b; }
    // (only the in-bank body uses these, which this policy never takes: kInbank = false)
    __device__ __forceinline__ double k_of(double dx, double ck) const { return dx / ck; }
    __device__ __forceinline__ double max_num(double a, double b) const { return a > b ? a : b; }
    __device__ __forceinline__ double sqrt_r(double x, bool) const { return ::sqrt(x); }
    __device__ __forceinline__ void div4(double n1, double n2, double n3, double n4, double d, double &q1, double &q2,
                                         double &q3, double &q4) const
    {
        q1 = n1 / d;
        q2 = n2 / d;
        q3 = n3 / d;
        q4 = n4 / d;
    }
};
// every kernel that evaluates segment steps stages the 512-byte power tables into LDS first
__device__ __forceinline__ const uint64_t *stage_pow_tables(uint64_t *s_tab)
{
    if (threadIdx.x < TRMC_POW_TAB_WORDS) s_tab[threadIdx.x] = d_pow_tab[threadIdx.x];
    __syncthreads();
    return s_tab;
}
// the once-per-segment-step part of DevMathF::div4's proof obligation (see there)
__device__ __forceinline__ bool coef_guard(float dt, float ql)
{
    const float n4 = ql * dt, an4 = __builtin_fabsf(n4);
    return (dt >= 0x1p-20f) && (dt <= 0x1p40f) && ((__float_as_uint(n4) == 0u) || (an4 >= 0x1p-60f && an4 <= 0x1p60f));
}
__device__ __forceinline__ bool coef_guard(double, double) { return false; }
// TOL: the plan's arithmetic is TRMC_ARITH_TOLERANCE (fp32 plans only)
template <class T, bool TOL = false> struct DevMath;
template <> struct DevMath<float, false> { using type = DevMathF; };
template <> struct DevMath<float, true> { using type = DevMathTol; };
template <> struct DevMath<double, false> { using type = DevMathD; };

constexpr int kBlock = 256;

// The threads of a block take the block's NB positions by DESCENDING cost class (cls: one byte per position; 8 = no row): a count
// per class in LDS, a prefix over the classes, a scatter of lane numbers.  Returns the lane whose position this thread takes.
// Every thread of the block must call it (three barriers).  Order inside a class is whatever the atomics give; results do not
// depend on which thread routes a row.  `none`: the position this thread takes is one of key 8 (no row there).
template <int NB> __device__ __forceinline__ int32_t block_partition_by_class(int32_t key, bool &none)
{
    __shared__ int32_t s_cnt[9], s_base[9];
    __shared__ int16_t s_lane[NB];
    if (threadIdx.x < 9) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int32_t rank = atomicAdd(&s_cnt[key], 1);
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t acc = 0;
        for (int b = 0; b < 9; ++b) {
            s_base[b] = acc;
            acc += s_cnt[b];
        }
    }
    __syncthreads();
    s_lane[s_base[key] + rank] = (int16_t)threadIdx.x;
    __syncthreads();
    none = (int32_t)threadIdx.x >= s_base[8];
    return (int32_t)s_lane[threadIdx.x];
}

// element of a column at a 32-bit BYTE offset from a (wave-uniform) base pointer
template <class T> __device__ __forceinline__ T &at(T *base, uint32_t byte_off)
{
    return *reinterpret_cast<T *>(reinterpret_cast<char *>(base) + byte_off);
}
template <class T> __device__ __forceinline__ const T &at(const T *base, uint32_t byte_off)
{
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off);
}

// ---------------------------------------------------------------- kernels
// COLD kernel arguments.  hipcc reads the arguments a kernel uses into SGPRs at entry, in tuples of up to sixteen, and when
// the arithmetic of a step needs those registers (the power's double constants alone take forty) it parks the tuples in
// VGPR lanes and reads them back WHOLE -- eight v_readlane for one pointer -- wherever a member is used: the lean kernel
// executed some 120 of them per step (the tables of the rare branches, the watchdog of every poll loop, what the epilogue
// stores).  What the time loop needs only in rare branches or after its end is therefore read where it is used, through an
// opaque pointer to the kernel-argument segment: one scalar load from the constant cache, no register held across the loop.
// (The argument struct is the kernel's first parameter: offset 0 of the segment.)
template <class A> using ColdArgs = const __attribute__((address_space(4))) A *;
template <class A> __device__ __forceinline__ ColdArgs<A> cold_args()
{
    ColdArgs<A> p = (ColdArgs<A>)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}
template <class T> struct StepArgs {
    const T *dx, *bw, *twcc, *n, *ncc, *s0;            // raw channel parameters the step still reads
    const T *z, *bfd, *sqrt_s0, *sq1pz2, *s0_n, *s0_ncc, *inv_n; // segment-invariant constants formed at plan time (k_make_const)
    const T *dt_col; // nullptr -> uniform dt
    T dt;
    const int32_t *up_ptr, *up_idx, *level;
    const int2 *up2; // first two upstream positions of every position (see k_mc_step)
    const int32_t *lag; // LAG form of the short-timestep kernel: position s is at step diag - lag[s]
    const T *qlat_tm;
    T *q_tm, *v_tm, *d_tm;
    uint8_t *it_prev; // secant iterations each position needed on the LAST step of the window (trmc_download_iterations)
    uint16_t *it_sum; // nullptr, or: sum over the window of min(iterations, 3) per position (trmc_plan_collect_cost)
    bool sane;        // every channel parameter of the plan lies in the range DevMathF::fast_ok's argument needs
    // level-pool reservoirs (nullptr = none): reservoir index of a position, parameters [nres][9],
    // inflow series [nres][nsteps] (the reference's upstream_array rows), routing period
    const int32_t *res_of_pos;
    const T *res_par;
    T *res_inflow;
    T res_dt;
    // streamflow nudging at gage positions (nullptr = off), tables [gage][nsteps], see trmc_set_nudging
    const int32_t *gage_of_pos;
    const uint8_t *da_mode;
    const T *da_a, *da_w;
    T *da_nudge;
    // gages INSIDE a reach, general mode only (trmc_set_nudging_successors): the segment below such a gage reads the gage
    // segment's flow of the current step as it was BEFORE the nudge (the reference nudges after the whole reach,
    // mc_reach.pyx:133-137,:761-796): raw_of_pos = gage whose raw flow a position reads (-1 = none), da_raw [gage][nsteps]
    const int32_t *raw_of_pos;
    T *da_raw;
    int64_t nseg_pad;
    int32_t nsteps, qts;
    // k_mc_tile writes its rows' results straight into the caller's layout out[row][step][q,v,d]
    T *out;
    const int32_t *row_of_pos;
    bool out_vec; // the runs it writes are 16-byte aligned (float, nsteps % 4 == 0, K % 4 == 0)
    // k_mc_tile: non-null = the threads of a block take the block's rows by descending cost class (see the kernel's prologue):
    // the class of every row at the last step it was routed in a tile, min(iterations, 3) + 4 if over bank
    uint8_t *cls_last;
    // k_mc_tile<.., DEC>: every dec_stride-th step of its rows' (q, v, d) also goes to dec[row][k][3], k = t / dec_stride - 1 <
    // dec_keep (trmc_plan_set_output_stride: what the reference's writers take of a window, written where it is produced)
    T *dec;
    int32_t dec_stride, dec_keep;
    // k_mc_tile, hot rows: rows of class >= 3 at the end of a tile are routed by blocks of their own in the next one.  Three
    // lists of positions [3][hot_cap] and their lengths [3], used in turn: a launch reads `hot_cur`, appends to the next and
    // clears the length of the one after; bit 7 of cls_last = "this row is in the list the next launch reads".
    int32_t *hot_list, *hot_cnt;
    int32_t hot_cap, hot_cur, hot_home; // (hot_home: the launch's first so many blocks take the list, the ones behind them positions)
    int32_t hot_wave_rows;              // rows of the list per wavefront of those blocks (trmc_plan_options.hot_wave_rows)
    // STREAM of windows (trmc_stream_*; tile kernels only): the launch index counts tiles over ALL days -- a position `lag` tiles
    // behind works on tile (seq_day * seq_tpd + tile) - lag of the stream: day d = that / seq_tpd, in the buffers of slot d %
    // seq_slots (q_tm, d_tm, qlat_tm, out, dec: slot s begins s * slot_* elements behind the pointer above).  A row that ends a day
    // also writes its state into time row 0 of the next slot.  seq_slots <= 1: one window, no ring (everything above as it is).
    int32_t seq_slots, seq_tpd, seq_day, seq_days; // slots; tiles per day; day of the launch's tile index; days pushed so far
    int32_t seq_day_min;                           // days before this one have been queued to their end (trmc_stream_flush)
    int64_t slot_tm, slot_qlat, slot_out, slot_dec;
};

// which tile of which day a position `lag` tiles behind works on in a launch of the stream: false = none (before the first
// day, or behind the last one pushed)
template <class T>
__device__ __forceinline__ bool seq_locate(const StepArgs<T> &a, int32_t tile, int32_t lag, int32_t &behind, int32_t &slot, int32_t &slot_next)
{
    behind = tile - lag;
    slot = slot_next = 0;
    if (a.seq_slots <= 1) return behind >= 0;
    int32_t d = a.seq_day;
    while (behind < 0) {
        behind += a.seq_tpd;
        --d;
    }
    if (d < a.seq_day_min || d >= a.seq_days) return false;
    slot = d % a.seq_slots;
    slot_next = slot + 1 == a.seq_slots ? 0 : slot + 1;
    return true;
}

// One launch = one timestep (SHORT) or one wavefront diagonal (!SHORT) over the plan
// positions [s_begin, s_end); thread w of the launch takes position s_begin + w.
//
// Divergence control is the PLAN's business, not the kernel's: the secant loop runs 0 (no flow), 1 (depth below
// the 1 cm floor: early exit, f90:120-122), 2 (wet channel) or, rarely, 3+ iterations per segment-step, a segment
// repeats its count from one step to the next 99.3 % of the time, and a plan built with a cost hint
// (trmc_plan_create_hinted) stores the rows of a level grouped by that cost, so that a wavefront holds rows of one
// class.  (Rounds 1-2 also carried a per-block partition by the previous step's class for plans without a hint --
// ballots, a shuffle scan and three barriers; since the step's control flow got cheaper it cost more than the mixed
// wavefronts it avoided, 23.0 against 21.7 ms per CONUS day, and it is gone.)
#ifndef TRMC_STEP_BLOCK // threads per block of the step kernel (measured on MI355X, CONUS: 64 -> 100.4 us per launch,
// 128 -> 97.2, 192 -> 99.1, 256 -> 100.2, 512 -> 115.5, 1024 -> 132: small blocks free their wave slots sooner)
#define TRMC_STEP_BLOCK 128
#endif
constexpr int kStepBlock = TRMC_STEP_BLOCK;
#ifndef TRMC_EMIT_TILE // timesteps per overlapped transpose launch (a multiple of kEmitSteps).  The last tile's
// transpose trails the last step launch: CONUS day 22.3 ms with tiles of 128 steps, 21.7 with 64, 21.5 with 32
#define TRMC_EMIT_TILE 32
#endif
template <class T, bool SHORT, bool LAG = false, bool TOL = false>
__global__ void __launch_bounds__(kStepBlock)
k_mc_step(const StepArgs<T> a, const int32_t s_begin, const int32_t s_end, const int32_t diag, const int32_t ql_col)
{   // ql_col: the lateral-inflow column (diag - 1) / qts of a launch whose rows are all at step diag (SHORT, no lag) -- formed
    // by the host: an integer division by a run-time divisor is some 35 instructions per thread
    using M = typename DevMath<T, TOL>::type;
    __shared__ uint64_t s_tab[TRMC_POW_TAB_WORDS];
    // Issue priority over whatever else is resident: beside the wide tiles (k_mc_tile) these launches are the narrow tail of
    // the level order -- 288 launches that wait for each other, the critical path of the window -- and a wavefront of theirs
    // that shares its SIMD with four tile wavefronts at equal priority needs 43 us for a step it does in 20 us alone.  (Alone
    // on the device every wavefront has the same priority and nothing changes.)
    __builtin_amdgcn_s_setprio(3);
    M m{stage_pow_tables(s_tab), false};
    m.sane = a.sane;

    {
        const int32_t s = s_begin + (int32_t)blockIdx.x * kStepBlock + (int32_t)threadIdx.x;
        // (the block's rows dealt to its threads by the class of the step before, as k_mc_tile does once per K steps, was
        // built and measured here in round 5 -- some fifty instructions and four barriers per step: untuned plan 19.39 ms per
        // day against 19.44 without, tuned plan 16.15 against 16.09, tolerance arithmetic 13.5 against 12.4 -- and removed.
        // So were k_mc_tile's hot rows per STEP -- the rows of three or more iterations in the step before routed by the
        // launch's first blocks, an atomic append per hot row and launch: cost-ordered plan 16.67 ms per day against 16.2,
        // unordered 20.1 against 17.3 -- these launches are the window's dependent chain, and what lengthens one lengthens it.)
        if (s >= s_end) return;
        const int32_t t = SHORT ? (LAG ? diag - a.lag[s] : diag) : diag - a.level[s];
        if (t < 1 || t > a.nsteps) return;

        // 32-bit unsigned position: with uniform (SGPR) array bases every load below is
        // `global_load v, v_off, s[base]` with ONE shared byte offset instead of a 64-bit add per array
        const uint32_t su = (uint32_t)s;
        uint32_t ob = su * (uint32_t)sizeof(T); // byte offset of position s in any T column (nseg_pad * 8 < 2**32)
        const size_t row_p = (size_t)(t - 1) * (size_t)a.nseg_pad; // previous time level
        const size_t row_c = (size_t)t * (size_t)a.nseg_pad;       // current time level
        const T *const q_prev = a.q_tm + row_p;
        const T *const q_curr = a.q_tm + row_c;

        trmc::ChannelParams<T> p;
        p.dt = a.dt_col ? at(a.dt_col, ob) : a.dt;
        // (the zero-extension of the offset has to be visible in the basic block of the loads for the
        // SGPR-base addressing form to be selected: re-introduce it after every branch)
        asm volatile("" : "+v"(ob));
        p.dx = at(a.dx, ob);
        p.bw = at(a.bw, ob);
        p.twcc = at(a.twcc, ob);
        p.n = at(a.n, ob);
        p.ncc = at(a.ncc, ob);
        p.s0 = at(a.s0, ob);
        p.tw = p.cs = T(0); // only enter the constants below
        trmc::ChannelConst<T> c;
        c.z = at(a.z, ob);
        c.bfd = at(a.bfd, ob);
        c.sqrt_s0 = at(a.sqrt_s0, ob);
        c.sq1pz2 = at(a.sq1pz2, ob);
        c.s0_n = at(a.s0_n, ob);
        c.s0_ncc = at(a.s0_ncc, ob);
        c.inv_n = at(a.inv_n, ob);
        trmc::derive_const(c, p);

        trmc::Inflow<T> f;
        f.qdp = at(q_prev, ob);
        const T depthp = at(a.d_tm + row_p, ob);
        f.ql = at(a.qlat_tm + (size_t)((SHORT && !LAG) ? ql_col : (t - 1) / a.qts) * (size_t)a.nseg_pad, ob);

        // junction sums in the reference's order (mc_reach.pyx:499-502).  The first two upstream positions of a row
        // sit in a table of their own (-1 = none; bit 30 of the second = "the CSR list has more"): one load beside
        // the parameter loads instead of the dependent chain up_ptr -> up_idx -> q, and no loop for fan-in <= 2.
        T qup = T(0), quc = T(0);
        {
            const int2 u = a.up2[su];
            if (u.x >= 0) {
                const uint32_t ub = (uint32_t)u.x * (uint32_t)sizeof(T);
                qup += at(q_prev, ub);
                if (!SHORT) quc += at(q_curr, ub);
            }
            if (u.y >= 0) {
                const uint32_t ub = (uint32_t)(u.y & 0x3fffffff) * (uint32_t)sizeof(T);
                qup += at(q_prev, ub);
                if (!SHORT) quc += at(q_curr, ub);
                if (u.y & 0x40000000) {
                    const int32_t k1 = a.up_ptr[su + 1];
                    for (int32_t k = a.up_ptr[su] + 2; k < k1; ++k) {
                        const uint32_t uk = (uint32_t)a.up_idx[k] * (uint32_t)sizeof(T);
                        qup += at(q_prev, uk);
                        if (!SHORT) quc += at(q_curr, uk);
                    }
                }
            }
        }
        if (!SHORT && a.raw_of_pos) {
            const int32_t g = a.raw_of_pos[s];
            if (g >= 0) quc = a.da_raw[(size_t)g * (size_t)a.nsteps + (size_t)(t - 1)]; // (its one upstream row is that gage)
        }
        f.qup = qup;
        f.quc = SHORT ? qup : quc;

        if (a.res_of_pos) { // reference loop branch mc_reach.pyx:507-510,:551-553,:706-710
            const int32_t ri = a.res_of_pos[s];
            if (ri >= 0) {
                const T *rp = a.res_par + (size_t)ri * 9;
                const trmc::LevelPoolParams<T> lp{rp[0], rp[1], rp[2], rp[3], rp[4], rp[5], rp[6], rp[7], rp[8]};
                T H = depthp; // a reservoir row keeps its water elevation in the depth slot
                const T outflow = trmc::levelpool_step<T, M>(f.quc, T(0), a.res_dt, H, lp, m);
                a.q_tm[row_c + s] = outflow;
                a.v_tm[row_c + s] = T(0);
                a.d_tm[row_c + s] = H;
                a.res_inflow[(size_t)ri * (size_t)a.nsteps + (size_t)(t - 1)] = f.quc;
                if (t == a.nsteps) a.it_prev[s] = 0;
                return;
            }
        }

        m.coef_ok = coef_guard(p.dt, f.ql);
        const trmc::StepResult<T> r = trmc::mc_segment_step<T, M>(p, c, f, depthp, m);
        T q_new = r.qdc;
        if (a.gage_of_pos) { // reference hook mc_reach.pyx:761-796; arithmetic of simple_da.pyx:47-76
            const int32_t g = a.gage_of_pos[s];
            if (g >= 0) {
                const size_t e = (size_t)g * (size_t)a.nsteps + (size_t)(t - 1);
                const uint8_t mode = a.da_mode[e];
                T nudge = T(0);
                if (!SHORT && a.da_raw) a.da_raw[e] = q_new;
                if (mode == 1) {            // valid observation: replace
                    nudge = a.da_a[e] - q_new;
                    q_new = a.da_a[e];
                } else if (mode == 2) {     // decay the last observation towards the model value
                    nudge = (a.da_a[e] - q_new) * a.da_w[e];
                    q_new = q_new + nudge;
                }
                a.da_nudge[e] = nudge;
            }
        }
        asm volatile("" : "+v"(ob));
        at(a.q_tm + row_c, ob) = q_new;
        at(a.d_tm + row_c, ob) = r.depthc;
        at(a.v_tm + row_c, ob) = r.velc;
        // (only trmc_download_iterations reads it, after the window: one byte-masked store per row and step would be
        // 3 % of the launch)
        if (t == a.nsteps) a.it_prev[su] = (uint8_t)min(r.iters, 255);
        // cost of the step for the plan's cost hint: the iteration class, plus 4 where the compound-channel branch ran
        // (a wavefront pays that branch -- two more divisions, one more power per evaluation -- as soon as one lane takes it)
        if (a.it_sum) a.it_sum[su] = (uint16_t)min(65535, (int)a.it_sum[su] + min(r.iters, 3) + (r.over ? 4 : 0));
    }
}

// The WIDE levels of a short-timestep window, K timesteps per launch, a row in ONE thread for all of them.
//
// With assume_short_ts a row at step t reads flows of step t - 1 only (mc_reach.pyx:504-505, :135-136).  So if level l
// of the network runs K steps BEHIND level l - 1, every flow a row reads during a tile of K steps was written by an
// EARLIER launch: launch `tile` routes level l through the steps ((tile - l) K, (tile - l + 1) K], and a row of that level
// reads its upstream rows (levels < l, hence at least K steps ahead) at steps it finds complete.  No flag, no poll, no
// barrier -- the kernel boundary is the only synchronisation -- and inside a launch a thread keeps its row's thirteen
// parameter / constant columns and its state (flow, depth) in registers, reads the forcing once per column and its
// upstream flows once per step, and writes (q, v, d) per step: 12 + 8 bytes of traffic per segment-step after the first
// instead of the 94 the one-step kernel moves, and no memory phase that the arithmetic of a 70-microsecond launch cannot hide.
// The level skew costs `wide - 1` partly filled launches at either end of the window; only levels wide enough to fill the
// device by themselves are routed this way (route_advance_t picks them), the narrow tail of the level order keeps the
// one-step launches of k_mc_step, trailing the last wide level.  Results are the same bits: the same segment steps on the
// same inputs, visited in another order (tests run both paths against the oracle).
// Results go straight into the caller's layout out[row][step][q,v,d]: a thread stages kTileStage steps in LDS (lane-
// contiguous columns: conflict-free) and writes them as one 96-byte run -- three whole 32-byte sectors -- so these rows
// need no transposing pass (k_emit skips them) and no velocity plane at all; of the time-major planes only the flow row
// of every step (what downstream rows and the gathers read) and the depth row of a tile's last step (where the row's next
// tile, or the final state, picks it up) are written.
#ifndef TRMC_TILE_PARTITION_UNHINTED
#define TRMC_TILE_PARTITION_UNHINTED 1
#endif
constexpr int kTileStage = 8;
#ifndef TRMC_TILE_BLOCK // threads per block of k_mc_tile: also the group its in-block partition deals rows in
#define TRMC_TILE_BLOCK 128
#endif
constexpr int kTileBlock = TRMC_TILE_BLOCK;
constexpr int32_t kWideMaxLevels = 64; // at most this many leading levels are routed by k_mc_tile (wide_levels, default 16, is capped by it)
// in-block partition of a tile's rows by cost class: on.  Measured on the CONUS sequence (ms per day, on / off): plan built from
// the topology alone 19.3 / 20.4, tuned plan on days whose forcing is drawn anew 19.8 / 20.8, tuned plan on its own kind of days
// 16.7 / 16.6, tolerance arithmetic 12.55 / 12.54 -- what a stale or missing cost hint loses, the partition wins back in part
constexpr bool kTilePartitionDefault(bool hinted) { return hinted || TRMC_TILE_PARTITION_UNHINTED; }
constexpr int32_t kMidMaxLevels = 32;  // ... and at most this many more by its second tier (mid_levels)
constexpr int64_t kMidDefaultRowsPerCu = 0; // default threshold of the second tier in rows per compute unit; 0 = off unless asked for
#ifndef TRMC_TILE_WAVES // wavefronts per SIMD the register allocation of k_mc_tile must allow.  Measured on the CONUS day by
// padding the blocks' LDS (TRMC_TILE_LDS_PAD) and by this cap: 1 wavefront per SIMD 36.7 ms, 2: 23.9, 3: 20.7, 3.5: 19.5,
// 4: 18.45, 5 (95 registers, 4 spilled): 17.9, 6 (80 registers, 27 spilled): 19.4 -- the curve of a kernel that hides its
// latencies with other wavefronts and is close to its issue limit at four
#define TRMC_TILE_WAVES 5
#endif
#ifndef TRMC_HOT_WAVE_MAX // a wavefront with at least so many hot rows keeps them (k_mc_tile's epilogue); measured on the CONUS
// sequence, ms per day on the cost-ordered / the unordered plan: 6: 16.06 / 17.55, 16: 16.07 / 17.33, 40: 16.06 / 17.36
#define TRMC_HOT_WAVE_MAX 16
#endif
template <class T, bool TOL = false, bool DEC = false>
__global__ void __launch_bounds__(kTileBlock, sizeof(T) == 4 ? TRMC_TILE_WAVES : 1)
k_mc_tile(const StepArgs<T> a, const int32_t s_begin, const int32_t s_end, const int32_t tile, const int32_t K)
{
    using M = typename DevMath<T, TOL>::type;
    const ColdArgs<StepArgs<T>> cold = cold_args<StepArgs<T>>(); // (see cold_args: what the loop rarely needs is not kept in registers)
    __shared__ uint64_t s_tab[TRMC_POW_TAB_WORDS];
    __shared__ T s_out[3 * kTileStage * kTileBlock]; // [step slot * 3 + c][thread]
    M m{stage_pow_tables(s_tab), false};
    m.sane = a.sane;

    int32_t s;
    bool from_hot = false;
    int32_t *const hot_list = a.cls_last ? cold->hot_list : nullptr;
    const int32_t hot_blocks = hot_list ? cold->hot_home : 0; // (the launch's FIRST blocks: the costliest rows start first)
    if ((int32_t)blockIdx.x < hot_blocks) {
        // HOT ROWS.  The first blocks of the launch take the list the tile before left: the rows that ended it in
        // class 3 or above -- three or more secant iterations, over bank; 1.5 % of the rows of an unordered CONUS plan, and one of
        // them in a wavefront makes all 64 lanes wait through its extra iterations (they sat in half of the wavefronts: 903
        // instructions per wavefront-step against 619 on the cost-ordered plan).  Gathered here they pace each other only.
        // Bookkeeping: the mark (bit 7 of the row's class byte) is only ever set together with an entry in the list the NEXT
        // launch reads, and that launch's list thread either routes the row -- and rewrites the byte at its end -- or clears
        // the mark: no row is left marked without being listed.  A row that a list thread has routed AND found cooled down
        // before a late block of the same launch looks at its byte is routed a second time by that block: the same steps from
        // the same inputs (a tile reads nothing it writes), so the same values are stored twice -- work, not a difference.
        // A wavefront of these blocks takes hot_wave_rows entries (its other lanes leave): a wavefront's step costs what its
        // costliest row's does, and where the launch does not fill the device many times over (a rank of a multi-GPU job) the
        // K dependent steps of its slowest wavefront ARE the launch -- sixteen rows per wavefront have a row of five or six
        // iterations among them a quarter as often as sixty-four.
        const int32_t cur = cold->hot_cur, cap = cold->hot_cap, H = cold->hot_wave_rows;
        const int32_t lane = (int32_t)(threadIdx.x & 63u), wave = (int32_t)blockIdx.x * (kTileBlock / 64) + (int32_t)(threadIdx.x >> 6);
        const int32_t i = wave * H + lane;
        const int32_t nlist = min(cold->hot_cnt[cur], cap);
        if (lane >= H || i >= nlist) return;
        if (lane == 0) atomicAdd(&cold->hot_cnt[3], min(nlist - i, H)); // (trmc_plan_hot_rows: a running total)
        s = hot_list[(size_t)cur * (size_t)cap + (size_t)i];
        from_hot = true;
        if (s < s_begin || s >= s_end) { // (listed by a window whose tiled levels reached further: back to where it is routed now)
            cold->cls_last[s] &= 0x7f;
            return;
        }
    } else {
        const int32_t home = (int32_t)blockIdx.x - hot_blocks;
        const int32_t s_mine = s_begin + home * kTileBlock + (int32_t)threadIdx.x;
        s = s_mine;
        if (a.cls_last) {
            // Which row a thread takes: the block's kTileBlock positions dealt out by DESCENDING cost class -- the class every row
            // showed at the end of the tile before (a row repeats its secant iteration count from step to step 99.3 % of the time)
            // -- so that a wavefront holds rows of one class whatever the forcing does and however old the plan's cost hint is.
            // Once per K steps, inside the launch: a count per class in LDS, a prefix over the eight classes, a scatter of lane
            // numbers.  (Until round 5 a launch of its own between the tiles, k_tile_perm, over groups of 256 positions: 13-25 us
            // of the tile stream per tile, 0.55 ms of a CONUS day spent between tiles.)  Order inside a class is whatever the
            // atomics give; results do not depend on which thread routes a row.  (Bit 7: the row is in the hot list.)
            if (hot_list && home == 0 && threadIdx.x == 0) cold->hot_cnt[(cold->hot_cur + 2) % 3] = 0; // (the list after next)
            const int32_t c = s_mine < s_end ? (int32_t)a.cls_last[s_mine] : 0x80;
            const int32_t key = (c & 0x80) ? 8 : 7 - min(c, 7); // bucket 0 = the costliest; 8 = no row
            bool none;
            s = s_begin + home * kTileBlock + block_partition_by_class<kTileBlock>(key, none);
            if (none) return; // (behind the tier's last position, or routed by a block of the hot list)
        }
        if (s >= s_end) return;
    }
    int32_t behind, slot, slot_next;
    const bool in_range = seq_locate(a, tile, a.level[s], behind, slot, slot_next);
    const int32_t t_lo = behind * K + 1, t_hi = min(behind * K + K, a.nsteps);
    if (!in_range || t_lo > t_hi) {
        if (from_hot) cold->cls_last[s] &= 0x7f; // (not routed in this launch: back to its block, which does that bookkeeping)
        return;
    }
    // Issue priority by cost.  A launch cannot end before its slowest wavefront has made its K dependent steps, and a step of
    // rows that take three secant iterations or run over bank is some 2 200 instructions against 600-900 for the others: on a
    // device that the launch does not fill many times over (one rank of a multi-GPU job: five wavefronts per SIMD, all
    // resident at once) those wavefronts ARE the launch -- 12.5 us per step when they share their SIMD's issue slots equally
    // with four cheaper ones, measured as 200 us per launch of 16 steps whatever the number of rows.  So the costlier a
    // wavefront's rows showed themselves in the tile before, the higher its priority (the list's blocks: the highest).
    if (a.cls_last) {
        const int32_t cp = from_hot ? 7 : (int32_t)(a.cls_last[s] & 0x7f);
        if (__any(cp >= 3)) __builtin_amdgcn_s_setprio(3);
        else if (__any(cp == 2)) __builtin_amdgcn_s_setprio(1);
    }

    const uint32_t su = (uint32_t)s;
    uint32_t ob = su * (uint32_t)sizeof(T);
    const size_t np = (size_t)a.nseg_pad;
    trmc::ChannelParams<T> p;
    p.dt = a.dt_col ? at(a.dt_col, ob) : a.dt;
    asm volatile("" : "+v"(ob));
    p.dx = at(a.dx, ob);
    p.bw = at(a.bw, ob);
    p.twcc = at(a.twcc, ob);
    p.n = at(a.n, ob);
    p.ncc = at(a.ncc, ob);
    p.s0 = at(a.s0, ob);
    p.tw = p.cs = T(0);
    trmc::ChannelConst<T> c;
    c.z = at(a.z, ob);
    c.bfd = at(a.bfd, ob);
    c.sqrt_s0 = at(a.sqrt_s0, ob);
    c.sq1pz2 = at(a.sq1pz2, ob);
    c.s0_n = at(a.s0_n, ob);
    c.s0_ncc = at(a.s0_ncc, ob);
    c.inv_n = at(a.inv_n, ob);
    trmc::derive_const(c, p);
    const int2 u = a.up2[su];
    const int32_t ri = a.res_of_pos ? a.res_of_pos[s] : -1;
    const int32_t gi = a.gage_of_pos ? a.gage_of_pos[s] : -1;

    // (a stream of windows: this row's day lives in its slot of the ring)
    T *const q_tm = a.q_tm + (size_t)slot * (size_t)a.slot_tm;
    const size_t ql_base = (size_t)slot * (size_t)a.slot_qlat;
    T q_prev = at(q_tm + (size_t)(t_lo - 1) * np, ob);
    T d_prev = at(a.d_tm + (size_t)slot * (size_t)a.slot_tm + (size_t)(t_lo - 1) * np, ob);
    T *const out_row = a.out + (size_t)slot * (size_t)a.slot_out + (size_t)a.row_of_pos[su] * (size_t)a.nsteps * 3;
    // the lateral-inflow column of step t is (t - 1) / qts: found by division once, by a counter from then on
    int32_t ql_col = (t_lo - 1) / a.qts, ql_left = a.qts - (t_lo - 1) % a.qts;
    T ql = at(a.qlat_tm + ql_base + (size_t)ql_col * np, ob);
    m.coef_ok = coef_guard(p.dt, ql); // (depends on the forcing column only: formed when that changes, not every step)
    const bool count_cost = a.it_sum != nullptr;
    int32_t it_acc = 0, it_last = 0, staged = 0;
    bool over_last = false;
    // flows of the step before (complete: earlier launches); advanced a row per step -- t differs from lane to lane (the
    // level skew), and (size_t)t * np in vector registers is a 64-bit multiplication per step
    T *q_up = q_tm + (size_t)(t_lo - 1) * np;
    // (asking for the upstream flows of step t + 1 while step t is computed -- they were all written by earlier launches -- was built
    // and measured in round 6: the CONUS stream 14.3 ms per day against 13.9, the ranks of an 8-way partition 2.76 against 2.70: two
    // more live registers in a kernel that already spills four cost more than the L2 trip they take off the chain)
    for (int32_t t = t_lo; t <= t_hi; ++t, q_up += np) {
        if (ql_left == 0) {
            ++ql_col;
            ql = at(cold->qlat_tm + ql_base + (size_t)ql_col * np, ob);
            m.coef_ok = coef_guard(p.dt, ql);
            ql_left = cold->qts;
        }
        --ql_left;
        // junction sum in the reference's order (mc_reach.pyx:499-502); see k_mc_step for the table of the first two
        T qup = T(0);
        if (u.x >= 0) qup += at(q_up, (uint32_t)u.x * (uint32_t)sizeof(T));
        if (u.y >= 0) {
            qup += at(q_up, (uint32_t)(u.y & 0x3fffffff) * (uint32_t)sizeof(T));
            if (u.y & 0x40000000) {
                const int32_t *const up_ptr = cold->up_ptr, *const up_idx = cold->up_idx;
                const int32_t k1 = up_ptr[su + 1];
                for (int32_t k = up_ptr[su] + 2; k < k1; ++k) qup += at(q_up, (uint32_t)up_idx[k] * (uint32_t)sizeof(T));
            }
        }
        T q_new, v_new, d_new;
        if (ri >= 0) { // level-pool reservoir row (see k_mc_step)
            const T *rp = cold->res_par + (size_t)ri * 9;
            const trmc::LevelPoolParams<T> lp{rp[0], rp[1], rp[2], rp[3], rp[4], rp[5], rp[6], rp[7], rp[8]};
            T H = d_prev;
            q_new = trmc::levelpool_step<T, M>(qup, T(0), cold->res_dt, H, lp, m);
            v_new = T(0);
            d_new = H;
            cold->res_inflow[(size_t)ri * (size_t)cold->nsteps + (size_t)(t - 1)] = qup;
            it_last = 0;
        } else {
            trmc::Inflow<T> f;
            f.qup = qup;
            f.quc = qup;
            f.qdp = q_prev;
            f.ql = ql;
            const trmc::StepResult<T> r = trmc::mc_segment_step<T, M>(p, c, f, d_prev, m);
            q_new = r.qdc;
            v_new = r.velc;
            d_new = r.depthc;
            it_last = r.iters;
            over_last = r.over;
            if (count_cost) it_acc += min(r.iters, 3) + (r.over ? 4 : 0);
            if (gi >= 0) { // streamflow nudging (see k_mc_step)
                const size_t e = (size_t)gi * (size_t)cold->nsteps + (size_t)(t - 1);
                const T *const da_a = cold->da_a;
                const uint8_t mode = cold->da_mode[e];
                T nudge = T(0);
                if (mode == 1) {
                    nudge = da_a[e] - q_new;
                    q_new = da_a[e];
                } else if (mode == 2) {
                    nudge = (da_a[e] - q_new) * cold->da_w[e];
                    q_new = q_new + nudge;
                }
                cold->da_nudge[e] = nudge;
            }
        }
        asm volatile("" : "+v"(ob));
        at(q_up + np, ob) = q_new;
        if (t == t_hi) {
            at(cold->d_tm + (size_t)slot * (size_t)cold->slot_tm + (size_t)t * np, ob) = d_new;
            if (cold->seq_slots > 1 && t == cold->nsteps) { // the day ends: the next one starts from here (its slot's time row 0)
                at(cold->q_tm + (size_t)slot_next * (size_t)cold->slot_tm, ob) = q_new;
                at(cold->d_tm + (size_t)slot_next * (size_t)cold->slot_tm, ob) = d_new;
            }
        }
        q_prev = q_new;
        d_prev = d_new;
        if (DEC || a.out) { // stage (q, v, d) of step t; a run ends when kTileStage steps are staged and at the tile's last step
            // (a.out == nullptr: a stream of windows whose callers take products only -- nothing of the full result is assembled)
            T *so = s_out + (size_t)(staged * 3) * kTileBlock + threadIdx.x;
            so[0] = q_new;
            so[kTileBlock] = v_new;
            so[2 * kTileBlock] = d_new;
            ++staged;
            if (staged == kTileStage || t == t_hi) {
                T *dst = out_row + (size_t)(t - staged) * 3;
                const T *si = s_out + threadIdx.x;
                if (!a.out) {
                } else if (a.out_vec && (staged & 3) == 0) { // (float: 3 * staged values = 3 * staged / 4 pieces of 16 bytes)
                    for (int j = 0; j < 3 * staged / 4; ++j) {
                        float4 v;
                        v.x = (float)si[(4 * j + 0) * kTileBlock];
                        v.y = (float)si[(4 * j + 1) * kTileBlock];
                        v.z = (float)si[(4 * j + 2) * kTileBlock];
                        v.w = (float)si[(4 * j + 3) * kTileBlock];
                        reinterpret_cast<float4 *>(dst)[j] = v;
                    }
                } else {
                    for (int32_t e = 0; e < 3 * staged; ++e) dst[e] = si[e * kTileBlock];
                }
                if constexpr (DEC) {
                    // the kept steps among the ones just written, (t - staged, t]: the multiples of dec_stride, newest first
                    const int32_t ds = cold->dec_stride;
                    for (int32_t k = t / ds; k >= 1 && k * ds > t - staged; --k) {
                        if (k > cold->dec_keep) continue;
                        const int32_t kslot = k * ds - (t - staged) - 1;
                        T *dd = cold->dec + (size_t)slot * (size_t)cold->slot_dec + ((size_t)cold->row_of_pos[su] * (size_t)cold->dec_keep + (size_t)(k - 1)) * 3;
                        dd[0] = si[(kslot * 3 + 0) * kTileBlock];
                        dd[1] = si[(kslot * 3 + 1) * kTileBlock];
                        dd[2] = si[(kslot * 3 + 2) * kTileBlock];
                    }
                }
                staged = 0;
            }
        }
    }
    if (t_hi == cold->nsteps) cold->it_prev[su] = (uint8_t)min(it_last, 255);
    if (uint8_t *const cls = cold->cls_last) {
        uint8_t c = (uint8_t)(min(it_last, 3) + (over_last ? 4 : 0));
        // (a row that has finished the window starts the next one in its block; and a wavefront that holds sixteen or more of
        // them -- the first blocks of every level of a cost-ordered plan -- keeps them: they pace each other where they are)
        const bool hot = hot_list && c >= 3 && (t_hi < cold->nsteps || cold->seq_slots > 1);
        if (hot && (from_hot || __builtin_popcountll(__ballot(hot)) < TRMC_HOT_WAVE_MAX)) {
            const int32_t nxt = (cold->hot_cur + 1) % 3, cap = cold->hot_cap;
            const int32_t i = atomicAdd(&cold->hot_cnt[nxt], 1);
            if (i < cap) {
                hot_list[(size_t)nxt * (size_t)cap + (size_t)i] = s;
                c |= 0x80;
            }
        }
        cls[su] = c;
    }
    if (uint16_t *const it_sum = cold->it_sum) it_sum[su] = (uint16_t)min(65535, (int)it_sum[su] + it_acc);
}

#include "k_mc_ctile.inc"

// plan time: the segment-invariant constants of mc_segment.hpp::make_const, one thread per position,
// written as six more SoA columns behind the nine parameter columns (same device arithmetic the
// step kernel would otherwise repeat every timestep: 4 divisions and 2 square roots per segment-step)
constexpr int kConstCols = 7, kTotalCols = TRMC_NPARAM + kConstCols;
template <class T>
__global__ void __launch_bounds__(kBlock)
k_make_const(T *cols, int32_t nseg, int64_t nseg_pad)
{
    using M = typename DevMath<T>::type;
    const int32_t s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= nseg) return;
    const M m{nullptr, false}; // make_const uses sqrt and divide only, never the power tables
    trmc::ChannelParams<T> p;
    p.dt = cols[(size_t)TRMC_P_DT * nseg_pad + s];
    p.dx = cols[(size_t)TRMC_P_DX * nseg_pad + s];
    p.bw = cols[(size_t)TRMC_P_BW * nseg_pad + s];
    p.tw = cols[(size_t)TRMC_P_TW * nseg_pad + s];
    p.twcc = cols[(size_t)TRMC_P_TWCC * nseg_pad + s];
    p.n = cols[(size_t)TRMC_P_N * nseg_pad + s];
    p.ncc = cols[(size_t)TRMC_P_NCC * nseg_pad + s];
    p.cs = cols[(size_t)TRMC_P_CS * nseg_pad + s];
    p.s0 = cols[(size_t)TRMC_P_S0 * nseg_pad + s];
    const trmc::ChannelConst<T> c = trmc::make_const<T, M>(p, m);
    T *o = cols + (size_t)TRMC_NPARAM * nseg_pad + s;
    o[0 * nseg_pad] = c.z;
    o[1 * nseg_pad] = c.bfd;
    o[2 * nseg_pad] = c.sqrt_s0;
    o[3 * nseg_pad] = c.sq1pz2;
    o[4 * nseg_pad] = c.s0_n;
    o[5 * nseg_pad] = c.s0_ncc;
    o[6 * nseg_pad] = c.inv_n;
}

// forcing: in[row][nq] (caller order) -> qlat_tm[j][pos]; LDS tile of 64 positions x 32 columns
template <class T>
__global__ void __launch_bounds__(kBlock)
k_prep_qlat(const T *__restrict__ in, const int32_t *__restrict__ row_of_pos, T *__restrict__ qlat_tm,
            int32_t nseg, int64_t nseg_pad, int32_t nq)
{
    __shared__ T tile[32][65];
    const int32_t p0 = blockIdx.x * 64;
    const int32_t j0 = blockIdx.y * 32;
    const int32_t nj = min(32, nq - j0);
    for (int32_t i = threadIdx.x; i < 64 * 32; i += kBlock) {
        const int32_t pl = i / 32, jl = i % 32;
        const int32_t p = p0 + pl;
        if (p < nseg && jl < nj) tile[jl][pl] = in[(size_t)row_of_pos[p] * nq + j0 + jl];
    }
    __syncthreads();
    for (int32_t i = threadIdx.x; i < 64 * 32; i += kBlock) {
        const int32_t jl = i / 64, pl = i % 64;
        const int32_t p = p0 + pl;
        if (p < nseg && jl < nj) qlat_tm[(size_t)(j0 + jl) * nseg_pad + p] = tile[jl][pl];
    }
}

// forcing from packed CHRTOUT columns: raw_a/raw_b [nq][nfeat] int32 (file order) -> qlat_tm[j][pos], decoding,
// the join on feature id (feat_of_pos) and the transposition in one pass.  Unpacking follows netCDF4-python's
// default read (nhd_io.py:397-434 get_ql_from_chrtout: masked where == _FillValue / missing_value or outside the
// valid range, filled with 0; otherwise raw * scale_factor + add_offset evaluated in double), the sum of the two
// variables in double, then the reference's cast to float32 (compute.py / qlat_sub.values.astype("float32")).
struct PackSpec {
    double scale, offset;
    int32_t fill1, fill2, vmin, vmax;
    int32_t use1, use2; // a fill value that is absent from the file arrives as NaN
};
__device__ __forceinline__ double unpack_cf(int32_t raw, const PackSpec &k)
{
    if ((k.use1 && raw == k.fill1) || (k.use2 && raw == k.fill2) || raw < k.vmin || raw > k.vmax) return 0.0;
    return (double)raw * k.scale + k.offset;
}
template <class T>
__global__ void __launch_bounds__(kBlock)
k_ingest_packed(const int32_t *__restrict__ raw_a, const int32_t *__restrict__ raw_b, const PackSpec ka, const PackSpec kb,
                const int32_t *__restrict__ feat_of_pos, T *__restrict__ qlat_tm, int32_t nseg, int64_t nseg_pad, int32_t nq,
                int64_t nfeat)
{
    const int32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= nseg) return;
    const int32_t f = feat_of_pos[p];
    for (int32_t j = 0; j < nq; ++j) {
        double v = 0.0;
        if (f >= 0) {
            v = unpack_cf(raw_a[(size_t)j * nfeat + f], ka);
            if (raw_b) v = v + unpack_cf(raw_b[(size_t)j * nfeat + f], kb);
        }
        qlat_tm[(size_t)j * nseg_pad + p] = (T)(float)v;
    }
}

// initial state: time row 0 <- q0[row] = (qu0, qd0, h0)   (mc_reach.pyx:361)
template <class T>
__global__ void __launch_bounds__(kBlock)
k_init_state(const T *__restrict__ q0, const int32_t *__restrict__ row_of_pos, T *q_tm, T *v_tm, T *d_tm,
             int32_t nseg)
{
    const int32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= nseg) return;
    const size_t r = (size_t)row_of_pos[p] * 3;
    q_tm[p] = q0[r + 0];
    v_tm[p] = q0[r + 1];
    d_tm[p] = q0[r + 2];
}

// boundary rows: prescribed hydrographs bfvd[b][t-1][c] -> time rows 1..nsteps at position b
template <class T>
__global__ void __launch_bounds__(kBlock)
k_fill_boundary(const T *__restrict__ bfvd, T *q_tm, T *v_tm, T *d_tm, int32_t nboundary, int32_t nsteps,
                int64_t nseg_pad)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= (int64_t)nboundary * nsteps) return;
    const int32_t b = (int32_t)(i / nsteps), t = (int32_t)(i % nsteps) + 1;
    const size_t src = ((size_t)b * nsteps + (t - 1)) * 3;
    const size_t dst = (size_t)t * nseg_pad + b;
    q_tm[dst] = bfvd[src + 0];
    v_tm[dst] = bfvd[src + 1];
    d_tm[dst] = bfvd[src + 2];
}

// result: time-major SoA -> out[row][t-1][3]; tile = 64 positions x kEmitSteps steps through LDS
// (32 steps: 24.8 KB of LDS per block -> 6 blocks per CU keep enough loads in flight; row chunks of
// 384 contiguous bytes on the store side)
#ifndef TRMC_EMIT_STEPS
#define TRMC_EMIT_STEPS 32
#endif
constexpr int kEmitSteps = TRMC_EMIT_STEPS;
// Index arithmetic is what this kernel's VALU instructions are, and it runs beside the VALU-bound step launches, so
// both passes are written to need little of it: a thread keeps ONE position through the load pass (its time-major
// addresses advance by a constant), and the store pass moves 16 bytes per lane -- a row's run of 32 steps is 24 such
// pieces, two rows per wave instruction -- when the result's rows are 16-byte aligned (nsteps % 4 == 0 in fp32).
template <class T>
__global__ void __launch_bounds__(kBlock)
k_emit(const T *__restrict__ q_tm, const T *__restrict__ v_tm, const T *__restrict__ d_tm,
       const int32_t *__restrict__ row_of_pos, T *__restrict__ out, int32_t nseg, int64_t nseg_pad,
       int32_t nsteps, int32_t t_begin, int32_t t_end, int32_t shift_from, int32_t shift, int32_t skip_lo, int32_t skip_hi)
{   // positions [skip_lo, skip_hi) have written their results themselves (k_mc_tile): they are passed over, and the whole
    // 64-position blocks inside that range are not launched at all (blocks from position shift_from on move up by `shift`)
    static_assert(kBlock == 256 && kEmitSteps % 4 == 0, "the passes below assume 4 waves and whole groups of 4 steps");
    constexpr int kRow = 3 * kEmitSteps + 4; // row stride in elements: a multiple of 4, so that 16-byte reads are aligned
    __shared__ __attribute__((aligned(16))) T tile[64][kRow]; // [position][step*3 + c]
    int32_t p0 = (int32_t)blockIdx.x * 64;
    if (p0 >= shift_from) p0 += shift;
    const int32_t t0 = t_begin + (int32_t)blockIdx.y * kEmitSteps; // zero-based output step
    const int32_t nt = min(kEmitSteps, t_end - t0);
    const int32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    {
        const int32_t p = p0 + lane;
        if (p < nseg && !(p >= skip_lo && p < skip_hi)) {
            size_t src = (size_t)(t0 + 1 + wave) * (size_t)nseg_pad + (size_t)p;
            T *dst = &tile[lane][wave * 3];
            for (int32_t tl = wave; tl < nt; tl += 4) {
                dst[0] = q_tm[src];
                dst[1] = v_tm[src];
                dst[2] = d_tm[src];
                src += 4 * (size_t)nseg_pad;
                dst += 12;
            }
        }
    }
    __syncthreads();
    const bool vec = sizeof(T) == 4 && nt == kEmitSteps && ((size_t)nsteps * 3 * sizeof(T)) % 16 == 0 && (t0 % 4) == 0;
    if (vec) {
        // 24 lanes per row (96 floats = 24 x 16 B), two rows per pass: lanes 0-23 and 24-47
        constexpr int kVecPerRow = 3 * kEmitSteps / 4;
        const int32_t half = lane / kVecPerRow, j = lane - half * kVecPerRow;
        if (half < 2) {
            for (int32_t pl = wave * 16 + half; pl < wave * 16 + 16; pl += 2) {
                const int32_t p = p0 + pl;
                if (p >= nseg) break;
                if (p >= skip_lo && p < skip_hi) continue;
                const float4 v = *reinterpret_cast<const float4 *>(&tile[pl][4 * j]);
                float4 *dst = reinterpret_cast<float4 *>(out + ((size_t)row_of_pos[p] * nsteps + t0) * 3);
                dst[j] = v;
            }
        }
    } else {
        for (int32_t pl = wave; pl < 64; pl += kBlock / 64) {
            const int32_t p = p0 + pl;
            if (p >= nseg) break;
            if (p >= skip_lo && p < skip_hi) continue;
            T *dst = out + ((size_t)row_of_pos[p] * nsteps + t0) * 3;
            for (int32_t e = lane; e < nt * 3; e += 64) dst[e] = tile[pl][e];
        }
    }
}

// (`qs`: element stride of the flow plane -- 1 for the time-major planes of the level engine, 2 for the granule plane
// of the dataflow engine, whose elements are {flow, tag} pairs; `d_tm` is then the depth-state column, d_row = 0)
template <class T>
__global__ void __launch_bounds__(kBlock)
k_final_state(const T *__restrict__ q_tm, const T *__restrict__ d_tm, const int32_t *__restrict__ row_of_pos,
              T *__restrict__ q0_out, int32_t nseg, int64_t nseg_pad, int32_t nsteps, int32_t qs, int32_t d_row)
{
    const int32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= nseg) return;
    const size_t src = (size_t)nsteps * nseg_pad + p;
    const size_t r = (size_t)row_of_pos[p] * 3;
    const T q = q_tm[src * qs];
    q0_out[r + 0] = q;
    q0_out[r + 1] = q;
    q0_out[r + 2] = d_tm[(size_t)d_row * nseg_pad + p];
}

template <class T>
__global__ void __launch_bounds__(kBlock)
k_gather_rows(const T *__restrict__ q_tm, const int32_t *__restrict__ pos, T *__restrict__ out, int64_t nrows,
              int64_t nseg_pad, int32_t nsteps, int32_t qs)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= nrows * nsteps) return;
    const int64_t r = i / nsteps;
    const int32_t t = (int32_t)(i % nsteps) + 1;
    out[i] = q_tm[((size_t)t * nseg_pad + pos[r]) * qs];
}

// flows of selected positions over the steps (t_begin, t_end]: out[r * stride + (t - 1 - t_begin)]
template <class T>
__global__ void __launch_bounds__(kBlock)
k_gather_range(const T *__restrict__ q_tm, const int32_t *__restrict__ pos, T *__restrict__ out, int64_t nrows,
               int64_t nseg_pad, int32_t t_begin, int32_t t_end, int64_t stride, int32_t qs)
{
    const int32_t w = t_end - t_begin;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= nrows * w) return;
    const int64_t r = i / w;
    const int32_t k = (int32_t)(i % w);
    out[r * stride + k] = q_tm[((size_t)(t_begin + 1 + k) * nseg_pad + pos[r]) * qs];
}

// boundary positions (the first nboundary of the plan order) <- q[b * stride + (t - 1 - t_begin)], t in (t_begin, t_end]
template <class T>
__global__ void __launch_bounds__(kBlock)
k_fill_boundary_range(const T *__restrict__ q, T *q_tm, T *v_tm, T *d_tm, int32_t nboundary, int64_t nseg_pad,
                      int32_t t_begin, int32_t t_end, int64_t stride, const int64_t *__restrict__ src_index)
{
    const int32_t w = t_end - t_begin;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= (int64_t)nboundary * w) return;
    const int32_t b = (int32_t)(i / w), k = (int32_t)(i % w);
    const size_t dst = (size_t)(t_begin + 1 + k) * nseg_pad + b;
    q_tm[dst] = q[(size_t)(src_index ? src_index[b] : b) * stride + k]; // (src_index: which source row feeds boundary row b)
    v_tm[dst] = T(0);
    d_tm[dst] = T(0);
}

// independent single-segment steps: in[n][15] -> out[n][6] (with courant), cf. reach.pyx:66-103
template <class T, bool TOL = false>
__global__ void __launch_bounds__(kBlock)
k_segments(const T *__restrict__ in, T *__restrict__ out, int32_t *__restrict__ iters_out, int64_t n)
{
    using M = typename DevMath<T, TOL>::type;
    using MX = typename DevMath<T, false>::type; // (the segment-invariant constants are exact in either arithmetic, as in a plan: k_make_const)
    __shared__ uint64_t s_tab[TRMC_POW_TAB_WORDS];
    M m{stage_pow_tables(s_tab), false};
    const MX mx{s_tab, false};
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const T *x = in + i * 15;
    trmc::ChannelParams<T> p;
    trmc::Inflow<T> f;
    p.dt = x[0]; f.qup = x[1]; f.quc = x[2]; f.qdp = x[3]; f.ql = x[4];
    p.dx = x[5]; p.bw = x[6]; p.tw = x[7]; p.twcc = x[8]; p.n = x[9]; p.ncc = x[10];
    p.cs = x[11]; p.s0 = x[12];
    const T depthp = x[14];
    m.coef_ok = coef_guard(p.dt, f.ql);
    const trmc::ChannelConst<T> c = trmc::make_const<T, MX>(p, mx);
    const trmc::StepResult<T> r = trmc::mc_segment_step<T, M>(p, c, f, depthp, m);
    T ck, cn;
    trmc::courant_at<T, M>(r.h, p, c, ck, cn, m);
    T *o = out + i * 6;
    o[0] = r.qdc; o[1] = r.velc; o[2] = r.depthc; o[3] = ck; o[4] = cn; o[5] = r.X;
    if (iters_out) iters_out[i] = r.iters;
}

// ---------------------------------------------------------------- dataflow engine (fp32)
// The level engine above makes a kernel boundary of every dependence: one launch per timestep (or per
// wavefront diagonal), 12 parameter loads and 3 state loads per segment-step, and between launches the whole
// device drains.  The dataflow engine keeps a row in ONE thread for a whole routing window instead:
//   * rows are laid out in BLOCK ORDER (topology.hpp): depth-first post-order cut into blocks of kFlowBlock rows, rows
//     of a block grouped by cost.  A block is a workgroup; it takes a ticket when it starts, so block k only ever
//     needs flows of blocks <= k, all of which are running or done: no deadlock whatever the dispatch order;
//   * a thread loads its row's parameters, constants and state ONCE, then steps through time in registers;
//   * the only thing rows exchange is the flow they pass downstream.  Every row publishes it per step as an 8-byte
//     GRANULE {flow bits, tag = tag_base + step} with one agent-scope store into gran[step][position]; a row reads its
//     upstream rows' granules of the step it needs and, where the tag is not there yet, waits for it (relaxed
//     agent-scope polls, MI355X_MICROARCH.md "handoff-1to1": a self-validating word needs no fence and no flag).
//     A kernel boundary is never needed: producers run ahead of consumers, a consumer that catches up sleeps;
//   * results go straight into the caller's layout out[row][step][q,v,d]: a thread stages 8 steps in LDS and writes
//     them as one 96-byte run (whole 32-byte sectors) -- no time-major planes, no transposing pass.
// Timestep modes: with assume_short_ts a row at step t needs its upstream rows at step t-1; without, also at step t
// (mc_reach.pyx:499-505) -- then rows of one wavefront that feed each other cannot be at the same step, and every
// row trails by its level rank inside the block (lane i works on step t0 + k - rank_i in round k).
// HBM traffic per segment-step: 8 B granule + 12 B result written, <= 16 B of upstream granules read (L2), the
// forcing every qts-th step -- against 64 B algorithmic; what bounds the engine is VALU issue.
#ifndef TRMC_FLOW_BLOCK
#define TRMC_FLOW_BLOCK 256
#endif
#ifndef TRMC_FLOW_WAVES // minimum waves per SIMD the register allocation must allow (workgroups per CU = this * 256 / block)
#define TRMC_FLOW_WAVES 4
#endif
constexpr int kFlowBlock = TRMC_FLOW_BLOCK;
#ifndef TRMC_FLOW_STAGE
#define TRMC_FLOW_STAGE 8
#endif
constexpr int kFlowStage = TRMC_FLOW_STAGE; // steps staged per thread before they are written to `out`

struct FlowArgs {
    const float *dx, *bw, *twcc, *n, *ncc, *s0;
    const float *z, *bfd, *sqrt_s0, *sq1pz2, *s0_n, *s0_ncc, *inv_n;
    const float *dt_col;
    float dt;
    const int32_t *up_ptr, *up_idx;
    const int2 *up2;
    const int32_t *lag;            // short-timestep mode: steps a row trails by (trmc_plan_set_lag); general mode: its
                                   // level rank inside the block; nullptr = none
    const float *qlat_tm;
    unsigned long long *gran;      // [nsteps + 1][nseg_pad] granules
    float *d_state;                // [nseg_pad] depth at the last step each row has completed
    unsigned long long *d_gran;    // [nseg_pad] the same as a granule {depth bits, tag of that step}: hand-over between
                                   // consecutive launches of one window that overlap in time (k_mc_flow_lean)
    float *out;                    // [nseg][nsteps][3]
    const int32_t *row_of_pos;
    uint8_t *it_prev;
    uint16_t *it_sum;
    bool sane, out_vec;            // out_vec: the 8-step runs of `out` are 16-byte aligned (nsteps % 4 == 0)
    const int32_t *res_of_pos;
    const float *res_par;
    float *res_inflow;
    float res_dt;
    const int32_t *gage_of_pos;
    const uint8_t *da_mode;
    const float *da_a, *da_w;
    float *da_nudge;
    int64_t nseg_pad;
    int32_t nsteps, qts, nseg, first; // first: position of the first routed row (= number of boundary rows)
    uint32_t tag_base;
    int32_t *ticket;               // [0] block tickets of this launch, [1] abort flag of the window
    uint64_t watchdog_ticks;       // wall_clock64 ticks (100 MHz) a row may wait for one granule
    const uint8_t *prio;           // issue priority 0..3 of every wavefront of the block order (topology.cpp)
    // blocks dealt to compute units by cost (lean kernels, see flow_place_blocks): queue q holds cuq_blk[cuq_ptr[q] ..
    // cuq_ptr[q + 1]), cuq_head[q] counts what has been taken, cu_index maps hw_cu_key() to a queue; nullptr = block tickets
    const int32_t *cuq_ptr, *cuq_blk, *cu_index;
    const uint8_t *cuq_perm;       // [nblocks] row group of the block for SIMD s: bits 2s+1..2s
    int32_t *cuq_head;
    int32_t ncuq;
    unsigned long long *dbg;       // nullptr, or [nblocks][2]: wall clock at the start and the end of every block (TRMC_FLOW_DEBUG)
    int32_t nblocks_dbg;
    const int32_t *ticket_map;     // general mode: ticket -> block, the blocks of the long main stems first (topology.hpp,
                                   // stem_min_rows); nullptr = block tickets in order
};

using FlowCold = ColdArgs<FlowArgs>;

// which compute unit a wavefront runs on: XCC_ID[3:0] | HW_ID {se_id[15:13], sh_id[12], cu_id[11:8]} -> 12 bits
__device__ __forceinline__ uint32_t hw_cu_key()
{
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID
    const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID
    return ((xcc & 15u) << 8) | ((hw >> 8) & 0xffu);
}
__device__ __forceinline__ unsigned long long gran_load(const unsigned long long *g)
{
    return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// flow of position `u` at the step whose tag is `want`; waits until it has been published.  A waiting row costs
// the others as little as possible: long sleeps, and a poll counter for a watchdog instead of a clock read per poll
// (the clock is only consulted every 1024th poll).
#ifndef TRMC_FLOW_SLEEP
#define TRMC_FLOW_SLEEP 16 // x 64 clocks between two polls of a granule that is not there yet
#endif
__device__ __forceinline__ bool flow_watchdog(uint32_t &polls, uint64_t &t_start, FlowCold a)
{
    if ((++polls & 1023u) != 0u) return false;
    if (t_start == 0) {
        t_start = wall_clock64();
        return false;
    }
    if (wall_clock64() - t_start > a->watchdog_ticks
        || __hip_atomic_load(a->ticket + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
        __hip_atomic_store(a->ticket + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return true;
    }
    return false;
}
// what the first row to give up was waiting for, for the host's error message: ticket[2..5] = {granule index in the
// plane (low, high word), wanted tag, tag found}
__device__ __forceinline__ void flow_report(FlowCold a, const unsigned long long *g, uint32_t want, unsigned long long v)
{
    if (atomicCAS(a->ticket + 6, 0, 1) == 0) {
        const unsigned long long idx = (unsigned long long)(g - a->gran);
        a->ticket[2] = (int32_t)(idx & 0xffffffffull);
        a->ticket[3] = (int32_t)(idx >> 32);
        a->ticket[4] = (int32_t)want;
        a->ticket[5] = (int32_t)(v >> 32);
    }
}
__device__ __forceinline__ float flow_wait(const unsigned long long *g, uint32_t want, FlowCold a, bool &dead)
{
    unsigned long long v = gran_load(g);
    if ((uint32_t)(v >> 32) != want) {
        uint32_t polls = 0;
        uint64_t t_start = 0;
        do {
            __builtin_amdgcn_s_sleep(TRMC_FLOW_SLEEP);
            v = gran_load(g);
            if (flow_watchdog(polls, t_start, a)) {
                dead = true;
                flow_report(a, g, want, v);
            }
        } while ((uint32_t)(v >> 32) != want && !dead);
    }
    return __uint_as_float((uint32_t)v);
}

// One upstream edge of a row.  In-block edges are read from the block's LDS ring while producer and consumer run in
// step; an edge whose producer runs well ahead (its ring slot is already overwritten: a cheap row feeding a costly one)
// or lives in another block is read from the granule plane, one step ahead of its use (the load is in flight during the
// arithmetic of the current step).
struct FlowEdge {
    int32_t u;                // plan position of the upstream row, -1 = none
    int32_t l;                // its index in the block's LDS ring, -1 = not eligible (other block, other lag)
    bool ahead;               // read through the granule plane, prefetched
    bool pre_ok;
    unsigned long long pre;   // the prefetched granule
};
#ifndef TRMC_FLOW_RING
#define TRMC_FLOW_RING 4
#endif
constexpr int kFlowRing = TRMC_FLOW_RING;  // steps the LDS ring of a block holds (a power of two)

__device__ __forceinline__ float flow_edge_get(FlowEdge &e, const unsigned long long *plane_row, unsigned long long *ring,
                                               int32_t ws, uint32_t want, FlowCold a, bool &dead)
{
    unsigned long long v = e.pre;
    if (!(e.ahead && e.pre_ok && (uint32_t)(v >> 32) == want)) {
        bool got = false;
        if (e.l >= 0) {
            const unsigned long long *slot = ring + (size_t)(ws & (kFlowRing - 1)) * kFlowBlock + e.l;
            v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if ((int32_t)((uint32_t)(v >> 32) - want) < 0) { // not produced yet: the producer is a wave of this block
                uint32_t polls = 0;
                uint64_t t_start = 0;
                do {
                    __builtin_amdgcn_s_sleep(TRMC_FLOW_SLEEP);
                    v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (flow_watchdog(polls, t_start, a)) dead = true;
                } while ((int32_t)((uint32_t)(v >> 32) - want) < 0 && !dead);
            }
            got = (uint32_t)(v >> 32) == want;
            e.ahead = !got; // overwritten: the producer is ahead by more than the ring holds
        }
        if (!got) {
            const float q = flow_wait(plane_row + e.u, want, a, dead);
            return q;
        }
    }
    return __uint_as_float((uint32_t)v);
}

template <bool SHORT, bool TOL = false>
__global__ void __launch_bounds__(kFlowBlock, TRMC_FLOW_WAVES)
k_mc_flow(const FlowArgs a, const int32_t t0, const int32_t t1) // routes the launches / steps (t0, t1] of the window
{
    using M = std::conditional_t<TOL, DevMathTolFlow, DevMathFlow>;
    const FlowCold cold = cold_args<FlowArgs>(); // (see cold_args: what the loop rarely needs is not kept in registers)
    __shared__ uint64_t s_tab[TRMC_POW_TAB_WORDS];
    __shared__ float s_out[3 * kFlowStage * kFlowBlock];             // [step slot * 3 + c][thread]
    __shared__ unsigned long long s_ring[kFlowRing * kFlowBlock];    // [step % kFlowRing][thread] granules
    __shared__ int32_t s_blk;
    if (threadIdx.x == 0) {
        // General mode on a plan laid out for it: the first tickets go to the blocks of the long main stems.  They wait for
        // their inflows in place -- a stem then advances behind the sweep over its basin instead of after it -- and every
        // other block still only needs blocks that took their tickets before it or are among those few, all resident.
        const int32_t tk = atomicAdd(a.ticket, 1);
        s_blk = (!SHORT && cold->ticket_map) ? cold->ticket_map[tk] : tk;
    }
#pragma unroll
    for (int j = 0; j < kFlowRing; ++j) s_ring[j * kFlowBlock + threadIdx.x] = 0ull; // tag 0: older than any live tag
    M m{stage_pow_tables(s_tab), false}; // (its barrier also publishes s_blk and the cleared ring)
    m.sane = a.sane;

    const int32_t blk_base = a.first + s_blk * kFlowBlock;
    const int32_t pos = blk_base + (int32_t)threadIdx.x;
    const bool valid = pos < a.nseg;
    const uint32_t su = valid ? (uint32_t)pos : (uint32_t)a.first;
    const uint32_t ob = su * 4u;
    const int32_t lag = a.lag ? a.lag[su] : 0;
    if (cold->dbg && threadIdx.x == 0) cold->dbg[2 * s_blk] = ((unsigned long long)hw_cu_key() << 48) | (wall_clock64() & 0xffffffffffffull);
    // the steps [t_lo, t_hi] this row covers in this launch, and the round it starts in
    const int32_t t_lo = SHORT ? max(t0 - lag, 0) + 1 : t0 + 1;
    const int32_t t_hi = valid ? (SHORT ? min(t1 - lag, a.nsteps) : min(t1, a.nsteps)) : 0;
    const int32_t delay = SHORT ? 0 : lag;
    // General mode: a wavefront whose rows trail each other deeply sits on a long chain of the network -- the critical
    // path of the window (a row at level l cannot finish step t before l rows have, one after the other).  It gets
    // issue priority over the wavefronts it shares its SIMD with, so that the chain advances at the pace of one
    // wavefront alone while the bulk of the network fills the remaining issue slots.
    if (!SHORT && __any(delay >= 16)) __builtin_amdgcn_s_setprio(3);

    trmc::ChannelParams<float> p;
    p.dt = a.dt_col ? at(a.dt_col, ob) : a.dt;
    p.dx = at(a.dx, ob);
    p.bw = at(a.bw, ob);
    p.twcc = at(a.twcc, ob);
    p.n = at(a.n, ob);
    p.ncc = at(a.ncc, ob);
    p.s0 = at(a.s0, ob);
    p.tw = p.cs = 0.0f; // only enter the constants below
    trmc::ChannelConst<float> c;
    c.z = at(a.z, ob);
    c.bfd = at(a.bfd, ob);
    c.sqrt_s0 = at(a.sqrt_s0, ob);
    c.sq1pz2 = at(a.sq1pz2, ob);
    c.s0_n = at(a.s0_n, ob);
    c.s0_ncc = at(a.s0_ncc, ob);
    c.inv_n = at(a.inv_n, ob);
    trmc::derive_const(c, p);
    const int2 up = a.up2[su];
    const bool more = up.y >= 0 && (up.y & 0x40000000);
    FlowEdge e0, e1;
    {
        auto init = [&](FlowEdge &e, int32_t u) {
            e.u = valid ? u : -1;
            e.l = -1;
            e.pre = 0ull;
            e.pre_ok = false;
            if (e.u >= 0) {
                const bool inb = u >= blk_base && u < blk_base + kFlowBlock;
                // (a skewed row and a row in step never share the ring: their step windows differ)
                const bool same = !SHORT || !a.lag || a.lag[u] == lag;
                if (inb && same) e.l = u - blk_base;
            }
            e.ahead = e.l < 0;
        };
        init(e0, up.x);
        init(e1, up.y >= 0 ? (up.y & 0x3fffffff) : -1);
    }
    const int32_t ri = a.res_of_pos ? a.res_of_pos[su] : -1;
    const int32_t gi = a.gage_of_pos ? a.gage_of_pos[su] : -1;
    const size_t np = (size_t)a.nseg_pad;
    float *const out_row = a.out + (size_t)a.row_of_pos[su] * (size_t)a.nsteps * 3;

    float q_prev = 0.0f, d_prev = 0.0f, ql = 0.0f, xp0 = 0.0f, xp1 = 0.0f;
    int32_t ql_col = -1, ql_left = 0, staged = 0, it_acc = 0, it_last = 0;
    bool have_state = false, dead = false;

    for (int32_t k = 0;; ++k) {
        const int32_t t = t_lo + k - delay;
        if (!__any(t <= t_hi) || __any(dead)) break;
        if (t < t_lo || t > t_hi) continue;
        const uint32_t tag_p = a.tag_base + (uint32_t)(t - 1);
        const unsigned long long *g_prev = a.gran + (size_t)(t - 1) * np;
        unsigned long long *g_curr = a.gran + (size_t)t * np;
        if (!have_state) { // the state this row was left in: its own granule of step t - 1, its depth column
            q_prev = flow_wait(g_prev + su, tag_p, cold, dead);
            d_prev = cold->d_state[su];
            __hip_atomic_store(s_ring + (size_t)((t - 1) & (kFlowRing - 1)) * kFlowBlock + threadIdx.x,
                               ((unsigned long long)tag_p << 32) | (unsigned long long)__float_as_uint(q_prev),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (!SHORT) { // the general mode also needs its upstream rows at the step before its first one
                if (e0.u >= 0) xp0 = flow_wait(g_prev + e0.u, tag_p, cold, dead);
                if (e1.u >= 0) xp1 = flow_wait(g_prev + e1.u, tag_p, cold, dead);
            }
            // the lateral-inflow column of step t is (t - 1) / qts: found by division once, by a counter from then on
            const int32_t qts = cold->qts;
            ql_col = (t - 1) / qts;
            ql_left = qts - (t - 1) % qts;
            ql = cold->qlat_tm[(size_t)ql_col * np + su];
            have_state = true;
        }
        if (ql_left == 0) {
            ++ql_col;
            ql = cold->qlat_tm[(size_t)ql_col * np + su];
            ql_left = cold->qts;
        }
        --ql_left;
        // (Evaluating the part of the step that needs the row's OWN state only -- step_pre: the bracket and its two hydraulic
        // points -- BEFORE the look-up below, off the dependence chain, was built and measured: the general-mode CONUS day
        // 55 ms instead of 38, a lone chain 8.3 us per row instead of 7.4.  The chain mostly runs through the lanes of ONE
        // wavefront, which advance a row per round whatever the order inside the round, and the two points held across the
        // look-up cost registers the kernel does not have.  What stayed: the flow is published before the velocity is formed.)
        trmc::StepPre<float> pre;
        pre.have = false;
        // junction sums in the reference's order (mc_reach.pyx:499-502): with assume_short_ts the upstream flows of
        // step t - 1 (they are also `quc`, :504-505), without it those of step t and -- kept from the round before --
        // of step t - 1
        const int32_t ws = SHORT ? t - 1 : t;
        const uint32_t want = SHORT ? tag_p : tag_p + 1u;
        const unsigned long long *g_want = SHORT ? g_prev : g_curr;
        float x0 = 0.0f, x1 = 0.0f;
        if (e0.u >= 0) x0 = flow_edge_get(e0, g_want, s_ring, ws, want, cold, dead);
        if (e1.u >= 0) x1 = flow_edge_get(e1, g_want, s_ring, ws, want, cold, dead);
        // edges read through the granule plane: next step's granule starts its way here now
        if (t < t_hi) {
            if (e0.u >= 0 && e0.ahead) e0.pre = gran_load(g_want + np + e0.u);
            if (e1.u >= 0 && e1.ahead) e1.pre = gran_load(g_want + np + e1.u);
        }
        e0.pre_ok = e1.pre_ok = t < t_hi;
        float qup = 0.0f, quc = 0.0f;
        if (e0.u >= 0) {
            qup += SHORT ? x0 : xp0;
            quc += x0;
        }
        if (e1.u >= 0) {
            qup += SHORT ? x1 : xp1;
            quc += x1;
        }
        if (more) { // fan-in above two (0.2 % of junctions): straight from the granule plane
            const int32_t *const up_ptr = cold->up_ptr, *const up_idx = cold->up_idx;
            const int32_t k1 = up_ptr[su + 1];
            for (int32_t e = up_ptr[su] + 2; e < k1; ++e) {
                const int32_t ue = up_idx[e];
                qup += flow_wait(g_prev + ue, tag_p, cold, dead);
                if (!SHORT) quc += flow_wait(g_curr + ue, tag_p + 1u, cold, dead);
            }
        }
        xp0 = x0;
        xp1 = x1;
        trmc::Inflow<float> f;
        f.qup = qup;
        f.quc = SHORT ? qup : quc;
        f.qdp = q_prev;
        f.ql = ql;

        float q_new, v_new = 0.0f, d_new;
        bool routed = false;
        if (ri >= 0) { // level-pool reservoir row, mc_reach.pyx:507-510,:551-553,:706-710 (see k_mc_step)
            const float *rp = cold->res_par + (size_t)ri * 9;
            const trmc::LevelPoolParams<float> lp{rp[0], rp[1], rp[2], rp[3], rp[4], rp[5], rp[6], rp[7], rp[8]};
            float H = d_prev;
            q_new = trmc::levelpool_step<float, M>(f.quc, 0.0f, cold->res_dt, H, lp, m);
            d_new = H;
            cold->res_inflow[(size_t)ri * (size_t)cold->nsteps + (size_t)(t - 1)] = f.quc;
            it_last = 0;
        } else {
            q_new = 0.0f;
            d_new = 0.0f;
            it_last = 0;
            if (trmc::step_has_flow(f)) {
                m.coef_ok = coef_guard(p.dt, f.ql);
                const trmc::StepSolve<float> r = trmc::step_solve<float, M>(p, c, f, d_prev, pre, m);
                q_new = r.qdc;
                d_new = r.h;
                routed = true;
                it_last = min(r.iters, 255);
                it_acc += min(r.iters, 3) + (r.over ? 4 : 0);
            }
            if (gi >= 0) { // streamflow nudging, mc_reach.pyx:761-796 / simple_da.pyx:47-76 (see k_mc_step)
                const size_t e = (size_t)gi * (size_t)cold->nsteps + (size_t)(t - 1);
                const float *const da_a = cold->da_a;
                const uint8_t mode = cold->da_mode[e];
                float nudge = 0.0f;
                if (mode == 1) {
                    nudge = da_a[e] - q_new;
                    q_new = da_a[e];
                } else if (mode == 2) {
                    nudge = (da_a[e] - q_new) * cold->da_w[e];
                    q_new = q_new + nudge;
                }
                cold->da_nudge[e] = nudge;
            }
        }
        // publish the flow as soon as it exists -- the block's ring, and one 8-byte agent-scope store into the plane; tag
        // in the high word -- and only then form the velocity (a power, a square root, a division: f90:163-169), which no
        // other row reads
        {
            const unsigned long long g = ((unsigned long long)(tag_p + 1u) << 32) | (unsigned long long)__float_as_uint(q_new);
            __hip_atomic_store(s_ring + (size_t)(t & (kFlowRing - 1)) * kFlowBlock + threadIdx.x, g, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(g_curr + su, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (routed) v_new = trmc::step_velocity<float, M>(p, c, d_new, m);
        q_prev = q_new;
        d_prev = d_new;
        // stage (q, v, d) of step t; a run ends at every 8th step of the window and at the last step of the launch
        {
            const int32_t slot = (t - 1) & (kFlowStage - 1);
            float *so = s_out + (size_t)(slot * 3) * kFlowBlock + threadIdx.x;
            so[0] = q_new;
            so[kFlowBlock] = v_new;
            so[2 * kFlowBlock] = d_new;
            ++staged;
            if (slot == kFlowStage - 1 || t == t_hi) {
                const int32_t s_first = slot + 1 - staged; // first staged slot
                float *dst = out_row + (size_t)(t - staged) * 3;
                const float *si = s_out + (size_t)(s_first * 3) * kFlowBlock + threadIdx.x;
                if (staged == kFlowStage && a.out_vec) {
#pragma unroll
                    for (int j = 0; j < 3 * kFlowStage / 4; ++j) {
                        float4 v;
                        v.x = si[(4 * j + 0) * kFlowBlock];
                        v.y = si[(4 * j + 1) * kFlowBlock];
                        v.z = si[(4 * j + 2) * kFlowBlock];
                        v.w = si[(4 * j + 3) * kFlowBlock];
                        reinterpret_cast<float4 *>(dst)[j] = v;
                    }
                } else {
                    for (int32_t e = 0; e < 3 * staged; ++e) dst[e] = si[e * kFlowBlock];
                }
                staged = 0;
            }
        }
    }
    if (cold->dbg) atomicMax(cold->dbg + 2 * s_blk + 1, (unsigned long long)wall_clock64());
    if (valid && have_state) {
        cold->d_state[su] = d_prev;
        if (t_hi == cold->nsteps) cold->it_prev[su] = (uint8_t)it_last;
        if (uint16_t *const it_sum = cold->it_sum) it_sum[su] = (uint16_t)min(65535, (int)it_sum[su] + it_acc);
    }
}

// The short-timestep form of the engine, written for occupancy: with assume_short_ts a row only ever needs flows of the
// step before, all rows of a block advance together, and the whole window is VALU-bound -- what decides the pace of a
// partly filled device (one rank of a multi-GPU job: 340 k rows are 5 455 wavefronts against 4 096 slots at four per
// SIMD) is whether every wavefront of the job is resident at once.  So a thread keeps only what it must in registers
// (its flow, its depth, its forcing, two upstream positions, one result offset): the twelve parameter and constant
// columns of its row wait in LDS (48 B per row, read back at every step), results are stored step by step (12 B;
// neighbouring steps of a row merge in the L2 / Infinity Cache), the LDS ring holds two steps.  LAG: rows with a skew
// (trmc_plan_set_lag) -- then the step is a per-lane quantity; without, it is wave-uniform and lives in SGPRs.
#ifndef TRMC_LEAN_WAVES
#define TRMC_LEAN_WAVES 6
#endif
#ifndef TRMC_LEAN_RING_SLEEP // x 64 clocks between two polls of the block's LDS ring
#define TRMC_LEAN_RING_SLEEP 2
#endif
constexpr int kLeanRing = 2;
constexpr int kLeanCols = 13;

__device__ __forceinline__ float lean_edge_get(int32_t u, int32_t l, uint32_t &flags, uint32_t ahead_bit, uint32_t never_bit,
                                               const unsigned long long *plane_row, const unsigned long long *ring, int32_t ws,
                                               uint32_t want, FlowCold a, bool &dead)
{
    if (!(flags & (ahead_bit | never_bit))) {
        const unsigned long long *slot = ring + (size_t)(ws & (kLeanRing - 1)) * kFlowBlock + l;
        unsigned long long v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((int32_t)((uint32_t)(v >> 32) - want) < 0) { // not produced yet: the producer is a wave of this block
            uint32_t polls = 0;
            uint64_t t_start = 0;
            do {
                __builtin_amdgcn_s_sleep(TRMC_LEAN_RING_SLEEP);
                v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (flow_watchdog(polls, t_start, a)) dead = true;
            } while ((int32_t)((uint32_t)(v >> 32) - want) < 0 && !dead);
        }
        if ((uint32_t)(v >> 32) == want) return __uint_as_float((uint32_t)v);
        flags |= ahead_bit; // overwritten: the producer runs ahead of what the ring holds
    }
    return flow_wait(plane_row + u, want, a, dead);
}

// Which block a workgroup of a lean launch routes.  Every block of such a launch is resident at once (flow_lean), each
// stays on its compute unit for the whole launch, and a SIMD issues for its wavefronts one instruction at a time: the launch
// lasts as long as the most heavily loaded compute unit needs.  The hardware deals workgroups out by COUNT (5 or 6 per
// unit for a 340 k-row rank); by block tickets the costliest unit of such a rank carried 1.5 times the mean (measured
// per-unit end times 0.8 .. 4.5 ms, correlation with the modelled load 0.8).  So the host deals the blocks to per-unit queues
// by cost (flow_place_blocks), a workgroup finds out where it runs (hw_cu_key) and takes the next block of that unit's
// queue -- or, when the hardware sent the unit more workgroups than its queue holds, of the nearest queue that has one left.
// Any assignment is correct: nothing depends on the order blocks start in while all of them are resident.
__device__ __forceinline__ int32_t lean_pick_block(const FlowArgs &a)
{
    if (!a.cuq_blk) return atomicAdd(a.ticket, 1);
    const int32_t Q = a.ncuq;
    const uint32_t key = hw_cu_key();
    int32_t q = a.cu_index[key];
    if (q < 0) q = (int32_t)(key % (uint32_t)Q);
    // cuq_head[q]: low half = blocks taken from the front (by workgroups of the unit itself: its costliest first), high half =
    // blocks taken from the back (by workgroups of other units whose own queue had run out: the cheapest); both only grow,
    // and a take counts if front + back was below the queue length before it
    for (int32_t k = 0; k < Q; ++k) {
        const int32_t qq = q + k < Q ? q + k : q + k - Q;
        const int32_t lo = a.cuq_ptr[qq], n = a.cuq_ptr[qq + 1] - lo;
        const uint32_t seen = (uint32_t)__hip_atomic_load(a.cuq_head + qq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int32_t)((seen & 0xffffu) + (seen >> 16)) >= n) continue;
        const uint32_t was = (uint32_t)atomicAdd(a.cuq_head + qq, k == 0 ? 1 : 0x10000);
        const int32_t front = (int32_t)(was & 0xffffu), back = (int32_t)(was >> 16);
        if (front + back < n) return a.cuq_blk[lo + (k == 0 ? front : n - 1 - back)];
    }
    return -1; // (cannot happen: as many workgroups as blocks)
}

template <bool LAG, bool TOL = false>
__global__ void __launch_bounds__(kFlowBlock, TRMC_LEAN_WAVES)
k_mc_flow_lean(const FlowArgs a, const int32_t t0, const int32_t t1)
{
    using M = std::conditional_t<TOL, DevMathTolFlow, DevMathFlow>;
    const FlowCold cold = cold_args<FlowArgs>(); // (see cold_args: what the loop rarely needs is not kept in registers)
    __shared__ uint64_t s_tab[TRMC_POW_TAB_WORDS];
    __shared__ float s_par[kLeanCols * kFlowBlock];                 // [column][thread]
    __shared__ unsigned long long s_ring[kLeanRing * kFlowBlock];   // [step % 2][thread] granules
    __shared__ int32_t s_blk;
    __shared__ int32_t s_grp[kFlowBlock / 64];
    if (threadIdx.x == 0) s_blk = lean_pick_block(a);
    if (threadIdx.x < kFlowBlock / 64) s_grp[threadIdx.x] = -1;
#pragma unroll
    for (int j = 0; j < kLeanRing; ++j) s_ring[j * kFlowBlock + threadIdx.x] = 0ull;
    M m{stage_pow_tables(s_tab), false}; // (its barrier also publishes s_blk and the cleared ring)
    m.sane = a.sane;
    if (s_blk < 0) return;
    // Which 64 rows of the block this wavefront takes.  A block's rows are grouped by descending cost, so its first
    // wavefront is its costliest (1 010 instructions per step against 730 for the others, 8-way CONUS rank) -- and if that
    // always lands on the same SIMD of its unit, that SIMD carries 1.3 times the others' load.  The host has matched the
    // block's four row groups with the four SIMDs of the unit it dealt the block to (costliest group to the SIMD carrying
    // least, flow_place_blocks); a wavefront reads which SIMD it is on and takes that SIMD's group.  Should two wavefronts
    // of the workgroup share a SIMD, everybody keeps the plain order.
    int32_t grp = (int32_t)(threadIdx.x >> 6);
    uint32_t my_simd = 0;
    if (a.cuq_perm) {
        my_simd = (__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4) >> 4) & 3u; // HW_REG_HW_ID.simd_id
        const int32_t g = (int32_t)((a.cuq_perm[s_blk] >> (2u * my_simd)) & 3u);
        if ((threadIdx.x & 63u) == 0) s_grp[g] = (int32_t)(threadIdx.x >> 6);
        __syncthreads();
        bool all_taken = true;
#pragma unroll
        for (int j = 0; j < kFlowBlock / 64; ++j) all_taken = all_taken && s_grp[j] >= 0;
        if (all_taken) grp = g;
    }
    const int32_t tix = grp * 64 + (int32_t)(threadIdx.x & 63u); // the row of the block this thread routes
    const int32_t blk_base = a.first + s_blk * kFlowBlock;
    if (cold->dbg && tix == 0) cold->dbg[2 * s_blk] = ((unsigned long long)hw_cu_key() << 48) | (wall_clock64() & 0xffffffffffffull);
    if (cold->dbg && (threadIdx.x & 63u) == 0) // which SIMD every row group ran on
        cold->dbg[2 * (size_t)(cold->nblocks_dbg + 1) + 4 * (size_t)s_blk + grp] = 1ull + my_simd + ((unsigned long long)(threadIdx.x >> 6) << 8);
    if (a.prio) { // the costlier a wavefront, the higher its issue priority (topology.cpp)
        const int pr = __builtin_amdgcn_readfirstlane((int)a.prio[(s_blk * kFlowBlock + tix) >> 6]);
        if (pr == 1) __builtin_amdgcn_s_setprio(1);
        else if (pr == 2) __builtin_amdgcn_s_setprio(2);
        else if (pr == 3) __builtin_amdgcn_s_setprio(3);
    }
    if (blk_base + tix >= a.nseg) return; // (no block-wide barrier below)
    const uint32_t su = (uint32_t)(blk_base + tix);
    {
        const uint32_t ob = su * 4u;
        float *sp = s_par + tix;
        sp[0 * kFlowBlock] = at(a.dx, ob);
        sp[1 * kFlowBlock] = at(a.bw, ob);
        sp[2 * kFlowBlock] = at(a.twcc, ob);
        sp[3 * kFlowBlock] = at(a.n, ob);
        sp[4 * kFlowBlock] = at(a.ncc, ob);
        sp[5 * kFlowBlock] = at(a.s0, ob);
        sp[6 * kFlowBlock] = at(a.z, ob);
        sp[7 * kFlowBlock] = at(a.bfd, ob);
        sp[8 * kFlowBlock] = at(a.sqrt_s0, ob);
        sp[9 * kFlowBlock] = at(a.sq1pz2, ob);
        sp[10 * kFlowBlock] = at(a.s0_n, ob);
        sp[11 * kFlowBlock] = at(a.s0_ncc, ob);
        sp[12 * kFlowBlock] = at(a.inv_n, ob);
    }
    const float dt = a.dt_col ? a.dt_col[su] : a.dt;
    const int32_t lag = LAG ? a.lag[su] : 0;
    const int32_t t_lo = LAG ? max(t0 - lag, 0) + 1 : t0 + 1;
    const int32_t t_hi = LAG ? min(t1 - lag, a.nsteps) : min(t1, a.nsteps);
    // flag bits: 0 edge 0 ahead, 1 edge 0 never through the ring, 2 / 3 the same for edge 1, 4 more than two upstream
    // rows, 5 reservoir row, 6 gage row
    uint32_t flags = 0;
    int32_t u0, u1;
    {
        const int2 up = a.up2[su];
        u0 = up.x;
        u1 = up.y >= 0 ? (up.y & 0x3fffffff) : -1;
        if (up.y >= 0 && (up.y & 0x40000000)) flags |= 16u;
        auto ring_ok = [&](int32_t u) { return u >= blk_base && u < blk_base + kFlowBlock && (!LAG || a.lag[u] == lag); };
        if (u0 >= 0 && !ring_ok(u0)) flags |= 2u;
        if (u1 >= 0 && !ring_ok(u1)) flags |= 8u;
        if (a.res_of_pos && a.res_of_pos[su] >= 0) flags |= 32u;
        if (a.gage_of_pos && a.gage_of_pos[su] >= 0) flags |= 64u;
    }
    const size_t np = (size_t)a.nseg_pad;
    const uint32_t out_idx = (uint32_t)a.row_of_pos[su] * (uint32_t)a.nsteps * 3u; // (the host checks nseg * nsteps * 3 < 2**32)
    bool dead = false;
    if (t_lo > t_hi) return;
    // the state this row was left in -- possibly by a launch that is still running (consecutive time chunks of a window
    // overlap on two streams): its own flow granule of step t_lo - 1 and the depth granule tagged with the same step
    float q_prev = flow_wait(a.gran + (size_t)(t_lo - 1) * np + su, a.tag_base + (uint32_t)(t_lo - 1), cold, dead);
    float d_prev = flow_wait(cold->d_gran + su, a.tag_base + (uint32_t)(t_lo - 1), cold, dead);
    __hip_atomic_store(s_ring + (size_t)((t_lo - 1) & (kLeanRing - 1)) * kFlowBlock + tix,
                       ((unsigned long long)(a.tag_base + (uint32_t)(t_lo - 1)) << 32) | (unsigned long long)__float_as_uint(q_prev),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    // the lateral-inflow column of step t is (t - 1) / qts: the first one is read now, the next ones as the counter runs out
    int32_t ql_col = (t_lo - 1) / a.qts;
    int32_t ql_left = a.qts - (t_lo - 1) % a.qts;
    float ql = a.qlat_tm[(size_t)ql_col * np + su];
    ++ql_col;
    uint32_t its = 0; // iterations: low 24 bits the sum of min(iterations, 3), high 8 bits those of the last step
    for (int32_t t = t_lo; t <= t_hi && !dead; ++t) {
        const uint32_t tag_p = a.tag_base + (uint32_t)(t - 1);
        const unsigned long long *g_prev = a.gran + (size_t)(t - 1) * np;
        if (ql_left == 0) { // (a counter, not (t - 1) % qts and (t - 1) / qts: two integer divisions per step otherwise)
            ql = cold->qlat_tm[(size_t)ql_col * np + su];
            ql_left = cold->qts;
            ++ql_col;
        }
        --ql_left;
        if ((t & 15) == 0) flags &= ~5u; // a producer that ran ahead may have been caught up with: try the ring again
        // junction sum in the reference's order (mc_reach.pyx:499-505)
        float qup = 0.0f;
        if (u0 >= 0) qup += lean_edge_get(u0, u0 - blk_base, flags, 1u, 2u, g_prev, s_ring, t - 1, tag_p, cold, dead);
        if (u1 >= 0) qup += lean_edge_get(u1, u1 - blk_base, flags, 4u, 8u, g_prev, s_ring, t - 1, tag_p, cold, dead);
        if (flags & 16u) {
            const int32_t *const up_ptr = cold->up_ptr, *const up_idx = cold->up_idx;
            const int32_t k1 = up_ptr[su + 1];
            for (int32_t e = up_ptr[su] + 2; e < k1; ++e) qup += flow_wait(g_prev + up_idx[e], tag_p, cold, dead);
        }
        float q_new = 0.0f, v_new = 0.0f, d_new = 0.0f;
        bool routed = false;
        trmc::ChannelParams<float> p;
        trmc::ChannelConst<float> c;
        if (flags & 32u) { // level-pool reservoir row (see k_mc_step)
            const int32_t ri = cold->res_of_pos[su];
            const int32_t nsteps = cold->nsteps;
            const float *rp = cold->res_par + (size_t)ri * 9;
            const trmc::LevelPoolParams<float> lp{rp[0], rp[1], rp[2], rp[3], rp[4], rp[5], rp[6], rp[7], rp[8]};
            float H = d_prev;
            q_new = trmc::levelpool_step<float, M>(qup, 0.0f, cold->res_dt, H, lp, m);
            d_new = H;
            cold->res_inflow[(size_t)ri * (size_t)nsteps + (size_t)(t - 1)] = qup;
            its &= 0x00ffffffu;
        } else {
            trmc::Inflow<float> f;
            f.qup = qup;
            f.quc = qup;
            f.qdp = q_prev;
            f.ql = ql;
            uint32_t it_now = 0, it_cost = 0;
            if (trmc::step_has_flow(f)) {
                // (with assume_short_ts a row reads flows its upstream rows published a step ago: no dependence chain runs
                // through the step, so nothing of it is hoisted above the look-up -- see k_mc_flow -- and the two points of
                // the bracket are formed inside step_solve)
                const float *sp = s_par + tix;
                p.dt = dt;
                p.dx = sp[0 * kFlowBlock];
                p.bw = sp[1 * kFlowBlock];
                p.twcc = sp[2 * kFlowBlock];
                p.n = sp[3 * kFlowBlock];
                p.ncc = sp[4 * kFlowBlock];
                p.s0 = sp[5 * kFlowBlock];
                p.tw = p.cs = 0.0f;
                c.z = sp[6 * kFlowBlock];
                c.bfd = sp[7 * kFlowBlock];
                c.sqrt_s0 = sp[8 * kFlowBlock];
                c.sq1pz2 = sp[9 * kFlowBlock];
                c.s0_n = sp[10 * kFlowBlock];
                c.s0_ncc = sp[11 * kFlowBlock];
                c.inv_n = sp[12 * kFlowBlock];
                trmc::derive_const(c, p);
                trmc::StepPre<float> pre;
                pre.have = false;
                m.coef_ok = coef_guard(p.dt, f.ql);
                const trmc::StepSolve<float> r = trmc::step_solve<float, M>(p, c, f, d_prev, pre, m);
                q_new = r.qdc;
                d_new = r.h;
                routed = true;
                it_now = (uint32_t)min(r.iters, 255);
                it_cost = (uint32_t)min(r.iters, 3) + (r.over ? 4u : 0u);
            }
            its = ((its + it_cost) & 0x00ffffffu) | (it_now << 24);
            if (flags & 64u) { // streamflow nudging (see k_mc_step)
                const size_t e = (size_t)cold->gage_of_pos[su] * (size_t)cold->nsteps + (size_t)(t - 1);
                const float *const da_a = cold->da_a;
                const uint8_t mode = cold->da_mode[e];
                float nudge = 0.0f;
                if (mode == 1) {
                    nudge = da_a[e] - q_new;
                    q_new = da_a[e];
                } else if (mode == 2) {
                    nudge = (da_a[e] - q_new) * cold->da_w[e];
                    q_new = q_new + nudge;
                }
                cold->da_nudge[e] = nudge;
            }
        }
        // the flow goes out as soon as it exists; the velocity (a power, a square root, a division no other row waits for)
        // is formed after the granule is on its way
        {
            const unsigned long long g = ((unsigned long long)(tag_p + 1u) << 32) | (unsigned long long)__float_as_uint(q_new);
            __hip_atomic_store(s_ring + (size_t)(t & (kLeanRing - 1)) * kFlowBlock + tix, g, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(a.gran + (size_t)t * np + su, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (routed) v_new = trmc::step_velocity<float, M>(p, c, d_new, m);
        q_prev = q_new;
        d_prev = d_new;
        if (t == t_hi) // hand the depth over to the next launch of the window
            __hip_atomic_store(cold->d_gran + su, ((unsigned long long)(tag_p + 1u) << 32) | (unsigned long long)__float_as_uint(d_new),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        {
            float *o = a.out + (size_t)(out_idx + (uint32_t)(t - 1) * 3u);
            o[0] = q_new;
            o[1] = v_new;
            o[2] = d_new;
        }
    }
    if (cold->dbg) atomicMax(cold->dbg + 2 * s_blk + 1, (unsigned long long)wall_clock64());
    cold->d_state[su] = d_prev;
    if (t_hi == cold->nsteps) cold->it_prev[su] = (uint8_t)(its >> 24);
    if (uint16_t *const it_sum = cold->it_sum) it_sum[su] = (uint16_t)min(65535u, (uint32_t)it_sum[su] + (its & 0x00ffffffu));
}

// which compute units exist: every workgroup marks the key of the unit it runs on
__global__ void __launch_bounds__(64) k_cu_probe(uint8_t *seen)
{
    if (threadIdx.x == 0) seen[hw_cu_key()] = 1;
}
// initial state of the dataflow engine: granule row 0 <- qu0 (mc_reach.pyx:361), depth column <- h0
__global__ void __launch_bounds__(kBlock)
k_flow_init(const float *__restrict__ q0, const int32_t *__restrict__ row_of_pos, unsigned long long *gran, float *d_state,
            unsigned long long *d_gran, int32_t nseg, uint32_t tag_base)
{
    const int32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= nseg) return;
    const size_t r = (size_t)row_of_pos[p] * 3;
    gran[p] = ((unsigned long long)tag_base << 32) | (unsigned long long)__float_as_uint(q0[r + 0]);
    d_state[p] = q0[r + 2];
    d_gran[p] = ((unsigned long long)tag_base << 32) | (unsigned long long)__float_as_uint(q0[r + 2]);
}
// boundary rows of the dataflow engine: hydrographs -> granules of the steps (t_begin, t_end] and the rows' result.
// src[b * stride_b + (t - 1 - t_begin) * stride_t + c]: bfvd[b][t-1][c] (stride_t = 3, ncomp = 3) or a flow block
// [b][t - 1 - t_begin] (stride_t = 1, ncomp = 1: velocity and depth of a boundary row are not inputs of anything, 0)
__global__ void __launch_bounds__(kBlock)
k_flow_boundary(const float *__restrict__ src, unsigned long long *gran, float *__restrict__ out,
                const int32_t *__restrict__ row_of_pos, int32_t nboundary, int32_t nsteps, int64_t nseg_pad, int32_t t_begin,
                int32_t t_end, int64_t stride_b, int32_t stride_t, int32_t ncomp, uint32_t tag_base,
                const int64_t *__restrict__ src_index)
{
    const int32_t w = t_end - t_begin;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= (int64_t)nboundary * w) return;
    const int32_t b = (int32_t)(i / w), k = (int32_t)(i % w), t = t_begin + 1 + k;
    const float *v = src + (size_t)(src_index ? src_index[b] : b) * stride_b + (size_t)k * stride_t;
    const float q = v[0];
    gran[(size_t)t * nseg_pad + b] = ((unsigned long long)(tag_base + (uint32_t)t) << 32) | (unsigned long long)__float_as_uint(q);
    float *o = out + ((size_t)row_of_pos[b] * nsteps + (t - 1)) * 3;
    o[0] = q;
    o[1] = ncomp > 1 ? v[1] : 0.0f;
    o[2] = ncomp > 2 ? v[2] : 0.0f;
}

// trmc_plan_set_stamps: the device's constant-rate clock (100 MHz) at four points of a window, written straight into
// page-locked host memory -- a timeline of consecutive windows of several plans without a profiler attached
__global__ void k_stamp(unsigned long long *slot) { *slot = wall_clock64(); }

// Every `stride`-th step of the result, out[row][t][3] -> dec[row][k][3] with t = stride (k + 1) - 1 (0-based): what the
// reference's writers consume of a window (nwm_routing/output.py:209-216 and :232-240 keep the steps whose END falls on a
// multiple of dt * qts_subdivisions).  One thread per (row, kept step): a 12-byte triple read at a stride of 12 * stride
// bytes, written densely.
template <class T>
__global__ void __launch_bounds__(kBlock)
k_decimate(const T *__restrict__ out, T *__restrict__ dec, int64_t nseg, int32_t nsteps, int32_t stride, int32_t nkeep)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= nseg * nkeep) return;
    const int64_t row = i / nkeep;
    const int32_t k = (int32_t)(i - row * nkeep);
    const T *src = out + ((size_t)row * (size_t)nsteps + (size_t)(stride * (k + 1) - 1)) * 3;
    T *dst = dec + (size_t)i * 3;
    dst[0] = src[0];
    dst[1] = src[1];
    dst[2] = src[2];
}

// The rows a window's tiles did NOT decimate as they went (k_mc_tile<.., DEC>): their kept steps from the time-major planes --
// coalesced by position -- into dec[row][k][3]; positions [skip_lo, skip_hi) are the tiles'.
template <class T>
__global__ void __launch_bounds__(kBlock)
k_decimate_planes(const T *__restrict__ q_tm, const T *__restrict__ v_tm, const T *__restrict__ d_tm, const int32_t *__restrict__ row_of_pos,
                  T *__restrict__ dec, int32_t nseg, int64_t nseg_pad, int32_t stride, int32_t nkeep, int32_t skip_lo, int32_t skip_hi)
{
    const int32_t p = (int32_t)blockIdx.x * kBlock + (int32_t)threadIdx.x;
    if (p >= nseg || (p >= skip_lo && p < skip_hi)) return;
    T *dst = dec + (size_t)row_of_pos[p] * (size_t)nkeep * 3;
    size_t src = (size_t)stride * (size_t)nseg_pad + (size_t)p;
    for (int32_t k = 0; k < nkeep; ++k, src += (size_t)stride * (size_t)nseg_pad, dst += 3) {
        dst[0] = q_tm[src];
        dst[1] = v_tm[src];
        dst[2] = d_tm[src];
    }
}

// ---------------------------------------------------------------- plan
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    bool borrowed = false; // the memory belongs to another plan (trmc_plan_clone): never freed, never regrown here
    void borrow(const DevBuf &o)
    {
        release();
        p = o.p;
        bytes = o.bytes;
        borrowed = true;
    }
    int ensure(size_t need, bool zero_new = false)
    {
        if (need <= bytes) return 0;
        if (borrowed) return fail(TRMC_ESTATE, "a buffer shared with the plan this one was cloned from would have to grow");
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        hipError_t e = hipMalloc(&p, need ? need : 1);
        if (e != hipSuccess) return fail(TRMC_ENOMEM, std::string("hipMalloc(") + std::to_string(need) + "): " + hipGetErrorString(e));
        bytes = need;
        if (zero_new && need) { // (granule planes: recycled memory must not hold a tag that could pass for a live one)
            // (hipMemset of device memory may return before it has run, on the null stream -- which the plans'
            // non-blocking streams do not wait for: without the synchronisation the fill can land AFTER the first
            // kernels of a window have written into the buffer; seen with two processes sharing one GPU)
            e = hipMemset(p, 0, need);
            if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
            if (e != hipSuccess) return fail(TRMC_EHIP, std::string("hipMemset: ") + hipGetErrorString(e));
        }
        return 0;
    }
    void release()
    {
        if (p && !borrowed) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        borrowed = false;
    }
};

} // namespace

struct RouteRun { // the routing window in progress (route_begin_t .. route_end_t)
    bool active = false;
    int32_t nsteps = 0, qts = 1, short_ts = 0;
    int32_t t_done = 0;           // launches ("diagonals") 1..t_done are queued: rows without lag are at step t_done,
                                  // lagged rows at step t_done - maxlag; the window ends at nsteps + maxlag
    int32_t boundary_through = 0; // boundary rows hold their hydrographs for steps 1..boundary_through
    int32_t tiles_done = 0, launches = 0;
    // wide levels routed K steps per launch with a skew of K steps per level (k_mc_tile); 0 = every level one step per launch
    int32_t wide = 0, wide_k = 0, wide_next = 0, wide_through = 0; // levels; K; tiles queued; last tile the tail waits for
    // second tier: the `mid` levels right below the wide ones, mid_k steps per launch under their own skew (k_mc_tile again),
    // queued on the plan's stream between the tail's launches
    int32_t mid = 0, mid_k = 0, mid_next = 0;
    // cluster tiles (k_mc_ctile): the rows below the wide levels K steps per launch too, cluster level c another tile behind;
    // `wide` is then the plan's cl_from_level (possibly 0: no slices at all), there is no one-step tail and no transposing pass
    bool cl = false;
    int32_t cl_next = 0;          // index of the next cluster tile to queue (they begin at tile `wide`)
    bool tail_active = false;     // the tail launches of this window go to the tail stream
    bool end_queued = false;      // route_end_queue has run for this window
    int32_t dec_stride = 0, dec_keep = 0; // the tiles of this window write the kept steps of their rows into the plan's `dec`
    int32_t dec_lo = 0, dec_hi = 0;       // ... the plan positions [dec_lo, dec_hi) that do
};

struct StreamRun;
struct trmc_plan {
    int device = 0;
    int precision = 32;
    size_t esz = 4;
    trmc::Topology topo;
    int64_t nseg = 0, nseg_pad = 0, nrouted = 0;
    bool dt_uniform = true;
    double dt = 0.0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t stream2 = nullptr;       // result transpose, overlapped with the step launches
    hipStream_t wstream = nullptr;       // the wide tiles (k_mc_tile), ordinary priority: the narrow tail of the level order runs one
                                         // step per launch on the plan's own high-priority stream BESIDE them -- the tail's 288
                                         // dependent launches are the latency-critical part, the tiles fill whatever they leave
    hipEvent_t ev_tail = nullptr;        // "every tile queued so far is complete" (tile stream -> plan stream)
    std::vector<hipEvent_t> wide_t0, wide_t1; // timing events around the wide launches of the window (trmc_stats.ms_wide)
    std::vector<hipEvent_t> tile_ev;     // "time tile b is complete" (main stream -> stream2)
    hipEvent_t ev_emit = nullptr;        // "all tiles emitted" (stream2 -> main stream)
    // static, plan order
    DevBuf params; // 9 columns x nseg_pad
    DevBuf up_ptr, up_idx, up2, level, row_of_pos, pos_of_row, it_prev, lag;
    DevBuf lagk, cblk_ptr;               // cluster order (topology.hpp): tiles every position runs behind level 0; the cluster blocks
    DevBuf it_sum;                       // per-position cost of the window (trmc_plan_collect_cost)
    bool collect_cost = false;
    bool hinted = false;                 // created with a cost hint: rows of a level are grouped by cost
    bool params_sane = false;            // see DevMathF::fast_ok
    int32_t cost_nsteps = -1;            // nsteps of the window it_sum was collected over
    int32_t maxlag = 0;                  // trmc_plan_set_lag: rows routed `maxlag` launches behind the others
    int64_t wide_safe_pos = -1;          // first plan position that is lagged or fed by a boundary row (-1: not looked for yet)
    // trmc_plan_chain_from: the next window's initial state has been set on the device from another plan's window
    bool chain_staged = false;
    hipEvent_t ev_chain[4] = {nullptr, nullptr, nullptr, nullptr}; // source's tiles / tail done; this plan's two copies done
    // as the SOURCE of a hand-over: "the receiver has read my planes" (its two copies), waited for at my next trmc_route_begin
    hipEvent_t ev_released[2] = {nullptr, nullptr};
    bool released_pending[2] = {false, false};
    std::vector<int32_t> lag_of_row;
    DevBuf gage_of_pos, da_mode, da_a, da_w, da_nudge; // nudging tables of the staged window
    DevBuf res_of_pos, res_par, res_inflow;             // level-pool reservoirs of the plan
    int64_t nres = 0;
    double res_dt = 0.0;
    int64_t ngage = 0;
    int64_t nraw = 0;                    // gages inside a reach whose successor reads the un-nudged flow (general mode)
    DevBuf raw_of_pos, da_raw;
    int32_t da_nsteps = -1;
    // per window
    DevBuf in_qlat, in_q0, in_bfvd, qlat_tm, tm, out, scratch, gathered;
    size_t gathered_bytes = 0;
    int64_t nq = 0;
    bool qlat_direct = false;   // qlat_tm was filled by trmc_upload_forcing_packed: no transpose at route time
    DevBuf qlat_alt;            // sequence mode: the STAGED forcing already transposed (trmc_stage_forcing, behind its copy) into
    bool qlat_alt_ready = false; // the buffer the window in progress does not read; the next window's set-up swaps the two
    bool have_boundary = true;  // boundary hydrographs present for the staged window
    int32_t staged_nsteps = -1; // nsteps the staged forcing was uploaded for
    int32_t routed_nsteps = -1; // nsteps of the last completed route
    // dataflow engine (k_mc_flow): the plan is in block order, `tm` holds the granule plane
    bool flow = false;
    uint32_t tag_base = 1;               // tag of step 0 of the current window (0 is never a live tag)
    int32_t tag_span = 0;                // tags the current window may use (nsteps + 1)
    DevBuf prio;                         // issue priority per wavefront
    DevBuf d_gran;                       // depth hand-over granules (k_mc_flow_lean)
    hipStream_t fstream = nullptr;       // second compute stream: consecutive time chunks of a resident window overlap
    hipEvent_t ev_chunk[2] = {nullptr, nullptr}; // the last launch queued on {stream, fstream}
    hipEvent_t ev_ctl = nullptr;         // ordering of the two compute streams against each other
    int flow_next = 0;                   // which of the two the next trmc_route_advance uses (only toggles in overlap mode)
    int flow_last = 0;                   // ... and which one the last launch went to
    DevBuf ticket_map;                   // [nblocks] ticket -> block of a general-mode plan with long main stems (topo.early_blocks)
    DevBuf d_state, ticket, rank, dbg;   // depth column; {block ticket, abort flag}; level rank of a position inside its block
    DevBuf cuq_ptr, cuq_blk, cuq_head, cu_index, cuq_perm; // blocks dealt to compute units by cost (flow_place_blocks); heads: one set per compute stream
    int32_t ncuq = 0;                    // number of queues (= compute units found), 0 = block tickets
    uint64_t watchdog_ticks = 3000000000ull; // 30 s of wall_clock64 (100 MHz): long enough for a device that is shared or profiled (TRMC_FLOW_WATCHDOG_MS)
    // trmc_plan_options, resolved at creation (a clone copies them)
    struct Opt {
        bool tol = false;                    // TRMC_ARITH_TOLERANCE
        int64_t wide_min_rows = 0;           // <= 0: no wide tier
        int32_t wide_levels = 16, wide_k = 16;
        int64_t mid_min_rows = 0;            // <= 0: no second tier
        int32_t mid_levels = 12, mid_k = 4;
        int32_t tile_perm_group = -1;        // -1: the default (see route_advance_t); 0: off; 1: on
        int32_t hot_rows = -1;               // -1 (default) or 1: with the partition; 0: off
        int32_t cluster_rows = 0;            // rows per cluster block of a short-timestep plan's deeper rows; 0: no cluster order
        int32_t cluster_late_lag = 0;        // tiles the rows fed by boundary rows run behind at least (cluster order)
        int32_t stream_split = 0;            // a stream's slices from this level on ride on the clusters' stream (0: all on the tile stream)
        int32_t hot_wave_rows = 0;           // rows of the hot list per wavefront (0: by the plan's size)
        bool sequence = false;
        bool flow_overlap = false;
        int32_t flow_lean = 0;
        bool flow_debug = false;
    } opt;
    trmc_stats stats{};
    RouteRun run;
    // asynchronous fetch of what a throughput-mode caller consumes (trmc_fetch_begin / trmc_fetch_wait)
    // trmc_plan_set_output_stride: windows write every out_stride-th step of every row's (q, v, d) into `dec` as they go (the
    // tiled rows from k_mc_tile; the others are gathered from the time-major planes when the block is fetched)
    int32_t out_stride = 0;
    DevBuf dec;
    int32_t dec_stride_done = 0, dec_keep_done = 0, dec_nsteps_done = 0, dec_lo_done = 0, dec_hi_done = 0; // ... what the last window left there
    unsigned long long *stamps = nullptr; // trmc_plan_set_stamps: [nstamp_windows][4] in page-locked host memory (the caller's)
    int32_t nstamp_windows = 0;
    int64_t stamp_seq = -1;               // windows begun since the ring was set, minus one
    DevBuf fetch_hyd, fetch_q0, fetch_fvd;
    hipEvent_t ev_dec = nullptr;         // "the copy stream has read `out`" (a fetch of the decimated result): the next window's
    bool dec_pending = false;            // set-up goes behind it
    hipStream_t cstream = nullptr;       // copy stream: D2H of window k runs beside the kernels of window k + 1
    hipStream_t hstream = nullptr;       // ... and the one of the other direction: a staged forcing on its way to the device
    hipEvent_t ev_fetch_ready = nullptr, ev_fetch_done = nullptr;
    bool fetch_pending = false;
    // "the last gather queued on the plan's stream after a window has read the planes" -- what a set-up queued on ANOTHER stream
    // (TRMC_SETUP_ASIDE) waits for before it lets a new window's tiles overwrite them
    hipEvent_t ev_gather = nullptr;
    bool gather_pending = false;
    // trmc_plan_clone: a second set of WINDOW buffers on the static data (topology, parameters) of `parent`
    trmc_plan *parent = nullptr;
    int32_t clones = 0;                  // live clones of this plan
    bool zombie = false;                 // destroyed while clones were alive: freed with the last of them
    // trmc_stage_forcing: the next window's forcing is on its way to in_qlat on the copy stream
    hipEvent_t ev_forcing = nullptr;
    bool forcing_pending = false;
    bool state_missing = false;          // ... and there is no initial state yet: trmc_plan_chain_from must supply it
    bool q0_staged = false;              // in_q0 holds the initial state of the window that is staged (an upload's q0, or the last
                                         // window's final state gathered by an upload with q0 = NULL / trmc_stage_forcing): valid
                                         // until a window consumes it, whatever routed_nsteps says in the meantime
    struct StreamRun *seq = nullptr;     // trmc_stream_*: a stream of windows on a ring of day slots (stream.inc)
    DevBuf hot_list, hot_cnt;            // k_mc_tile's hot rows (StepArgs::hot_list): [3][hot_cap] positions, [3] lengths
    int32_t hot_cap = 0;
    int64_t tile_seq = 0;                // tile launches of the wide tier so far, all windows: which of the three lists is read
    DevBuf cls_last;                     // the cost class every wide row showed at the end of its last tile (k_mc_tile's in-block partition)
    std::vector<DevBuf> rowsets;        // positions of registered row sets (trmc_rowset_create)
    std::vector<int64_t> rowset_n;
    std::vector<int32_t> rowset_lag;
    std::vector<int32_t> rowset_lagk;   // cluster order: the largest tile lag among the set's rows (trmc_stream_gather)
};

namespace {

template <class T> T *col(trmc_plan *pl, int c) { return (T *)pl->params.p + (size_t)c * pl->nseg_pad; }

template <class T> int upload_params(trmc_plan *pl, const float *params)
{
    const int64_t n = pl->nseg, np = pl->nseg_pad;
    std::vector<T> host((size_t)TRMC_NPARAM * np, T(0));
    const size_t all_cols = (size_t)kTotalCols * np;
    // a benign channel for the padding lanes (never routed, never read back)
    bool sane = true;
    auto in_range = [](float v) { return v >= 0x1p-14f && v <= 0x1p17f; };
    auto within = [](float v, float lo, float hi) { return v >= lo && v <= hi; };
    for (int64_t p = 0; p < n; ++p) {
        const float *src = params + (size_t)pl->topo.row_of_pos[p] * TRMC_NPARAM;
        for (int c = 0; c < TRMC_NPARAM; ++c) host[(size_t)c * np + p] = (T)src[c];
        const float cs = src[TRMC_P_CS];
        // (the operands of those divisions are made of bw, the side slope, n, ncc, twcc and the depth only)
        sane = sane && in_range(src[TRMC_P_BW]) && in_range(src[TRMC_P_N]) && (cs == 0.0f || in_range(cs))
               && (src[TRMC_P_TWCC] == 0.0f || in_range(src[TRMC_P_TWCC])) && (src[TRMC_P_NCC] == 0.0f || in_range(src[TRMC_P_NCC]))
               // (... and the Muskingum K of an in-bank point of dt, dx, s0 too: DevMathF::k_of)
               && within(src[TRMC_P_DT], 0x1p-20f, 0x1p40f) && within(src[TRMC_P_DX], 0x1p-10f, 0x1p19f)
               && within(src[TRMC_P_S0], 0x1p-30f, 0x1p10f);
    }
    pl->params_sane = sane && sizeof(T) == 4;
    if (int rc = pl->params.ensure(all_cols * sizeof(T))) return rc;
    HIP_TRY(hipMemcpy(pl->params.p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    if (n > 0) {
        hipLaunchKernelGGL((k_make_const<T>), dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, pl->stream,
                           (T *)pl->params.p, (int32_t)n, np);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(pl->stream));
    }
    return 0;
}

int upload_i32(DevBuf &b, const std::vector<int32_t> &v, size_t min_elems)
{
    const size_t n = v.size() > min_elems ? v.size() : min_elems;
    if (int rc = b.ensure(n * sizeof(int32_t))) return rc;
    if (!v.empty()) HIP_TRY(hipMemcpy(b.p, v.data(), v.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    return 0;
}

int use_device(const trmc_plan *pl)
{
    HIP_TRY(hipSetDevice(pl->device));
    return 0;
}

// a kernel that reads the result planes of the last window has just been queued on the plan's stream (outside a window)
int note_gather(trmc_plan *pl, bool also_in_window = false)
{
    if (pl->run.active && !also_in_window) return 0;
    if (!pl->ev_gather) HIP_TRY(hipEventCreateWithFlags(&pl->ev_gather, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(pl->ev_gather, pl->stream));
    pl->gather_pending = true;
    return 0;
}

// diagnosis: slot `which` (0 tiles begin, 1 tiles end, 2 tail begins, 3 window ends) of the current window's stamps
inline void stamp(trmc_plan *pl, hipStream_t st, int which)
{
    if (!pl->stamps || pl->stamp_seq < 0) return;
    hipLaunchKernelGGL(k_stamp, dim3(1), dim3(1), 0, st, pl->stamps + (size_t)(pl->stamp_seq % pl->nstamp_windows) * 4 + which);
}

template <class T> StepArgs<T> step_args(trmc_plan *pl, int nsteps, int qts)
{
    StepArgs<T> a;
    a.dt_col = pl->dt_uniform ? nullptr : col<T>(pl, TRMC_P_DT);
    a.dt = (T)pl->dt;
    a.dx = col<T>(pl, TRMC_P_DX);
    a.bw = col<T>(pl, TRMC_P_BW);
    a.twcc = col<T>(pl, TRMC_P_TWCC);
    a.n = col<T>(pl, TRMC_P_N);
    a.ncc = col<T>(pl, TRMC_P_NCC);
    a.s0 = col<T>(pl, TRMC_P_S0);
    a.z = col<T>(pl, TRMC_NPARAM + 0);
    a.bfd = col<T>(pl, TRMC_NPARAM + 1);
    a.sqrt_s0 = col<T>(pl, TRMC_NPARAM + 2);
    a.sq1pz2 = col<T>(pl, TRMC_NPARAM + 3);
    a.s0_n = col<T>(pl, TRMC_NPARAM + 4);
    a.s0_ncc = col<T>(pl, TRMC_NPARAM + 5);
    a.inv_n = col<T>(pl, TRMC_NPARAM + 6);
    a.dec = nullptr; // (set for the tile launches of a window that decimates as it goes: route_advance_t)
    a.dec_stride = a.dec_keep = 0;
    a.hot_list = a.hot_cnt = nullptr;
    a.hot_cap = a.hot_cur = a.hot_home = 0;
    a.hot_wave_rows = 64;
    a.seq_slots = a.seq_tpd = a.seq_day = a.seq_days = a.seq_day_min = 0;
    a.slot_tm = a.slot_qlat = a.slot_out = a.slot_dec = 0;
    a.up_ptr = (const int32_t *)pl->up_ptr.p;
    a.up_idx = (const int32_t *)pl->up_idx.p;
    a.up2 = (const int2 *)pl->up2.p;
    a.level = (const int32_t *)pl->level.p;
    a.lag = pl->maxlag > 0 ? (const int32_t *)pl->lag.p : nullptr;
    a.it_prev = (uint8_t *)pl->it_prev.p;
    a.it_sum = pl->collect_cost ? (uint16_t *)pl->it_sum.p : nullptr;
    a.sane = pl->params_sane;
    a.res_of_pos = pl->nres > 0 ? (const int32_t *)pl->res_of_pos.p : nullptr;
    a.res_par = (const T *)pl->res_par.p;
    a.res_inflow = (T *)pl->res_inflow.p;
    a.res_dt = (T)pl->res_dt;
    const bool da = pl->ngage > 0;
    a.gage_of_pos = da ? (const int32_t *)pl->gage_of_pos.p : nullptr;
    a.da_mode = (const uint8_t *)pl->da_mode.p;
    a.da_a = (const T *)pl->da_a.p;
    a.da_w = (const T *)pl->da_w.p;
    a.da_nudge = (T *)pl->da_nudge.p;
    a.raw_of_pos = da && pl->nraw > 0 ? (const int32_t *)pl->raw_of_pos.p : nullptr;
    a.da_raw = da && pl->nraw > 0 ? (T *)pl->da_raw.p : nullptr;
    a.qlat_tm = (const T *)pl->qlat_tm.p;
    const size_t plane = (size_t)(nsteps + 1) * pl->nseg_pad;
    a.q_tm = (T *)pl->tm.p;
    a.v_tm = a.q_tm + plane;
    a.d_tm = a.v_tm + plane;
    a.nseg_pad = pl->nseg_pad;
    a.nsteps = nsteps;
    a.qts = qts;
    a.out = (T *)pl->out.p;
    a.row_of_pos = (const int32_t *)pl->row_of_pos.p;
    a.out_vec = sizeof(T) == 4 && nsteps % 4 == 0 && pl->run.wide_k % 4 == 0;
    a.cls_last = nullptr; // (route_advance_t switches the in-block partition on for its wide tiles)
    return a;
}

inline unsigned blocks_for(int64_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }
// rows of the hot list per wavefront of the blocks that route it (trmc_plan_options.hot_wave_rows)
inline int32_t hot_wave_rows_of(const trmc_plan *pl) { return pl->opt.hot_wave_rows > 0 ? pl->opt.hot_wave_rows : (pl->nrouted >= 1000000 ? 64 : 16); }

template <class T, bool SHORT, bool TOL>
inline void launch_step_m(hipStream_t st, const StepArgs<T> &a, int32_t s0, int32_t s1, int32_t d)
{
    const int64_t n = (int64_t)s1 - s0;
    const dim3 grid((unsigned)((n + kStepBlock - 1) / kStepBlock)), block(kStepBlock);
    if (SHORT && a.lag)
        hipLaunchKernelGGL((k_mc_step<T, SHORT, SHORT, TOL>), grid, block, 0, st, a, s0, s1, d, (d - 1) / a.qts);
    else
        hipLaunchKernelGGL((k_mc_step<T, SHORT, false, TOL>), grid, block, 0, st, a, s0, s1, d, (d - 1) / a.qts);
}
// (tol: the plan's arithmetic is TRMC_ARITH_TOLERANCE -- precision-32 plans only, trmc_plan_create_opt sees to that)
template <class T, bool SHORT>
inline void launch_step(hipStream_t st, const StepArgs<T> &a, int32_t s0, int32_t s1, int32_t d, bool tol)
{
    if constexpr (sizeof(T) == 4) {
        if (tol) return launch_step_m<T, SHORT, true>(st, a, s0, s1, d);
    }
    launch_step_m<T, SHORT, false>(st, a, s0, s1, d);
}
// one launch of k_mc_tile: positions [p0, p1), `tile` = launch index + the first level of the tier, K steps
template <class T>
inline void launch_tile(hipStream_t st, const StepArgs<T> &a, int32_t p0, int32_t p1, int32_t tile, int32_t K, bool tol)
{
    // (with hot rows: the first a.hot_home blocks take the list, the blocks behind them positions)
    const unsigned home = (unsigned)((p1 - p0 + kTileBlock - 1) / kTileBlock);
    const dim3 grid(home + (a.hot_list ? (unsigned)a.hot_home : 0u)), block(kTileBlock);
    const bool dec = a.dec != nullptr;
    if constexpr (sizeof(T) == 4) {
        if (tol) {
            if (dec) hipLaunchKernelGGL((k_mc_tile<T, true, true>), grid, block, 0, st, a, p0, p1, tile, K);
            else hipLaunchKernelGGL((k_mc_tile<T, true, false>), grid, block, 0, st, a, p0, p1, tile, K);
            return;
        }
    }
    if (dec) hipLaunchKernelGGL((k_mc_tile<T, false, true>), grid, block, 0, st, a, p0, p1, tile, K);
    else hipLaunchKernelGGL((k_mc_tile<T, false, false>), grid, block, 0, st, a, p0, p1, tile, K);
}

// one launch of k_mc_ctile: the cluster blocks [b0, b1) of the plan at tile index `tile`
template <class T>
inline void launch_ctile(hipStream_t st, const StepArgs<T> &a, const int32_t *cblk_ptr, int32_t b0, int32_t b1, int32_t tile, int32_t K,
                         bool tol)
{
    const dim3 grid((unsigned)(b1 - b0)), block(kTileBlock);
    const bool dec = a.dec != nullptr;
    if constexpr (sizeof(T) == 4) {
        if (tol) {
            if (dec) hipLaunchKernelGGL((k_mc_ctile<T, true, true>), grid, block, 0, st, a, cblk_ptr, b0, tile, K);
            else hipLaunchKernelGGL((k_mc_ctile<T, true, false>), grid, block, 0, st, a, cblk_ptr, b0, tile, K);
            return;
        }
    }
    if (dec) hipLaunchKernelGGL((k_mc_ctile<T, false, true>), grid, block, 0, st, a, cblk_ptr, b0, tile, K);
    else hipLaunchKernelGGL((k_mc_ctile<T, false, false>), grid, block, 0, st, a, cblk_ptr, b0, tile, K);
}

// A routing window runs in three parts so that a caller can interleave other device work (the multi-GPU
// hand-off of cut-edge hydrographs, distributed.py) with it, everything asynchronous on the plan's stream:
//   route_begin_t    forcing transpose, initial state, boundary rows (if already staged)
//   route_advance_t  the step launches for the timesteps (t_done, t_end]
//   route_end_t      the rest of the result transpose, completion, timing
template <class T> int emit_tiles_through(trmc_plan *pl, int32_t t_complete) // all steps <= t_complete are queued
{
    // The result transpose (memory-bound) runs on a second stream, one time tile at a time, as soon as the
    // launches that complete the tile have been queued: it overlaps with the VALU-bound step kernels
    // instead of trailing them.
    constexpr int32_t kTile = TRMC_EMIT_TILE;
    RouteRun &r = pl->run;
    const int32_t n = (int32_t)pl->nseg, nsteps = r.nsteps;
    const int32_t ntiles = (nsteps + kTile - 1) / kTile;
    const size_t plane = (size_t)(nsteps + 1) * pl->nseg_pad;
    const T *q_tm = (const T *)pl->tm.p;
    while (r.tiles_done < ntiles && ((r.tiles_done + 1) * kTile <= t_complete || t_complete >= nsteps)) {
        // (with wide tiles: the tail, on the plan's stream, trails them -- its progress is everybody's; without a tail the
        // tile stream's is)
        HIP_TRY(hipEventRecord(pl->tile_ev[r.tiles_done], (r.wide > 0 && !r.tail_active) ? pl->wstream : pl->stream));
        HIP_TRY(hipStreamWaitEvent(pl->stream2, pl->tile_ev[r.tiles_done], 0));
        // (rows of the wide levels wrote their results themselves, k_mc_tile: their positions are left out)
        // (with cluster tiles every routed row has: only boundary rows are left to this pass)
        const int32_t skip_lo = (r.wide > 0 || r.cl) ? pl->topo.lvl_ptr[0] : 0;
        const int32_t skip_hi = r.cl ? pl->topo.lvl_ptr[pl->topo.nlevels] : (r.wide > 0 ? pl->topo.lvl_ptr[r.wide + r.mid] : 0);
        const int32_t shift_from = (skip_lo + 63) / 64 * 64, shift = std::max(0, (skip_hi - shift_from) / 64 * 64);
        const int32_t n_emit = n - shift;
        if (n_emit > 0) {
            const int32_t tb = r.tiles_done * kTile, te = min(nsteps, tb + kTile);
            hipLaunchKernelGGL((k_emit<T>), dim3((n_emit + 63) / 64, (unsigned)((te - tb + kEmitSteps - 1) / kEmitSteps)),
                               dim3(kBlock), 0, pl->stream2, q_tm, q_tm + plane, q_tm + 2 * plane,
                               (const int32_t *)pl->row_of_pos.p, (T *)pl->out.p, n, pl->nseg_pad, nsteps, tb, te, shift_from, shift,
                               skip_lo, skip_hi);
        }
        ++r.tiles_done;
    }
    return 0;
}

template <class T> int route_begin_t(trmc_plan *pl, int nsteps, int qts, int short_ts)
{
    const trmc::Topology &tp = pl->topo;
    const int32_t n = (int32_t)pl->nseg;
    const int64_t np = pl->nseg_pad;
    hipStream_t st = pl->stream;
    const size_t plane = (size_t)(nsteps + 1) * np;
    if (int rc = pl->tm.ensure(3 * plane * sizeof(T))) return rc;
    const bool qlat_early = pl->qlat_alt_ready; // (the staged forcing is in plan order already: the other buffer becomes this window's)
    if (qlat_early) std::swap(pl->qlat_tm, pl->qlat_alt);
    pl->qlat_alt_ready = false;
    if (int rc = pl->qlat_tm.ensure((size_t)pl->nq * np * sizeof(T))) return rc;
    if (int rc = pl->out.ensure((size_t)pl->nseg * nsteps * 3 * sizeof(T))) return rc;
    if (pl->nres > 0)
        if (int rc = pl->res_inflow.ensure((size_t)pl->nres * nsteps * sizeof(T))) return rc;
    if (pl->collect_cost) {
        if (int rc = pl->it_sum.ensure((size_t)np * sizeof(uint16_t))) return rc;
        pl->cost_nsteps = nsteps;
    }
    StepArgs<T> a = step_args<T>(pl, nsteps, qts);
    const int32_t *row_of_pos = (const int32_t *)pl->row_of_pos.p;
    constexpr int32_t kTile = TRMC_EMIT_TILE;
    const int32_t ntiles = (nsteps + kTile - 1) / kTile;
    while ((int32_t)pl->tile_ev.size() < ntiles) {
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        pl->tile_ev.push_back(e);
    }

    for (int i = 0; i < 2; ++i) // a receiver of this plan's last window (trmc_plan_chain_from) has read what this window overwrites
        if (pl->released_pending[i]) {
            HIP_TRY(hipStreamWaitEvent(st, pl->ev_released[i], 0));
            if (pl->wstream) HIP_TRY(hipStreamWaitEvent(pl->wstream, pl->ev_released[i], 0));
            pl->released_pending[i] = false;
        }
    HIP_TRY(hipEventRecord(pl->ev[0], st));
    // Sequence mode (trmc_plan_options.sequence_mode): the window's set-up (forcing transpose, initial state, boundary rows)
    // goes to the TILE stream instead of the plan's own.  For one plan it is all the same; for two plans that take turns on a device (two ensemble
    // members) it is what lets the tiles of one member's next window start behind the other
    // member's tiles while that member's tail is still running: with one hardware queue per stream priority the two plans'
    // high-priority streams share a queue, in order of submission, and a set-up queued there would sit behind the other
    // member's 288 tail launches -- and the tiles behind the set-up.
    const bool setup_aside = pl->opt.sequence && short_ts && pl->wstream != nullptr;
    hipStream_t const plan_st = st;
    if (setup_aside) {
        st = pl->wstream;
        // the gathers queued on the plan's own stream since the last window (trmc_fetch_begin, trmc_gather_flow_rows, the
        // final state) read planes this window's tiles overwrite: the set-up goes behind the last of them (an event
        // recorded when that gather was queued -- not ev[0] above, which sits behind whatever another plan has put into
        // the shared high-priority queue since)
        if (pl->gather_pending) HIP_TRY(hipStreamWaitEvent(st, pl->ev_gather, 0));
    }
    pl->gather_pending = false;
    if (pl->dec_pending) { // (trmc_fetch_begin_fvd: the copy stream reads `out`, which this window's kernels overwrite)
        HIP_TRY(hipStreamWaitEvent(st, pl->ev_dec, 0));
        pl->dec_pending = false;
    }
    if (pl->forcing_pending) { // (trmc_stage_forcing: the copy into in_qlat runs on the copy stream)
        HIP_TRY(hipStreamWaitEvent(st, pl->ev_forcing, 0));
        pl->forcing_pending = false;
    }
    HIP_TRY(hipMemsetAsync(pl->it_prev.p, 0, (size_t)np, st)); // no history at the start of a window
    if (pl->collect_cost) HIP_TRY(hipMemsetAsync(pl->it_sum.p, 0, (size_t)np * sizeof(uint16_t), st));
    // every element the result reads is written below: time row 0 by k_init_state, rows 1..nsteps
    // of routed positions by k_mc_step and of boundary positions by k_fill_boundary (the padding
    // lanes of each row are never read), so the reference's zero fill (mc_reach.pyx:253) is moot
    if (n > 0) {
        if (!pl->qlat_direct && !qlat_early)
            hipLaunchKernelGGL((k_prep_qlat<T>), dim3((n + 63) / 64, (unsigned)((pl->nq + 31) / 32)), dim3(kBlock), 0, st,
                               (const T *)pl->in_qlat.p, row_of_pos, (T *)pl->qlat_tm.p, n, np, (int32_t)pl->nq);
        if (!pl->chain_staged) // (else: time row 0 was set on the device by trmc_plan_chain_from)
            hipLaunchKernelGGL((k_init_state<T>), dim3(blocks_for(n)), dim3(kBlock), 0, st, (const T *)pl->in_q0.p,
                               row_of_pos, a.q_tm, a.v_tm, a.d_tm, n);
    }
    pl->chain_staged = false;
    pl->q0_staged = false; // (consumed: the next staging gathers this window's final state)
    RouteRun &r = pl->run;
    r = RouteRun{};
    r.active = true;
    r.nsteps = nsteps;
    r.qts = qts;
    r.short_ts = short_ts ? 1 : 0;
    if (pl->stamps) ++pl->stamp_seq;
    r.boundary_through = tp.nboundary > 0 ? 0 : nsteps;
    if (tp.nboundary > 0 && pl->have_boundary) {
        hipLaunchKernelGGL((k_fill_boundary<T>), dim3(blocks_for(tp.nboundary * (int64_t)nsteps)), dim3(kBlock), 0, st,
                           (const T *)pl->in_bfvd.p, a.q_tm, a.v_tm, a.d_tm, (int32_t)tp.nboundary, nsteps, np);
        r.boundary_through = nsteps;
    }
    HIP_TRY(hipEventRecord(pl->ev[1], st));
    if (setup_aside) {
        st = plan_st;
        HIP_TRY(hipStreamWaitEvent(st, pl->ev[1], 0)); // the plan's stream continues behind the set-up
    }
    // Short-timestep windows of a wide network: the leading levels that can fill the device by themselves are routed K
    // steps per launch (k_mc_tile), the rest one step per launch behind them.  Needs every boundary hydrograph up front
    // (wide rows run ahead of the window's progress) and no lagged rows (the multi-GPU trunk has its own skew).
    // TRMC_WIDE_MIN_ROWS (rows a level must have, default 384 per compute unit; 0 switches the path off), TRMC_WIDE_LEVELS (at
    // most, default 16) and TRMC_WIDE_K (steps per launch, default 16) are measurement / test knobs.
    // Rows with a lag (the trunk of a cut basin riding in its owner's launches) and rows fed by boundary rows whose values
    // arrive chunk by chunk stay in the TAIL: the leading levels are only routed ahead if none of them is among their rows
    // (plans built for assume_short_ts keep such rows below level kWideMaxLevels: topology.hpp, boundary_floor).
    if (pl->wide_safe_pos < 0) { // first plan position that is lagged or reads a boundary row (host arrays; once per plan / lag)
        int64_t first = pl->nseg;
        for (int64_t p = tp.nboundary; p < pl->nseg && first == pl->nseg; ++p) {
            bool unsafe = pl->maxlag > 0 && !pl->lag_of_row.empty() && pl->lag_of_row[(size_t)tp.row_of_pos[p]] != 0;
            for (int32_t k = tp.up_ptr[p]; !unsafe && k < tp.up_ptr[p + 1]; ++k) unsafe = tp.up_idx[k] < tp.nboundary;
            if (unsafe) first = p;
        }
        pl->wide_safe_pos = first;
    }
    if (short_ts && pl->nrouted > 0) {
        // (measured on the CONUS day, MI355X, with every tile queued up front: levels of at least 32 768 rows -- eleven of them
        // -- and K = 12: 17.5 ms; eight levels 16.9; six 16.6; five 16.5 with K = 12 and 16.25 with K = 16; four 16.6; three 16.9;
        // K = 24: 17.0.  Fewer wide levels shorten the ramps of the level skew and give the tail launches more rows to fill the
        // device with after the last tile; the threshold that picks five levels there is 30 % of the rows the device holds at
        // five wavefronts per SIMD.  DESIGN.md lists what else was tried on this schedule.)
        const int64_t min_rows = pl->opt.wide_min_rows;
        int32_t W = 0, M = 0;
        const bool all_in_place = pl->maxlag == 0 && r.boundary_through == nsteps; // (then every level may run ahead)
        const int32_t level_cap = tp.ncl > 0 ? tp.cl_from_level : (tp.tail_from_level > 0 ? tp.tail_from_level : tp.nlevels); // (deeper rows are not in level slices)
        auto level_ok = [&](int32_t l, int64_t need) {
            return l < level_cap && tp.lvl_ptr[l + 1] - tp.lvl_ptr[l] >= need && (all_in_place || (int64_t)tp.lvl_ptr[l + 1] <= pl->wide_safe_pos);
        };
        if (min_rows > 0)
            while (W < std::min<int32_t>(pl->opt.wide_levels, kWideMaxLevels) && level_ok(W, min_rows)) ++W;
        // the second tier: the levels right below, fewer steps per launch (a skew of mid_k steps per level instead of wide_k)
        if (W > 0 && pl->opt.mid_min_rows > 0)
            while (M < pl->opt.mid_levels && level_ok(W + M, pl->opt.mid_min_rows)) ++M;
        // CLUSTER TILES (k_mc_ctile): a plan in cluster order (topology.hpp) routes its deeper rows K steps per launch too --
        // if, like the wide levels, they find every boundary hydrograph in place and carry no lag of their own.  The slices
        // are then exactly the plan's (the same rule picked them when the order was made).
        if (tp.ncl > 0 && all_in_place) {
            r.cl = true;
            W = tp.cl_from_level;
            M = 0;
        }
        if (W > 0 || r.cl) {
            r.wide = W;
            r.wide_k = std::max(1, std::min(nsteps, pl->opt.wide_k > 0 ? pl->opt.wide_k : std::max(1, std::min(16, nsteps / 8))));
            r.mid = M;
            r.mid_k = M > 0 ? std::max(1, std::min(r.wide_k, pl->opt.mid_k)) : 0;
            r.cl_next = W;
            if (!pl->wstream) {
                // ordinary priority: between the tail's step launches (high) and the result transpose (low); one hardware queue each
                HIP_TRY(hipStreamCreateWithFlags(&pl->wstream, hipStreamNonBlocking));
                HIP_TRY(hipEventCreateWithFlags(&pl->ev_tail, hipEventDisableTiming));
            }
            HIP_TRY(hipStreamWaitEvent(pl->wstream, pl->ev[1], 0)); // the tiles start behind the window's set-up
            const size_t ntile = (size_t)std::max(1, (nsteps + r.wide_k - 1) / r.wide_k + W - 1);
            while (pl->wide_t0.size() < ntile) { // (t0: only the first is used -- a tile starts where the one before it ended)
                hipEvent_t e0 = nullptr, e1 = nullptr;
                HIP_TRY(hipEventCreate(&e0));
                HIP_TRY(hipEventCreate(&e1));
                pl->wide_t0.push_back(e0);
                pl->wide_t1.push_back(e1);
            }
        }
    }
    HIP_TRY(hipGetLastError());
    pl->routed_nsteps = -1;
    return 0;
}

// what ends a window on the device -- the rest of the result transpose and the events the clock reads -- queued (once)
template <class T> int route_end_queue(trmc_plan *pl)
{
    RouteRun &r = pl->run;
    if (r.end_queued) return 0;
    hipStream_t st = pl->stream;
    stamp(pl, st, 3);
    HIP_TRY(hipEventRecord(pl->ev[2], st));
    if (int rc = emit_tiles_through<T>(pl, r.nsteps)) return rc; // whatever is left (at least the last tile)
    HIP_TRY(hipEventRecord(pl->ev_emit, pl->stream2));
    HIP_TRY(hipStreamWaitEvent(st, pl->ev_emit, 0));
    HIP_TRY(hipEventRecord(pl->ev[3], st));
    HIP_TRY(hipGetLastError());
    r.end_queued = true;
    return 0;
}

template <class T> int route_advance_t(trmc_plan *pl, int t_end)
{
    const trmc::Topology &tp = pl->topo;
    RouteRun &r = pl->run;
    constexpr int32_t kTile = TRMC_EMIT_TILE;
    const int32_t nsteps = r.nsteps, t0 = r.t_done;
    StepArgs<T> a = step_args<T>(pl, nsteps, r.qts);
    hipStream_t st = pl->stream;
    if (pl->nrouted > 0) {
        const int32_t L = tp.nlevels;
        if (r.short_ts && (r.wide > 0 || r.cl)) {
            // wide levels: K steps per launch, level l trailing level l - 1 by K steps (k_mc_tile), on the TILE stream
            // (ordinary priority); the narrow tail of the level order: one step per launch (k_mc_step) on the plan's own
            // high-priority stream, behind the last wide level (an event per tile).  The tail is what the window waits for
            // -- 288 launches that depend on each other -- so it gets the issue slots first and the tiles, which are pure
            // throughput, fill the rest; with the priorities the other way round the tail fell 140 steps behind and ran
            // 2.4 ms past the last tile.  A tile is queued when the tail needs it; at the end of the call the plan's stream
            // waits for the tiles queued so far, so whatever the caller queues next (a gather, the next window) sees every
            // row at t_end.
            const int32_t K = r.wide_k, W = r.wide, M = r.mid, K2 = r.mid_k;
            const int32_t w0 = tp.lvl_ptr[0], w1 = tp.lvl_ptr[W], m1 = tp.lvl_ptr[W + M], s1 = tp.lvl_ptr[L];
            const int32_t nt = (nsteps + K - 1) / K;
            const int32_t ntile = W > 0 ? nt + W - 1 : 0;
            const int32_t nmid = M > 0 ? (nsteps + K2 - 1) / K2 + M - 1 : 0;
            const bool tail = s1 > m1 && !r.cl;
            const bool tol = pl->opt.tol;
            hipStream_t ws = pl->wstream;
            r.tail_active = tail || M > 0 || r.cl;
            // Every tile of the window is queued at once, at the window's first call: the wide path needs all boundary
            // hydrographs up front (route_begin_t), so a tile depends on nothing but the tile before it.  (Queued one by one
            // as the tail came to need them, the last tiles of a window were late -- the host runs only a little ahead of the
            // device once the runtime's pool of dependency signals is in use -- and the tail, the critical path, waited
            // 1.2 ms for them.)  One event per tile: it ends the tile for the clock (a tile starts where the one before it
            // ended; the first has a start event of its own) and it is what the tail waits for.
            // what the tiles of this window are launched with: the in-block partition, the hot rows, the decimated output
            // (idempotent: the buffers are made and cleared on the first call only)
            StepArgs<T> at = a;
            if (r.cl) at.level = (const int32_t *)pl->lagk.p; // (tiles every position runs behind: the level in the slices)
            {
                // rows dealt to the threads of every block of a tile by the cost class they showed in the tile before (k_mc_tile's
                // prologue; trmc_plan_options.tile_perm_group: > 0 on, 0 off, < 0 the default below).
                const bool use_perm = pl->opt.tile_perm_group > 0 || (pl->opt.tile_perm_group < 0 && kTilePartitionDefault(pl->hinted));
                if (pl->out_stride > 0 && nsteps / pl->out_stride >= 1) {
                    r.dec_stride = pl->out_stride;
                    r.dec_keep = nsteps / pl->out_stride;
                    if (int rc = pl->dec.ensure((size_t)pl->nseg * r.dec_keep * 3 * sizeof(T))) return rc;
                    at.dec = (T *)pl->dec.p;
                    at.dec_stride = r.dec_stride;
                    at.dec_keep = r.dec_keep;
                    r.dec_lo = w0;
                    r.dec_hi = r.cl ? s1 : m1;
                }
                if (use_perm) {
                    const bool fresh = pl->cls_last.bytes < (size_t)pl->nseg_pad;
                    if (int rc = pl->cls_last.ensure((size_t)pl->nseg_pad)) return rc;
                    if (fresh) HIP_TRY(hipMemsetAsync(pl->cls_last.p, 0, (size_t)pl->nseg_pad, ws)); // (no history yet: one class)
                    at.cls_last = (uint8_t *)pl->cls_last.p;
                    // hot rows (trmc_plan_options.hot_rows): with the partition unless switched off.  Measured on the CONUS sequence
                    // (ms per day, with / without): plan built from the topology alone 17.4 / 19.5; cost-ordered plan on its own
                    // kind of days 16.22 / 16.34 -- once the list's blocks were made the FIRST of the launch: behind the others
                    // (the costliest rows of all started last, every launch ended on them) it was 16.7 / 16.2 and 18.3 / 19.4.
                    if (pl->opt.hot_rows != 0 && W > 0) {
                        const int32_t cap = std::max<int32_t>(kTileBlock, ((w1 - w0) / 32 + kTileBlock - 1) / kTileBlock * kTileBlock);
                        if (pl->hot_cap != cap || !pl->hot_list.p) { // (a tier of another size: the lists start empty, the marks are cleared)
                            if (int rc = pl->hot_list.ensure((size_t)3 * cap * sizeof(int32_t))) return rc;
                            const bool first = !pl->hot_cnt.p;
                            if (int rc = pl->hot_cnt.ensure(4 * sizeof(int32_t))) return rc;
                            HIP_TRY(hipMemsetAsync(pl->hot_cnt.p, 0, (first ? 4 : 3) * sizeof(int32_t), ws));
                            if (!fresh) HIP_TRY(hipMemsetAsync(pl->cls_last.p, 0, (size_t)pl->nseg_pad, ws));
                            pl->hot_cap = cap;
                        }
                        at.hot_list = (int32_t *)pl->hot_list.p;
                        at.hot_cnt = (int32_t *)pl->hot_cnt.p;
                        at.hot_cap = cap;
                        at.hot_wave_rows = hot_wave_rows_of(pl);
                        at.hot_home = (cap + at.hot_wave_rows * (kTileBlock / 64) - 1) / (at.hot_wave_rows * (kTileBlock / 64));
                    }
                }
            }
            if (r.wide_next == 0 && W > 0) {
                stamp(pl, ws, 0);
                HIP_TRY(hipEventRecord(pl->wide_t0[0], ws));
                for (int32_t j = 0; j < ntile; ++j) {
                    at.hot_cur = (int32_t)(pl->tile_seq++ % 3);
                    launch_tile<T>(ws, at, w0, w1, j, K, tol);
                    HIP_TRY(hipEventRecord(pl->wide_t1[(size_t)j], ws));
                    ++r.launches;
                }
                stamp(pl, ws, 1);
                stamp(pl, st, 2);
                r.wide_next = ntile;
                r.wide_through = -1; // (from here on: the last tile the tail has been told to wait for)
            }
            if (r.wide_next == 0 && W == 0) { // (cluster tiles only: no slices, nothing on the tile stream)
                stamp(pl, st, 2);
                r.wide_next = -1;
                r.wide_through = -1;
            }
            auto wait_tile = [&](int32_t need) -> int { // the plan's stream behind wide tile `need` (once per tile)
                if (ntile > 0 && need > r.wide_through) {
                    HIP_TRY(hipStreamWaitEvent(st, pl->wide_t1[(size_t)std::min(need, ntile - 1)], 0));
                    r.wide_through = need;
                }
                return 0;
            };
            // The SECOND tier (levels W .. W + M - 1, K2 steps per launch, level l trailing level l - 1 by K2 steps): the same
            // kernel on the plan's own stream, in order with the tail's launches.  Launch j2 routes level W + m through the
            // steps ((j2 - m) K2, (j2 - m + 1) K2]; its rows read the wide rows (any level below W) one step behind, so it goes
            // behind the tile that completes the LAST wide level through step (j2 + 1) K2 - 1; level W + m has completed
            // step s after launch m + ceil(s / K2) - 1.
            StepArgs<T> am = a;
            am.out_vec = a.out_vec && K2 % 4 == 0 && sizeof(T) == 4 && nsteps % 4 == 0;
            if (r.dec_stride > 0) {
                am.dec = (T *)pl->dec.p;
                am.dec_stride = r.dec_stride;
                am.dec_keep = r.dec_keep;
            }
            auto mid_through = [&](int32_t j2_last) -> int { // queue the second tier's launches up to index j2_last
                for (; r.mid_next <= std::min(j2_last, nmid - 1); ++r.mid_next) {
                    const int32_t j2 = r.mid_next;
                    const int32_t s_need = std::min((j2 + 1) * K2, nsteps) - 1; // the wide rows' step the launch reads up to
                    if (s_need >= 1)
                        if (int rc = wait_tile((s_need - 1) / K + W - 1)) return rc;
                    launch_tile<T>(st, am, w1, m1, j2 + W, K2, tol);
                    ++r.launches;
                }
                return 0;
            };
            if (r.cl) {
                // CLUSTER TILES on the plan's own stream: tile j routes cluster level c (W + c tiles behind level 0) through the
                // steps ((j - W - c) K, (j - W - c + 1) K]; its rows read rows outside their cluster at least one tile ahead of
                // them -- the last slice is there after wide tile j - 1, every cluster level above after cluster tile j - 1 (this
                // stream).  Every row has reached step t after tile W + C - 1 + ceil(t / K) - 1.
                const int32_t C = tp.ncl;
                const int32_t te = std::min(t_end, nsteps);
                const int32_t j_end = te <= 0 ? W - 1 : W + C - 1 + (te + K - 1) / K - 1;
                const int32_t *cblk_ptr = (const int32_t *)pl->cblk_ptr.p;
                StepArgs<T> ac = at;
                ac.hot_list = ac.hot_cnt = nullptr;
                for (; r.cl_next <= j_end; ++r.cl_next) {
                    const int32_t j = r.cl_next;
                    const int32_t c_lo = std::max(0, j - W - nt + 1), c_hi = std::min(C - 1, j - W);
                    if (c_lo > c_hi) continue;
                    if (j >= 1)
                        if (int rc = wait_tile(std::min(j - 1, ntile - 1))) return rc;
                    const int32_t b0 = tp.cblk_of_cl[(size_t)c_lo], b1 = tp.cblk_of_cl[(size_t)c_hi + 1];
                    if (b1 > b0) {
                        launch_ctile<T>(st, ac, cblk_ptr, b0, b1, j, K, tol);
                        ++r.launches;
                    }
                }
            }
            // (with lagged rows -- always in the tail, route_begin_t -- launch t routes the tail's other rows at step t and the
            // lagged ones at step t - maxlag, and the window ends at launch nsteps + maxlag: k_mc_step's LAG form)
            const int32_t lagmax = pl->maxlag;
            for (int32_t t = t0 + 1; t <= t_end && !r.cl; ++t) {
                const int32_t tn = std::min(t, nsteps);
                // the tail's step t reads the rows above it at step t - 1: the last level of the second tier is there after
                // launch M - 2 + ceil((t - 1) / K2) ...
                if (M > 0 && tn >= 2)
                    if (int rc = mid_through(M - 2 + (tn - 1 + K2 - 1) / K2)) return rc;
                if (tail) {
                    // ... and the last wide level after tile ceil((t - 1) / K) + W - 2 (and, with it, through step
                    // ceil((t - 1) / K) K: the tail waits once per K steps)
                    if (tn >= 2)
                        if (int rc = wait_tile((tn - 2) / K + W - 1)) return rc;
                    launch_step<T, true>(st, a, m1, s1, t, tol);
                    ++r.launches;
                }
                const int32_t t_all = t - lagmax; // every row has reached step t_all
                if (t_all > 0 && t_all % kTile == 0 && t_all < nsteps)
                    if (int rc = emit_tiles_through<T>(pl, t_all)) return rc;
            }
            // what the caller queues next on the plan's stream (a gather of cut-edge flows through t_end, the next window) must
            // see every WIDE row at t_end too: the tile that completes the last wide level through that step -- not every
            // tile queued, they are all queued up front and a chunk's hand-off must not wait for the window's last tile --
            // and every row of the second tier: its launches through the one that completes its last level
            {
                const int32_t te = std::min(t_end, nsteps);
                if (M > 0 && te >= 1)
                    if (int rc = mid_through(M - 2 + (te + K2 - 1) / K2)) return rc;
                const int32_t need_end = (te <= 0 || ntile == 0) ? -1 : std::min((te - 1) / K + W - 1, ntile - 1);
                if (need_end >= 0)
                    if (int rc = wait_tile(need_end)) return rc;
            }
        } else if (r.short_ts) {
            const int32_t s0 = tp.lvl_ptr[0], s1 = tp.lvl_ptr[L];
            const int32_t lagmax = pl->maxlag;
            for (int32_t t = t0 + 1; t <= t_end; ++t) { // launch t: rows at step t, lagged rows at step t - lagmax
                launch_step<T, true>(st, a, s0, s1, t, pl->opt.tol);
                ++r.launches;
                const int32_t t_all = t - lagmax; // every row has reached step t_all
                if (t_all > 0 && t_all % kTile == 0 && t_all < nsteps)
                    if (int rc = emit_tiles_through<T>(pl, t_all)) return rc;
            }
        } else {
            // level wavefront over the window (t0, t_end]: diagonal d runs (level l, step t0 + d - l)
            const int32_t W = t_end - t0;
            for (int32_t d = 1; d <= L - 1 + W; ++d) {
                const int32_t lo = d - W > 0 ? d - W : 0;
                const int32_t hi = d - 1 < L - 1 ? d - 1 : L - 1;
                const int32_t s0 = tp.lvl_ptr[lo], s1 = tp.lvl_ptr[hi + 1];
                if (s1 > s0) {
                    launch_step<T, false>(st, a, s0, s1, t0 + d, pl->opt.tol);
                    ++r.launches;
                }
                const int32_t t_all = t0 + d - (L - 1); // every level has reached step t_all
                if (t_all > t0 && t_all % kTile == 0 && t_all < nsteps)
                    if (int rc = emit_tiles_through<T>(pl, t_all)) return rc;
            }
        }
    }
    HIP_TRY(hipGetLastError());
    r.t_done = t_end;
    // Sequence mode (several plans taking turns on a device, route_begin_t): the window's end is queued with its last launch,
    // so that what ANOTHER plan queues next in the shared hardware queues comes after it -- trmc_route_end then waits for this
    // plan's window only, not for the other plan's as well
    if (t_end >= r.nsteps + pl->maxlag && pl->opt.sequence)
        if (int rc = route_end_queue<T>(pl)) return rc;
    return 0;
}

template <class T> int route_end_t(trmc_plan *pl)
{
    const trmc::Topology &tp = pl->topo;
    RouteRun &r = pl->run;
    const int32_t nsteps = r.nsteps;
    if (int rc = route_end_queue<T>(pl)) return rc;
    // (The STREAM, not the event behind the window's last launch: measured on the sequence with the decimated result among
    // the products -- 22.8 ms per day with the event against 19.5; 26.4 against 19.0 once the windows decimate as they go.  Behind a kernel queued after the window the stream's wait
    // returns when the OTHER plan's next window has ended; the host then queues every day a little late -- a pacing under which
    // the copies of consecutive days, 14 ms each on one PCIe direction, were observed not to run into each other.)
    HIP_TRY(hipStreamSynchronize(pl->stream));
    float ms01 = 0, ms12 = 0, ms23 = 0;
    HIP_TRY(hipEventElapsedTime(&ms01, pl->ev[0], pl->ev[1]));
    HIP_TRY(hipEventElapsedTime(&ms12, pl->ev[1], pl->ev[2]));
    HIP_TRY(hipEventElapsedTime(&ms23, pl->ev[2], pl->ev[3]));
    trmc_stats &s = pl->stats;
    s.nseg = pl->nseg;
    s.nseg_routed = pl->nrouted;
    s.nlevels = tp.nlevels;
    s.nsteps = nsteps;
    s.assume_short_ts = r.short_ts;
    s.main_launches = r.launches;
    s.segment_steps = pl->nrouted * (int64_t)nsteps;
    s.ms_prep = ms01;
    s.ms_main = ms12;
    s.ms_emit = ms23;
    s.ms_total = (double)ms01 + ms12 + ms23;
    s.wide_levels = r.wide;
    s.wide_k = r.wide_k;
    s.wide_launches = std::max(0, r.wide_next);
    s.mid_levels = r.mid;
    s.mid_k = r.mid_k;
    s.mid_launches = r.mid_next;
    s.arithmetic = pl->opt.tol ? TRMC_ARITH_TOLERANCE : TRMC_ARITH_EXACT;
    s.wide_segment_steps = r.cl ? (int64_t)(tp.lvl_ptr[tp.nlevels] - tp.lvl_ptr[0]) * nsteps
                                : (r.wide > 0 ? (int64_t)(tp.lvl_ptr[r.wide + r.mid] - tp.lvl_ptr[0]) * nsteps : 0);
    s.ms_wide = 0.0;
    {
        const size_t timed_n = std::min<size_t>((size_t)std::max(0, r.wide_next), pl->wide_t0.size());
        for (size_t i = 0; i < timed_n; ++i) { // (a tile's clock starts where the previous tile's stopped)
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, i == 0 ? pl->wide_t0[0] : pl->wide_t1[i - 1], pl->wide_t1[i]));
            s.ms_wide += ms;
        }
        if (timed_n > 0 && (int64_t)timed_n < (int64_t)r.wide_next) s.ms_wide *= (double)r.wide_next / (double)timed_n;
    }
    pl->routed_nsteps = nsteps;
    pl->dec_stride_done = r.dec_stride;
    pl->dec_keep_done = r.dec_keep;
    pl->dec_nsteps_done = nsteps;
    pl->dec_lo_done = r.dec_lo;
    pl->dec_hi_done = r.dec_hi;
    r.active = false;
    return 0;
}

// ---- dataflow engine, host side (fp32 plans in block order) ------------------------------------------------------
// Deal the blocks of a dataflow plan to the compute units by cost (see lean_pick_block).  Cost of a wavefront: the
// instructions of one of its steps by the hint of its costliest row (counters of the lean kernel by iteration class: dry
// 130, one iteration 600, two 890, three 1 150, over-bank rows about twice that) -- a hint is either the iteration class
// 0..3 (+ 4 over bank) or its window mean in sixteenths (ShardedRouter.iteration_hint); without one it is the drainage
// class 0..3, the same scale.  Blocks go out in rounds of one per unit, costliest first, each to the unit that carries least
// so far: every unit gets as many blocks as the hardware will send it workgroups (it deals those out by count).
int flow_place_blocks(trmc_plan *pl)
{
    pl->ncuq = 0;
    const int32_t nb = pl->topo.nblocks;
    if (nb <= 0) return 0;
    constexpr int kKeys = 4096;
    // which compute units are there?
    DevBuf seen;
    if (int rc = seen.ensure(kKeys)) return rc;
    HIP_TRY(hipMemset(seen.p, 0, kKeys));
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, pl->device);
    hipLaunchKernelGGL(k_cu_probe, dim3((unsigned)(64 * ncu)), dim3(64), 0, nullptr, (uint8_t *)seen.p);
    std::vector<uint8_t> hs(kKeys);
    hipError_t e = hipMemcpy(hs.data(), seen.p, kKeys, hipMemcpyDeviceToHost);
    seen.release();
    if (e != hipSuccess) return fail(TRMC_EHIP, std::string("compute-unit probe: ") + hipGetErrorString(e));
    std::vector<int32_t> cu_index(kKeys, -1);
    int32_t Q = 0;
    for (int k = 0; k < kKeys; ++k)
        if (hs[k]) cu_index[k] = Q++;
    if (Q < 2) return 0;
    // cost of every block
    const std::vector<uint8_t> &wc = pl->topo.cost_of_wave;
    int hi = 0;
    for (const uint8_t c : wc) hi = std::max(hi, (int)c);
    const bool sixteenths = hi > 7;
    auto units = [&](int hint) {
        const double c = sixteenths ? hint / 16.0 : (double)hint;
        static const double x[5] = {0, 1, 2, 3, 7}, y[5] = {130, 600, 890, 1150, 2200};
        for (int i = 1; i < 5; ++i)
            if (c <= x[i]) return y[i - 1] + (y[i] - y[i - 1]) * (c - x[i - 1]) / (x[i] - x[i - 1]);
        return y[4];
    };
    constexpr int wpb = kFlowBlock / 64;
    std::vector<double> cost((size_t)nb, 0.0), wcost((size_t)nb * wpb, 0.0);
    for (int32_t b = 0; b < nb; ++b)
        for (int q = 0; q < wpb; ++q) {
            const size_t w = (size_t)b * wpb + q;
            if (w < wc.size()) wcost[w] = units(wc[w]);
            cost[(size_t)b] += wcost[w];
        }
    std::vector<int32_t> order((size_t)nb);
    for (int32_t b = 0; b < nb; ++b) order[(size_t)b] = b;
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return cost[(size_t)x] > cost[(size_t)y]; });
    std::vector<double> load((size_t)Q, 0.0), sload((size_t)Q * 4, 0.0); // per unit; per SIMD of a unit
    std::vector<std::vector<int32_t>> queue((size_t)Q);
    std::vector<int32_t> cus((size_t)Q);
    std::vector<uint8_t> perm((size_t)nb + 4, 0xe4); // identity: group s for SIMD s
    for (int32_t r = 0; r < nb; r += Q) {
        for (int32_t c = 0; c < Q; ++c) cus[(size_t)c] = c;
        std::stable_sort(cus.begin(), cus.end(), [&](int32_t x, int32_t y) { return load[(size_t)x] < load[(size_t)y]; });
        for (int32_t j = 0; j < std::min(Q, nb - r); ++j) {
            const int32_t b = order[(size_t)(r + j)], c = cus[(size_t)j];
            queue[(size_t)c].push_back(b);
            load[(size_t)c] += cost[(size_t)b];
            if (wpb == 4) { // the block's row groups by descending cost meet the unit's SIMDs by ascending load
                int g[4] = {0, 1, 2, 3}, sd[4] = {0, 1, 2, 3};
                std::stable_sort(g, g + 4, [&](int x, int y) { return wcost[(size_t)b * 4 + x] > wcost[(size_t)b * 4 + y]; });
                std::stable_sort(sd, sd + 4, [&](int x, int y) { return sload[(size_t)c * 4 + x] < sload[(size_t)c * 4 + y]; });
                uint8_t pm = 0;
                for (int i = 0; i < 4; ++i) {
                    pm |= (uint8_t)(g[i] << (2 * sd[i]));
                    sload[(size_t)c * 4 + sd[i]] += wcost[(size_t)b * 4 + g[i]];
                }
                perm[(size_t)b] = pm;
            }
        }
    }
    std::vector<int32_t> ptr((size_t)Q + 1, 0), blk;
    blk.reserve((size_t)nb);
    for (int32_t c = 0; c < Q; ++c) {
        // (in the order they were dealt: by descending cost -- a unit's own workgroups take from the front, thieves from the back)
        for (const int32_t b : queue[(size_t)c]) blk.push_back(b);
        ptr[(size_t)c + 1] = (int32_t)blk.size();
    }
    if (int rc = pl->cuq_perm.ensure(perm.size())) return rc;
    HIP_TRY(hipMemcpy(pl->cuq_perm.p, perm.data(), perm.size(), hipMemcpyHostToDevice));
    if (int rc = upload_i32(pl->cuq_ptr, ptr, 1)) return rc;
    if (int rc = upload_i32(pl->cuq_blk, blk, 1)) return rc;
    if (int rc = upload_i32(pl->cu_index, cu_index, 1)) return rc;
    if (int rc = pl->cuq_head.ensure((size_t)Q * 2 * sizeof(int32_t))) return rc;
    pl->ncuq = Q;
    return 0;
}

FlowArgs flow_args(trmc_plan *pl, int nsteps, int qts, bool short_ts)
{
    FlowArgs a;
    a.dt_col = pl->dt_uniform ? nullptr : col<float>(pl, TRMC_P_DT);
    a.dt = (float)pl->dt;
    a.dx = col<float>(pl, TRMC_P_DX);
    a.bw = col<float>(pl, TRMC_P_BW);
    a.twcc = col<float>(pl, TRMC_P_TWCC);
    a.n = col<float>(pl, TRMC_P_N);
    a.ncc = col<float>(pl, TRMC_P_NCC);
    a.s0 = col<float>(pl, TRMC_P_S0);
    a.z = col<float>(pl, TRMC_NPARAM + 0);
    a.bfd = col<float>(pl, TRMC_NPARAM + 1);
    a.sqrt_s0 = col<float>(pl, TRMC_NPARAM + 2);
    a.sq1pz2 = col<float>(pl, TRMC_NPARAM + 3);
    a.s0_n = col<float>(pl, TRMC_NPARAM + 4);
    a.s0_ncc = col<float>(pl, TRMC_NPARAM + 5);
    a.inv_n = col<float>(pl, TRMC_NPARAM + 6);
    a.up_ptr = (const int32_t *)pl->up_ptr.p;
    a.up_idx = (const int32_t *)pl->up_idx.p;
    a.up2 = (const int2 *)pl->up2.p;
    // short-timestep mode: the skew of trmc_plan_set_lag; general mode: the level rank inside the block
    a.lag = short_ts ? (pl->maxlag > 0 ? (const int32_t *)pl->lag.p : nullptr)
                     : (pl->topo.maxrank > 0 ? (const int32_t *)pl->rank.p : nullptr);
    a.ticket_map = short_ts ? nullptr : (const int32_t *)pl->ticket_map.p;
    a.qlat_tm = (const float *)pl->qlat_tm.p;
    a.gran = (unsigned long long *)pl->tm.p;
    a.d_state = (float *)pl->d_state.p;
    a.d_gran = (unsigned long long *)pl->d_gran.p;
    a.out = (float *)pl->out.p;
    a.row_of_pos = (const int32_t *)pl->row_of_pos.p;
    a.it_prev = (uint8_t *)pl->it_prev.p;
    a.it_sum = pl->collect_cost ? (uint16_t *)pl->it_sum.p : nullptr;
    a.sane = pl->params_sane;
    a.out_vec = nsteps % 4 == 0;
    a.res_of_pos = pl->nres > 0 ? (const int32_t *)pl->res_of_pos.p : nullptr;
    a.res_par = (const float *)pl->res_par.p;
    a.res_inflow = (float *)pl->res_inflow.p;
    a.res_dt = (float)pl->res_dt;
    const bool da = pl->ngage > 0;
    a.gage_of_pos = da ? (const int32_t *)pl->gage_of_pos.p : nullptr;
    a.da_mode = (const uint8_t *)pl->da_mode.p;
    a.da_a = (const float *)pl->da_a.p;
    a.da_w = (const float *)pl->da_w.p;
    a.da_nudge = (float *)pl->da_nudge.p;
    a.nseg_pad = pl->nseg_pad;
    a.nsteps = nsteps;
    a.qts = qts;
    a.nseg = (int32_t)pl->nseg;
    a.first = (int32_t)pl->topo.nboundary;
    a.tag_base = pl->tag_base;
    a.ticket = (int32_t *)pl->ticket.p;
    a.watchdog_ticks = pl->watchdog_ticks;
    a.dbg = pl->opt.flow_debug ? (unsigned long long *)pl->dbg.p : nullptr; // (trmc_plan_options.flow_debug: when did every block run?)
    a.nblocks_dbg = pl->topo.nblocks;
    a.prio = (const uint8_t *)pl->prio.p;
    a.cuq_ptr = a.cuq_blk = a.cu_index = nullptr; // (flow_route_advance switches the queues on for lean launches)
    a.cuq_perm = nullptr;
    a.cuq_head = nullptr;
    a.ncuq = 0;
    return a;
}

int flow_route_begin(trmc_plan *pl, int nsteps, int qts, int short_ts)
{
    const trmc::Topology &tp = pl->topo;
    const int32_t n = (int32_t)pl->nseg;
    const int64_t np = pl->nseg_pad;
    hipStream_t st = pl->stream;
    if (int rc = pl->tm.ensure((size_t)(nsteps + 1) * np * sizeof(unsigned long long), true)) return rc;
    if (int rc = pl->d_state.ensure((size_t)np * sizeof(float))) return rc;
    if (int rc = pl->d_gran.ensure((size_t)np * sizeof(unsigned long long), true)) return rc;
    if (int rc = pl->ticket.ensure(16 * sizeof(int32_t))) return rc; // one set of 8 per compute stream
    if (pl->opt.flow_debug) {
        if (int rc = pl->dbg.ensure((size_t)(tp.nblocks + 1) * 6 * sizeof(unsigned long long))) return rc;
        HIP_TRY(hipMemsetAsync(pl->dbg.p, 0, pl->dbg.bytes, st));
    }
    if (int rc = pl->qlat_tm.ensure((size_t)pl->nq * np * sizeof(float))) return rc;
    if (int rc = pl->out.ensure((size_t)pl->nseg * nsteps * 3 * sizeof(float))) return rc;
    if (pl->nres > 0)
        if (int rc = pl->res_inflow.ensure((size_t)pl->nres * nsteps * sizeof(float))) return rc;
    if (pl->collect_cost) {
        if (int rc = pl->it_sum.ensure((size_t)np * sizeof(uint16_t))) return rc;
        pl->cost_nsteps = nsteps;
    }
    // a fresh range of tags for this window: nothing an earlier window left in the plane can pass for a live granule
    if ((uint64_t)pl->tag_base + (uint64_t)pl->tag_span + (uint64_t)nsteps + 2 >= 0xffffffffull) {
        HIP_TRY(hipMemsetAsync(pl->tm.p, 0, pl->tm.bytes, st));
        HIP_TRY(hipMemsetAsync(pl->d_gran.p, 0, pl->d_gran.bytes, st));
        pl->tag_base = 1;
        pl->tag_span = 0;
    }
    pl->tag_base += (uint32_t)pl->tag_span;
    pl->tag_span = nsteps + 1;
    const int32_t *row_of_pos = (const int32_t *)pl->row_of_pos.p;
    HIP_TRY(hipEventRecord(pl->ev[0], st));
    pl->q0_staged = false; // (consumed by this window)
    if (pl->dec_pending) { // (trmc_fetch_begin_fvd: the copy stream reads `out`)
        HIP_TRY(hipStreamWaitEvent(st, pl->ev_dec, 0));
        if (pl->fstream) HIP_TRY(hipStreamWaitEvent(pl->fstream, pl->ev_dec, 0));
        pl->dec_pending = false;
    }
    if (pl->forcing_pending) { // (trmc_stage_forcing: the copy into in_qlat runs on the copy stream)
        HIP_TRY(hipStreamWaitEvent(st, pl->ev_forcing, 0));
        pl->forcing_pending = false;
    }
    HIP_TRY(hipMemsetAsync(pl->it_prev.p, 0, (size_t)np, st));
    HIP_TRY(hipMemsetAsync(pl->ticket.p, 0, 16 * sizeof(int32_t), st));
    pl->flow_next = pl->flow_last = 0;
    if (pl->collect_cost) HIP_TRY(hipMemsetAsync(pl->it_sum.p, 0, (size_t)np * sizeof(uint16_t), st));
    if (n > 0) {
        if (!pl->qlat_direct)
            hipLaunchKernelGGL((k_prep_qlat<float>), dim3((n + 63) / 64, (unsigned)((pl->nq + 31) / 32)), dim3(kBlock), 0, st,
                               (const float *)pl->in_qlat.p, row_of_pos, (float *)pl->qlat_tm.p, n, np, (int32_t)pl->nq);
        hipLaunchKernelGGL(k_flow_init, dim3(blocks_for(n)), dim3(kBlock), 0, st, (const float *)pl->in_q0.p, row_of_pos,
                           (unsigned long long *)pl->tm.p, (float *)pl->d_state.p, (unsigned long long *)pl->d_gran.p, n,
                           pl->tag_base);
    }
    RouteRun &r = pl->run;
    r = RouteRun{};
    r.active = true;
    r.nsteps = nsteps;
    r.qts = qts;
    r.short_ts = short_ts ? 1 : 0;
    r.boundary_through = tp.nboundary > 0 ? 0 : nsteps;
    if (tp.nboundary > 0 && pl->have_boundary) {
        hipLaunchKernelGGL(k_flow_boundary, dim3(blocks_for(tp.nboundary * (int64_t)nsteps)), dim3(kBlock), 0, st,
                           (const float *)pl->in_bfvd.p, (unsigned long long *)pl->tm.p, (float *)pl->out.p, row_of_pos,
                           (int32_t)tp.nboundary, nsteps, np, 0, nsteps, (int64_t)nsteps * 3, 3, 3, pl->tag_base, (const int64_t *)nullptr);
        r.boundary_through = nsteps;
    }
    HIP_TRY(hipEventRecord(pl->ev[1], st));
    HIP_TRY(hipStreamWaitEvent(pl->fstream, pl->ev[1], 0)); // the second compute stream starts behind the window's set-up
    HIP_TRY(hipGetLastError());
    pl->routed_nsteps = -1;
    return 0;
}

// Which form of the short-timestep kernel a window uses, and whether consecutive launches of it may overlap.
// The lean form pays where every block of the launch is resident at once (6 workgroups per compute unit: one rank of a
// multi-GPU job, a regional network) -- there the pace is set by latency and by the slowest wavefront, and occupancy plus
// wavefront priorities win (349 k rows, 288 steps: 4.8 ms against 6.1 ms).  Where blocks run in many rounds (CONUS on one
// GPU: 10 661 blocks) throughput counts and the staged form, which spills nothing and writes whole sectors, is ahead
// (25.4 ms against 30.5 ms).
static bool flow_lean(const trmc_plan *pl, int nsteps)
{
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, pl->device);
    const int32_t force = pl->opt.flow_lean; // (trmc_plan_options.flow_lean: > 0 always, < 0 never -- A/B measurements)
    return (uint64_t)pl->nseg * (uint64_t)nsteps * 3ull < (1ull << 32)
           && (force ? force > 0 : pl->topo.nblocks <= TRMC_LEAN_WAVES * (256 / kFlowBlock) * ncu);
}
// Optional (trmc_plan_options.flow_overlap; off by default): consecutive launches of a short-timestep window on the lean kernel
// alternate between two compute streams.  Rows hand their state from launch to launch through granules (flow plane,
// d_gran), so launch c+1 needs no kernel boundary behind launch c -- its cheap blocks take the slots launch c's cheap blocks
// have left while c's costly blocks are still finishing.  Safe because every block of a launch is resident (flow_lean):
// launch c+2, behind c on its stream, starts when all of c+1 is.  Measured on 349 k-row ranks (8 chunks per window): no
// gain -- 4.9 ms either way once both streams have a hardware queue each, and the two streams share one under
// GPU_MAX_HW_QUEUES=1, which troute_amd.distributed sets -- hence opt-in.
static bool flow_overlap(const trmc_plan *pl)
{
    return pl->opt.flow_overlap && pl->run.short_ts && flow_lean(pl, pl->run.nsteps);
}
static hipStream_t flow_stream(const trmc_plan *pl, int which) { return which ? pl->fstream : pl->stream; }

// one launch routes every row through the launches / steps (t_done, t_end]
int flow_route_advance(trmc_plan *pl, int t_end)
{
    RouteRun &r = pl->run;
    if (pl->nrouted > 0 && t_end > r.t_done) {
        const bool lean = r.short_ts && flow_lean(pl, r.nsteps);
        const int which = flow_overlap(pl) ? pl->flow_next : 0;
        hipStream_t st = flow_stream(pl, which);
        FlowArgs a = flow_args(pl, r.nsteps, r.qts, r.short_ts != 0);
        a.ticket += 8 * which;                                       // every compute stream has its own block tickets
        HIP_TRY(hipMemsetAsync(a.ticket, 0, sizeof(int32_t), st));   // they restart; the abort flag stays
        if (lean && pl->ncuq > 0) { // blocks by compute unit (lean_pick_block)
            a.cuq_ptr = (const int32_t *)pl->cuq_ptr.p;
            a.cuq_blk = (const int32_t *)pl->cuq_blk.p;
            a.cu_index = (const int32_t *)pl->cu_index.p;
            a.cuq_perm = (const uint8_t *)pl->cuq_perm.p;
            a.cuq_head = (int32_t *)pl->cuq_head.p + (size_t)pl->ncuq * which;
            a.ncuq = pl->ncuq;
            HIP_TRY(hipMemsetAsync(a.cuq_head, 0, (size_t)pl->ncuq * sizeof(int32_t), st));
        }
        const dim3 grid((unsigned)pl->topo.nblocks), block(kFlowBlock);
        if (pl->opt.tol) {
            if (lean && a.lag)
                hipLaunchKernelGGL((k_mc_flow_lean<true, true>), grid, block, 0, st, a, r.t_done, t_end);
            else if (lean)
                hipLaunchKernelGGL((k_mc_flow_lean<false, true>), grid, block, 0, st, a, r.t_done, t_end);
            else if (r.short_ts)
                hipLaunchKernelGGL((k_mc_flow<true, true>), grid, block, 0, st, a, r.t_done, t_end);
            else
                hipLaunchKernelGGL((k_mc_flow<false, true>), grid, block, 0, st, a, r.t_done, t_end);
        } else if (lean && a.lag)
            hipLaunchKernelGGL((k_mc_flow_lean<true>), grid, block, 0, st, a, r.t_done, t_end);
        else if (lean)
            hipLaunchKernelGGL((k_mc_flow_lean<false>), grid, block, 0, st, a, r.t_done, t_end);
        else if (r.short_ts)
            hipLaunchKernelGGL((k_mc_flow<true>), grid, block, 0, st, a, r.t_done, t_end);
        else
            hipLaunchKernelGGL((k_mc_flow<false>), grid, block, 0, st, a, r.t_done, t_end);
        HIP_TRY(hipEventRecord(pl->ev_chunk[which], st));
        pl->flow_last = which;
        if (flow_overlap(pl)) pl->flow_next = 1 - which;
        ++r.launches;
    }
    HIP_TRY(hipGetLastError());
    r.t_done = t_end;
    return 0;
}

int flow_route_end(trmc_plan *pl)
{
    RouteRun &r = pl->run;
    hipStream_t st = pl->stream;
    if (r.launches > 0) { // whatever the second compute stream still runs belongs to the window
        HIP_TRY(hipStreamWaitEvent(st, pl->ev_chunk[0], 0));
        if (flow_overlap(pl) && r.launches > 1) HIP_TRY(hipStreamWaitEvent(st, pl->ev_chunk[1], 0));
    }
    HIP_TRY(hipEventRecord(pl->ev[2], st));
    HIP_TRY(hipEventRecord(pl->ev[3], st));
    HIP_TRY(hipStreamSynchronize(st));
    pl->flow_next = pl->flow_last = 0;
    int32_t both[16] = {0};
    HIP_TRY(hipMemcpy(both, pl->ticket.p, sizeof both, hipMemcpyDeviceToHost));
    const int32_t *flags = both[1] != 0 ? both : both + 8;
    if (flags[1] != 0) {
        r.active = false;
        const uint64_t idx = ((uint64_t)(uint32_t)flags[3] << 32) | (uint32_t)flags[2];
        const uint64_t np = (uint64_t)pl->nseg_pad;
        return fail(TRMC_EHIP, "dataflow engine: a row waited longer than the watchdog allows for an upstream flow "
                               "(boundary hydrographs missing for the steps routed, or an internal error); window abandoned"
                               " [waited for position " + std::to_string(idx % np) + " (" + (idx % np < (uint64_t)pl->topo.nboundary ? "a boundary row" : "a routed row")
                               + ") at step " + std::to_string(idx / np) + ", tag " + std::to_string((uint32_t)flags[4]) + ", found tag "
                               + std::to_string((uint32_t)flags[5]) + ", window tag base " + std::to_string(pl->tag_base) + "]");
    }
    if (pl->opt.flow_debug && pl->topo.nblocks > 0) { // developer aid: when did every block run?
        const int32_t nb = pl->topo.nblocks;
        std::vector<unsigned long long> d((size_t)nb * 2);
        HIP_TRY(hipMemcpy(d.data(), pl->dbg.p, d.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        std::vector<uint32_t> cu_key(nb);
        for (int32_t b = 0; b < nb; ++b) { // (the start word carries the compute unit in its top 16 bits)
            cu_key[b] = (uint32_t)(d[2 * b] >> 48);
            d[2 * b] &= 0xffffffffffffull;
        }
        std::vector<unsigned long long> dx((size_t)nb * 4, 0ull); // per row group: 1 + SIMD + 256 * wavefront of the workgroup
        HIP_TRY(hipMemcpy(dx.data(), (unsigned long long *)pl->dbg.p + 2 * (size_t)(nb + 1), dx.size() * sizeof(unsigned long long),
                          hipMemcpyDeviceToHost));
        if (const char *path = std::getenv("TRMC_FLOW_DEBUG_FILE")) { // one line per block: id, compute unit, start, end (ms)
            if (FILE *f = std::fopen(path, "a")) {
                std::fprintf(f, "# launch of %d blocks\n", nb);
                for (int32_t b = 0; b < nb; ++b)
                {
                    std::fprintf(f, "%d %u %.5f %.5f", b, cu_key[b], d[2 * b] * 1e-5, d[2 * b + 1] * 1e-5);
                    for (int q = 0; q < kFlowBlock / 64; ++q) {
                        const size_t w = (size_t)b * (kFlowBlock / 64) + q;
                        std::fprintf(f, " %d", w < pl->topo.cost_of_wave.size() ? (int)pl->topo.cost_of_wave[w] : -1);
                    }
                    for (int q = 0; q < kFlowBlock / 64; ++q) std::fprintf(f, " %llu", dx[(size_t)4 * b + q]);
                    std::fprintf(f, "\n");
                }
                std::fclose(f);
            }
        }
        unsigned long long t_min = ~0ull, t_max = 0;
        for (int32_t b = 0; b < nb; ++b) {
            t_min = std::min(t_min, d[2 * b]);
            t_max = std::max(t_max, d[2 * b + 1]);
        }
        std::vector<int32_t> by_end(nb);
        for (int32_t b = 0; b < nb; ++b) by_end[b] = b;
        std::sort(by_end.begin(), by_end.end(), [&](int32_t x, int32_t y) { return d[2 * x + 1] > d[2 * y + 1]; });
        std::fprintf(stderr, "[flow debug] %d blocks, span %.3f ms; blocks ending last (id, start ms, end ms, duration ms):\n", nb,
                     (t_max - t_min) * 1e-5);
        for (int32_t i = 0; i < std::min(nb, 12); ++i) {
            const int32_t b = by_end[i];
            std::fprintf(stderr, "   %6d  %9.3f %9.3f %9.3f\n", b, (d[2 * b] - t_min) * 1e-5, (d[2 * b + 1] - t_min) * 1e-5,
                         (d[2 * b + 1] - d[2 * b]) * 1e-5);
        }
        // how many blocks are running at a few instants
        for (int k = 1; k <= 10; ++k) {
            const unsigned long long tt = t_min + (t_max - t_min) * k / 11;
            int32_t live = 0;
            for (int32_t b = 0; b < nb; ++b) live += d[2 * b] <= tt && d[2 * b + 1] > tt;
            std::fprintf(stderr, "   at %8.3f ms: %d blocks live\n", (tt - t_min) * 1e-5, live);
        }
    }
    float ms01 = 0, ms12 = 0;
    HIP_TRY(hipEventElapsedTime(&ms01, pl->ev[0], pl->ev[1]));
    HIP_TRY(hipEventElapsedTime(&ms12, pl->ev[1], pl->ev[2]));
    trmc_stats &s = pl->stats;
    s.nseg = pl->nseg;
    s.nseg_routed = pl->nrouted;
    s.nlevels = pl->topo.nlevels;
    s.nsteps = r.nsteps;
    s.assume_short_ts = r.short_ts;
    s.main_launches = r.launches;
    s.segment_steps = pl->nrouted * (int64_t)r.nsteps;
    s.ms_prep = ms01;
    s.ms_main = ms12;
    s.ms_emit = 0.0; // results are written in the caller's layout by the routing kernel itself
    s.ms_total = (double)ms01 + ms12;
    s.wide_levels = s.wide_k = s.wide_launches = s.mid_levels = s.mid_k = s.mid_launches = 0;
    s.wide_segment_steps = 0;
    s.ms_wide = 0.0;
    s.arithmetic = pl->opt.tol ? TRMC_ARITH_TOLERANCE : TRMC_ARITH_EXACT;
    pl->routed_nsteps = r.nsteps;
    r.active = false;
    return 0;
}

template <class T> int segments_t(int64_t n, const void *in, void *out, bool tol, int32_t *iters_out)
{
    DevBuf din, dout, dit;
    int rc = din.ensure((size_t)n * 15 * sizeof(T));
    if (!rc) rc = dout.ensure((size_t)n * 6 * sizeof(T));
    if (!rc && iters_out) rc = dit.ensure((size_t)n * sizeof(int32_t));
    if (!rc) {
        hipError_t e = hipMemcpy(din.p, in, (size_t)n * 15 * sizeof(T), hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            bool launched = false;
            if constexpr (sizeof(T) == 4) {
                if (tol) {
                    hipLaunchKernelGGL((k_segments<T, true>), dim3(blocks_for(n)), dim3(kBlock), 0, 0, (const T *)din.p, (T *)dout.p,
                                       (int32_t *)dit.p, n);
                    launched = true;
                }
            }
            if (!launched)
                hipLaunchKernelGGL((k_segments<T, false>), dim3(blocks_for(n)), dim3(kBlock), 0, 0, (const T *)din.p, (T *)dout.p,
                                   (int32_t *)dit.p, n);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e == hipSuccess) e = hipMemcpy(out, dout.p, (size_t)n * 6 * sizeof(T), hipMemcpyDeviceToHost);
        if (e == hipSuccess && iters_out) e = hipMemcpy(iters_out, dit.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(TRMC_EHIP, std::string("trmc_segments: ") + hipGetErrorString(e));
    }
    din.release();
    dout.release();
    dit.release();
    return rc;
}

// trmc_plan_chain_from: time row `src_row` of one plan's planes -> time row 0 of another's, positions [lo, hi) (same order)
template <class T>
__global__ void __launch_bounds__(kBlock)
k_chain_state(const T *__restrict__ sq, const T *__restrict__ sv, const T *__restrict__ sd, T *__restrict__ dq,
              T *__restrict__ dv, T *__restrict__ dd, int32_t lo, int32_t hi)
{
    const int32_t p = lo + (int32_t)blockIdx.x * kBlock + (int32_t)threadIdx.x;
    if (p >= hi) return;
    dq[p] = sq[p];
    dv[p] = sv[p];
    dd[p] = sd[p];
}
template <class T> int chain_from_t(trmc_plan *dst, trmc_plan *src, int nsteps_dst)
{
    const trmc::Topology &tp = src->topo;
    const int64_t np = src->nseg_pad;
    const int32_t ns = src->run.nsteps;
    const size_t plane_s = (size_t)(ns + 1) * np, plane_d = (size_t)(nsteps_dst + 1) * np;
    if (int rc = dst->tm.ensure(3 * plane_d * sizeof(T))) return rc;
    const T *sq = (const T *)src->tm.p + (size_t)ns * np, *sv = sq + plane_s, *sd = sv + plane_s;
    T *dq = (T *)dst->tm.p, *dv = dq + plane_d, *dd = dv + plane_d;
    for (int i = 0; i < 4; ++i)
        if (!dst->ev_chain[i]) HIP_TRY(hipEventCreateWithFlags(&dst->ev_chain[i], hipEventDisableTiming));
    if (!dst->wstream) {
        HIP_TRY(hipStreamCreateWithFlags(&dst->wstream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&dst->ev_tail, hipEventDisableTiming));
    }
    // the rows of the source's wide levels are final when its last tile is (they never wrote a velocity row: the velocity
    // of the initial state is not an input of the step); the other rows when its tail is
    const int32_t W = src->run.short_ts ? src->run.wide : 0, M = W > 0 ? src->run.mid : 0;
    const int32_t b0 = 0, w1 = tp.lvl_ptr[W], s1 = (int32_t)src->nseg; // (boundary rows, if any, go with the tail's part)
    const int32_t w0 = W > 0 ? tp.lvl_ptr[0] : w1;
    const int32_t m1 = tp.lvl_ptr[W + M]; // (the second tier's rows, [w1, m1): routed on the source's own stream, no velocity row either)
    if (W > 0 && src->wstream) {
        HIP_TRY(hipEventRecord(dst->ev_chain[0], src->wstream));
        HIP_TRY(hipStreamWaitEvent(dst->wstream, dst->ev_chain[0], 0));
        hipLaunchKernelGGL((k_chain_state<T>), dim3(blocks_for(w1 - w0)), dim3(kBlock), 0, dst->wstream, sq, sq, sd, dq, dv, dd, w0, w1);
        HIP_TRY(hipEventRecord(dst->ev_chain[2], dst->wstream));
        // The source's NEXT window must not overwrite what is being read here.  Not as waits queued on the source's streams
        // now: with one hardware queue per stream priority the two plans' tile streams share an in-order queue, and a wait
        // placed there at this point -- behind the source's tiles, ahead of everything this plan is about to queue -- holds
        // THIS plan's set-up and tiles back until the event fires: the second one below only does when the source's tail
        // has ended, so the receiver's leading levels started after the source's whole window instead of behind its last
        // tile (the timeline of bench.py's sequence showed exactly that).  The source takes the events (its own objects)
        // and waits for them at its next trmc_route_begin -- a window later, when they have long fired.
        if (!src->ev_released[0]) HIP_TRY(hipEventCreateWithFlags(&src->ev_released[0], hipEventDisableTiming));
        HIP_TRY(hipEventRecord(src->ev_released[0], dst->wstream));
        src->released_pending[0] = true;
        // ... and the receiver's own stream does not read time row 0 of the wide rows before this copy has written it: its
        // tail's first launch (step 1) takes no tile wait and reads q_tm[0] of its upstream wide rows
        HIP_TRY(hipStreamWaitEvent(dst->stream, dst->ev_chain[2], 0));
    }
    HIP_TRY(hipEventRecord(dst->ev_chain[1], src->stream));
    HIP_TRY(hipStreamWaitEvent(dst->stream, dst->ev_chain[1], 0));
    if (w0 > b0) hipLaunchKernelGGL((k_chain_state<T>), dim3(blocks_for(w0 - b0)), dim3(kBlock), 0, dst->stream, sq, sv, sd, dq, dv, dd, b0, w0);
    if (m1 > w1) hipLaunchKernelGGL((k_chain_state<T>), dim3(blocks_for(m1 - w1)), dim3(kBlock), 0, dst->stream, sq, sq, sd, dq, dv, dd, w1, m1);
    // (rows routed by cluster tiles wrote no velocity row either)
    if (s1 > m1) hipLaunchKernelGGL((k_chain_state<T>), dim3(blocks_for(s1 - m1)), dim3(kBlock), 0, dst->stream, sq, src->run.cl ? sq : sv, sd, dq, dv, dd, m1, s1);
    HIP_TRY(hipEventRecord(dst->ev_chain[3], dst->stream));
    if (!src->ev_released[1]) HIP_TRY(hipEventCreateWithFlags(&src->ev_released[1], hipEventDisableTiming));
    HIP_TRY(hipEventRecord(src->ev_released[1], dst->stream));
    src->released_pending[1] = true;
    HIP_TRY(hipGetLastError());
    dst->chain_staged = true;
    return 0;
}

// trmc_selfcheck_fast_arith: the short forms of DevMathF against the operations they stand for (see trmc.h)
__global__ void __launch_bounds__(kBlock)
k_selfcheck_sqrt(uint32_t lo_bits, uint32_t hi_bits, unsigned long long *mismatches)
{
    DevMathF m{nullptr, false};
    unsigned long long bad = 0;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t b = (uint64_t)lo_bits + (uint64_t)blockIdx.x * kBlock + threadIdx.x; b <= hi_bits; b += stride) {
        const float x = __uint_as_float((uint32_t)b);
        bad += __float_as_uint(m.sqrt_r(x, true)) != __float_as_uint(::sqrtf(x));
    }
    if (bad) atomicAdd(mismatches, bad);
}
__device__ __forceinline__ uint64_t selfcheck_mix(uint64_t z) // splitmix64
{
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
// a float with a uniform significand and an exponent uniform over [e_lo, e_hi] (2**e_hi itself included as the top value)
__device__ __forceinline__ float selfcheck_draw(uint64_t r, int e_lo, int e_hi)
{
    const uint32_t span = (uint32_t)(e_hi - e_lo);
    const uint32_t e = (uint32_t)((r >> 32) % (span + 1));
    const uint32_t frac = e == span ? 0u : (uint32_t)r & 0x7fffffu;
    return __uint_as_float(((uint32_t)(e_lo + (int)e + 127) << 23) | frac);
}
__global__ void __launch_bounds__(kBlock)
k_selfcheck_div(int64_t n, uint64_t seed, unsigned long long *mismatches)
{
    DevMathF m{nullptr, false};
    unsigned long long bad = 0;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const uint64_t r0 = selfcheck_mix(seed + 3 * (uint64_t)i), r1 = selfcheck_mix(seed + 3 * (uint64_t)i + 1),
                       r2 = selfcheck_mix(seed + 3 * (uint64_t)i + 2);
        float a = selfcheck_draw(r0, -10, 19), b = selfcheck_draw(r1, -76, 62);
        const float c = selfcheck_draw(r2, -20, 40);
        // one draw in eight: a divisor (and in half of those a dividend too) with a significand of all ones, all ones but the
        // last bit, or all zeros -- the hard cases of reciprocal-based division
        if (((r2 >> 40) & 7u) == 0u) {
            const uint32_t pick = (uint32_t)(r2 >> 43) & 3u, frac = pick == 0 ? 0x7fffffu : pick == 1 ? 0x7ffffeu : pick == 2 ? 0u : 1u;
            b = __uint_as_float((__float_as_uint(b) & 0xff800000u) | frac);
            if ((r2 >> 45) & 1u) a = __uint_as_float((__float_as_uint(a) & 0xff800000u) | (0x7fffffu - frac));
        }
        const float q_fast = m.k_of(a, b), q = a / b;
        const float k_fast = m.max_num(c, q_fast), k = c > q ? c : q;
        bad += (__float_as_uint(q_fast) != __float_as_uint(q)) || (__float_as_uint(k_fast) != __float_as_uint(k));
    }
    if (bad) atomicAdd(mismatches, bad);
}

int check_device(int device)
{
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(TRMC_ENODEVICE, std::string("no HIP device available (") + hipGetErrorString(e)
                                        + "); this library has no CPU fallback");
    if (device < 0 || device >= count)
        return fail(TRMC_EINVAL, "device ordinal " + std::to_string(device) + " out of range [0," + std::to_string(count) + ")");
    return 0;
}

} // namespace

// ---------------------------------------------------------------- C ABI
extern "C" {

static void stream_release(trmc_plan *pl); // (stream.inc)
static bool stream_active(const trmc_plan *pl);

const char *trmc_last_error(void) { return g_err.c_str(); }
int trmc_abi_version(void) { return TRMC_ABI_VERSION; }

int trmc_device_count(int *count)
{
    if (!count) return fail(TRMC_EINVAL, "count is NULL");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        *count = 0;
        return fail(TRMC_ENODEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    *count = c;
    return 0;
}

int trmc_topology_levels(int64_t nseg, const int64_t *up_ptr, const int64_t *up_idx, const uint8_t *boundary,
                         int32_t *level_of_row, int64_t *plan_pos_of_row, int32_t *nlevels)
{
    return trmc_topology_levels_hinted(nseg, up_ptr, up_idx, boundary, nullptr, level_of_row, plan_pos_of_row, nlevels);
}

int trmc_topology_levels_hinted(int64_t nseg, const int64_t *up_ptr, const int64_t *up_idx, const uint8_t *boundary,
                                const uint8_t *cost_hint, int32_t *level_of_row, int64_t *plan_pos_of_row, int32_t *nlevels)
{
    trmc::Topology t;
    std::string err;
    const int rc = trmc::build_topology(nseg, up_ptr, up_idx, boundary, t, err, cost_hint);
    if (rc) return fail(rc == -2 ? TRMC_ECYCLE : TRMC_EINVAL, err);
    for (int64_t r = 0; r < nseg; ++r) {
        if (level_of_row) level_of_row[r] = t.level_of_row[r];
        if (plan_pos_of_row) plan_pos_of_row[r] = t.pos_of_row[r];
    }
    if (nlevels) *nlevels = t.nlevels;
    return 0;
}

int trmc_topology_clusters(int64_t nseg, const int64_t *up_ptr, const int64_t *up_idx, const uint8_t *boundary,
                           const uint8_t *cost_hint, int64_t wide_min_rows, int32_t wide_max_levels, int32_t cluster_rows,
                           int64_t *plan_pos_of_row, int32_t *lag_of_row, int32_t *block_of_row, int32_t *wide_levels,
                           int32_t *cluster_levels, int32_t *cluster_blocks)
{
    if (cluster_rows <= 0 || cluster_rows > kTileBlock) return fail(TRMC_EINVAL, "cluster_rows must be in [1, 128]");
    trmc::Topology t;
    std::string err;
    const int rc = trmc::build_topology(nseg, up_ptr, up_idx, boundary, t, err, cost_hint, 0, true, kWideMaxLevels, wide_min_rows,
                                        std::min<int32_t>(wide_max_levels, kWideMaxLevels), 0, 0, 0, cluster_rows);
    if (rc) return fail(rc == -2 ? TRMC_ECYCLE : TRMC_EINVAL, err);
    const int32_t nb = t.ncl > 0 ? (int32_t)t.cblk_ptr.size() - 1 : 0;
    std::vector<int32_t> blk_of_pos((size_t)nseg, -1);
    for (int32_t b = 0; b < nb; ++b)
        for (int32_t p = t.cblk_ptr[(size_t)b]; p < t.cblk_ptr[(size_t)b + 1]; ++p) blk_of_pos[(size_t)p] = b;
    for (int64_t r = 0; r < nseg; ++r) {
        const int32_t p = t.pos_of_row[r];
        if (plan_pos_of_row) plan_pos_of_row[r] = p;
        if (lag_of_row) lag_of_row[r] = t.level_of_row[r] < 0 ? -1 : t.lagk_of_pos[(size_t)p];
        if (block_of_row) block_of_row[r] = blk_of_pos[(size_t)p];
    }
    if (wide_levels) *wide_levels = t.cl_from_level;
    if (cluster_levels) *cluster_levels = t.ncl;
    if (cluster_blocks) *cluster_blocks = nb;
    return 0;
}

int trmc_topology_blocks(int64_t nseg, const int64_t *up_ptr, const int64_t *up_idx, const uint8_t *boundary,
                         const uint8_t *cost_hint, int cost_tiers, int64_t *plan_pos_of_row, int32_t *rank_of_row,
                         int32_t *block_rows, int32_t *nblocks)
{
    trmc::Topology t;
    std::string err;
    const int rc = trmc::build_topology(nseg, up_ptr, up_idx, boundary, t, err, cost_hint, kFlowBlock, cost_tiers != 0);
    if (rc) return fail(rc == -2 ? TRMC_ECYCLE : TRMC_EINVAL, err);
    for (int64_t r = 0; r < nseg; ++r) {
        if (plan_pos_of_row) plan_pos_of_row[r] = t.pos_of_row[r];
        if (rank_of_row) rank_of_row[r] = t.rank_of_pos[t.pos_of_row[r]];
    }
    if (block_rows) *block_rows = t.block_rows;
    if (nblocks) *nblocks = t.nblocks;
    return 0;
}

int trmc_topology_blocks_general(int64_t nseg, const int64_t *up_ptr, const int64_t *up_idx, const uint8_t *boundary,
                                 int32_t stem_min_rows, int64_t *plan_pos_of_row, int32_t *rank_of_row, int32_t *block_rows,
                                 int32_t *nblocks, int32_t *early_blocks, int32_t early_cap, int32_t *nearly)
{
    trmc::Topology t;
    std::string err;
    const int rc = trmc::build_topology(nseg, up_ptr, up_idx, boundary, t, err, nullptr, kFlowBlock, false, 0, 0, 0,
                                        std::max<int32_t>(0, stem_min_rows));
    if (rc) return fail(rc == -2 ? TRMC_ECYCLE : TRMC_EINVAL, err);
    for (int64_t r = 0; r < nseg; ++r) {
        if (plan_pos_of_row) plan_pos_of_row[r] = t.pos_of_row[r];
        if (rank_of_row) rank_of_row[r] = t.rank_of_pos[t.pos_of_row[r]];
    }
    if (block_rows) *block_rows = t.block_rows;
    if (nblocks) *nblocks = t.nblocks;
    if (nearly) *nearly = (int32_t)t.early_blocks.size();
    for (int32_t i = 0; early_blocks && i < early_cap && i < (int32_t)t.early_blocks.size(); ++i) early_blocks[i] = t.early_blocks[(size_t)i];
    return 0;
}

int trmc_plan_create(int64_t nseg, const int64_t *up_ptr, const int64_t *up_idx, const float *params,
                     const uint8_t *boundary, int precision, int device, trmc_plan **out)
{
    return trmc_plan_create_hinted(nseg, up_ptr, up_idx, params, boundary, nullptr, precision, device, out);
}

int trmc_plan_create_hinted(int64_t nseg, const int64_t *up_ptr, const int64_t *up_idx, const float *params,
                            const uint8_t *boundary, const uint8_t *cost_hint, int precision, int device, trmc_plan **out)
{
    return trmc_plan_create_ex(nseg, up_ptr, up_idx, params, boundary, cost_hint, precision, device, TRMC_ENGINE_AUTO, out);
}

int trmc_plan_engine(const trmc_plan *pl, int32_t *is_flow)
{
    if (!pl || !is_flow) return fail(TRMC_EINVAL, "plan/is_flow is NULL");
    *is_flow = pl->flow ? 1 : 0;
    return 0;
}

int trmc_plan_set_sequence_mode(trmc_plan *pl, int on)
{
    if (!pl) return fail(TRMC_EINVAL, "plan is NULL");
    if (pl->run.active) return fail(TRMC_ESTATE, "a routing window is in progress");
    pl->opt.sequence = on != 0;
    return 0;
}

int trmc_plan_hot_rows(trmc_plan *pl, int64_t *rows_out)
{
    if (!pl || !rows_out) return fail(TRMC_EINVAL, "plan/rows_out is NULL");
    *rows_out = 0;
    if (!pl->hot_cnt.p) return 0;
    if (int rc = use_device(pl)) return rc;
    int32_t n = 0;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(&n, (const int32_t *)pl->hot_cnt.p + 3, sizeof n, hipMemcpyDeviceToHost));
    *rows_out = n;
    return 0;
}

int trmc_plan_set_output_stride(trmc_plan *pl, int32_t stride)
{
    if (!pl) return fail(TRMC_EINVAL, "plan is NULL");
    if (pl->run.active) return fail(TRMC_ESTATE, "a routing window is open");
    if (stride < 0) return fail(TRMC_EINVAL, "stride must be >= 0");
    pl->out_stride = stride;
    return 0;
}

int trmc_plan_set_stamps(trmc_plan *pl, void *host_ring, int32_t nwindows)
{
    if (!pl) return fail(TRMC_EINVAL, "plan is NULL");
    if (pl->run.active) return fail(TRMC_ESTATE, "a routing window is open");
    if (host_ring && nwindows < 1) return fail(TRMC_EINVAL, "nwindows must be >= 1");
    pl->stamps = (unsigned long long *)host_ring;
    pl->nstamp_windows = host_ring ? nwindows : 0;
    pl->stamp_seq = -1;
    return 0;
}

int trmc_plan_arithmetic(const trmc_plan *pl, int32_t *arithmetic)
{
    if (!pl || !arithmetic) return fail(TRMC_EINVAL, "plan/arithmetic is NULL");
    *arithmetic = pl->opt.tol ? TRMC_ARITH_TOLERANCE : TRMC_ARITH_EXACT;
    return 0;
}

int trmc_plan_create_ex(int64_t nseg, const int64_t *up_ptr, const int64_t *up_idx, const float *params,
                        const uint8_t *boundary, const uint8_t *cost_hint, int precision, int device, int flags,
                        trmc_plan **out)
{
    return trmc_plan_create_opt(nseg, up_ptr, up_idx, params, boundary, cost_hint, precision, device, flags, nullptr, out);
}

int trmc_plan_create_opt(int64_t nseg, const int64_t *up_ptr, const int64_t *up_idx, const float *params,
                         const uint8_t *boundary, const uint8_t *cost_hint, int precision, int device, int flags,
                         const trmc_plan_options *options, trmc_plan **out)
{
    if (!out) return fail(TRMC_EINVAL, "out is NULL");
    trmc_plan_options o{};
    if (options) {
        if (options->struct_size < 8 || options->struct_size > (int32_t)sizeof(trmc_plan_options))
            return fail(TRMC_EINVAL, "trmc_plan_options.struct_size is not the size of a trmc_plan_options this library knows");
        std::memcpy(&o, options, (size_t)options->struct_size);
    }
    if (o.arithmetic != TRMC_ARITH_EXACT && o.arithmetic != TRMC_ARITH_TOLERANCE) return fail(TRMC_EINVAL, "bad trmc_plan_options.arithmetic");
    if (o.arithmetic == TRMC_ARITH_TOLERANCE && precision != 32)
        return fail(TRMC_EINVAL, "TRMC_ARITH_TOLERANCE is an arithmetic of precision-32 plans");
    *out = nullptr;
    if (precision != 32 && precision != 64) return fail(TRMC_EINVAL, "precision must be 32 or 64");
    if (nseg > 0 && !params) return fail(TRMC_EINVAL, "params is NULL");
    if ((flags & ~(TRMC_ENGINE_MASK | TRMC_PLAN_SHORT_TS | TRMC_PLAN_FULL_TS)) || (flags & TRMC_ENGINE_MASK) == 3
        || ((flags & TRMC_PLAN_SHORT_TS) && (flags & TRMC_PLAN_FULL_TS)))
        return fail(TRMC_EINVAL, "bad plan flags");
    int engine = flags & TRMC_ENGINE_MASK;
    if (engine == TRMC_ENGINE_FLOW && precision != 32) return fail(TRMC_EINVAL, "the dataflow engine runs precision-32 plans only");
    if (int rc = check_device(device)) return rc;

    trmc_plan *pl = new (std::nothrow) trmc_plan();
    if (!pl) return fail(TRMC_ENOMEM, "out of host memory");
    if (engine == TRMC_ENGINE_AUTO) {
        if (precision != 32)
            engine = TRMC_ENGINE_LEVELS;
        else {
            int64_t nb = 0;
            for (int64_t r = 0; boundary && r < nseg; ++r) nb += boundary[r] == 1;
            engine = ((flags & TRMC_PLAN_SHORT_TS) && nseg - nb >= 1000000) ? TRMC_ENGINE_LEVELS : TRMC_ENGINE_FLOW;
        }
    }
    pl->flow = engine == TRMC_ENGINE_FLOW;
    pl->watchdog_ticks = (uint64_t)(o.flow_watchdog_ms > 0 ? o.flow_watchdog_ms : 30000) * 100000ull;
    {   // the options, resolved once (see trmc.h)
        int ncu = 256;
        if (hipSetDevice(device) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device);
        trmc_plan::Opt &po = pl->opt;
        po.tol = o.arithmetic == TRMC_ARITH_TOLERANCE;
        // (default threshold of the wide tier, until round 5: 384 rows per compute unit -- five levels of the CONUS network -- in exact arithmetic,
        // where the tiles are what the device is busy with and the ramps of the skew cost; 256 -- seven levels -- in tolerance
        // arithmetic, where a tile's arithmetic is a quarter cheaper and the window waits for the tail's chain of launches:
        // 12.8 ms per CONUS day against 14.0, measured on the sequence of bench.py; 192: 13.0, 128: 13.4)
        // Round 5, with the partition of a tile's rows inside the launch (no launch between tiles any more): exact arithmetic
        // 288 rows per compute unit -- six levels of CONUS -- 16.7 ms per day against 17.2 with five, 16.85 with seven or eight
        // (three runs each on one box).
        po.wide_min_rows = o.wide_min_rows < 0 ? 0 : (o.wide_min_rows > 0 ? o.wide_min_rows : (po.tol ? 256L : 288L) * ncu);
        po.wide_levels = (int32_t)std::min<long>(o.wide_levels > 0 ? o.wide_levels : 16, kWideMaxLevels);
        po.wide_k = o.wide_k > 0 ? o.wide_k : 0; // (0: 16, or an eighth of a short window -- route_begin_t)
        po.mid_min_rows = o.mid_min_rows < 0 ? 0 : (o.mid_min_rows > 0 ? o.mid_min_rows : kMidDefaultRowsPerCu * (int64_t)ncu);
        po.mid_levels = (int32_t)std::min<long>(o.mid_levels > 0 ? o.mid_levels : 12, kMidMaxLevels);
        po.mid_k = o.mid_k > 0 ? o.mid_k : 4;
        po.tile_perm_group = o.tile_perm_group < 0 ? 0 : (o.tile_perm_group > 0 ? 1 : -1); // off / on / by the plan (default)
        po.hot_rows = o.hot_rows < 0 ? 0 : (o.hot_rows > 0 ? 1 : -1);
        // (the cluster order is for plans whose windows follow each other as a stream, trmc_stream_*: a single window pays for the
        // clusters' skew with some thirty small launches at its end -- CONUS: 20.7 ms alone against 16.8 with one launch per
        // step -- so a plan only gets it when asked)
        po.cluster_rows = o.cluster_rows > 0 ? std::min<int32_t>(o.cluster_rows, kTileBlock) : 0;
        po.cluster_late_lag = std::max(0, o.cluster_late_lag);
        po.stream_split = std::max(0, o.stream_split);
        po.hot_wave_rows = std::max(0, std::min(64, o.hot_wave_rows));
        po.sequence = o.sequence_mode != 0;
        po.flow_overlap = o.flow_overlap != 0;
        po.flow_lean = o.flow_lean;
        po.flow_debug = o.flow_debug != 0;
    }
    const bool tiers = (flags & TRMC_PLAN_SHORT_TS) != 0;
    std::string err;
    // (plans meant for assume_short_ts on the level engine: rows fed by boundary rows stay below the levels that may be routed
    // several timesteps per launch -- topology.hpp, boundary_floor; wide_levels never asks for more than kWideMaxLevels, and the
    // second tier only takes levels none of whose rows reads a boundary row: route_begin_t)
    // ... and (a hinted short-timestep plan of the level engine) the rows below the levels that can be routed several timesteps
    // per launch are ordered by cost across levels: the same rule picks those levels here as in route_begin_t, which never
    // takes more of them than the plan was ordered for.  tail_sort < 0 keeps the per-level order (A/B).
    int64_t wide_min_rows = 0, mid_min_rows = 0;
    int32_t wide_max_levels = 0, mid_max_levels = 0;
    if (tiers && !pl->flow && cost_hint && o.tail_sort >= 0) {
        wide_min_rows = pl->opt.wide_min_rows;
        wide_max_levels = pl->opt.wide_levels;
        mid_min_rows = pl->opt.mid_min_rows;
        mid_max_levels = pl->opt.mid_levels;
    }
    // ... or, if asked for, laid out in CLUSTERS (topology.hpp, cluster_rows): then they are routed several timesteps per launch
    // as well (k_mc_ctile) -- what a stream of windows needs (trmc_stream_*); the second tier has no part in that
    int32_t cluster_rows = 0;
    if (tiers && !pl->flow && pl->opt.cluster_rows > 0) {
        cluster_rows = pl->opt.cluster_rows;
        wide_min_rows = pl->opt.wide_min_rows;
        wide_max_levels = pl->opt.wide_levels;
        mid_min_rows = 0;
        mid_max_levels = 0;
        pl->opt.mid_min_rows = 0;
    }
    // (a dataflow plan built for the general mode: basins with a main stem of at least stem_min_rows rows -- default 1 024,
    // < 0 = off -- are laid out stem-last with the side tributaries from the top of the stem down, and their stems' blocks take
    // the first tickets: topology.hpp, stem_min_rows)
    int32_t stem_min_rows = 0;
    if (pl->flow && (flags & TRMC_PLAN_FULL_TS)) stem_min_rows = o.stem_min_rows < 0 ? 0 : (o.stem_min_rows > 0 ? o.stem_min_rows : 1024);
    const int trc = trmc::build_topology(nseg, up_ptr, up_idx, boundary, pl->topo, err, cost_hint, pl->flow ? kFlowBlock : 0, tiers,
                                         (tiers && !pl->flow) ? kWideMaxLevels : 0, wide_min_rows, wide_max_levels, stem_min_rows,
                                         mid_min_rows, mid_max_levels, cluster_rows, pl->opt.cluster_late_lag);
    if (trc) {
        delete pl;
        return fail(trc == -2 ? TRMC_ECYCLE : TRMC_EINVAL, err);
    }
    pl->device = device;
    pl->hinted = cost_hint != nullptr && (!pl->flow || tiers);
    pl->precision = precision;
    pl->esz = precision == 32 ? 4 : 8;
    pl->nseg = nseg;
    pl->nseg_pad = (nseg + 63) / 64 * 64;
    if (pl->nseg_pad == 0) pl->nseg_pad = 64;
    pl->nrouted = nseg - pl->topo.nboundary;
    pl->dt_uniform = true;
    pl->dt = nseg ? params[TRMC_P_DT] : 0.0;
    for (int64_t r = 1; r < nseg; ++r)
        if (params[(size_t)r * TRMC_NPARAM + TRMC_P_DT] != params[TRMC_P_DT]) {
            pl->dt_uniform = false;
            break;
        }

    auto bail = [&](int rc) {
        trmc_plan_destroy(pl);
        return rc;
    };
    {
        hipError_t e = hipSetDevice(device);
        // step launches on a high-priority stream, the overlapped transpose on a low-priority one:
        // the transpose should fill gaps, not displace the VALU-bound step kernel
        int prio_lo = 0, prio_hi = 0;
        if (e == hipSuccess) e = hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        if (e == hipSuccess) e = hipStreamCreateWithPriority(&pl->stream, hipStreamNonBlocking, prio_hi);
        if (e == hipSuccess) e = hipStreamCreateWithPriority(&pl->stream2, hipStreamNonBlocking, prio_lo);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&pl->ev_emit, hipEventDisableTiming);
        if (e == hipSuccess && pl->flow) {
            e = hipStreamCreateWithPriority(&pl->fstream, hipStreamNonBlocking, prio_hi);
            for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&pl->ev_chunk[i], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&pl->ev_ctl, hipEventDisableTiming);
        }
        for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipEventCreate(&pl->ev[i]);
        if (e != hipSuccess) return bail(fail(TRMC_EHIP, std::string("stream/event setup: ") + hipGetErrorString(e)));
    }
    int rc = precision == 32 ? upload_params<float>(pl, params) : upload_params<double>(pl, params);
    if (rc) return bail(rc);
    std::vector<int32_t> level_plan((size_t)pl->nseg_pad, 0);
    for (int64_t p = 0; p < nseg; ++p) level_plan[p] = pl->topo.level_of_row[pl->topo.row_of_pos[p]];
    if ((rc = upload_i32(pl->level, level_plan, 1))) return bail(rc);
    if (pl->topo.ncl > 0) {
        std::vector<int32_t> lagk(pl->topo.lagk_of_pos);
        lagk.resize((size_t)pl->nseg_pad, 0);
        if ((rc = upload_i32(pl->lagk, lagk, 1))) return bail(rc);
        if ((rc = upload_i32(pl->cblk_ptr, pl->topo.cblk_ptr, 1))) return bail(rc);
    }
    if ((rc = upload_i32(pl->up_ptr, pl->topo.up_ptr, 1))) return bail(rc);
    if ((rc = upload_i32(pl->up_idx, pl->topo.up_idx, 1))) return bail(rc);
    {
        if (nseg >= (int64_t)1 << 30) return bail(fail(TRMC_EINVAL, "more than 2**30 segments in one plan"));
        // (the step kernels address a column with a 32-bit BYTE offset: position * element size)
        if ((uint64_t)pl->nseg_pad * pl->esz >= (1ull << 32))
            return bail(fail(TRMC_EINVAL, "too many segments for one plan at this precision (column size reaches 4 GiB)"));
        std::vector<int32_t> up2((size_t)pl->nseg_pad * 2, -1);
        for (int64_t p = 0; p < nseg; ++p) {
            const int32_t k0 = pl->topo.up_ptr[p], k1 = pl->topo.up_ptr[p + 1];
            if (k1 - k0 >= 1) up2[2 * p] = pl->topo.up_idx[k0];
            if (k1 - k0 >= 2) up2[2 * p + 1] = pl->topo.up_idx[k0 + 1] | (k1 - k0 > 2 ? 0x40000000 : 0);
        }
        if ((rc = upload_i32(pl->up2, up2, 2))) return bail(rc);
    }
    if (pl->flow && (rc = upload_i32(pl->rank, pl->topo.rank_of_pos, 1))) return bail(rc);
    if (pl->flow) {
        if ((rc = pl->prio.ensure(pl->topo.prio_of_wave.size() + 1))) return bail(rc);
        if (!pl->topo.prio_of_wave.empty()
            && hipMemcpy(pl->prio.p, pl->topo.prio_of_wave.data(), pl->topo.prio_of_wave.size(), hipMemcpyHostToDevice) != hipSuccess)
            return bail(fail(TRMC_EHIP, "uploading the wavefront priorities failed"));
        if ((rc = flow_place_blocks(pl))) return bail(rc);
        // The stems' blocks take the first tickets and wait IN PLACE for their inflows: nothing deadlocks as long as the device
        // holds more workgroups of the kernel at once than there are such blocks -- the others then still find slots.  Asked
        // of the runtime for THIS device (occupancy x compute units; a smaller part, a masked one); a plan whose stems would
        // take more than half of the slots routes in the plain ticket order instead.
        if (!pl->topo.early_blocks.empty()) {
            int per_cu = 0, ncu = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_mc_flow<false, false>, kFlowBlock, 0) != hipSuccess) per_cu = 0;
            (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, pl->device);
            if ((int64_t)pl->topo.early_blocks.size() * 2 > (int64_t)per_cu * ncu) pl->topo.early_blocks.clear();
        }
        if (!pl->topo.early_blocks.empty()) { // ticket -> block: the stems' blocks, then everybody else in order
            const int32_t nb = pl->topo.nblocks;
            std::vector<int32_t> map;
            map.reserve((size_t)nb);
            std::vector<uint8_t> early((size_t)nb, 0);
            for (const int32_t b : pl->topo.early_blocks) {
                early[(size_t)b] = 1;
                map.push_back(b);
            }
            for (int32_t b = 0; b < nb; ++b)
                if (!early[(size_t)b]) map.push_back(b);
            if ((rc = upload_i32(pl->ticket_map, map, 1))) return bail(rc);
        }
    }
    if ((rc = upload_i32(pl->row_of_pos, pl->topo.row_of_pos, 1))) return bail(rc);
    if ((rc = upload_i32(pl->pos_of_row, pl->topo.pos_of_row, 1))) return bail(rc);
    if ((rc = pl->it_prev.ensure((size_t)pl->nseg_pad))) return bail(rc);
    *out = pl;
    return 0;
}

void trmc_plan_destroy(trmc_plan *pl)
{
    if (!pl) return;
    if (pl->clones > 0) {
        // its clones still use its STATIC buffers (topology, parameter columns, placement tables, the lag table): those go with
        // the last of them.  Everything else -- the window buffers, gigabytes of planes and results -- is freed now (hipFree
        // waits for the device); the handle is dead for the caller from here on.
        if (!pl->zombie) {
            (void)hipSetDevice(pl->device);
            for (DevBuf &b : pl->rowsets) b.release();
            pl->rowsets.clear();
            for (DevBuf *b : {&pl->dec, &pl->fetch_hyd, &pl->fetch_q0, &pl->fetch_fvd, &pl->it_prev, &pl->it_sum, &pl->d_state, &pl->ticket, &pl->dbg, &pl->cuq_head, &pl->d_gran,
                              &pl->raw_of_pos, &pl->da_raw, &pl->gage_of_pos, &pl->da_mode, &pl->da_a, &pl->da_w, &pl->da_nudge, &pl->res_of_pos,
                              &pl->res_par, &pl->res_inflow, &pl->in_qlat, &pl->in_q0, &pl->in_bfvd, &pl->qlat_tm, &pl->qlat_alt, &pl->tm, &pl->out, &pl->scratch,
                              &pl->gathered, &pl->cls_last, &pl->hot_list, &pl->hot_cnt})
                b->release();
        }
        pl->zombie = true;
        return;
    }
    trmc_plan *const parent = pl->parent;
    (void)hipSetDevice(pl->device);
    stream_release(pl);
    if (pl->ev_forcing) (void)hipEventDestroy(pl->ev_forcing);
    for (DevBuf &b : pl->rowsets) b.release();
    pl->fetch_hyd.release();
    pl->fetch_q0.release();
    pl->fetch_fvd.release();
    pl->dec.release();
    if (pl->ev_dec) (void)hipEventDestroy(pl->ev_dec);
    if (pl->cstream) (void)hipStreamDestroy(pl->cstream);
    if (pl->hstream) (void)hipStreamDestroy(pl->hstream);
    if (pl->ev_fetch_ready) (void)hipEventDestroy(pl->ev_fetch_ready);
    if (pl->ev_fetch_done) (void)hipEventDestroy(pl->ev_fetch_done);
    if (pl->ev_gather) (void)hipEventDestroy(pl->ev_gather);
    for (DevBuf *b : {&pl->lagk, &pl->cblk_ptr, &pl->params, &pl->up_ptr, &pl->up_idx, &pl->up2, &pl->level, &pl->row_of_pos, &pl->pos_of_row, &pl->it_prev, &pl->it_sum, &pl->lag, &pl->d_state, &pl->ticket, &pl->ticket_map, &pl->rank, &pl->dbg, &pl->prio, &pl->cuq_ptr, &pl->cuq_blk, &pl->cuq_head, &pl->cu_index, &pl->cuq_perm, &pl->d_gran, &pl->raw_of_pos, &pl->da_raw, &pl->gage_of_pos,
                      &pl->da_mode, &pl->da_a, &pl->da_w, &pl->da_nudge, &pl->res_of_pos, &pl->res_par, &pl->res_inflow,
                      &pl->in_qlat, &pl->in_q0, &pl->in_bfvd, &pl->qlat_tm, &pl->qlat_alt, &pl->tm, &pl->out, &pl->scratch, &pl->gathered, &pl->cls_last, &pl->hot_list, &pl->hot_cnt})
        b->release();
    for (auto &e : pl->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto &e : pl->tile_ev)
        if (e) (void)hipEventDestroy(e);
    if (pl->ev_emit) (void)hipEventDestroy(pl->ev_emit);
    if (pl->stream2) (void)hipStreamDestroy(pl->stream2);
    for (hipEvent_t e : pl->ev_chain)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : pl->ev_released)
        if (e) (void)hipEventDestroy(e);
    if (pl->wstream) (void)hipStreamDestroy(pl->wstream);
    for (auto *v : {&pl->wide_t0, &pl->wide_t1})
        for (auto &e : *v)
            if (e) (void)hipEventDestroy(e);
    if (pl->ev_tail) (void)hipEventDestroy(pl->ev_tail);
    if (pl->fstream) (void)hipStreamDestroy(pl->fstream);
    for (auto &e : pl->ev_chunk)
        if (e) (void)hipEventDestroy(e);
    if (pl->ev_ctl) (void)hipEventDestroy(pl->ev_ctl);
    if (pl->stream) (void)hipStreamDestroy(pl->stream);
    delete pl;
    if (parent && --parent->clones == 0 && parent->zombie) {
        parent->zombie = false;
        trmc_plan_destroy(parent);
    }
}

static int final_state_into(trmc_plan *pl, void *dst, int32_t nsteps_of_window = -1);
static int ensure_copy_stream(trmc_plan *pl);

int trmc_plan_clone(trmc_plan *src, trmc_plan **out)
{
    if (!out) return fail(TRMC_EINVAL, "out is NULL");
    *out = nullptr;
    if (!src) return fail(TRMC_EINVAL, "plan is NULL");
    if (src->parent) src = src->parent; // (a clone of a clone shares the same original)
    if (src->zombie) return fail(TRMC_ESTATE, "the plan has been destroyed");
    if (int rc = use_device(src)) return rc;
    trmc_plan *pl = new (std::nothrow) trmc_plan();
    if (!pl) return fail(TRMC_ENOMEM, "out of host memory");
    pl->device = src->device;
    pl->precision = src->precision;
    pl->esz = src->esz;
    pl->topo = src->topo;
    pl->nseg = src->nseg;
    pl->nseg_pad = src->nseg_pad;
    pl->nrouted = src->nrouted;
    pl->dt_uniform = src->dt_uniform;
    pl->dt = src->dt;
    pl->hinted = src->hinted;
    pl->params_sane = src->params_sane;
    pl->flow = src->flow;
    pl->watchdog_ticks = src->watchdog_ticks;
    pl->opt = src->opt;
    pl->ncuq = src->ncuq;
    pl->parent = src;
    ++src->clones;
    auto bail = [&](int rc) {
        trmc_plan_destroy(pl);
        return rc;
    };
    {
        hipError_t e = hipSuccess;
        int prio_lo = 0, prio_hi = 0;
        e = hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        if (e == hipSuccess) e = hipStreamCreateWithPriority(&pl->stream, hipStreamNonBlocking, prio_hi);
        if (e == hipSuccess) e = hipStreamCreateWithPriority(&pl->stream2, hipStreamNonBlocking, prio_lo);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&pl->ev_emit, hipEventDisableTiming);
        if (e == hipSuccess && pl->flow) {
            e = hipStreamCreateWithPriority(&pl->fstream, hipStreamNonBlocking, prio_hi);
            for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&pl->ev_chunk[i], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&pl->ev_ctl, hipEventDisableTiming);
        }
        for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipEventCreate(&pl->ev[i]);
        if (e != hipSuccess) return bail(fail(TRMC_EHIP, std::string("stream/event setup: ") + hipGetErrorString(e)));
    }
    // static data: the original's buffers (plan order, parameters and constants, the dataflow engine's placement tables)
    pl->params.borrow(src->params);
    pl->up_ptr.borrow(src->up_ptr);
    pl->up_idx.borrow(src->up_idx);
    pl->up2.borrow(src->up2);
    pl->level.borrow(src->level);
    pl->lagk.borrow(src->lagk);
    pl->cblk_ptr.borrow(src->cblk_ptr);
    pl->row_of_pos.borrow(src->row_of_pos);
    pl->pos_of_row.borrow(src->pos_of_row);
    pl->rank.borrow(src->rank);
    pl->prio.borrow(src->prio);
    pl->ticket_map.borrow(src->ticket_map);
    pl->cuq_ptr.borrow(src->cuq_ptr);
    pl->cuq_blk.borrow(src->cuq_blk);
    pl->cu_index.borrow(src->cu_index);
    pl->cuq_perm.borrow(src->cuq_perm);
    // the lag of its rows (trmc_plan_set_lag: the trunk of a cut basin riding behind its owner's sub-basins) is part of how
    // the plan routes: the clone routes the same way (row sets and the cost collection are per plan and start empty)
    if (src->maxlag > 0) {
        pl->lag.borrow(src->lag);
        pl->lag_of_row = src->lag_of_row;
        pl->maxlag = src->maxlag;
    }
    if (int rc = pl->it_prev.ensure((size_t)pl->nseg_pad)) return bail(rc);
    if (pl->ncuq > 0)
        if (int rc = pl->cuq_head.ensure((size_t)pl->ncuq * 2 * sizeof(int32_t))) return bail(rc);
    *out = pl;
    return 0;
}

// The next window's forcing on its way to the device WHILE another plan's window (or this plan's copy of results) runs:
// the host-to-device copy is queued on the plan's copy stream and nothing is waited for; the next trmc_route_begin orders
// the window's set-up behind it.  The initial state is not touched: it is what trmc_plan_chain_from hands over, or (q0 =
// NULL semantics of trmc_upload_forcing) the state this plan's last window left.
int trmc_stage_forcing(trmc_plan *pl, int nsteps, const void *qlat, int64_t nq)
{
    if (!pl) return fail(TRMC_EINVAL, "plan is NULL");
    // Also WHILE the plan routes a window that has been queued to its end: the staging area is only read by the window's
    // set-up (the transposing pass at its start), so the next window's forcing may land in it behind that -- a sequence that
    // alternates between a plan and its clone stages a day's forcing two days ahead, right after it has queued the plan's
    // current day, and the copy is through long before the plan's next set-up asks for it.  (Then there is no state of
    // "the plan's last window" yet: trmc_plan_chain_from must supply one.)
    const bool busy = pl->run.active;
    if (busy && pl->run.t_done < pl->run.nsteps + (pl->run.short_ts ? pl->maxlag : 0))
        return fail(TRMC_ESTATE, "the window in progress has not been queued to its end (trmc_route_advance)");
    if (busy && (pl->ngage > 0 || pl->nres > 0))
        return fail(TRMC_ESTATE, "a window with nudging tables or reservoirs is in progress: its series are read after trmc_route_end; "
                                 "stage the next forcing then");
    if (nsteps < 1) return fail(TRMC_EINVAL, "nsteps must be >= 1");
    if (nq < 1) return fail(TRMC_EINVAL, "qlat needs at least one column");
    if (pl->nseg > 0 && !qlat) return fail(TRMC_EINVAL, "qlat is NULL");
    if (int rc = use_device(pl)) return rc;
    if (int rc = ensure_copy_stream(pl)) return rc;
    const size_t bytes = (size_t)pl->nseg * nq * pl->esz;
    if (busy && bytes > pl->in_qlat.bytes)
        return fail(TRMC_ESTATE, "the staging area would have to grow while the window in progress may still read it");
    if (int rc = pl->in_qlat.ensure(bytes)) return rc;
    // the initial state unless trmc_plan_chain_from replaces it: what this plan's last window left (gathered now, on the
    // plan's stream, before anything overwrites the planes); a plan that has routed nothing must be chained to
    if (int rc = pl->in_q0.ensure((size_t)pl->nseg * 3 * pl->esz)) return rc;
    // (a second staging before the window is routed -- a corrected forcing -- finds routed_nsteps < 0 but the state the first
    // one gathered still in in_q0: q0_staged)
    pl->state_missing = busy || (pl->routed_nsteps < 0 && !pl->q0_staged);
    if (pl->nseg > 0 && !pl->state_missing && pl->routed_nsteps >= 0)
        if (int rc = final_state_into(pl, pl->in_q0.p)) return rc;
    pl->q0_staged = !pl->state_missing;
    // (a forcing staged earlier and not routed yet is still on its way on the same stream: the new copy lands behind it)
    // (behind the set-up of the window in progress, which reads the staging area)
    if (busy) HIP_TRY(hipStreamWaitEvent(pl->hstream, pl->ev[1], 0));
    if (pl->nseg > 0) HIP_TRY(hipMemcpyAsync(pl->in_qlat.p, qlat, bytes, hipMemcpyHostToDevice, pl->hstream));
    // Sequence mode, level engine: the transpose into plan order right behind the copy, into the forcing buffer the window in
    // progress does not read -- 0.27 ms of a CONUS day that the next window's set-up, which sits in the tile queue between
    // one day's last tile and the next day's first, no longer has to do
    pl->qlat_alt_ready = false;
    if (pl->opt.sequence && !pl->flow && pl->nseg > 0) {
        if (int rc = pl->qlat_alt.ensure((size_t)nq * pl->nseg_pad * pl->esz)) return rc;
        const int32_t n = (int32_t)pl->nseg;
        const dim3 grid((n + 63) / 64, (unsigned)((nq + 31) / 32));
        if (pl->precision == 32)
            hipLaunchKernelGGL((k_prep_qlat<float>), grid, dim3(kBlock), 0, pl->hstream, (const float *)pl->in_qlat.p,
                               (const int32_t *)pl->row_of_pos.p, (float *)pl->qlat_alt.p, n, pl->nseg_pad, (int32_t)nq);
        else
            hipLaunchKernelGGL((k_prep_qlat<double>), grid, dim3(kBlock), 0, pl->hstream, (const double *)pl->in_qlat.p,
                               (const int32_t *)pl->row_of_pos.p, (double *)pl->qlat_alt.p, n, pl->nseg_pad, (int32_t)nq);
        HIP_TRY(hipGetLastError());
        pl->qlat_alt_ready = true;
    }
    HIP_TRY(hipEventRecord(pl->ev_forcing, pl->hstream));
    pl->forcing_pending = true;
    pl->qlat_direct = false;
    pl->nq = nq;
    // (a plan with boundary rows: their hydrographs of the staged window arrive in ranges while it runs --
    // trmc_set_boundary_flow_range, the multi-GPU hand-off -- or with trmc_set_boundary_flow_device before it begins)
    pl->have_boundary = pl->topo.nboundary == 0;
    pl->ngage = 0; // nudging tables belong to one window (the launches of a window in progress carry their own copy of the pointers)
    pl->nraw = 0;
    pl->staged_nsteps = nsteps;
    if (!busy) pl->routed_nsteps = -1; // (a window in progress sets it when it ends)
    return 0;
}

int trmc_plan_info(const trmc_plan *pl, int64_t *nseg, int64_t *nseg_routed, int32_t *nlevels, int32_t *precision,
                   int32_t *device)
{
    if (!pl) return fail(TRMC_EINVAL, "plan is NULL");
    if (nseg) *nseg = pl->nseg;
    if (nseg_routed) *nseg_routed = pl->nrouted;
    if (nlevels) *nlevels = pl->topo.nlevels;
    if (precision) *precision = pl->precision;
    if (device) *device = pl->device;
    return 0;
}

int trmc_plan_lags(const trmc_plan *pl, int32_t *lag_of_row, int32_t *wide_levels, int32_t *cluster_levels)
{
    if (!pl) return fail(TRMC_EINVAL, "plan is NULL");
    const trmc::Topology &t = pl->topo;
    if (t.cl_rows <= 0) return fail(TRMC_EINVAL, "the plan is not in cluster order (trmc_plan_options.cluster_rows)");
    for (int64_t r = 0; lag_of_row && r < pl->nseg; ++r)
        lag_of_row[r] = t.level_of_row[r] < 0 ? -1 : t.lagk_of_pos[(size_t)t.pos_of_row[r]];
    if (wide_levels) *wide_levels = t.cl_from_level;
    if (cluster_levels) *cluster_levels = t.ncl;
    return 0;
}

int trmc_plan_levels(const trmc_plan *pl, int32_t *level_of_row, int64_t *plan_pos_of_row)
{
    if (!pl) return fail(TRMC_EINVAL, "plan is NULL");
    for (int64_t r = 0; r < pl->nseg; ++r) {
        if (level_of_row) level_of_row[r] = pl->topo.level_of_row[r];
        if (plan_pos_of_row) plan_pos_of_row[r] = pl->topo.pos_of_row[r];
    }
    return 0;
}

// (q_T, q_T, depth_T) of the last routed window, row order, into device memory `dst` [nseg][3]; queued on the plan stream
static int final_state_into(trmc_plan *pl, void *dst, int32_t nsteps_of_window)
{
    const int32_t n = (int32_t)pl->nseg, T_ = nsteps_of_window >= 0 ? nsteps_of_window : pl->routed_nsteps;
    const size_t plane = (size_t)(T_ + 1) * pl->nseg_pad;
    const int32_t *rop = (const int32_t *)pl->row_of_pos.p;
    if (pl->flow) {
        hipLaunchKernelGGL((k_final_state<float>), dim3(blocks_for(n)), dim3(kBlock), 0, pl->stream, (const float *)pl->tm.p,
                           (const float *)pl->d_state.p, rop, (float *)dst, n, pl->nseg_pad, T_, 2, 0);
    } else if (pl->precision == 32) {
        const float *q = (const float *)pl->tm.p;
        hipLaunchKernelGGL((k_final_state<float>), dim3(blocks_for(n)), dim3(kBlock), 0, pl->stream, q, q + 2 * plane, rop,
                           (float *)dst, n, pl->nseg_pad, T_, 1, T_);
    } else {
        const double *q = (const double *)pl->tm.p;
        hipLaunchKernelGGL((k_final_state<double>), dim3(blocks_for(n)), dim3(kBlock), 0, pl->stream, q, q + 2 * plane, rop,
                           (double *)dst, n, pl->nseg_pad, T_, 1, T_);
    }
    HIP_TRY(hipGetLastError());
    return note_gather(pl, nsteps_of_window >= 0);
}

// initial state (or warm start), boundary hydrographs, and the bookkeeping common to every forcing upload
static int stage_state(trmc_plan *pl, int nsteps, int64_t nq, const void *q0, const void *boundary_fvd)
{
    const size_t e = pl->esz;
    if (int rc = pl->in_q0.ensure((size_t)pl->nseg * 3 * e)) return rc;
    if (pl->nseg > 0) {
        if (q0) {
            HIP_TRY(hipMemcpyAsync(pl->in_q0.p, q0, (size_t)pl->nseg * 3 * e, hipMemcpyHostToDevice, pl->stream));
        } else if (pl->routed_nsteps >= 0) { // warm start in HBM: (q_T, q_T, depth_T) of the previous window, AbstractNetwork.py:182-190
            if (int rc = final_state_into(pl, pl->in_q0.p)) return rc;
        } // (else: q0_staged -- the state an earlier staging of this window put into in_q0 stands, upload_check)
    }
    if (pl->topo.nboundary > 0 && boundary_fvd) {
        const size_t b = (size_t)pl->topo.nboundary * nsteps * 3 * e;
        if (int rc = pl->in_bfvd.ensure(b)) return rc;
        HIP_TRY(hipMemcpyAsync(pl->in_bfvd.p, boundary_fvd, b, hipMemcpyHostToDevice, pl->stream));
    }
    HIP_TRY(hipStreamSynchronize(pl->stream));
    pl->nq = nq;
    pl->have_boundary = pl->topo.nboundary == 0 || boundary_fvd != nullptr;
    pl->ngage = 0; // nudging tables belong to one window: trmc_set_nudging() after each upload
    pl->nraw = 0;
    pl->staged_nsteps = nsteps;
    pl->routed_nsteps = -1;
    pl->state_missing = false;
    pl->q0_staged = true;
    return 0;
}

static int upload_check(trmc_plan *pl, int nsteps, int64_t nq, const void *q0)
{
    if (!pl) return fail(TRMC_EINVAL, "plan is NULL");
    if (pl->run.active) return fail(TRMC_ESTATE, "a routing window is in progress");
    if (stream_active(pl)) return fail(TRMC_ESTATE, "a stream of windows is in progress (trmc_stream_end it first)");
    if (nsteps < 1) return fail(TRMC_EINVAL, "nsteps must be >= 1");
    if (nq < 1) return fail(TRMC_EINVAL, "qlat needs at least one column");
    if (pl->nseg > 0 && !q0 && pl->routed_nsteps < 0 && !pl->q0_staged)
        return fail(TRMC_ESTATE, "q0 is NULL (continue from the resident state) but nothing has been routed yet");
    if (int rc = use_device(pl)) return rc;
    if (pl->forcing_pending) {
        // a forcing staged with trmc_stage_forcing is (or may still be) on its way into in_qlat on the copy stream: this upload
        // replaces it, and its own copy must not be overtaken by the older one
        HIP_TRY(hipEventSynchronize(pl->ev_forcing));
        pl->forcing_pending = false;
    }
    return 0;
}

int trmc_upload_forcing(trmc_plan *pl, int nsteps, const void *qlat, int64_t nq, const void *q0,
                        const void *boundary_fvd)
{
    if (int rc = upload_check(pl, nsteps, nq, q0)) return rc;
    if (pl->nseg > 0 && !qlat) return fail(TRMC_EINVAL, "qlat is NULL");
    // boundary_fvd may be NULL here when trmc_set_boundary_flow_device() supplies the hydrographs later
    const size_t e = pl->esz;
    if (int rc = pl->in_qlat.ensure((size_t)pl->nseg * nq * e)) return rc;
    if (pl->nseg > 0)
        HIP_TRY(hipMemcpyAsync(pl->in_qlat.p, qlat, (size_t)pl->nseg * nq * e, hipMemcpyHostToDevice, pl->stream));
    pl->qlat_direct = false;
    pl->qlat_alt_ready = false; // (a staged forcing, transposed already, is replaced by this one)
    return stage_state(pl, nsteps, nq, q0, boundary_fvd);
}

int trmc_upload_forcing_packed(trmc_plan *pl, int nsteps, int64_t nq, int64_t nfeat, const int32_t *raw_a,
                               const int32_t *raw_b, const double *pack_a, const double *pack_b,
                               const int64_t *feat_of_row, const void *q0, const void *boundary_fvd)
{
    if (int rc = upload_check(pl, nsteps, nq, q0)) return rc;
    if (nfeat < 0 || (pl->nseg > 0 && (!raw_a || !pack_a || !feat_of_row))) return fail(TRMC_EINVAL, "raw_a/pack_a/feat_of_row is NULL");
    if (raw_b && !pack_b) return fail(TRMC_EINVAL, "pack_b is NULL");
    if (nfeat > INT32_MAX) return fail(TRMC_EINVAL, "feature axis too long");
    auto spec = [](const double *k) {
        PackSpec s;
        s.scale = k[0];
        s.offset = k[1];
        s.use1 = k[2] == k[2];
        s.use2 = k[3] == k[3];
        s.fill1 = s.use1 ? (int32_t)k[2] : 0;
        s.fill2 = s.use2 ? (int32_t)k[3] : 0;
        s.vmin = k[4] == k[4] ? (int32_t)k[4] : INT32_MIN; // NaN: no lower / upper bound in the file
        s.vmax = k[5] == k[5] ? (int32_t)k[5] : INT32_MAX;
        return s;
    };
    const PackSpec ka = spec(pack_a), kb = raw_b ? spec(pack_b) : PackSpec{1.0, 0.0, 0, 0, INT32_MIN, INT32_MAX, 0, 0};
    const int64_t n = pl->nseg;
    std::vector<int32_t> feat_of_pos((size_t)(n > 0 ? n : 1), -1);
    for (int64_t p = 0; p < n; ++p) {
        const int64_t f = feat_of_row[pl->topo.row_of_pos[p]];
        if (f >= nfeat) return fail(TRMC_EINVAL, "feat_of_row entry outside the feature axis");
        feat_of_pos[(size_t)p] = f < 0 ? -1 : (int32_t)f;
    }
    const size_t raw_bytes = (size_t)nq * (size_t)nfeat * sizeof(int32_t);
    const size_t pbytes = ((size_t)(n > 0 ? n : 1) * sizeof(int32_t) + 255) / 256 * 256;
    if (int rc = pl->in_qlat.ensure((raw_b ? 2 : 1) * raw_bytes)) return rc; // staging for the raw columns
    if (int rc = pl->scratch.ensure(pbytes)) return rc;
    if (int rc = pl->qlat_tm.ensure((size_t)nq * pl->nseg_pad * pl->esz)) return rc;
    if (n > 0) {
        int32_t *da = (int32_t *)pl->in_qlat.p, *db = raw_b ? da + (size_t)nq * nfeat : nullptr;
        HIP_TRY(hipMemcpyAsync(da, raw_a, raw_bytes, hipMemcpyHostToDevice, pl->stream));
        if (raw_b) HIP_TRY(hipMemcpyAsync(db, raw_b, raw_bytes, hipMemcpyHostToDevice, pl->stream));
        HIP_TRY(hipMemcpyAsync(pl->scratch.p, feat_of_pos.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, pl->stream));
        if (pl->precision == 32)
            hipLaunchKernelGGL((k_ingest_packed<float>), dim3(blocks_for(n)), dim3(kBlock), 0, pl->stream, da, db, ka, kb,
                               (const int32_t *)pl->scratch.p, (float *)pl->qlat_tm.p, (int32_t)n, pl->nseg_pad, (int32_t)nq, nfeat);
        else
            hipLaunchKernelGGL((k_ingest_packed<double>), dim3(blocks_for(n)), dim3(kBlock), 0, pl->stream, da, db, ka, kb,
                               (const int32_t *)pl->scratch.p, (double *)pl->qlat_tm.p, (int32_t)n, pl->nseg_pad, (int32_t)nq, nfeat);
        HIP_TRY(hipGetLastError());
    }
    pl->qlat_direct = true;
    pl->qlat_alt_ready = false;
    // (feat_of_pos is pageable host memory: the copy above must have been consumed before it goes out of scope;
    // stage_state ends with a stream synchronisation)
    return stage_state(pl, nsteps, nq, q0, boundary_fvd);
}

int trmc_set_boundary_flow_device(trmc_plan *pl, int nsteps, const void *q_dev)
{
    if (!pl) return fail(TRMC_EINVAL, "plan is NULL");
    if (pl->staged_nsteps < 0) return fail(TRMC_ESTATE, "trmc_upload_forcing must precede trmc_set_boundary_flow_device");
    if (nsteps != pl->staged_nsteps) return fail(TRMC_EINVAL, "nsteps differs from the staged forcing");
    const int64_t nb = pl->topo.nboundary;
    if (nb == 0) return 0;
    if (!q_dev) return fail(TRMC_EINVAL, "q_dev is NULL");
    if (int rc = use_device(pl)) return rc;
    const size_t e = pl->esz;
    if (int rc = pl->in_bfvd.ensure((size_t)nb * nsteps * 3 * e)) return rc;
    // expand [b][t] -> [b][t][q,0,0] on the device (velocity and depth of a boundary row are never read)
    HIP_TRY(hipMemsetAsync(pl->in_bfvd.p, 0, (size_t)nb * nsteps * 3 * e, pl->stream));
    HIP_TRY(hipMemcpy2DAsync(pl->in_bfvd.p, 3 * e, q_dev, e, e, (size_t)nb * nsteps, hipMemcpyDeviceToDevice, pl->stream));
    HIP_TRY(hipStreamSynchronize(pl->stream));
    pl->have_boundary = true;
    pl->routed_nsteps = -1;
    return 0;
}

int trmc_set_reservoirs(trmc_plan *pl, int64_t nres, const int64_t *res_rows, const void *par, double routing_period)
{
    if (!pl) return fail(TRMC_EINVAL, "plan is NULL");
    if (nres < 0) return fail(TRMC_EINVAL, "nres < 0");
    pl->nres = 0;
    if (nres == 0) return 0;
    if (!res_rows || !par) return fail(TRMC_EINVAL, "res_rows/par is NULL");
    if (int rc = use_device(pl)) return rc;
    std::vector<int32_t> r_of_pos((size_t)pl->nseg_pad, -1);
    for (int64_t i = 0; i < nres; ++i) {
        const int64_t r = res_rows[i];
        if (r < 0 || r >= pl->nseg) return fail(TRMC_EINVAL, "reservoir row out of range");
        if (pl->topo.level_of_row[r] < 0) return fail(TRMC_EINVAL, "reservoir on a boundary row");
        if (r_of_pos[pl->topo.pos_of_row[r]] >= 0) return fail(TRMC_EINVAL, "two reservoirs on one row");
        r_of_pos[pl->topo.pos_of_row[r]] = (int32_t)i;
    }
    if (int rc = upload_i32(pl->res_of_pos, r_of_pos, 1)) return rc;
    const size_t bytes = (size_t)nres * 9 * pl->esz;
    if (int rc = pl->res_par.ensure(bytes)) return rc;
    HIP_TRY(hipMemcpy(pl->res_par.p, par, bytes, hipMemcpyHostToDevice));
    pl->nres = nres;
    pl->res_dt = routing_period;
    pl->routed_nsteps = -1;
    return 0;
}

int trmc_download_reservoir_inflow(trmc_plan *pl, void *inflow_out)
{
    if (!pl) return fail(TRMC_EINVAL, "plan is NULL");
    if (pl->routed_nsteps < 0) return fail(TRMC_ESTATE, "nothing routed yet");
    if (pl->nres == 0) return 0;
    if (!inflow_out) return fail(TRMC_EINVAL, "inflow_out is NULL");
    if (int rc = use_device(pl)) return rc;
    HIP_TRY(hipMemcpy(inflow_out, pl->res_inflow.p, (size_t)pl->nres * pl->routed_nsteps * pl->esz, hipMemcpyDeviceToHost));
    return 0;
}

int trmc_set_nudging(trmc_plan *pl, int nsteps, int64_t ngage, const int64_t *gage_rows, const uint8_t *mode,
                     const void *a, const void *w)
{
    if (!pl) return fail(TRMC_EINVAL, "plan is NULL");
    if (pl->staged_nsteps < 0) return fail(TRMC_ESTATE, "trmc_upload_forcing must precede trmc_set_nudging");
    if (ngage < 0 || nsteps < 1) return fail(TRMC_EINVAL, "bad ngage/nsteps");
    pl->ngage = 0;
    if (ngage == 0) return 0;
    if (!gage_rows || !mode || !a || !w) return fail(TRMC_EINVAL, "nudging table pointer is NULL");
    if (int rc = use_device(pl)) return rc;
    std::vector<int32_t> g_of_pos((size_t)pl->nseg_pad, -1);
    std::vector<int32_t> res_of_pos;
    if (pl->nres > 0) {
        res_of_pos.resize((size_t)pl->nseg_pad);
        HIP_TRY(hipMemcpy(res_of_pos.data(), pl->res_of_pos.p, (size_t)pl->nseg_pad * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    for (int64_t g = 0; g < ngage; ++g) {
        const int64_t r = gage_rows[g];
        if (r < 0 || r >= pl->nseg) return fail(TRMC_EINVAL, "gage row out of range");
        if (pl->topo.level_of_row[r] < 0) return fail(TRMC_EINVAL, "gage on a boundary row");
        // (the reservoir branch of the kernels ends a row's step before the nudging hook)
        if (!res_of_pos.empty() && res_of_pos[(size_t)pl->topo.pos_of_row[r]] >= 0)
            return fail(TRMC_EINVAL, "a gage on a reservoir row is not supported (row " + std::to_string(r) + ")");
        g_of_pos[pl->topo.pos_of_row[r]] = (int32_t)g; // one gage per segment: the last listed wins, as reach_has_gage does
    }
    const size_t n = (size_t)ngage * nsteps, e = pl->esz;
    if (int rc = upload_i32(pl->gage_of_pos, g_of_pos, 1)) return rc;
    if (int rc = pl->da_mode.ensure(n)) return rc;
    if (int rc = pl->da_a.ensure(n * e)) return rc;
    if (int rc = pl->da_w.ensure(n * e)) return rc;
    if (int rc = pl->da_nudge.ensure(n * e)) return rc;
    HIP_TRY(hipMemcpy(pl->da_mode.p, mode, n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(pl->da_a.p, a, n * e, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(pl->da_w.p, w, n * e, hipMemcpyHostToDevice));
    HIP_TRY(hipMemsetAsync(pl->da_nudge.p, 0, n * e, pl->stream)); // (on the plan's stream: ordered before its kernels)
    pl->ngage = ngage;
    pl->nraw = 0;
    pl->da_nsteps = nsteps;
    pl->routed_nsteps = -1;
    return 0;
}

int trmc_set_nudging_successors(trmc_plan *pl, int64_t ngage, const int64_t *successor_rows)
{
    if (!pl) return fail(TRMC_EINVAL, "plan is NULL");
    if (pl->ngage == 0 || ngage != pl->ngage) return fail(TRMC_ESTATE, "trmc_set_nudging (same ngage) must precede trmc_set_nudging_successors");
    if (!successor_rows) return fail(TRMC_EINVAL, "successor_rows is NULL");
    if (pl->run.active) return fail(TRMC_ESTATE, "a routing window is in progress");
    pl->nraw = 0;
    int64_t n = 0;
    for (int64_t g = 0; g < ngage; ++g) n += successor_rows[g] >= 0;
    if (n == 0) return 0;
    if (pl->flow)
        return fail(TRMC_ESTATE, "gages inside a reach without assume_short_ts need the level engine (create the plan with "
                                       "TRMC_ENGINE_LEVELS)");
    if (int rc = use_device(pl)) return rc;
    std::vector<int32_t> g_of_pos((size_t)pl->nseg_pad, -1), raw((size_t)pl->nseg_pad, -1);
    HIP_TRY(hipMemcpy(g_of_pos.data(), pl->gage_of_pos.p, (size_t)pl->nseg_pad * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (int64_t g = 0; g < ngage; ++g) {
        const int64_t r = successor_rows[g];
        if (r < 0) continue;
        if (r >= pl->nseg || pl->topo.level_of_row[r] < 0) return fail(TRMC_EINVAL, "successor row out of range or a boundary row");
        const int32_t p = pl->topo.pos_of_row[r];
        // the segment below a gage inside a reach has exactly one upstream row: that gage's segment
        if (pl->topo.up_ptr[p + 1] - pl->topo.up_ptr[p] != 1 || g_of_pos[(size_t)pl->topo.up_idx[pl->topo.up_ptr[p]]] != (int32_t)g)
            return fail(TRMC_EINVAL, "successor row " + std::to_string(r) + " is not the segment directly below gage " + std::to_string(g));
        raw[(size_t)p] = (int32_t)g;
    }
    if (int rc = upload_i32(pl->raw_of_pos, raw, 1)) return rc;
    if (int rc = pl->da_raw.ensure((size_t)ngage * pl->da_nsteps * pl->esz)) return rc;
    HIP_TRY(hipMemsetAsync(pl->da_raw.p, 0, (size_t)ngage * pl->da_nsteps * pl->esz, pl->stream));
    pl->nraw = n;
    return 0;
}

int trmc_download_nudge(trmc_plan *pl, void *nudge_out)
{
    if (!pl) return fail(TRMC_EINVAL, "plan is NULL");
    if (pl->routed_nsteps < 0) return fail(TRMC_ESTATE, "nothing routed yet");
    if (pl->ngage == 0) return 0;
    if (!nudge_out) return fail(TRMC_EINVAL, "nudge_out is NULL");
    if (int rc = use_device(pl)) return rc;
    HIP_TRY(hipMemcpy(nudge_out, pl->da_nudge.p, (size_t)pl->ngage * pl->da_nsteps * pl->esz, hipMemcpyDeviceToHost));
    return 0;
}

static int route_check(trmc_plan *pl, int nsteps, int qts_subdivisions, bool boundary_later)
{
    if (!pl) return fail(TRMC_EINVAL, "plan is NULL");
    if (pl->staged_nsteps < 0) return fail(TRMC_ESTATE, "trmc_upload_forcing must precede routing");
    if (pl->run.active) return fail(TRMC_ESTATE, "a routing window is already in progress (trmc_route_end it first)");
    if (stream_active(pl)) return fail(TRMC_ESTATE, "a stream of windows is in progress (trmc_stream_end it first)");
    if (nsteps < 1) return fail(TRMC_EINVAL, "nsteps must be >= 1");
    if (qts_subdivisions < 1) return fail(TRMC_EINVAL, "qts_subdivisions must be >= 1");
    if (pl->topo.nboundary > 0 && nsteps != pl->staged_nsteps)
        return fail(TRMC_EINVAL, "nsteps differs from the staged boundary hydrographs");
    if (!pl->have_boundary && !boundary_later)
        return fail(TRMC_ESTATE, "plan has boundary rows but no boundary hydrographs were supplied");
    if (pl->ngage > 0 && pl->da_nsteps != nsteps) return fail(TRMC_EINVAL, "nudging tables were set for a different nsteps");
    if (pl->state_missing && !pl->chain_staged)
        return fail(TRMC_ESTATE, "the forcing was staged on a plan that has routed nothing (or whose window has not ended yet): "
                                 "trmc_plan_chain_from must hand it a state, or trmc_route_end the window first");
    // the reference's precondition, mc_reach.pyx:246-247
    if ((int64_t)(nsteps - 1) / qts_subdivisions >= pl->nq)
        return fail(TRMC_EINVAL, "Number of columns (timesteps) in Qlat is incorrect: need "
                                     + std::to_string((nsteps - 1) / qts_subdivisions + 1) + ", got " + std::to_string(pl->nq));
    return use_device(pl);
}

static int lag_check(trmc_plan *pl, int assume_short_ts)
{
    if ((pl->topo.tail_from_level > 0 || pl->topo.ncl > 0) && !assume_short_ts)
        return fail(TRMC_EINVAL, "this plan was created for assume_short_ts (TRMC_PLAN_SHORT_TS on the level engine: its deeper rows are "
                                 "ordered in clusters or by cost, not by level); create a plan for the general mode");
    if (pl->maxlag > 0 && !assume_short_ts)
        return fail(TRMC_EINVAL, "a plan with lagged rows (trmc_plan_set_lag) routes with assume_short_ts only");
    return 0;
}

// A forcing staged WHILE the plan's own window was in flight (trmc_stage_forcing on a busy plan) had no state to take then; if
// that window has ended since and nobody has handed a state over (trmc_plan_chain_from), the window continues from it: the
// gather of (q_T, q_T, depth_T) that an idle staging does at once is done here, at the head of the new window's queue.
static int settle_deferred_state(trmc_plan *pl)
{
    if (!pl || !pl->state_missing || pl->chain_staged || pl->run.active || pl->routed_nsteps < 0) return 0;
    if (int rc = use_device(pl)) return rc;
    if (pl->nseg > 0)
        if (int rc = final_state_into(pl, pl->in_q0.p)) return rc;
    pl->state_missing = false;
    pl->q0_staged = true;
    return 0;
}

int trmc_route_device(trmc_plan *pl, int nsteps, int qts_subdivisions, int assume_short_ts)
{
    if (int rc = settle_deferred_state(pl)) return rc;
    if (int rc = route_check(pl, nsteps, qts_subdivisions, false)) return rc;
    if (int rc = lag_check(pl, assume_short_ts)) return rc;
    const bool f = pl->precision == 32;
    const int t_last = nsteps + (assume_short_ts ? pl->maxlag : 0);
    int rc;
    if (pl->flow) {
        rc = flow_route_begin(pl, nsteps, qts_subdivisions, assume_short_ts);
        if (!rc) rc = flow_route_advance(pl, t_last);
        if (!rc) rc = flow_route_end(pl);
    } else {
        rc = f ? route_begin_t<float>(pl, nsteps, qts_subdivisions, assume_short_ts)
               : route_begin_t<double>(pl, nsteps, qts_subdivisions, assume_short_ts);
        if (!rc) rc = f ? route_advance_t<float>(pl, t_last) : route_advance_t<double>(pl, t_last);
        if (!rc) rc = f ? route_end_t<float>(pl) : route_end_t<double>(pl);
    }
    if (rc) pl->run.active = false;
    return rc;
}

int trmc_route_begin(trmc_plan *pl, int nsteps, int qts_subdivisions, int assume_short_ts)
{
    if (int rc = settle_deferred_state(pl)) return rc;
    if (int rc = route_check(pl, nsteps, qts_subdivisions, true)) return rc;
    if (int rc = lag_check(pl, assume_short_ts)) return rc;
    const int rc = pl->flow ? flow_route_begin(pl, nsteps, qts_subdivisions, assume_short_ts)
                   : pl->precision == 32 ? route_begin_t<float>(pl, nsteps, qts_subdivisions, assume_short_ts)
                                         : route_begin_t<double>(pl, nsteps, qts_subdivisions, assume_short_ts);
    if (rc) pl->run.active = false;
    return rc;
}

int trmc_route_advance(trmc_plan *pl, int t_end)
{
    if (!pl) return fail(TRMC_EINVAL, "plan is NULL");
    if (!pl->run.active) return fail(TRMC_ESTATE, "trmc_route_begin must precede trmc_route_advance");
    const int32_t lagmax = pl->run.short_ts ? pl->maxlag : 0;
    if (t_end < pl->run.t_done || t_end > pl->run.nsteps + lagmax)
        return fail(TRMC_EINVAL, "t_end outside [launches done, nsteps + lag]");
    // boundary rows feed the lagged rows when the plan has a lag (trmc_plan_set_lag), all rows otherwise
    const int32_t need = t_end - lagmax < pl->run.nsteps ? t_end - lagmax : pl->run.nsteps;
    if (need > pl->run.boundary_through)
        return fail(TRMC_ESTATE, "boundary hydrographs are staged through step " + std::to_string(pl->run.boundary_through)
                                     + " only (trmc_set_boundary_flow_range)");
    if (t_end == pl->run.t_done) return 0;
    if (int rc = use_device(pl)) return rc;
    if (pl->flow) return flow_route_advance(pl, t_end);
    return pl->precision == 32 ? route_advance_t<float>(pl, t_end) : route_advance_t<double>(pl, t_end);
}

int trmc_route_end(trmc_plan *pl)
{
    if (!pl) return fail(TRMC_EINVAL, "plan is NULL");
    if (!pl->run.active) return fail(TRMC_ESTATE, "no routing window in progress");
    if (pl->run.t_done != pl->run.nsteps + (pl->run.short_ts ? pl->maxlag : 0)) {
        // abandon the window: drain the queue so the plan can be reused
        (void)hipStreamSynchronize(pl->stream);
        (void)hipStreamSynchronize(pl->stream2);
        if (pl->fstream) (void)hipStreamSynchronize(pl->fstream);
        pl->run.active = false;
        return fail(TRMC_ESTATE, "trmc_route_end before every timestep was queued; window abandoned");
    }
    if (int rc = use_device(pl)) return rc;
    const int rc = pl->flow ? flow_route_end(pl) : pl->precision == 32 ? route_end_t<float>(pl) : route_end_t<double>(pl);
    if (rc) pl->run.active = false;
    return rc;
}

int trmc_plan_stream(trmc_plan *pl, void **stream_out)
{
    if (!pl || !stream_out) return fail(TRMC_EINVAL, "plan/stream_out is NULL");
    // inside a window of the dataflow engine whose launches alternate between two compute streams: the stream the NEXT
    // trmc_route_advance uses (afterwards: the one it used, on which trmc_gather_flow_range is queued as well)
    *stream_out = (void *)((pl->flow && pl->run.active && flow_overlap(pl)) ? flow_stream(pl, pl->flow_next) : pl->stream);
    return 0;
}

int trmc_plan_set_lag(trmc_plan *pl, const int32_t *lag_of_row)
{
    if (!pl) return fail(TRMC_EINVAL, "plan is NULL");
    if (pl->run.active) return fail(TRMC_ESTATE, "a routing window is in progress");
    if (!pl->rowsets.empty()) return fail(TRMC_ESTATE, "set the lag before registering row sets");
    if (pl->clones > 0 || pl->parent) return fail(TRMC_ESTATE, "set the lag before cloning the plan (a clone shares the lag table)");
    pl->maxlag = 0;
    pl->lag_of_row.clear();
    pl->wide_safe_pos = -1;
    if (!lag_of_row) return 0;
    int32_t mx = 0;
    for (int64_t r = 0; r < pl->nseg; ++r) {
        if (lag_of_row[r] < 0) return fail(TRMC_EINVAL, "negative lag");
        mx = std::max(mx, lag_of_row[r]);
    }
    if (mx == 0) return 0;
    // two classes only: in step, or `mx` launches behind; an upstream neighbour is never behind its consumer
    std::vector<int32_t> by_pos((size_t)pl->nseg_pad, 0);
    for (int64_t r = 0; r < pl->nseg; ++r) {
        if (lag_of_row[r] != 0 && lag_of_row[r] != mx) return fail(TRMC_EINVAL, "lag must be 0 or one common value");
        by_pos[(size_t)pl->topo.pos_of_row[r]] = lag_of_row[r];
    }
    for (int64_t p = pl->topo.nboundary; p < pl->nseg; ++p)
        for (int32_t k = pl->topo.up_ptr[p]; k < pl->topo.up_ptr[p + 1]; ++k) {
            const int32_t u = pl->topo.up_idx[k];
            if (u >= pl->topo.nboundary && by_pos[(size_t)u] > by_pos[(size_t)p])
                return fail(TRMC_EINVAL, "a lagged row feeds a row that is not lagged");
            if (u < pl->topo.nboundary && by_pos[(size_t)p] != mx)
                return fail(TRMC_EINVAL, "with a lag, boundary rows may only feed lagged rows");
        }
    if (int rc = use_device(pl)) return rc;
    if (int rc = pl->lag.ensure((size_t)pl->nseg_pad * sizeof(int32_t))) return rc;
    HIP_TRY(hipMemcpy(pl->lag.p, by_pos.data(), (size_t)pl->nseg_pad * sizeof(int32_t), hipMemcpyHostToDevice));
    pl->lag_of_row.assign(lag_of_row, lag_of_row + pl->nseg);
    pl->maxlag = mx;
    return 0;
}

int trmc_rowset_create(trmc_plan *pl, const int64_t *rows, int64_t nrows, int32_t *id_out)
{
    if (!pl || !id_out) return fail(TRMC_EINVAL, "plan/id_out is NULL");
    if (nrows < 0 || (nrows > 0 && !rows)) return fail(TRMC_EINVAL, "rows is NULL");
    if (int rc = use_device(pl)) return rc;
    std::vector<int32_t> pos((size_t)nrows);
    for (int64_t i = 0; i < nrows; ++i) {
        if (rows[i] < 0 || rows[i] >= pl->nseg) return fail(TRMC_EINVAL, "row out of range");
        pos[i] = pl->topo.pos_of_row[rows[i]];
    }
    DevBuf b;
    if (int rc = b.ensure((size_t)(nrows > 0 ? nrows : 1) * sizeof(int32_t))) return rc;
    if (nrows > 0) HIP_TRY(hipMemcpy(b.p, pos.data(), (size_t)nrows * sizeof(int32_t), hipMemcpyHostToDevice));
    int32_t rs_lag = 0; // a set with lagged rows is complete `maxlag` launches later
    if (pl->maxlag > 0)
        for (int64_t i = 0; i < nrows; ++i) rs_lag = std::max(rs_lag, pl->lag_of_row[(size_t)rows[i]]);
    pl->rowsets.push_back(b);
    pl->rowset_n.push_back(nrows);
    pl->rowset_lag.push_back(rs_lag);
    int32_t lk = 0;
    if (!pl->topo.lagk_of_pos.empty())
        for (int64_t i = 0; i < nrows; ++i) lk = std::max(lk, pl->topo.lagk_of_pos[(size_t)pos[i]]);
    pl->rowset_lagk.push_back(lk);
    *id_out = (int32_t)pl->rowsets.size() - 1;
    return 0;
}

int trmc_gather_flow_range(trmc_plan *pl, int32_t rowset, int t_begin, int t_end, void *dst_dev, int64_t dst_stride)
{
    if (!pl) return fail(TRMC_EINVAL, "plan is NULL");
    if (rowset < 0 || rowset >= (int32_t)pl->rowsets.size()) return fail(TRMC_EINVAL, "unknown row set");
    const int32_t through = pl->run.active ? std::min(pl->run.nsteps, pl->run.t_done - (pl->run.short_ts ? pl->rowset_lag[rowset] : 0))
                                           : pl->routed_nsteps;
    if (t_begin < 0 || t_end < t_begin || t_end > through)
        return fail(TRMC_ESTATE, "steps (t_begin, t_end] are not all routed yet");
    const int64_t nrows = pl->rowset_n[rowset];
    if (nrows == 0 || t_end == t_begin) return 0;
    if (!dst_dev || dst_stride < t_end - t_begin) return fail(TRMC_EINVAL, "dst_dev is NULL or dst_stride too small");
    if (int rc = use_device(pl)) return rc;
    const int64_t work = nrows * (t_end - t_begin);
    hipStream_t gst = pl->stream;
    if (pl->flow && pl->run.active && flow_overlap(pl) && pl->run.launches > 0) {
        // on the stream of the last launch, and behind the last launch of the other one (which holds the steps before)
        gst = flow_stream(pl, pl->flow_last);
        if (pl->run.launches > 1) HIP_TRY(hipStreamWaitEvent(gst, pl->ev_chunk[1 - pl->flow_last], 0));
    }
    if (pl->precision == 32)
        hipLaunchKernelGGL((k_gather_range<float>), dim3(blocks_for(work)), dim3(kBlock), 0, gst, (const float *)pl->tm.p,
                           (const int32_t *)pl->rowsets[rowset].p, (float *)dst_dev, nrows, pl->nseg_pad, t_begin, t_end, dst_stride,
                           pl->flow ? 2 : 1);
    else
        hipLaunchKernelGGL((k_gather_range<double>), dim3(blocks_for(work)), dim3(kBlock), 0, pl->stream, (const double *)pl->tm.p,
                           (const int32_t *)pl->rowsets[rowset].p, (double *)dst_dev, nrows, pl->nseg_pad, t_begin, t_end, dst_stride, 1);
    HIP_TRY(hipGetLastError());
    return gst == pl->stream ? note_gather(pl) : 0;
}

int trmc_set_boundary_flow_range(trmc_plan *pl, int t_begin, int t_end, const void *q_dev, int64_t src_stride,
                                 void *stream)
{
    return trmc_set_boundary_flow_range_indexed(pl, t_begin, t_end, q_dev, src_stride, nullptr, stream);
}

int trmc_set_boundary_flow_range_indexed(trmc_plan *pl, int t_begin, int t_end, const void *q_dev, int64_t src_stride,
                                         const int64_t *src_index_dev, void *stream)
{
    if (!pl) return fail(TRMC_EINVAL, "plan is NULL");
    if (!pl->run.active) return fail(TRMC_ESTATE, "trmc_route_begin must precede trmc_set_boundary_flow_range");
    RouteRun &r = pl->run;
    if (t_begin != r.boundary_through || t_end < t_begin || t_end > r.nsteps)
        return fail(TRMC_EINVAL, "ranges must continue where the staged boundary hydrographs end (step "
                                     + std::to_string(r.boundary_through) + ")");
    const int64_t nb = pl->topo.nboundary;
    if (nb == 0 || t_end == t_begin) {
        r.boundary_through = t_end;
        return 0;
    }
    if (!q_dev || src_stride < t_end - t_begin) return fail(TRMC_EINVAL, "q_dev is NULL or src_stride too small");
    if (int rc = use_device(pl)) return rc;
    const size_t plane = (size_t)(r.nsteps + 1) * pl->nseg_pad;
    const int64_t work = nb * (t_end - t_begin);
    hipStream_t st = stream ? (hipStream_t)stream : pl->stream;
    if (!stream && pl->flow && flow_overlap(pl)) {
        // no stream given: before the next launch, and behind whatever was queued after the last one (a gather that
        // produced q_dev, typically)
        st = flow_stream(pl, pl->flow_next);
        if (r.launches > 0 && pl->flow_next != pl->flow_last) {
            HIP_TRY(hipEventRecord(pl->ev_ctl, flow_stream(pl, pl->flow_last)));
            HIP_TRY(hipStreamWaitEvent(st, pl->ev_ctl, 0));
        }
    }
    if (pl->flow) {
        hipLaunchKernelGGL(k_flow_boundary, dim3(blocks_for(work)), dim3(kBlock), 0, st, (const float *)q_dev,
                           (unsigned long long *)pl->tm.p, (float *)pl->out.p, (const int32_t *)pl->row_of_pos.p, (int32_t)nb,
                           r.nsteps, pl->nseg_pad, t_begin, t_end, src_stride, 1, 1, pl->tag_base, src_index_dev);
    } else if (pl->precision == 32) {
        float *q = (float *)pl->tm.p;
        hipLaunchKernelGGL((k_fill_boundary_range<float>), dim3(blocks_for(work)), dim3(kBlock), 0, st, (const float *)q_dev,
                           q, q + plane, q + 2 * plane, (int32_t)nb, pl->nseg_pad, t_begin, t_end, src_stride, src_index_dev);
    } else {
        double *q = (double *)pl->tm.p;
        hipLaunchKernelGGL((k_fill_boundary_range<double>), dim3(blocks_for(work)), dim3(kBlock), 0, st, (const double *)q_dev,
                           q, q + plane, q + 2 * plane, (int32_t)nb, pl->nseg_pad, t_begin, t_end, src_stride, src_index_dev);
    }
    HIP_TRY(hipGetLastError());
    r.boundary_through = t_end;
    return 0;
}

int trmc_download_fvd(trmc_plan *pl, void *fvd_out)
{
    if (!pl) return fail(TRMC_EINVAL, "plan is NULL");
    if (pl->routed_nsteps < 0) return fail(TRMC_ESTATE, "nothing routed yet");
    if (pl->nseg == 0) return 0;
    if (!fvd_out) return fail(TRMC_EINVAL, "fvd_out is NULL");
    if (int rc = use_device(pl)) return rc;
    HIP_TRY(hipMemcpy(fvd_out, pl->out.p, (size_t)pl->nseg * pl->routed_nsteps * 3 * pl->esz, hipMemcpyDeviceToHost));
    return 0;
}

int trmc_download_fvd_strided(trmc_plan *pl, int stride, void *fvd_out)
{
    if (!pl) return fail(TRMC_EINVAL, "plan is NULL");
    if (pl->routed_nsteps < 0) return fail(TRMC_ESTATE, "nothing routed yet");
    if (stride < 1) return fail(TRMC_EINVAL, "stride must be >= 1");
    if (stride == 1) return trmc_download_fvd(pl, fvd_out);
    const int32_t nkeep = pl->routed_nsteps / stride;
    if (pl->nseg == 0 || nkeep == 0) return 0;
    if (!fvd_out) return fail(TRMC_EINVAL, "fvd_out is NULL");
    if (int rc = use_device(pl)) return rc;
    const size_t bytes = (size_t)pl->nseg * nkeep * 3 * pl->esz;
    if (int rc = pl->gathered.ensure(bytes)) return rc; // (the plan's scratch block for gathers of its result)
    pl->gathered_bytes = 0;
    const int64_t work = pl->nseg * (int64_t)nkeep;
    if (pl->precision == 32)
        hipLaunchKernelGGL((k_decimate<float>), dim3(blocks_for(work)), dim3(kBlock), 0, pl->stream, (const float *)pl->out.p,
                           (float *)pl->gathered.p, pl->nseg, pl->routed_nsteps, stride, nkeep);
    else
        hipLaunchKernelGGL((k_decimate<double>), dim3(blocks_for(work)), dim3(kBlock), 0, pl->stream, (const double *)pl->out.p,
                           (double *)pl->gathered.p, pl->nseg, pl->routed_nsteps, stride, nkeep);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(fvd_out, pl->gathered.p, bytes, hipMemcpyDeviceToHost, pl->stream));
    HIP_TRY(hipStreamSynchronize(pl->stream));
    return 0;
}

int trmc_host_alloc(size_t bytes, void **ptr_out)
{
    if (!ptr_out) return fail(TRMC_EINVAL, "ptr_out is NULL");
    *ptr_out = nullptr;
    if (bytes == 0) return 0;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(TRMC_ENODEVICE, "no HIP device available");
    const hipError_t e = hipHostMalloc(ptr_out, bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        *ptr_out = nullptr;
        (void)hipGetLastError();
        return fail(TRMC_ENOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e));
    }
    return 0;
}

int trmc_host_free(void *ptr)
{
    if (ptr) HIP_TRY(hipHostFree(ptr));
    return 0;
}

int trmc_plan_collect_cost(trmc_plan *pl, int enable)
{
    if (!pl) return fail(TRMC_EINVAL, "plan is NULL");
    if (pl->run.active) return fail(TRMC_ESTATE, "a routing window is open");
    pl->collect_cost = enable != 0;
    if (!pl->collect_cost) pl->cost_nsteps = -1;
    return 0;
}

int trmc_download_cost(trmc_plan *pl, uint16_t *cost_out, int32_t *nsteps_out)
{
    if (!pl || !cost_out) return fail(TRMC_EINVAL, "plan/cost_out is NULL");
    if (!pl->collect_cost || pl->cost_nsteps < 0 || pl->routed_nsteps != pl->cost_nsteps)
        return fail(TRMC_ESTATE, "no window has been routed with cost collection on");
    if (int rc = use_device(pl)) return rc;
    std::vector<uint16_t> by_pos((size_t)pl->nseg_pad);
    HIP_TRY(hipMemcpy(by_pos.data(), pl->it_sum.p, (size_t)pl->nseg_pad * sizeof(uint16_t), hipMemcpyDeviceToHost));
    for (int64_t p = 0; p < pl->nseg; ++p) cost_out[pl->topo.row_of_pos[p]] = by_pos[(size_t)p];
    if (nsteps_out) *nsteps_out = pl->cost_nsteps;
    return 0;
}

int trmc_download_iterations(trmc_plan *pl, uint8_t *iters_out)
{
    if (!pl || !iters_out) return fail(TRMC_EINVAL, "plan/iters_out is NULL");
    if (pl->routed_nsteps < 0) return fail(TRMC_ESTATE, "nothing routed yet");
    if (int rc = use_device(pl)) return rc;
    std::vector<uint8_t> by_pos((size_t)pl->nseg_pad);
    HIP_TRY(hipMemcpy(by_pos.data(), pl->it_prev.p, (size_t)pl->nseg_pad, hipMemcpyDeviceToHost));
    for (int64_t p = 0; p < pl->nseg; ++p) iters_out[pl->topo.row_of_pos[p]] = by_pos[(size_t)p];
    return 0;
}

int trmc_download_final_state(trmc_plan *pl, void *q0_out)
{
    if (!pl) return fail(TRMC_EINVAL, "plan is NULL");
    if (pl->routed_nsteps < 0) return fail(TRMC_ESTATE, "nothing routed yet");
    if (pl->nseg == 0) return 0;
    if (!q0_out) return fail(TRMC_EINVAL, "q0_out is NULL");
    if (int rc = use_device(pl)) return rc;
    const size_t bytes = (size_t)pl->nseg * 3 * pl->esz;
    if (int rc = pl->scratch.ensure(bytes)) return rc;
    if (int rc = final_state_into(pl, pl->scratch.p)) return rc;
    HIP_TRY(hipMemcpyAsync(q0_out, pl->scratch.p, bytes, hipMemcpyDeviceToHost, pl->stream));
    HIP_TRY(hipStreamSynchronize(pl->stream));
    return 0;
}

// the plan's copy stream (results to the host beside the next window; the next window's forcing to the device beside this one)
static int ensure_copy_stream(trmc_plan *pl)
{
    if (pl->cstream) return 0;
    // LOW priority: with one hardware queue per priority (GPU_MAX_HW_QUEUES=1) a copy on a stream of ordinary
    // priority shares the queue of the tile stream, and the barrier packet that orders the copy holds the window's tile
    // launches back for as long as the copy runs (CONUS: 50 MB, 0.9 ms of an 18 ms window).  The low-priority queue only
    // carries the result transposes.
    int prio_lo = 0, prio_hi = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    HIP_TRY(hipStreamCreateWithPriority(&pl->cstream, hipStreamNonBlocking, prio_lo));
    // (a stream per direction: a day's decimated result on its way out -- 0.8 GB for a CONUS day, most of a window -- does not
    // hold the next day's forcing back, which the plan's next set-up waits for)
    HIP_TRY(hipStreamCreateWithPriority(&pl->hstream, hipStreamNonBlocking, prio_lo));
    HIP_TRY(hipEventCreateWithFlags(&pl->ev_fetch_ready, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&pl->ev_fetch_done, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&pl->ev_forcing, hipEventDisableTiming));
    return 0;
}

int trmc_fetch_begin(trmc_plan *pl, int32_t rowset, void *hyd_host, void *q0_host)
{
    return trmc_fetch_begin_fvd(pl, rowset, hyd_host, q0_host, 0, nullptr);
}

int trmc_fetch_begin_fvd(trmc_plan *pl, int32_t rowset, void *hyd_host, void *q0_host, int stride, void *fvd_host)
{
    if (!pl) return fail(TRMC_EINVAL, "plan is NULL");
    if (fvd_host && stride < 1) return fail(TRMC_EINVAL, "stride must be >= 1");
    // After a window (trmc_route_end), or -- level engine -- WITH a window that has been queued to its end: the gathers then go
    // right behind the window's last launch on the plan's stream.  That is where a sequence alternating between two plans
    // wants them: queued after the window has been waited for, they would sit in the shared high-priority hardware queue
    // behind the OTHER plan's 288 tail launches, run a whole window late, and this plan's next window -- which overwrites
    // the planes they read -- could not start before (the timeline of bench.py's sequence showed exactly that).
    const bool in_window = pl->run.active;
    if (in_window) {
        if (pl->flow) return fail(TRMC_ESTATE, "a fetch queued with the window needs the level engine");
        if (pl->run.t_done < pl->run.nsteps + (pl->run.short_ts ? pl->maxlag : 0))
            return fail(TRMC_ESTATE, "the window in progress has not been queued to its end (trmc_route_advance)");
    } else if (pl->routed_nsteps < 0) {
        return fail(TRMC_ESTATE, "nothing routed yet");
    }
    if (pl->fetch_pending) return fail(TRMC_ESTATE, "a fetch is in flight (trmc_fetch_wait it first)");
    if (hyd_host && (rowset < 0 || rowset >= (int32_t)pl->rowsets.size())) return fail(TRMC_EINVAL, "unknown row set");
    if (int rc = use_device(pl)) return rc;
    if (int rc = ensure_copy_stream(pl)) return rc;
    if (in_window) { // (the window's own end -- transposing launches, the clock's events -- first)
        if (int rc = pl->precision == 32 ? route_end_queue<float>(pl) : route_end_queue<double>(pl)) return rc;
    }
    const int32_t T_ = in_window ? pl->run.nsteps : pl->routed_nsteps;
    const int64_t nrows = hyd_host ? pl->rowset_n[rowset] : 0;
    const size_t hb = (size_t)nrows * T_ * pl->esz, qb = q0_host ? (size_t)pl->nseg * 3 * pl->esz : 0;
    if (hb) {
        if (int rc = pl->fetch_hyd.ensure(hb)) return rc;
        if (pl->precision == 32)
            hipLaunchKernelGGL((k_gather_rows<float>), dim3(blocks_for(nrows * T_)), dim3(kBlock), 0, pl->stream, (const float *)pl->tm.p,
                               (const int32_t *)pl->rowsets[rowset].p, (float *)pl->fetch_hyd.p, nrows, pl->nseg_pad, T_, pl->flow ? 2 : 1);
        else
            hipLaunchKernelGGL((k_gather_rows<double>), dim3(blocks_for(nrows * T_)), dim3(kBlock), 0, pl->stream, (const double *)pl->tm.p,
                               (const int32_t *)pl->rowsets[rowset].p, (double *)pl->fetch_hyd.p, nrows, pl->nseg_pad, T_, 1);
        HIP_TRY(hipGetLastError());
        if (int rc = note_gather(pl, in_window)) return rc;
    }
    if (qb) {
        if (int rc = pl->fetch_q0.ensure(qb)) return rc;
        if (int rc = final_state_into(pl, pl->fetch_q0.p, in_window ? T_ : -1)) return rc;
    }
    // Every stride-th step of (q, v, d) of every row: decimated on the plan's stream, which at this point follows everything
    // that writes `out` (the tiles and the transposes: route_end_queue).  The kernel reads the whole result once (12 bytes of
    // every 144 at stride 12: every cache line) -- about 2 ms of a CONUS day's period wherever it runs; measured on the sequence
    // with hourly output (ms per day; 17.3 with neither kernel nor copy): here 19.4; on the transpose stream (low priority) 19.3
    // with the copy on a stream of its own and 28 with the copy on the copy stream (the in-order hardware queue of that
    // priority then also holds the next day's transposes and forcing behind the 14-ms copy); without the kernel 17.5.  The copy
    // on a stream of its own with no event behind it (its end polled with hipStreamQuery): days of 24.7 and 15 ms in turn,
    // 20.7 on average -- the copies of consecutive days overlap on the one PCIe direction.
    const int32_t nkeep = fvd_host ? T_ / stride : 0;
    const size_t fb = (size_t)pl->nseg * nkeep * 3 * pl->esz;
    const void *fvd_src = pl->out.p;
    // ... unless the window decimated as it went (trmc_plan_set_output_stride): the tiled rows' kept steps are in `dec` already,
    // the others are gathered from the time-major planes (coalesced; 0.2 GB of a CONUS day instead of 9.4)
    const bool have_dec = fb && !pl->flow && pl->dec.p
                          && (in_window ? pl->run.dec_stride == stride && pl->run.dec_keep == nkeep
                                        : pl->dec_stride_done == stride && pl->dec_keep_done == nkeep && pl->dec_nsteps_done == T_);
    bool from_dec = false;
    if (have_dec) {
        const int32_t lo = in_window ? pl->run.dec_lo : pl->dec_lo_done, hi = in_window ? pl->run.dec_hi : pl->dec_hi_done;
        const size_t plane = (size_t)(T_ + 1) * pl->nseg_pad;
        if (pl->precision == 32) {
            const float *q = (const float *)pl->tm.p;
            hipLaunchKernelGGL((k_decimate_planes<float>), dim3(blocks_for(pl->nseg)), dim3(kBlock), 0, pl->stream, q, q + plane, q + 2 * plane,
                               (const int32_t *)pl->row_of_pos.p, (float *)pl->dec.p, (int32_t)pl->nseg, pl->nseg_pad, stride, nkeep, lo, hi);
        } else {
            const double *q = (const double *)pl->tm.p;
            hipLaunchKernelGGL((k_decimate_planes<double>), dim3(blocks_for(pl->nseg)), dim3(kBlock), 0, pl->stream, q, q + plane, q + 2 * plane,
                               (const int32_t *)pl->row_of_pos.p, (double *)pl->dec.p, (int32_t)pl->nseg, pl->nseg_pad, stride, nkeep, lo, hi);
        }
        HIP_TRY(hipGetLastError());
        if (int rc = note_gather(pl, in_window)) return rc;
        fvd_src = pl->dec.p;
        from_dec = true;
    } else if (fb && stride > 1) {
        if (int rc = pl->fetch_fvd.ensure(fb)) return rc;
        const int64_t work = pl->nseg * (int64_t)nkeep;
        if (pl->precision == 32)
            hipLaunchKernelGGL((k_decimate<float>), dim3(blocks_for(work)), dim3(kBlock), 0, pl->stream, (const float *)pl->out.p,
                               (float *)pl->fetch_fvd.p, pl->nseg, T_, stride, nkeep);
        else
            hipLaunchKernelGGL((k_decimate<double>), dim3(blocks_for(work)), dim3(kBlock), 0, pl->stream, (const double *)pl->out.p,
                               (double *)pl->fetch_fvd.p, pl->nseg, T_, stride, nkeep);
        HIP_TRY(hipGetLastError());
        if (int rc = note_gather(pl, in_window)) return rc; // (the next window's set-up, wherever it is queued, goes behind it)
        fvd_src = pl->fetch_fvd.p;
    }
    HIP_TRY(hipEventRecord(pl->ev_fetch_ready, pl->stream));
    HIP_TRY(hipStreamWaitEvent(pl->cstream, pl->ev_fetch_ready, 0));
    // (Queued WITH the window the copies have a dependence that is still pending, and hipMemcpyAsync device-to-host then keeps
    // the calling thread until it is resolved: trmc_fetch_begin returns when the window ends.  A copy by a kernel that writes
    // the page-locked arrays through the device's mapping never waits on the host, but was no gain on the CONUS sequence --
    // 17.1 ms per day against 16.7: a day's narrow levels can only start when the day before has ended -- and is gone.)
    auto to_host = [&](void *dst_host, const void *src_dev, size_t bytes) -> int {
        HIP_TRY(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, pl->cstream));
        return 0;
    };
    if (hb)
        if (int rc = to_host(hyd_host, pl->fetch_hyd.p, hb)) return rc;
    if (qb)
        if (int rc = to_host(q0_host, pl->fetch_q0.p, qb)) return rc;
    if (fb) {
        if (int rc = to_host(fvd_host, fvd_src, fb)) return rc;
        if (stride == 1 || from_dec) { // (copied from `out` / `dec` themselves: the plan's next window, which writes them, starts behind the copy)
            if (!pl->ev_dec) HIP_TRY(hipEventCreateWithFlags(&pl->ev_dec, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(pl->ev_dec, pl->cstream));
            pl->dec_pending = true;
        }
    }
    HIP_TRY(hipEventRecord(pl->ev_fetch_done, pl->cstream));
    pl->fetch_pending = true;
    return 0;
}

int trmc_fetch_wait(trmc_plan *pl)
{
    if (!pl) return fail(TRMC_EINVAL, "plan is NULL");
    if (!pl->fetch_pending) return 0;
    if (int rc = use_device(pl)) return rc;
    pl->fetch_pending = false;
    HIP_TRY(hipEventSynchronize(pl->ev_fetch_done));
    return 0;
}

int trmc_gather_flow_rows(trmc_plan *pl, const int64_t *rows, int64_t nrows, void *out, int dst_is_device)
{
    if (!pl) return fail(TRMC_EINVAL, "plan is NULL");
    if (pl->routed_nsteps < 0) return fail(TRMC_ESTATE, "nothing routed yet");
    if (nrows == 0) return 0;
    if (!rows || (!out && !dst_is_device)) return fail(TRMC_EINVAL, "rows/out is NULL");
    if (int rc = use_device(pl)) return rc;
    std::vector<int32_t> pos((size_t)nrows);
    for (int64_t i = 0; i < nrows; ++i) {
        if (rows[i] < 0 || rows[i] >= pl->nseg) return fail(TRMC_EINVAL, "row out of range");
        pos[i] = pl->topo.pos_of_row[rows[i]];
    }
    const int32_t T_ = pl->routed_nsteps;
    const size_t obytes = (size_t)nrows * T_ * pl->esz;
    const size_t pbytes = ((size_t)nrows * sizeof(int32_t) + 255) / 256 * 256;
    if (int rc = pl->scratch.ensure(pbytes + (dst_is_device ? 0 : obytes))) return rc;
    HIP_TRY(hipMemcpyAsync(pl->scratch.p, pos.data(), (size_t)nrows * sizeof(int32_t), hipMemcpyHostToDevice, pl->stream));
    if (dst_is_device && !out) { // keep the block in plan-owned HBM; fetch with trmc_download_gathered()
        if (int rc = pl->gathered.ensure(obytes)) return rc;
        out = pl->gathered.p;
        pl->gathered_bytes = obytes;
    }
    void *dst = dst_is_device ? out : (void *)((char *)pl->scratch.p + pbytes);
    if (pl->precision == 32)
        hipLaunchKernelGGL((k_gather_rows<float>), dim3(blocks_for(nrows * T_)), dim3(kBlock), 0, pl->stream,
                           (const float *)pl->tm.p, (const int32_t *)pl->scratch.p, (float *)dst, nrows, pl->nseg_pad, T_,
                           pl->flow ? 2 : 1);
    else
        hipLaunchKernelGGL((k_gather_rows<double>), dim3(blocks_for(nrows * T_)), dim3(kBlock), 0, pl->stream,
                           (const double *)pl->tm.p, (const int32_t *)pl->scratch.p, (double *)dst, nrows, pl->nseg_pad, T_, 1);
    HIP_TRY(hipGetLastError());
    if (!dst_is_device) HIP_TRY(hipMemcpyAsync(out, dst, obytes, hipMemcpyDeviceToHost, pl->stream));
    HIP_TRY(hipStreamSynchronize(pl->stream));
    return 0;
}

int trmc_download_gathered(trmc_plan *pl, void *out)
{
    if (!pl || !out) return fail(TRMC_EINVAL, "plan/out is NULL");
    if (pl->gathered_bytes == 0) return fail(TRMC_ESTATE, "no device-resident gather to download");
    if (int rc = use_device(pl)) return rc;
    HIP_TRY(hipMemcpy(out, pl->gathered.p, pl->gathered_bytes, hipMemcpyDeviceToHost));
    return 0;
}

int trmc_get_stats(const trmc_plan *pl, trmc_stats *stats)
{
    if (!pl || !stats) return fail(TRMC_EINVAL, "plan/stats is NULL");
    *stats = pl->stats;
    return 0;
}

int trmc_route(trmc_plan *pl, int nsteps, int qts_subdivisions, int assume_short_ts, const void *qlat, int64_t nq,
               const void *q0, const void *boundary_fvd, void *fvd_out)
{
    if (int rc = trmc_upload_forcing(pl, nsteps, qlat, nq, q0, boundary_fvd)) return rc;
    if (int rc = trmc_route_device(pl, nsteps, qts_subdivisions, assume_short_ts)) return rc;
    return trmc_download_fvd(pl, fvd_out);
}

int trmc_segments(int device, int precision, int64_t n, const void *in, void *out)
{
    return trmc_segments_ex(device, precision, TRMC_ARITH_EXACT, n, in, out, nullptr);
}

int trmc_segments_ex(int device, int precision, int arithmetic, int64_t n, const void *in, void *out, int32_t *iters_out)
{
    if (arithmetic != TRMC_ARITH_EXACT && arithmetic != TRMC_ARITH_TOLERANCE) return fail(TRMC_EINVAL, "bad arithmetic");
    if (arithmetic == TRMC_ARITH_TOLERANCE && precision != 32)
        return fail(TRMC_EINVAL, "TRMC_ARITH_TOLERANCE is an arithmetic of precision 32");
    if (precision != 32 && precision != 64) return fail(TRMC_EINVAL, "precision must be 32 or 64");
    if (n < 0) return fail(TRMC_EINVAL, "n < 0");
    if (n == 0) return 0;
    if (!in || !out) return fail(TRMC_EINVAL, "in/out is NULL");
    if (int rc = check_device(device)) return rc;
    HIP_TRY(hipSetDevice(device));
    const bool tol = arithmetic == TRMC_ARITH_TOLERANCE;
    return precision == 32 ? segments_t<float>(n, in, out, tol, iters_out) : segments_t<double>(n, in, out, false, iters_out);
}

// The reference's own C binding of one segment-step -- c_muskingcungenwm, src/kernel/muskingum/pyMCsingleSegStime_NoLoop.f90:8-21
// (header src/troute-routing/troute/routing/fast_reach/pyMCsingleSegStime_NoLoop.h:1-21, declared to Cython at
// fast_reach/fortran_wrappers.pxd:19-40): 21 float pointers, 15 in, 6 out, no return value.  One step on the device per call
// (trmc_segments with n = 1): the drop-in for reach.pyx:37-94's call site, not a fast path -- the fast paths are the batch
// form and the plans.  Like the Fortran it cannot signal: on failure the six outputs are NaN and trmc_last_error() says why.
// qdc is taken as 0 on entry (reach.pyx:55 passes 0; f90:74 reads it).  Device: TRMC_DEVICE (default 0).
void trmc_muskingcungenwm(float *dt, float *qup, float *quc, float *qdp, float *ql, float *dx, float *bw, float *tw,
                          float *twcc, float *n, float *ncc, float *cs, float *s0, float *velp, float *depthp, float *qdc,
                          float *velc, float *depthc, float *ck, float *cn, float *X)
{
    float *outs[6] = {qdc, velc, depthc, ck, cn, X};
    const float *ins[15] = {dt, qup, quc, qdp, ql, dx, bw, tw, twcc, n, ncc, cs, s0, velp, depthp};
    bool ok = true;
    for (const float *p : ins) ok = ok && p != nullptr;
    for (float *p : outs) ok = ok && p != nullptr;
    float in[15], out[6];
    int rc = TRMC_EINVAL;
    if (ok) {
        for (int i = 0; i < 15; ++i) in[i] = *ins[i];
        const char *d = std::getenv("TRMC_DEVICE");
        rc = trmc_segments(d ? std::atoi(d) : 0, 32, 1, in, out);
    } else {
        (void)fail(TRMC_EINVAL, "trmc_muskingcungenwm: an argument is NULL");
    }
    for (int i = 0; i < 6; ++i)
        if (outs[i]) *outs[i] = rc == 0 ? out[i] : std::nanf("");
}

int trmc_plan_chain_from(trmc_plan *dst, trmc_plan *src)
{
    if (!dst || !src || dst == src) return fail(TRMC_EINVAL, "two different plans are needed");
    if (dst->flow || src->flow) return fail(TRMC_EINVAL, "plans of the level engine only");
    if (dst->run.active) return fail(TRMC_ESTATE, "a routing window of the receiving plan is in progress");
    if (dst->staged_nsteps < 1) return fail(TRMC_ESTATE, "the receiving plan has no forcing staged (trmc_upload_forcing)");
    if (src->run.nsteps < 1 || (!src->run.active && src->routed_nsteps < 0))
        return fail(TRMC_ESTATE, "the source plan has routed nothing");
    if (src->run.active && src->run.t_done < src->run.nsteps + (src->run.short_ts ? src->maxlag : 0))
        return fail(TRMC_ESTATE, "the source plan's window has not been queued to its end (trmc_route_advance)");
    if (dst->nseg != src->nseg || dst->precision != src->precision || dst->device != src->device
        || dst->topo.row_of_pos != src->topo.row_of_pos)
        return fail(TRMC_EINVAL, "the two plans must hold the same network in the same order (same inputs, same cost hint)");
    if (int rc = use_device(dst)) return rc;
    return dst->precision == 32 ? chain_from_t<float>(dst, src, dst->staged_nsteps) : chain_from_t<double>(dst, src, dst->staged_nsteps);
}

int trmc_selfcheck_fast_arith(int device, int what, int64_t n, uint64_t seed, int64_t *checked_out, int64_t *mismatches_out)
{
    if (!checked_out || !mismatches_out) return fail(TRMC_EINVAL, "checked_out/mismatches_out is NULL");
    if (what != 0 && what != 1) return fail(TRMC_EINVAL, "what must be 0 (square root) or 1 (division, maximum)");
    if (what == 1 && n < 0) return fail(TRMC_EINVAL, "n < 0");
    if (int rc = check_device(device)) return rc;
    HIP_TRY(hipSetDevice(device));
    DevBuf cnt;
    if (int rc = cnt.ensure(sizeof(unsigned long long))) return rc;
    hipError_t e = hipMemset(cnt.p, 0, sizeof(unsigned long long));
    int64_t checked = 0;
    if (e == hipSuccess) {
        if (what == 0) {
            const uint32_t lo = (uint32_t)(127 - 60) << 23, hi = (uint32_t)(127 + 63) << 23; // 2**-60 .. 2**63 inclusive
            checked = (int64_t)hi - (int64_t)lo + 1;
            hipLaunchKernelGGL(k_selfcheck_sqrt, dim3(8192), dim3(kBlock), 0, 0, lo, hi, (unsigned long long *)cnt.p);
        } else {
            checked = n;
            if (n > 0) hipLaunchKernelGGL(k_selfcheck_div, dim3(8192), dim3(kBlock), 0, 0, n, seed, (unsigned long long *)cnt.p);
        }
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    unsigned long long bad = 0;
    if (e == hipSuccess) e = hipMemcpy(&bad, cnt.p, sizeof bad, hipMemcpyDeviceToHost);
    cnt.release();
    if (e != hipSuccess) return fail(TRMC_EHIP, std::string("trmc_selfcheck_fast_arith: ") + hipGetErrorString(e));
    *checked_out = checked;
    *mismatches_out = (int64_t)bad;
    return 0;
}

#include "stream.inc"

} // extern "C"
