// mc_segment.hpp -- one Muskingum-Cunge segment-timestep, precision-generic.
//
// Device arithmetic of the engine.  Semantics follow the reference Fortran
// (paths relative to the T-Route tree):
//   src/kernel/muskingum/MCsingleSegStime_f2py_NOLOOP.f90
//     muskingcungenwm :8-186, secant2_h :198-334, courant :342-367,
//     hydraulic_geometry :374-444
// with the canonical initialisation Qj_0 = 0 at label 110 (the shipped routine
// reads Qj_0 uninitialised at :92-93; its parent sets it,
// src/kernel/muskingum/MUSKINGCUNGE.f90:100, and
// src/kernel/muskingum/test_MC_kernel.py:36-40 asserts equality with the parent).
//
// Layout of the computation (not of the reference's call tree):
//   * ChannelConst   everything that depends only on the channel parameters is
//                    formed once per segment-step (side distance z, bankfull
//                    depth, sqrt(s0), sqrt(1+z^2), the two Manning factors);
//   * HydraulicPoint everything a residual evaluation needs that depends on the depth alone;
//   * step_pre       the two points the secant iteration starts from -- they depend on the row's OWN depth of the step
//                    before only, never on what flows in, so a row that waits for its upstream rows (the dataflow
//                    engine) evaluates them while it waits;
//   * step_solve     the bracketed secant iteration and the outflow (what the row below waits for);
//   * step_velocity  the velocity of the result row, which no other row reads;
//   * mc_segment_step = the three in sequence.
// The same operations, in the same order and rounding as the reference, so the
// fp32 instantiation is bit-comparable with the fp32 Fortran wherever pow()
// agrees; compile with -ffp-contract=off.
//
// Math policy M:  m.sqrt(x); m.pow(x, y); and, so that x**a and x**b can share
// one logarithm, M::Log, m.log_of(x), m.pow_l(log_of(x), x, y) == m.pow(x, y);
// m.log_of_r(x, ok) / m.pow_l_r(l, x, y, ok): the same for a hydraulic radius x; with `ok` (see fast_ok) the policy may
// skip the special-value tests of its power.
// m.div4(n1, n2, n3, n4, d, q1, q2, q3, q4): qi = ni / d, the four correctly rounded quotients by one
// divisor (the Muskingum coefficients) -- a policy may share the reciprocal between them;
// (a policy whose fast_ok can be true promises with it: h > 0 and a positive wetted perimeter)
// m.fast_ok(h, h_in, h_over) -> ok; m.div2(a1, a2, b, ok, q1, q2): qi = ai / b; m.div1(a, b, ok) = a / b: the
// divisions of the hydraulic point, for which a policy may use a cheaper exact sequence when `ok` (its own test of
// the operand ranges) holds.
// m.k_of(dx, ck) = dx / ck and m.max_num(a, b) = max(a, b) for the Muskingum K of an in-bank point, where the policy's
// fast_ok has shown the celerity positive, normal and within 2**95 of dx (DevMathF: a division without scaling or fix-up,
// one v_max); m.sqrt_r(x, ok): sqrt(x), with `ok` for an x the policy knows to be an ordinary number.
// M::kInbank: whether the policy wants the in-bank body of the hydraulic point at all (hydraulics_inbank: fewer instructions,
// more code and registers -- the level kernels take it, the dataflow kernels, whose pace is one wavefront's latency, do not).
// m.all(pred): true when `pred` holds for every row that is evaluated together with this one (a wavefront's active
// lanes) -- a scalar condition: the in-bank body of the hydraulic point is chosen by ONE uniform branch per wavefront.
//
// The header is host/device neutral (no HIP construct outside MC_HD); the shipped library only ever instantiates
// it in device code, and its results are checked against the oracle through the C ABI (tests/test_gpu_parity.py).
#pragma once

#if defined(__HIPCC__)
#define MC_HD __host__ __device__ __forceinline__
#else
#define MC_HD inline
#endif

namespace trmc {

template <class T> MC_HD T mc_max(T a, T b) { return a > b ? a : b; }
template <class T> MC_HD T mc_min(T a, T b) { return a < b ? a : b; }
// |a|.  Every magnitude of the step is only ever COMPARED (the absolute error against the depth floor, the lateral loss
// against the routed sum), so the sign of a zero and of a NaN cannot show: the sign bit is cleared, which costs nothing
// on the device (an operand modifier) where the select `a < 0 ? -a : a` was a compare and a move.
MC_HD float mc_abs(float a) { return __builtin_fabsf(a); }
MC_HD double mc_abs(double a) { return __builtin_fabs(a); }

// Channel parameters of one segment, as the reference names them.
template <class T> struct ChannelParams {
    T dt, dx, bw, tw, twcc, n, ncc, cs, s0;
};

// Segment-invariant quantities.
template <class T> struct ChannelConst {
    T z;          // side distance 1/cs (1 when cs == 0)            f90:49-53
    T bfd;        // bankfull depth                                 f90:55-61
    T sqrt_s0;    // sqrt(s0)
    T sq1pz2;     // sqrt(1 + z*z)
    T s0_n;       // sqrt(s0)/n
    T s0_ncc;     // sqrt(s0)/ncc
    T two_sq;     // 2*sqrt(1 + z*z)
    T half_dt;    // dt/2
    T inv_n;      // 1/n, the factor of the velocity formula (f90:169)
    T z2;         // 2*z
    bool fp_ok;   // twcc > 0 && ncc > 0 : flood-plain terms may activate
};

// the constants that are one operation away from the stored ones (the kernels load z, bfd, sqrt_s0, sq1pz2, s0_n, s0_ncc,
// inv_n from the plan's columns and form these per row)
template <class T> MC_HD void derive_const(ChannelConst<T> &c, const ChannelParams<T> &p)
{
    c.two_sq = T(2) * c.sq1pz2;
    c.z2 = T(2) * c.z;
    c.half_dt = p.dt / T(2);
    c.fp_ok = (p.twcc > T(0)) && (p.ncc > T(0));
}

template <class T, class M>
MC_HD ChannelConst<T> make_const(const ChannelParams<T> &p, const M &m)
{
    ChannelConst<T> c;
    c.z = (p.cs == T(0)) ? T(1) : T(1) / p.cs;
    if (p.bw > p.tw)
        c.bfd = p.bw / T(0.00001);
    else if (p.bw == p.tw)
        c.bfd = p.bw / (T(2) * c.z);
    else
        c.bfd = (p.tw - p.bw) / (T(2) * c.z);
    c.sqrt_s0 = m.sqrt(p.s0);
    c.sq1pz2 = m.sqrt(T(1) + c.z * c.z);
    c.s0_n = c.sqrt_s0 / p.n;
    c.s0_ncc = c.sqrt_s0 / p.ncc;
    c.inv_n = T(1) / p.n;
    derive_const(c, p);
    return c;
}

// Wetted geometry at depth h (f90:374-444).
template <class T> struct Section {
    T twl, area, areac, wp, wpc, R, h_in, h_over;
};

template <class T, class M, bool WITH_R = true>
MC_HD Section<T> section_at(T h, const ChannelParams<T> &p, const ChannelConst<T> &c, const M &m)
{
    Section<T> s;
    s.twl = p.bw + T(2) * c.z * h;
    s.h_over = mc_max(h - c.bfd, T(0));
    s.h_in = mc_min(c.bfd, h);
    {
        const bool extend = s.h_over > T(0) && p.twcc <= T(0); // NWM 3.0: no flood plain -> extend trapezoid
        s.h_over = extend ? T(0) : s.h_over;
        s.h_in = extend ? h : s.h_in;
    }
    s.area = (p.bw + s.h_in * c.z) * s.h_in;
    s.wp = p.bw + T(2) * s.h_in * c.sq1pz2;
    s.areac = p.twcc * s.h_over;
    s.wpc = (s.h_over > T(0)) ? p.twcc + (T(2) * s.h_over) : T(0);
    if (WITH_R) s.R = (s.area + s.areac) / (s.wp + s.wpc);
    return s;
}

// Muskingum coefficients carried from one residual evaluation to the next
// (the reference keeps them alive through intent(out) dummies, f90:92-95).
template <class T> struct MuskCoef {
    T C1, C2, C3, C4, X;
};

// Boundary flows of one segment-step.
template <class T> struct Inflow {
    T qup, quc, qdp, ql;
};

// Everything the residual needs that depends on the depth alone (f90:244-275): wetted geometry,
// R**(2/3), kinematic celerity Ck and the Muskingum K.  The secant loop evaluates the residual at
// (h_0, h) and then moves h_0 <- h, so the point computed for h in one iteration IS the point for
// h_0 in the next: it is kept instead of being recomputed (same operations on the same inputs, hence
// the same bits -- only work is saved).
template <class T> struct HydraulicPoint {
    T ck, km;      // kinematic celerity, Muskingum K
    T denom;       // 2 * width * s0 * Ck * dx, the divisor of the X formula (width = twcc over bank, else twl)
    T q_manning;   // (1/n_composite) * (AREA+AREAC) * R**(2/3) * sqrt(s0); 0-flagged by has_wp
    bool has_wp;   // WP + WPC > 0
    bool over;     // evaluated with the flood plain active (the compound-channel branch)
};

// (h - bfd)**(2/3) of the over-bank celerity.  Under fast_ok the over-bank depth is h - bfd itself (the branch is only taken
// with a flood plain, so the NWM-3.0 exception has not zeroed it) and lies in [2**-30, 2**17]: an ordinary number like the
// hydraulic radius, for which the policy's power needs no special-value tests.
template <class T, class M, bool OK> MC_HD T over_bank_pow(T x, T y, const M &m)
{
    if (OK) return m.pow_l_r(m.log_of_r(x, true), x, y, true);
    return m.pow(x, y);
}
// (OK: the policy's fast_ok() verdict as a compile-time constant -- the point is evaluated by one of two straight-line
// bodies chosen by ONE branch, instead of testing the same flag at each of its six divisions and powers)
template <class T, class M, bool OK>
MC_HD HydraulicPoint<T> hydraulics_core(T h, Section<T> &s, const ChannelParams<T> &p, const ChannelConst<T> &c, const M &m)
{
    const T c23 = T(2) / T(3), c53 = T(5) / T(3);
    HydraulicPoint<T> hp;
    const bool over = (h > c.bfd) && c.fp_ok;
    // the hydraulic radius and the composite Manning n are quotients by the same wetted perimeter (f90:417, :329)
    const bool ok = OK;
    T n_comp;
    m.div2(s.area + s.areac, (s.wp * p.n) + (s.wpc * p.ncc), s.wp + s.wpc, ok, s.R, n_comp);

    // R**(2/3) and R**(5/3) share one logarithm (M::Log), see det_pow.h; `ok` also tells the policy that R is an
    // ordinary number (DevMathF::fast_ok: 2**-64 < R < 2**63), for which a power needs no special-value tests
    const typename M::Log lr = m.log_of_r(s.R, ok);
    const T r23 = m.pow_l_r(lr, s.R, c23, ok);
    if (over) {
        hp.ck = mc_max(T(0), (c.s0_n * (c53 * r23 - (c23 * m.pow_l_r(lr, s.R, c53, ok)
                                                     * (c.two_sq / (p.bw + T(2) * c.bfd * c.z))))
                                  * s.area
                              + (c.s0_ncc * c53 * over_bank_pow<T, M, OK>(h - c.bfd, c23, m)) * s.areac)
                                 / (s.area + s.areac));
    } else if (OK || h > T(0)) { // (fast_ok: the in-bank depth is at least 2**-30, the bottom width at least 2**-14)
        hp.ck = mc_max(T(0), c.s0_n * (c53 * r23 - (c23 * m.pow_l_r(lr, s.R, c53, ok)
                                                    * m.div1(c.two_sq, p.bw + T(2) * h * c.z, ok))));
    } else {
        hp.ck = T(0);
    }
    {
        const T kq = mc_max(p.dt, m.divx(p.dx, hp.ck)); // (a zero celerity: inf or NaN, discarded)
        hp.km = (hp.ck > T(0)) ? kq : p.dt;
    }
    hp.denom = T(2) * (over ? p.twcc : s.twl) * p.s0 * hp.ck * p.dx;
    hp.has_wp = OK || (s.wp + s.wpc) > T(0);
    hp.over = over;
    hp.q_manning = m.div1(T(1), n_comp, ok) * (s.area + s.areac) * r23 * c.sqrt_s0;
    return hp;
}
template <class T, class M>
MC_HD HydraulicPoint<T> hydraulics_general(T h, const ChannelParams<T> &p, const ChannelConst<T> &c, const M &m)
{
    Section<T> s = section_at<T, M, false>(h, p, c, m);
    if (m.fast_ok(h, s.h_in, s.h_over)) return hydraulics_core<T, M, true>(h, s, p, c, m);
    return hydraulics_core<T, M, false>(h, s, p, c, m);
}

// The point of a depth that lies in the channel (h <= bfd) and in the range of the policy's fast_ok: what
// hydraulics_core<OK = true> computes there, with everything the flood plain contributes written out of it.  With
// h <= bfd: h_over = max(h - bfd, 0) = 0 and h_in = min(bfd, h) = h; the NWM-3.0 exception needs h_over > 0; areac =
// twcc * 0 = +0 and wpc = 0 (twcc, ncc finite: the policy's parameter check), so area + areac = area, wp*n + wpc*ncc =
// wp*n and wp + wpc = wp bit for bit (x + 0 = x for x > 0); `over` is false.  Three products are taken in another
// association that rounds the same because a factor of two is exact:  (2 z) h = (2 h) z  and  (2 h) sq = h (2 sq) --
// so the top width term of the celerity, bw + 2 h z, IS twl.  Same bits as the general body, some forty instructions
// fewer, most of them compares and selects.
template <class T, class M> MC_HD bool inbank_fast(T h, const ChannelConst<T> &c, const M &m)
{
    return m.fast_ok(h, h, T(0)) && h <= c.bfd;
}
template <class T, class M>
MC_HD HydraulicPoint<T> hydraulics_inbank(T h, const ChannelParams<T> &p, const ChannelConst<T> &c, const M &m)
{
    const T c23 = T(2) / T(3), c53 = T(5) / T(3);
    HydraulicPoint<T> hp;
    const T twl = p.bw + c.z2 * h;
    const T area = (p.bw + h * c.z) * h;
    const T wp = p.bw + h * c.two_sq;
    T R, n_comp;
    m.div2(area, wp * p.n, wp, true, R, n_comp);
    const typename M::Log lr = m.log_of_r(R, true);
    const T r23 = m.pow_l_r(lr, R, c23, true);
    // The celerity of an in-bank point under fast_ok is POSITIVE and an ordinary number (the proof is at DevMathF::fast_ok:
    // the bracket is r23 (5/3 - 2/3 [A 2 sq / (WP twl)]) with the square bracket below one), so max(0, .) returns it, the
    // guard `ck > 0` of K holds, and K = max(dt, dx / ck) is one division of ordinary numbers and one maximum.
    hp.ck = c.s0_n * (c53 * r23 - (c23 * m.pow_l_r(lr, R, c53, true) * m.div1(c.two_sq, twl, true)));
    hp.km = m.max_num(p.dt, m.k_of(p.dx, hp.ck));
    hp.denom = T(2) * twl * p.s0 * hp.ck * p.dx;
    hp.has_wp = true;
    hp.over = false;
    hp.q_manning = m.div1(T(1), n_comp, true) * area * r23 * c.sqrt_s0;
    return hp;
}

// (The same body for the bracket's TWO depths at once on two-element vectors -- v_pk_add / v_pk_mul / v_pk_fma_f32 for both
// points -- was built and measured in round 3: 671 instead of 701 instructions per wavefront-step and slower, a packed
// instruction issues for two passes; removed.)
// one point: the in-bank body when every row evaluated together is in bank (one uniform branch), else the general one
template <class T, class M>
MC_HD HydraulicPoint<T> hydraulics_at(T h, const ChannelParams<T> &p, const ChannelConst<T> &c, const M &m)
{
    if (M::kInbank && m.all(inbank_fast<T, M>(h, c, m))) return hydraulics_inbank<T, M>(h, p, c, m);
    return hydraulics_general<T, M>(h, p, c, m);
}

// Residual Q_mc(h) - Q_manning(h) at the hydraulic point of depth h (f90:277-332).
//   LOWER=false: "interval 1": X from the previous residual `qj_prev`, clamp [0,0.5]
//   LOWER=true : "interval 2": X from the coefficients just left in `k`, clamp [0.25,0.5]
template <class T, class M, bool LOWER>
MC_HD T secant_residual(const HydraulicPoint<T> &hp, T qj_prev, const ChannelParams<T> &p,
                        const ChannelConst<T> &c, const Inflow<T> &f, MuskCoef<T> &k, const M &m)
{
    const T km = hp.km;
    T x;
    {
        // The weighting factor is formed whether or not the celerity is positive and selected afterwards (a zero celerity
        // has a zero `denom`: the quotient is NaN or inf and is discarded) -- a guarded division is a divergent branch,
        // and the guard nearly always holds.  Same for K = dx / celerity and the secant update below.
        const T denom = hp.denom;
        T xv;
        if (!LOWER) {
            if (qj_prev == T(0) && (denom > T(0) || denom < T(0)))
                xv = T(0.5);
            else
                xv = mc_min(T(0.5), mc_max(T(0), T(0.5) * (T(1) - m.divx(qj_prev, denom))));
        } else
            xv = mc_min(T(0.5),
                        mc_max(T(0.25),
                               T(0.5) * (T(1) - m.divx(((k.C1 * f.qup) + (k.C2 * f.quc) + (k.C3 * f.qdp) + k.C4),
                                                        denom))));
        x = (hp.ck > T(0)) ? xv : T(0.5);
    }

    const T d = (km * (T(1) - x) + c.half_dt);
    m.div4(km * x + c.half_dt, c.half_dt - km * x, km * (T(1) - x) - c.half_dt, f.ql * p.dt, d,
           k.C1, k.C2, k.C3, k.C4);
    k.X = x;
    if (LOWER) {
        const T w = (k.C1 * f.qup) + (k.C2 * f.quc) + (k.C3 * f.qdp);
        k.C4 = ((k.C4 < T(0)) && (mc_abs(k.C4) > w)) ? -w : k.C4;
    }

    const T res = ((k.C1 * f.qup) + (k.C2 * f.quc) + (k.C3 * f.qdp) + k.C4) - hp.q_manning;
    return hp.has_wp ? res : T(0);
}

// Kinematic celerity and Courant number at depth h (f90:342-367; unguarded).
template <class T, class M>
MC_HD void courant_at(T h, const ChannelParams<T> &p, const ChannelConst<T> &c, T &ck, T &cn, const M &m)
{
    const T c23 = T(2) / T(3), c53 = T(5) / T(3);
    const Section<T> s = section_at<T, M>(h, p, c, m);
    const typename M::Log lr = m.log_of(s.R);
    ck = mc_max(T(0), (c.s0_n * (c53 * m.pow_l(lr, s.R, c23)
                                  - (c23 * m.pow_l(lr, s.R, c53) * (c.two_sq / (p.bw + T(2) * s.h_in * c.z))))
                           * s.area
                       + (c.s0_ncc * c53 * m.pow(s.h_over, c23)) * s.areac)
                          / (s.area + s.areac));
    cn = ck * (p.dt / p.dx);
}

template <class T> struct StepResult {
    T qdc, velc, depthc; // outflow, velocity, depth at the new time level
    T h;                 // depth the iteration ended on (courant uses it even with no flow)
    T X;                 // last weighting factor (0 when nothing was routed)
    int iters;           // secant iterations spent (all retries together)
    bool over;           // some evaluation of the step took the compound-channel (over-bank) branch: cost diagnostics only
};

// ---- the step in three parts -------------------------------------------------------------------------------------
// The bracket the iteration starts from and its two hydraulic points (f90:69-71, first pass of :83-95): functions of
// the depth at the previous time level and the channel alone.
template <class T> struct StepPre {
    T h, h_0;
    HydraulicPoint<T> at_h0, at_h;
    bool have; // the two points have been evaluated
};
template <class T> MC_HD void step_bracket(T depthp, T &h, T &h_0)
{
    const T depth0 = mc_max(depthp, T(0));
    h = (depth0 * T(1.33)) + T(0.01);
    h_0 = (depth0 * T(0.67));
}
template <class T, class M>
MC_HD StepPre<T> step_pre(const ChannelParams<T> &p, const ChannelConst<T> &c, T depthp, const M &m)
{
    StepPre<T> s;
    step_bracket(depthp, s.h, s.h_0);
    // (0 <= h_0 <= h: the lower bound of the range is tested on h_0, the upper bound and the bank on h)
    if (M::kInbank && m.all(m.fast_ok(s.h, s.h_0, T(0)) && s.h <= c.bfd)) {
        s.at_h0 = hydraulics_inbank<T, M>(s.h_0, p, c, m);
        s.at_h = hydraulics_inbank<T, M>(s.h, p, c, m);
    } else {
        s.at_h0 = hydraulics_general<T, M>(s.h_0, p, c, m);
        s.at_h = hydraulics_general<T, M>(s.h, p, c, m);
    }
    s.have = true;
    return s;
}
// whether anything is routed at all (f90:73-74; the caller passes qdc = 0, reach.pyx:55)
template <class T> MC_HD bool step_has_flow(const Inflow<T> &f)
{
    return f.ql > T(0) || f.qup > T(0) || f.quc > T(0) || f.qdp > T(0);
}
// ... and whether that is already certain from the row's own state
template <class T> MC_HD bool step_has_own_flow(T ql, T qdp) { return ql > T(0) || qdp > T(0); }

template <class T> struct StepSolve {
    T qdc, h, X;
    int iters;
    bool over;
};
// what a secant iteration carries to the next one (f90:92-117; the coefficients and both residual slots live on through
// the reference's intent(out) dummies)
template <class T> struct SecantState {
    T h, h_0, qj_0, aerror;
    bool rel_open;
    MuskCoef<T> k;
    HydraulicPoint<T> at_h0; // the point of h_0: after an iteration, the one just evaluated for h (h_0 <- h)
};
// one iteration, the hydraulic point of the current h supplied (f90:92-117)
template <class T, class M>
MC_HD void secant_iterate(SecantState<T> &s, const HydraulicPoint<T> &at_h, const ChannelParams<T> &p, const ChannelConst<T> &c,
                          const Inflow<T> &f, const M &m)
{
    s.qj_0 = secant_residual<T, M, false>(s.at_h0, s.qj_0, p, c, f, s.k, m);
    const T qj = secant_residual<T, M, true>(at_h, T(0), p, c, f, s.k, m);
    const T h = s.h, h_0 = s.h_0;
    T h_1;
    {
        const T hq = h - m.divx(qj * (h_0 - h), s.qj_0 - qj); // (equal residuals: NaN or inf, discarded)
        h_1 = (s.qj_0 - qj != T(0)) ? hq : h;
        h_1 = (h_1 < T(0)) ? h : h_1;
    }
    {
        // The relative error |(h_1 - h) / h| (f90:108) is only ever compared with 0.01 (f90:83): `rel_open` is that
        // comparison, decided without the division whenever the ratio is clearly on one side (a band of 1e-4 around the
        // threshold against a rounding error of 6e-8; a NaN fails both tests and takes the division).
        const T dh = mc_abs(h_1 - h);
        const bool hi = dh > T(0.010001) * h, lo = dh < T(0.009999) * h;
        bool open = hi;
        if (!hi && !lo) open = mc_abs((h_1 - h) / h) > T(0.01); // (the band around the threshold, or a NaN: rare)
        s.rel_open = (h > T(0)) ? open : false;                   // h = 0: rerror = 0
        s.aerror = (h > T(0)) ? dh : T(0.9);
    }
    s.at_h0 = at_h;
    s.h_0 = mc_max(T(0), h);
    s.h = mc_max(T(0), h_1);
}
// The secant iteration (f90:83-134) and the outflow (f90:149-161).  `pre` = step_pre of the same row and depth (evaluated
// here when it has not been); requires step_has_flow(f).  The first iteration of the first pass is written out on its
// own: it is always entered (rerror = 1, aerror = 0.01, iter = 0), both of its points come from `pre`, and its first
// residual starts from Qj_0 = 0 -- the compiler folds that; 40 % of the CONUS segment-steps are done after it (depth under
// the 1 cm floor).  Retries (f90:126-134: more than 100 iterations, never seen on a river network) go round the same loop.
template <class T, class M>
MC_HD StepSolve<T> step_solve(const ChannelParams<T> &p, const ChannelConst<T> &c, const Inflow<T> &f, T depthp,
                              const StepPre<T> &pre_in, const M &m)
{
    const T mindepth = T(0.01);
    StepSolve<T> out;
    const StepPre<T> pre = pre_in.have ? pre_in : step_pre<T, M>(p, c, depthp, m);
    SecantState<T> s;
    s.h = pre.h;
    s.h_0 = pre.h_0;
    s.qj_0 = T(0);
    s.k = MuskCoef<T>{T(0), T(0), T(0), T(0), T(0)};
    s.at_h0 = pre.at_h0;
    bool any_over = pre.at_h0.over || pre.at_h.over;
    int maxiter = 100, tries = 0, iter = 1, total_iter = 1;
    secant_iterate<T, M>(s, pre.at_h, p, c, f, m);
    bool skip = s.h < mindepth; // (f90:120-122 after the first iteration)
    for (;;) {
        if (!skip) {
            while (s.rel_open && s.aerror >= mindepth && iter <= maxiter) {
                const HydraulicPoint<T> at_h = hydraulics_at<T, M>(s.h, p, c, m);
                any_over = any_over || at_h.over;
                secant_iterate<T, M>(s, at_h, p, c, f, m);
                ++iter;
                ++total_iter;
                if (s.h < mindepth) break;
            }
        }
        skip = false;
        if (iter >= maxiter && ++tries <= 4) { // widen the bracket and retry
            s.h = s.h * T(1.33);
            s.h_0 = s.h_0 * T(0.67);
            maxiter += 25;
            s.qj_0 = T(0);
            iter = 0;
            if (s.rel_open && s.aerror >= mindepth && iter <= maxiter) {
                s.at_h0 = hydraulics_at<T, M>(s.h_0, p, c, m);
                any_over = any_over || s.at_h0.over;
            }
            continue;
        }
        break;
    }

    const MuskCoef<T> &k = s.k;
    const T w3 = (k.C1 * f.qup) + (k.C2 * f.quc) + (k.C3 * f.qdp);
    {   // (f90:126-141; all three candidates are a few additions: formed, then selected)
        const T q_neg = mc_max(((k.C1 * f.qup) + (k.C2 * f.quc) + k.C4), ((k.C1 * f.qup) + (k.C3 * f.qdp) + k.C4));
        const T q_lo = ((k.C4 < T(0)) && (mc_abs(k.C4) > w3)) ? T(0) : q_neg;
        out.qdc = ((w3 + k.C4) < T(0)) ? q_lo : w3 + k.C4;
    }
    out.h = s.h;
    out.X = k.X;
    out.iters = total_iter;
    out.over = any_over;
    return out;
}

// velocity at the depth the iteration ended on, from the trapezoid-only hydraulic radius (f90:163-169)
template <class T, class M>
MC_HD T step_velocity(const ChannelParams<T> &p, const ChannelConst<T> &c, T h, const M &m)
{
    const T twl = p.bw + T(2) * c.z * h;
    const T a = (twl - p.bw) / T(2);
    // (numerator in [2**-44, 2**52], denominator in [2**-14, 2**36]: see fast_ok; one uniform branch, two straight-line bodies)
    if (m.all(m.fast_ok(h, h, T(0)))) {
        // (a = z h up to rounding: a*a + h*h lies in [2**-60, 2**63], an ordinary number)
        const T R = m.div1(h * (p.bw + twl) / T(2), p.bw + T(2) * m.sqrt_r(a * a + h * h, true), true);
        return c.inv_n * m.pow_l_r(m.log_of_r(R, true), R, T(2) / T(3), true) * c.sqrt_s0;
    }
    const T R = m.div1(h * (p.bw + twl) / T(2), p.bw + T(2) * m.sqrt(a * a + h * h), false);
    return c.inv_n * m.pow_l_r(m.log_of_r(R, false), R, T(2) / T(3), false) * c.sqrt_s0;
}

// One segment, one timestep (f90:8-186), with the segment-invariant constants supplied.
// want_velocity = false: velc is left 0 -- for callers that hand nobody this step's velocity (it feeds nothing: the next step
// starts from the flow and the depth, f90:8-186 takes velp and does not read it).
template <class T, class M>
MC_HD StepResult<T> mc_segment_step(const ChannelParams<T> &p, const ChannelConst<T> &c, const Inflow<T> &f,
                                    T depthp, const M &m, bool want_velocity = true)
{
    StepResult<T> out;
    if (!step_has_flow(f)) {
        T h, h_0;
        step_bracket(depthp, h, h_0);
        out.qdc = T(0); out.velc = T(0); out.depthc = T(0); out.h = h; out.X = T(0); out.iters = 0;
        out.over = false;
        return out;
    }
    StepPre<T> none;
    none.have = false;
    const StepSolve<T> s = step_solve<T, M>(p, c, f, depthp, none, m);
    out.qdc = s.qdc;
    out.velc = want_velocity ? step_velocity<T, M>(p, c, s.h, m) : T(0);
    out.depthc = s.h;
    out.h = s.h;
    out.X = s.X;
    out.iters = s.iters;
    out.over = s.over;
    return out;
}

// One segment, one timestep, constants formed on the spot.
template <class T, class M>
MC_HD StepResult<T> mc_segment_step(const ChannelParams<T> &p, const Inflow<T> &f, T depthp, const M &m)
{
    return mc_segment_step<T, M>(p, make_const<T, M>(p, m), f, depthp, m);
}

} // namespace trmc
