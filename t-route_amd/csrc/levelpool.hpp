// levelpool.hpp -- level-pool reservoir step, precision-generic device function.
//
// Semantics: LEVELPOOL_PHYSICS, src/kernel/reservoir/Level_Pool/module_levelpool.F:233-427
// (LAKE_OPT 2, third-order Runge-Kutta on a vertically walled pool with an orifice, a weir and a dam
// crest), reached from the network loop through run_lp_c -> route -> run_lp
// (src/troute-network/troute/network/reservoirs/levelpool/levelpool.pyx:23-24,
//  src/kernel/reservoir/bind_lp.f90), which hands the current inflow over as the previous one too.
// Same operations and order as the reference's fp32 object code: powf for **(3./2.), sqrtf for the
// orifice law (Math policy: mc_segment.hpp).
#pragma once
#include "mc_segment.hpp"

namespace trmc {

template <class T> struct LevelPoolParams {
    T area, max_depth, orifice_area, orifice_coefficient, orifice_elevation, weir_coefficient, weir_elevation,
        weir_length, dam_length;
};

// storage discharge with the pool at h_eval; the overtopping test uses the elevation at the start of
// the step (h_top), as the reference does in every Runge-Kutta stage
template <class T, class M>
MC_HD T levelpool_discharge(T h_eval, T h_top, T max_weir_depth, const LevelPoolParams<T> &p, const M &m)
{
    T dh = h_eval - p.weir_elevation;
    if (dh > max_weir_depth) dh = max_weir_depth;
    const T orifice = p.orifice_coefficient * p.orifice_area * m.sqrt(T(2.) * T(9.81) * (h_eval - p.orifice_elevation));
    const T weir = p.weir_coefficient * p.weir_length * m.pow(dh, T(3.) / T(2.));
    if (h_top > p.max_depth)
        return orifice + weir
               + (p.weir_coefficient * (p.weir_length * p.dam_length) * m.pow(h_top - p.max_depth, T(3.) / T(2.)));
    if (dh > T(0)) return orifice + weir;
    if (h_eval > p.orifice_elevation) return orifice;
    return T(0);
}

// one routing period: water elevation H in/out, returns the outflow
template <class T, class M>
MC_HD T levelpool_step(T inflow, T lateral, T dt, T &H, const LevelPoolParams<T> &p, const M &m)
{
    const T qi0 = inflow, qi1 = inflow;
    const T it_0 = qi0;
    const T it_1 = qi0 + ((qi1 + lateral - qi0) * T(0.33));
    const T it_2 = qi0 + ((qi1 + lateral - qi0) * T(0.67));
    const T mwd = p.max_depth - p.weir_elevation;
    const T sap = p.area * T(1.0E6);
    const T h = H;
    T q = levelpool_discharge<T, M>(h, h, mwd, p, m);
    const T dh1 = (sap > T(0)) ? ((it_0 - q) / sap) * dt : T(0);
    q = levelpool_discharge<T, M>(h + dh1 / T(3), h, mwd, p, m);
    const T dh2 = (sap > T(0)) ? ((it_1 - q) / sap) * dt : T(0);
    q = levelpool_discharge<T, M>(h + (T(0.667) * dh2), h, mwd, p, m);
    const T dh3 = (sap > T(0)) ? ((it_2 - q) / sap) * dt : T(0);
    H = h + ((dh1 / T(4.)) + (T(0.75) * dh3));
    return levelpool_discharge<T, M>(H, H, mwd, p, m);
}

} // namespace trmc
