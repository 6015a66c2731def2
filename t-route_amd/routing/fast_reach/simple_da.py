"""Streamflow nudging -- host side of the drop-in for ``troute.routing.fast_reach.simple_da``.

Reference: src/troute-routing/troute/routing/fast_reach/simple_da.pyx
  obs_persist_shift :98-118, simple_da_with_decay :85-96, simple_da :22-95,
and the hook in the network loop, mc_reach.pyx:380-411 (setup) and :761-796 (per reach, per step).

The scalar functions below keep the reference's arithmetic (float32 everywhere, the decay weight
through libc's double ``exp``) and exist for callers/tests of the reference's Python entry
(``simple_da_with_decay_py``).  The engine itself does not run them per step: which branch of
``simple_da`` is taken at (gage, timestep) depends on the observation record alone, never on the
modelled flow, so ``resolve_tables`` decides it once per routing window on the host and the GPU
only applies  flow := a  or  flow := model + (a - model) * w  (include/trmc.h, trmc_set_nudging).
"""
import math

import numpy as np

F = np.float32


def obs_persist_shift(last_valid_obs, model_val, minutes_since_last_valid, decay_coeff):
    da_weight = F(math.exp(abs(float(F(minutes_since_last_valid))) / float(-F(decay_coeff))))
    return F(F(last_valid_obs) - F(model_val)) * da_weight


def simple_da_with_decay_py(last_valid_obs, model_val, minutes_since_last_valid, decay_coeff):
    return F(model_val) + obs_persist_shift(last_valid_obs, model_val, minutes_since_last_valid, decay_coeff)


def simple_da(timestep, routing_period, decay_coeff, gage_maxtimestep, target_val, model_val, lastobs_time,
              lastobs_val):
    """(replacement_val, nudge_val, lastobs_time, lastobs_val), simple_da.pyx:22-95."""
    timestep, routing_period, target_val, model_val = F(timestep), F(routing_period), F(target_val), F(model_val)
    lastobs_time, lastobs_val = F(lastobs_time), F(lastobs_val)
    if timestep <= F(gage_maxtimestep) and not np.isnan(target_val):
        return target_val, F(target_val - model_val), F(timestep * routing_period), target_val
    if np.isnan(target_val) and np.isnan(lastobs_val):
        return model_val, F(0.0), F(np.nan), F(np.nan)
    minutes = F(F(timestep * routing_period - lastobs_time) / F(60))
    shift = obs_persist_shift(lastobs_val, model_val, minutes, decay_coeff)
    return F(model_val + shift), shift, lastobs_time, lastobs_val


def resolve_tables(nsteps, routing_period, decay_coeff, usgs_values, lastobs_values_init, time_since_lastobs_init):
    """Decide simple_da's branch for every (gage, step) of a window.

    usgs_values [ngage, gage_maxtimestep] float32 (NaN = missing), indexed by routing timestep as the
    reference does (``usgs_values[gage_i, timestep]``, mc_reach.pyx:777).
    Returns mode uint8 [ngage, nsteps], a float32, w float32, and the final (lastobs_times, lastobs_values).
    """
    usgs_values = np.asarray(usgs_values, dtype=F)
    ngage = int(np.asarray(lastobs_values_init).shape[0])
    gmax = usgs_values.shape[1] if usgs_values.ndim == 2 else 0
    lt = np.array(time_since_lastobs_init, dtype=F, copy=True)
    lv = np.array(lastobs_values_init, dtype=F, copy=True)
    mode = np.zeros((ngage, nsteps), dtype=np.uint8)
    a = np.zeros((ngage, nsteps), dtype=F)
    w = np.zeros((ngage, nsteps), dtype=F)
    rp, neg_decay = F(routing_period), float(-F(decay_coeff))
    for t in range(1, nsteps + 1):
        target = usgs_values[:, t] if t < gmax else np.full(ngage, np.nan, dtype=F)   # NaN once t >= gmax
        now = F(F(t) * rp)
        replace = (F(t) <= F(gmax)) & ~np.isnan(target)
        passthru = ~replace & np.isnan(target) & np.isnan(lv)
        decay = ~replace & ~passthru
        mode[replace, t - 1] = 1
        a[replace, t - 1] = target[replace]
        lt[replace] = now
        lv[replace] = target[replace]
        lt[passthru] = np.nan
        lv[passthru] = np.nan
        if decay.any():
            minutes = ((now - lt[decay]).astype(F) / F(60)).astype(F)
            mode[decay, t - 1] = 2
            a[decay, t - 1] = lv[decay]
            w[decay, t - 1] = np.array([math.exp(abs(float(m)) / neg_decay) for m in minutes], dtype=F)
    return mode, a, w, lt, lv
