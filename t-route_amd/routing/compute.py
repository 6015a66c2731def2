"""Drop-in for the Muskingum-Cunge part of ``troute.routing.compute``.

``compute_nhd_routing_v02`` keeps the reference's signature and return value
(src/troute-routing/troute/routing/compute.py:507-546; results = a list of the
kernel callable's tuples, one per independent network, consumed by
``AbstractNetwork.new_q0`` (AbstractNetwork.py:177-191) and the output writers
(output.py:209-216)).

What is different, on purpose (MI355X-first, SURVEY.md 8b "Threading"):
  * the reference fans the independent networks out to joblib/loky worker
    processes and, for the big basins, to ordered sub-networks that hand
    tailwater hydrographs from one order to the next
    (compute.py:553-1209, :1211-1395).  Here ALL networks of the call are
    flattened into ONE device plan and routed by one sequence of launches: on a
    GPU the levels of every network run side by side, so neither processes nor
    sub-network orders are needed.  Every ``parallel_compute_method`` the
    reference accepts is therefore accepted and produces the same per-network
    result list (``cpu_pool`` and ``subnetwork_target_size`` are ignored).
  * the per-call argument preparation of the reference (pandas ``.loc`` slicing
    per network, compute.py:1399-1467) is done once for the whole table.
Out of scope (raise NotImplementedError): reservoirs / waterbodies, gage
nudging and every other data-assimilation input, the diffusive branch.
"""
from collections import defaultdict

import numpy as np

from .fast_reach.mc_reach import _KERNEL_COLS, compute_network_structured  # noqa: F401
from ..plan import RoutingPlan, csr_from_lists

_compute_func_map = defaultdict(
    lambda: compute_network_structured,
    {
        "V02-structured": compute_network_structured,   # the reference's only key (compute.py:21-26)
        "V02-hip": compute_network_structured,
    },
)

_PARALLEL_METHODS = ("serial", "by-network", "by-subnetwork", "by-subnetwork-jit",
                     "by-subnetwork-jit-clustered", "bmi")


def _is_empty(df):
    return df is None or getattr(df, "empty", True) or len(df) == 0


def compute_nhd_routing_v02(
    connections,
    rconn,
    wbody_conn,
    reaches_bytw,
    compute_func_name,
    parallel_compute_method,
    subnetwork_target_size,
    cpu_pool,
    t0,
    dt,
    nts,
    qts_subdivisions,
    independent_networks,
    param_df,
    q0,
    qlats,
    usgs_df,
    lastobs_df,
    reservoir_usgs_df,
    reservoir_usgs_param_df,
    reservoir_usace_df,
    reservoir_usace_param_df,
    reservoir_rfc_df,
    reservoir_rfc_param_df,
    great_lakes_df,
    great_lakes_param_df,
    great_lakes_climatology_df,
    da_parameter_dict,
    assume_short_ts,
    return_courant,
    waterbodies_df,
    data_assimilation_parameters,
    waterbody_types_df,
    waterbody_type_specified,
    subnetwork_list,
    flowveldepth_interorder={},
    from_files=True,
    *,
    precision=32,
    device=0,
):
    """Route every independent network of the call for ``nts`` timesteps.

    Returns the reference's ``results`` list: one 10-tuple per tailwater of ``reaches_bytw`` (in its
    iteration order), each shaped like ``compute_network_structured``'s return
    (mc_reach.pyx:811-845): ``(ids, flowveldepth[n, nts*3], 0, (..), (..), (..), upstream[n, nts],
    (..), nudge, (..))``.
    """
    if parallel_compute_method not in _PARALLEL_METHODS and parallel_compute_method is not None:
        raise ValueError(f"unknown parallel_compute_method {parallel_compute_method!r}")
    for name, df in (("waterbodies_df", waterbodies_df), ("usgs_df", usgs_df), ("lastobs_df", lastobs_df),
                     ("reservoir_usgs_df", reservoir_usgs_df), ("reservoir_usace_df", reservoir_usace_df),
                     ("reservoir_rfc_df", reservoir_rfc_df), ("great_lakes_df", great_lakes_df)):
        if not _is_empty(df):
            raise NotImplementedError(
                f"{name} is not empty: reservoirs and data assimilation are outside the Muskingum-Cunge "
                "hot path of this engine (run MC-only: break_network_at_waterbodies False, DA off)")
    if flowveldepth_interorder:
        raise NotImplementedError("flowveldepth_interorder hand-off is only needed by the reference's "
                                  "sub-network orders; call compute_network_structured for that")

    # compute.py:548-549
    param_df = param_df.copy()
    param_df["dt"] = dt
    param_df = param_df.astype("float32").sort_index()
    ids = param_df.index.values.astype("int64")
    nseg = ids.shape[0]
    params = np.ascontiguousarray(param_df[list(_KERNEL_COLS)].values, dtype=np.float32)
    q0_v = np.ascontiguousarray(q0.loc[param_df.index].values, dtype=np.float32)
    qlat_v = np.ascontiguousarray(qlats.loc[param_df.index].values, dtype=np.float32)
    if qlat_v.shape[1] < nts / qts_subdivisions:
        raise ValueError(
            f"Number of columns (timesteps) in Qlat is incorrect: expected at most ({nseg}), got "
            f"({qlat_v.shape[1]}). The number of columns in Qlat must be equal to or less than the number "
            "of routing timesteps")

    # one upstream list per row, reference order: head of reach <- independent_networks[tw][head],
    # inside a reach <- previous segment (mc_reach.pyx:288-289, :133-138)
    ups = [()] * nseg
    owner = np.full(nseg, -1, dtype=np.int64)
    tws = list(reaches_bytw.keys())
    flat_ids, flat_kind, head_lists = [], [], []
    for k, tw in enumerate(tws):
        net = independent_networks[tw]
        for reach in reaches_bytw[tw]:
            flat_ids.extend(reach)
            flat_kind.extend([k] * len(reach))
            head_lists.append((reach, net.get(reach[0], ())))
    flat_ids = np.asarray(flat_ids, dtype=np.int64)
    rows = np.searchsorted(ids, flat_ids)
    if rows.size and ((rows >= nseg).any() or (ids[np.minimum(rows, nseg - 1)] != flat_ids).any()):
        bad = flat_ids[(rows >= nseg) | (ids[np.minimum(rows, nseg - 1)] != flat_ids)][0]
        raise ValueError(f"element {bad} not found in {ids}")
    owner[rows] = np.asarray(flat_kind, dtype=np.int64)
    row_of = dict(zip(flat_ids.tolist(), rows.tolist()))
    pos = 0
    for reach, head_ups in head_lists:
        r0 = rows[pos]
        lst = []
        for u in head_ups:
            if u not in row_of:
                raise ValueError(f"element {u} not found in {ids}")
            lst.append(row_of[u])
        ups[r0] = lst
        for j in range(1, len(reach)):
            ups[rows[pos + j]] = (rows[pos + j - 1],)
        pos += len(reach)
    if (owner < 0).any():
        raise ValueError("param_df contains segments that are in no reach of reaches_bytw")

    up_ptr, up_idx = csr_from_lists(ups)
    with RoutingPlan(up_ptr, up_idx, params, None, precision, device) as plan:
        fvd = plan.route(nts, qts_subdivisions, assume_short_ts, qlat_v, q0_v)
    fvd = fvd.reshape(nseg, nts * 3)

    e_f1 = np.zeros(0, dtype="float32")
    e_i1 = np.zeros(0, dtype="int32")
    results = []
    for k in range(len(tws)):
        sel = np.flatnonzero(owner == k)
        results.append((
            ids[sel].astype(np.intp),
            fvd[sel],
            0,
            (np.asarray([], dtype=np.int64), np.full(0, np.nan, "float32"), np.full(0, np.nan, "float32")),
            (e_i1, e_f1, e_f1, e_f1, e_f1),
            (e_i1, e_f1, e_f1, e_f1, e_f1),
            np.zeros((sel.shape[0], nts), dtype="float32"),
            (e_i1, e_f1, e_i1),
            np.zeros((0, nts + 1), dtype="float32"),
            (e_i1, e_f1, e_i1, e_i1),
        ))
    return results


def new_q0(run_results, index=None):
    """State for the next routing window from a results list: ``[qu0, qd0, h0] = fvd[:, [-3,-3,-1]]``
    (AbstractNetwork.new_q0, AbstractNetwork.py:177-191), as a DataFrame indexed by segment id."""
    import pandas as pd
    ids = np.concatenate([r[0] for r in run_results])
    vals = np.concatenate([r[1][:, [-3, -3, -1]] for r in run_results])
    df = pd.DataFrame(vals, index=ids, columns=["qu0", "qd0", "h0"])
    return df if index is None else df.loc[index]
