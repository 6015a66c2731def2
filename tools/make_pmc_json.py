#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the counter passes of tools/profile_round.sh (rocpd databases -> tools/rocpd_summary.py
--json): HBM bytes per launch of the dominant routing kernel = 2 x FETCH_SIZE (gfx950: FETCH_SIZE reports half of streamed
reads, MI355X_MICROARCH.md "HBM"; checked on k_emit whose byte count is known) + WRITE_SIZE, in KB units of the counters.
    python tools/make_pmc_json.py <summary.json> <engine> <round> > profiles/pmc_traffic.json"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = json.load(open(sys.argv[1]))
engine, rnd = sys.argv[2], int(sys.argv[3])
pat = "k_mc_step" if engine == "levels" else "k_mc_flow"
best, c = None, {}
for db in res:
    for k in db.get("kernels", []):
        if pat in k["name"] and (best is None or k["total_ms"] > best["total_ms"]):
            best = k
for db in res:
    for e in db.get("counters", []):
        if best and e["kernel"] == best["name"]:
            c[e["counter"]] = e["mean"]
h = hashlib.sha256()
for f in ("trmc.hip", "mc_segment.hpp", "det_pow.h", "levelpool.hpp"):
    h.update(open(os.path.join(ROOT, "t-route_amd", "csrc", f), "rb").read())
out = {"round": rnd, "engine": engine, "kernel": best["name"] if best else None,
       "avg_launch_us": best["avg_us"] if best else None, "calls": best["calls"] if best else None,
       "fetch_size_kb_per_launch": c.get("FETCH_SIZE"), "write_size_kb_per_launch": c.get("WRITE_SIZE"),
       "fetch_correction": 2.0,
       "hbm_bytes_per_launch": (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 if "FETCH_SIZE" in c and "WRITE_SIZE" in c else None,
       "valu_instructions_per_launch": c.get("SQ_INSTS_VALU"),
       "source_sha256": h.hexdigest(),
       "source": "tools/profile_round.sh: separate rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU, --kernel-trace only) of the default bench command"}
print(json.dumps(out, indent=1))
