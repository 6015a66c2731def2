#!/usr/bin/env python3
"""Idle time between consecutive step-kernel dispatches in a rocprofv3 rocpd database (launch-gap check)."""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select s.kernel_name, d.start, d.end, d.queue_id, d.stream_id from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
last = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rows = rows[-last:]
t0 = rows[0][1]
prev_end = None
for name, st, en, q, sid in rows:
    short = name.split("EEv")[0][-28:]
    gap = (st - prev_end) / 1e3 if prev_end else 0
    print(f"{(st - t0) / 1e3:10.1f} us  dur {(en - st) / 1e3:8.1f}  gap {gap:8.1f}  q{q} s{sid}  {short}")
    prev_end = en if prev_end is None or en > prev_end else prev_end
