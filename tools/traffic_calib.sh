#!/bin/bash
# FETCH_SIZE / WRITE_SIZE factors for the routing kernels' access patterns (tools/traffic_calib.hip): two counter-only passes.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/traffic_calib
rm -rf "$out"; mkdir -p "$out" tools/_bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/_bin/traffic_calib tools/traffic_calib.hip || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d "$out/$c" -o p -- tools/_bin/traffic_calib > "$out/$c.log" 2>&1
done
python - "$out" <<'P'
import glob, json, os, sqlite3, sys
out = sys.argv[1]
n = (512 << 20) * 4
known = {"read4": n, "read16": n, "write4": n, "write96": ((512 << 20) // 864) * 96}
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    db = glob.glob(os.path.join(out, c, "**", "*.db"), recursive=True)[0]
    con = sqlite3.connect(db)
    for name, val, cnt in con.execute(
            "select s.kernel_name, sum(e.value), count(distinct d.id) from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
            "join rocpd_pmc_event e on e.event_id = d.event_id join rocpd_info_pmc p on e.pmc_id = p.id where p.name = ? group by s.kernel_name", (c,)):
        k = next((x for x in known if x in name), None)
        if k:
            res.setdefault(k, {})[c + "_KB_per_launch"] = val / cnt
for k, v in res.items():
    v["known_bytes_per_launch"] = known[k]
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        if v.get(c + "_KB_per_launch"):
            v["bytes_per_" + c + "_KB"] = known[k] / v[c + "_KB_per_launch"]
print(json.dumps(res, indent=1))
json.dump(res, open(os.path.join("gpurun_out", "traffic_calibration.json"), "w"), indent=1)
P
find "$out" -name '*.db' -delete
