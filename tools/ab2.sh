#!/bin/bash
# A/B of library variants on the headline window only: tools/ab2.sh "" _x ...  (suffixes of t-route_amd/libtrmc<suffix>.so)
cd "${GRAFT_REPO_ROOT:-.}"
for v in "$@"; do
  L=$PWD/t-route_amd/libtrmc$v.so
  TRMC_LIB_PATH=$L python bench.py --steps 5 --no-cpu-baseline --no-full-ts --no-diffusive --no-parity-mode --no-traffic ${AB_ARGS:-} 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('variant %-8s value(with d2h) %.3e ms/step %.2f | resident ms/step %.2f ms_main %.2f | frac %.3f untuned_ms %.2f cold_ms %.2f launches %d parity %s' % ('$v', d['value'], d['ms_per_step'], d['value_resident']['ms_per_step'], d['value_resident']['ms_main'], r['frac'], d['untuned']['ms_main'], d.get('cold_start',{}).get('ms_main',0), r['launches_per_step'], (d.get('parity_sample') or {}).get('bit_identical')))"
done
