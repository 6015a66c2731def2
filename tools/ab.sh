#!/bin/bash
# A/B timing of library variants on the GPU box: tools/ab.sh "" _e0 _e1 ...  (suffixes of t-route_amd/libtrmc<suffix>.so)
cd "${GRAFT_REPO_ROOT:-.}"
for v in "$@"; do
  L=$PWD/t-route_amd/libtrmc$v.so
  echo "== variant '$v'"
  TRMC_LIB_PATH=$L python bench.py --steps 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value %.3e ms/step %.2f ms_main %.2f avg_launch_us %.1f full_ts %.3e ms_main_full %.2f' % (d['value'], d['ms_per_step'], d['roofline']['ms_main'], d['roofline']['avg_launch_ms']*1e3, d['full_ts']['value'], d['full_ts']['ms_main']))"
done
