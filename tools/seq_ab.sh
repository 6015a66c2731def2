#!/bin/bash
# A/B of the timed sequence (bench.py --headline-only) under run-time settings: tools/seq_ab.sh "VAR=a" "VAR=b VAR2=c" ...
cd "${GRAFT_REPO_ROOT:-.}"
for v in "$@"; do
  echo "== $v"
  env $v python bench.py --steps 9 --warmup 1 --headline-only --no-traffic --no-parity-full 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.2f ms_main %.2f' % (d['ms_per_step'], d['ms_main']))"
done
