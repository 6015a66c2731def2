#!/bin/bash
# A/B timing of run-time knobs on the GPU box: tools/ab_env.sh "VAR=a VAR2=b" "VAR=c" ...  (one headline-only bench per setting;
# TRMC_LIB_PATH=<variant library> is a knob like any other)
cd "${GRAFT_REPO_ROOT:-.}"
for v in "$@"; do
  echo "== $v"
  env $v python bench.py --steps 5 --warmup 1 --headline-only --no-traffic --no-parity-sample 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.2f ms_main %.2f' % (d['ms_per_step'], d['ms_main']))"
done
