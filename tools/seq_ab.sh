#!/bin/bash
# A/B of the timed sequence (bench.py --headline-only) under run-time settings: tools/seq_ab.sh "VAR=a" "VAR=b VAR2=c" ...
# (an argument that starts with "--" goes to bench.py instead: tools/seq_ab.sh "--persistence 0.5" ...)
cd "${GRAFT_REPO_ROOT:-.}"
for v in "$@"; do
  echo "== $v"
  if [[ "$v" == --* ]]; then extra="$v"; envs=""; else extra=""; envs="$v"; fi
  env $envs python bench.py --steps ${SEQ_STEPS:-9} --warmup 1 --headline-only --no-traffic --no-parity-full $extra 2>/dev/null | tail -1 | python -c '
import sys, json
d = json.loads(sys.stdin.read())
print("ms/step %.2f ms_main %.2f days %s" % (d["ms_per_step"], d["ms_main"], d.get("day_ms")))'
done
