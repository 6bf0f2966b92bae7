#!/bin/bash
# A/B of run-time knobs on the sharded step's per-rank compute (bench.py --simultaneous, 512^3 x 6 views):
# every argument is one environment setting (VAR=value, several joined by commas), "-" = defaults
cd "$(dirname "$0")/.."
for cfg in "$@"; do
  echo "== $cfg"
  ( if [ "$cfg" != "-" ]; then IFS=, ; for kv in $cfg; do export "$kv"; done; unset IFS; fi
    python bench.py --simultaneous --no-side --no-cpu-baseline --steps ${AB_STEPS:-20} --warmup 3 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        o = json.loads(l)
        pk = o['roofline']['per_kernel']
        print('  %.3f ms/step  %.2f it/s ' % (o['ms_per_step'], o['value']), {k: v['avg_ms'] for k, v in pk.items()})
" )
done
