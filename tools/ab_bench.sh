#!/bin/bash
# A/B of run-time knobs with the bench itself (512^3 x 6 views, sampled per-kernel events): every
# argument is one environment setting (VAR=value, several joined by commas), "-" = defaults
cd "$(dirname "$0")/.."
for cfg in "$@"; do
  echo "== $cfg"
  ( if [ "$cfg" != "-" ]; then IFS=, ; for kv in $cfg; do export "$kv"; done; unset IFS; fi
    python bench.py --no-side --no-cpu-baseline --steps ${AB_STEPS:-20} --warmup 3 $AB_ARGS 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        o = json.loads(l)
        pk = (o.get('roofline') or {}).get('per_kernel') or {}
        print('  %.3f ms/step  %.2f it/s ' % (o['ms_per_step'], o['value']), {k: v['avg_ms'] for k, v in pk.items()})
" )
done
