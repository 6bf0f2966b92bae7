#!/bin/bash
# A/B of run-time knobs on the same box: every argument is one environment setting (VAR=value, or
# several joined by commas); prints per-kernel times of one view update at 512^3 (tools/sweep.py)
cd "$(dirname "$0")/.."
for cfg in "-" "$@" "-"; do
  echo "== $cfg"
  ( if [ "$cfg" != "-" ]; then IFS=, ; for kv in $cfg; do export "$kv"; done; fi
    python tools/sweep.py "" 2>&1 | grep view-iter )
done
