#!/bin/bash
# Same-box A/B of variant builds of the product library (make -C libmultiviewnative_amd/csrc variant NAME=x EXTRA=-D..):
#   tools/ab_libs.sh [bench.py arguments --] name1 name2 ...      ("-" = the product library itself)
# runs the headline bench on each library in turn, the product first and last, and prints ms per step and the
# per-kernel event times of each run.
cd "$(dirname "$0")/.."
ARGS="--no-abi --no-cpu-baseline --no-side --steps 20 --warmup 3"
if [[ " $* " == *" -- "* ]]; then ARGS="${*%% -- *}"; set -- ${*#* -- }; fi
for v in - "$@" -; do
  so=libmultiviewnative_amd/lib/libmultiviewnative.so
  [ "$v" != "-" ] && so=libmultiviewnative_amd/lib/libmultiviewnative_$v.so
  echo "== $v"
  MVN_PRODUCT_SO=$PWD/$so timeout -k 10 300 python bench.py $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  %.3f ms/step  %.2f it/s ' % (d['ms_per_step'], d['value']), {k: round(v['avg_ms'], 4) for k, v in ((d.get('roofline') or {}).get('per_kernel') or {}).items()})"
done
