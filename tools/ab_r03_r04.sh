#!/bin/bash
# Same-box comparison of round 3's build (lib/libmultiviewnative_r03.so, built from commit e581fb9) with the current
# one, with and without the Nyquist riders: short runs (20 steps, per-kernel events) and sustained ones (600 steps,
# no events: the loop runs at the package power cap, where only energy per iteration counts).
cd "$(dirname "$0")/.."
L=$PWD/libmultiviewnative_amd/lib
run() {  # $1 label, $2 library, $3 extra env, $4.. bench arguments
  local label=$1 so=$2 extra=$3; shift 3
  echo -n "== $label  "
  env MVN_PRODUCT_SO=$so $extra timeout -k 10 300 python bench.py "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.3f ms/step  %.2f it/s ' % (d['ms_per_step'], d['value']), {k: round(v['avg_ms'], 4) for k, v in ((d.get('roofline') or {}).get('per_kernel') or {}).items()})"
}
SHORT="--no-abi --no-cpu-baseline --no-side --steps 20 --warmup 3"
LONG="--no-abi --no-cpu-baseline --no-side --no-profile --steps 600 --warmup 3"
for rep in 1 2; do
  run "r03 short      " $L/libmultiviewnative_r03.so "A=1" $SHORT
  run "r04 short      " $L/libmultiviewnative.so "A=1" $SHORT
  run "r04 no riders  " $L/libmultiviewnative.so "MVN_NYQ_RIDE=0" $SHORT
done
for rep in 1 2; do
  run "r03 sustained      " $L/libmultiviewnative_r03.so "A=1" $LONG
  run "r04 sustained      " $L/libmultiviewnative.so "A=1" $LONG
  run "r04 no riders sust." $L/libmultiviewnative.so "MVN_NYQ_RIDE=0" $LONG
done
