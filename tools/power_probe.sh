#!/bin/bash
# Clock and power of the GPU while the RL loop runs (is the loop power-limited?): samples rocm-smi every 0.2 s
# beside `bench.py --steps N`.   tools/power_probe.sh [VAR=value ...]
cd "$(dirname "$0")/.."
for kv in "$@"; do export "$kv"; done
( for i in $(seq 1 60); do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | python3 -c "
import re, sys
t = sys.stdin.read()
g = lambda pat: (re.search(pat, t) or [None, '?'])[1]
print('sclk %s MHz  power %s W  junction %s C' % (g(r'sclk clock level: \S+ \((\d+)Mhz\)'), g(r'Power \(W\): ([0-9.]+)'), g(r'junction\) \(C\): ([0-9.]+)')))
"; sleep 0.2; done ) > /tmp/smi.log 2>&1 &
SMI=$!
python bench.py --no-side --no-cpu-baseline --steps 600 --warmup 3 --no-profile 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        o = json.loads(l); print('  %.3f ms/step  %.2f it/s' % (o['ms_per_step'], o['value']))
"
wait $SMI
awk 'NR % 3 == 0' /tmp/smi.log | head -16 | tr '\n' ';'; echo
