#!/bin/bash
# Packed Nyquist layout (MVN_NYQ_PACKED=1) against the split layout whose Nyquist lines ride in the dim1 launches
# (MVN_NYQ_PACKED=0) over a ladder of cube edges: one view update, PSF edge $PSF (default 15), same box.
cd "$(dirname "$0")/.."
PSF=${PSF:-15}
for n in "$@"; do
  for p in 1 0 1 0; do
    echo -n "edge $n packed=$p  "
    MVN_NYQ_PACKED=$p python tools/shape_probe.py $n $n $n $PSF 2>/dev/null | grep view-iter | cut -c1-260
  done
done
