#!/bin/bash
# registers / scratch / occupancy of the device kernels whose (demangled) name matches a pattern
#   tools/kernel_regs.sh [pattern]        e.g. tools/kernel_regs.sh 'kw_rows|kx_strided<512'
cd "$(dirname "$0")/../libmultiviewnative_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -fno-fast-math -fno-slp-vectorize \
  $MVN_EXTRA_FLAGS -S --cuda-device-only -o /tmp/mvn_kernels.s mvn_kernels.hip 2>/dev/null
awk '/^[_A-Za-z0-9]+:/ {name=$1; sub(":", "", name)}
     /; NumVgprs:/ {v=$3} /; ScratchSize:/ {s=$3}
     /; Occupancy:/ {print name, "vgprs", v, "scratch", s, "occupancy", $3}' /tmp/mvn_kernels.s \
  | c++filt | grep -E "${1:-.}" | sort -u
