// mvn_fixed_geom.hpp -- host-visible launch geometry of the fixed-length kernels (mvn_fixed.hpp)
#pragma once

#include <cstddef>

#include "mvn_fixed.hpp"

namespace mvn {

// power-of-two lengths plus the common 2^a 3^b 5^c sizes (1920 x 1920 x 320 SPIM stacks, "good"
// padded sizes: 96, 160, 288, 576 are where 64-, 128-, 256- and 512-blocks land with a 31-tap PSF);
// every length here costs three kernel instantiations
#define MVN_FIXED_STRIDED_LENGTHS(X) \
  X(64) X(128) X(256) X(512) X(1024) X(96) X(160) X(192) X(288) X(320) X(384) X(576) X(640) X(768) X(960) X(1280) X(1920)
// H = d2 / 2
#define MVN_FIXED_ROWS_LENGTHS(X) \
  X(32) X(64) X(128) X(256) X(512) X(1024) X(48) X(80) X(96) X(144) X(160) X(192) X(288) X(320) X(384) X(480) X(640) X(768) X(960)

inline bool fixed_strided_geom(int n, int* T, int* threads, size_t* lds_bytes) {
  switch (n) {
#define X(N)                                                        \
  case N:                                                           \
    *T = FxStridedCfg<N>::T;                                        \
    *threads = FxStridedCfg<N>::NT;                                 \
    *lds_bytes = sizeof(cfloat) * (size_t)FxStridedCfg<N>::lds_cfloats; \
    return true;
    MVN_FIXED_STRIDED_LENGTHS(X)
#undef X
    default: return false;
  }
}

inline bool fixed_rows_geom(int h, int* T, int* threads, size_t* lds_bytes) {
  switch (h) {
#define X(H)                                                     \
  case H:                                                        \
    *T = FxRowsCfg<H>::T;                                        \
    *threads = FxRowsCfg<H>::NT;                                 \
    *lds_bytes = sizeof(cfloat) * (size_t)FxRowsCfg<H>::lds_cfloats; \
    return true;
    MVN_FIXED_ROWS_LENGTHS(X)
#undef X
    default: return false;
  }
}

}  // namespace mvn
