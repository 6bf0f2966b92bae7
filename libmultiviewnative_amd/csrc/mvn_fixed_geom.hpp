// mvn_fixed_geom.hpp -- host-visible launch geometry of the fixed-length kernels (mvn_fixed.hpp)
#pragma once

#include <cstddef>

#include "mvn_fixed.hpp"

namespace mvn {

#define MVN_FIXED_STRIDED_LENGTHS(X) X(64) X(128) X(256) X(512) X(1024)
#define MVN_FIXED_ROWS_LENGTHS(X) X(32) X(64) X(128) X(256) X(512) X(1024)

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
