// mvn_plan.hpp -- host-side FFT plan construction (pure C++, no HIP).
//
// Counterpart of the reference's per-shape plan cache (inc/plan_store.cuh:20-216,
// inc/plan_store.h:25-235): a plan is everything a pass kernel needs that depends only on
// the logical shape (d0,d1,d2) -- radix schedule, digit-reversal tables, twiddles, tile and
// launch geometry.
//
// Device data layout ("split-Nyquist", even d2): the real volume is [d0][d1][d2] floats,
// unpadded, and doubles in place as the half-spectrum [d0][d1][d2/2] complex (bins
// 0..d2/2-1 of the last axis); the Nyquist bin d2/2 of every row lives in a separate
// [d0][d1] complex plane.  Together that is exactly the reference's in-place r2c footprint
// 4*d0*d1*2(d2/2+1) bytes (inc/image_stack_utils.h:24-42), but every row stays aligned and
// power-of-two extents stay power-of-two.  Odd d2 falls back to a padded row of (d2+1) floats
// = (d2+1)/2 complex bins and has no Nyquist plane.
#pragma once

#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "mvn_fft_core.hpp"

namespace mvn {

// next length >= n whose only prime factors are 2, 3, 5, 7
inline int next_smooth(int n) {
  for (;; ++n) {
    int m = n;
    for (int p : {2, 3, 5, 7})
      while (m % p == 0) m /= p;
    if (m == 1) return n;
  }
}

struct AxisPlanHost {
  int n = 0;
  int nfft = 0;            // length of the radix transform: n, or the chirp-z length (bluestein)
  bool bluestein = false;  // n has a prime factor > kMaxDirectPrime
  std::vector<cfloat> chirp, bhat;
  static constexpr int kMaxDirectPrime = 31;  // larger primes go through the chirp-z transform
  std::vector<int> radix;  // DIF order
  std::vector<int> M;
  std::vector<cfloat> tw;
  std::vector<cfloat> tws;  // per stage with M > 1: M rows of R entries exp(-2 pi i j2 k / (R M))
  std::vector<int> rev, inv;
  bool generic = false;
  int max_generic_radix = 0;

  // `composite`: the schedule of the fixed-length kernels (fx_radix, mvn_fixed.hpp), which adds the
  // register butterflies 12, 6, 10, 15 (prime-factor maps) to the run-time-radix schedule.  Axes
  // that run the run-time-radix kernels keep the plain schedule: those kernels are compiled without
  // the composite cases.
  static std::vector<int> factorize(int n, bool composite = false) {
    std::vector<int> f;
    // big inline radices first: fewer stages = fewer LDS round trips
    if (composite) {
      for (int r : {8, 12, 4, 6, 10, 2, 15, 9})
        while (n % r == 0) { f.push_back(r); n /= r; }
    } else {
      for (int r : {8, 4, 2, 9})
        while (n % r == 0) { f.push_back(r); n /= r; }
    }
    for (int p = 3; (long)p * p <= n; p += 2)
      while (n % p == 0) { f.push_back(p); n /= p; }
    if (n > 1) f.push_back(n);
    return f;
  }

  bool composite = false;
  explicit AxisPlanHost(int n_, bool composite_ = false) : n(n_), composite(composite_) {
    if (n < 1) throw std::invalid_argument("mvn: FFT length must be >= 1");
    nfft = n;
    {
      std::vector<int> f = factorize(n);
      for (int r : f)
        if (r > kMaxDirectPrime) bluestein = true;
    }
    if (bluestein) {
      nfft = next_smooth(2 * n - 1);
      build_radix_tables(nfft);
      // identity position tables: a bluestein axis keeps natural order
      rev.resize(n);
      inv.resize(n);
      for (int j = 0; j < n; ++j) rev[j] = inv[j] = j;
      build_chirp();
      return;
    }
    build_radix_tables(n);
  }

  // radix schedule, strides, twiddles and position tables of a length-`len` radix transform
  void build_radix_tables(int len) {
    const int n = len;  // shadows the member on purpose: everything below is about `len`
    radix = factorize(n, composite);
    if (radix.empty()) radix.push_back(1);  // n == 1: a single no-op "radix-1" generic stage
    if ((int)radix.size() > MVN_MAX_STAGES)
      throw std::invalid_argument("mvn: too many radix stages for length " + std::to_string(n));
    int prod = 1;
    for (size_t s = 0; s < radix.size(); ++s) {
      prod *= radix[s];
      M.push_back(n / prod);
      if (!mvn_inline_radix(radix[s])) {
        generic = true;
        if (radix[s] > max_generic_radix) max_generic_radix = radix[s];
      }
    }
    tw.resize(n);
    for (int j = 0; j < n; ++j) {
      double a = -2.0 * M_PI * (double)j / (double)n;
      tw[j].x = (float)std::cos(a);
      tw[j].y = (float)std::sin(a);
    }
    for (size_t s = 0; s < radix.size(); ++s) {
      if (M[s] <= 1) continue;
      const int R = radix[s], L = R * M[s];
      for (int j2 = 0; j2 < M[s]; ++j2) {
        for (int k = 0; k < R; ++k) {
          double a = -2.0 * M_PI * (double)((long)j2 * k % L) / (double)L;
          cfloat w;
          w.x = (float)std::cos(a);
          w.y = (float)std::sin(a);
          tws.push_back(w);
        }
        if (R & 1) {  // rows are padded to an even number of entries (16-byte alignment)
          cfloat z;
          z.x = 0.f;
          z.y = 0.f;
          tws.push_back(z);
        }
      }
    }
    rev.resize(n);
    inv.resize(n);
    for (int p = 0; p < n; ++p) {
      int k = 0, w = 1;
      for (size_t s = 0; s < radix.size(); ++s) {
        int digit = (p / M[s]) % radix[s];
        k += digit * w;
        w *= radix[s];
      }
      rev[p] = k;
      inv[k] = p;
    }
  }

  // chirp[j] = exp(-i pi j^2 / n) and the pre-transformed, pre-scaled kernel of the chirp-z
  // convolution in the position order of the length-nfft forward stages
  void build_chirp() {
    const int m = nfft;
    std::vector<std::pair<double, double>> b((size_t)m, {0.0, 0.0});
    chirp.resize(n);
    for (int j = 0; j < n; ++j) {
      const long jj = ((long)j * j) % (2L * n);  // keeps the angle small
      const double a = M_PI * (double)jj / (double)n;
      chirp[j].x = (float)std::cos(a);
      chirp[j].y = (float)(-std::sin(a));
      b[j] = {std::cos(a), std::sin(a)};              // conj(chirp)
      if (j) b[m - j] = b[j];
    }
    // plain O(m^2) DFT in double: done once per plan on the host
    std::vector<std::pair<double, double>> B((size_t)m);
    std::vector<double> cs((size_t)m), sn((size_t)m);
    for (int t = 0; t < m; ++t) {
      cs[t] = std::cos(-2.0 * M_PI * t / m);
      sn[t] = std::sin(-2.0 * M_PI * t / m);
    }
    for (int k = 0; k < m; ++k) {
      double re = 0, im = 0;
      for (int j = 0; j < m; ++j) {
        const int t = (int)(((long)j * k) % m);
        re += b[j].first * cs[t] - b[j].second * sn[t];
        im += b[j].first * sn[t] + b[j].second * cs[t];
      }
      B[k] = {re / m, im / m};
    }
    // position order of the length-m stages: position p holds bin rev_m[p]; the member `rev` was
    // overwritten with the identity, so recompute the digit reversal here
    bhat.resize(m);
    for (int p = 0; p < m; ++p) {
      int k = 0, w = 1;
      for (size_t s = 0; s < radix.size(); ++s) {
        int digit = (p / M[s]) % radix[s];
        k += digit * w;
        w *= radix[s];
      }
      bhat[p].x = (float)B[k].first;
      bhat[p].y = (float)B[k].second;
    }
  }

  // kernel-visible plan; table pointers are filled by whoever owns the (device) copies
  AxisPlan view(const cfloat* tw_p, const int* rev_p, const int* inv_p,
                const cfloat* tws_p = nullptr, const cfloat* chirp_p = nullptr,
                const cfloat* bhat_p = nullptr) const {
    AxisPlan a;
    a.n = n;
    a.nfft = nfft;
    a.bluestein = bluestein ? 1 : 0;
    a.chirp = chirp_p;
    a.bhat = bhat_p;
    a.nstages = (int)radix.size();
    a.generic = generic ? 1 : 0;
    for (int s = 0; s < MVN_MAX_STAGES; ++s) {
      a.radix[s] = s < a.nstages ? radix[s] : 1;
      a.M[s] = s < a.nstages ? M[s] : 1;
      a.Mmul[s] = mvn_fastdiv_mul((unsigned)a.M[s]);
    }
    a.nmul = mvn_fastdiv_mul((unsigned)nfft);
    a.tw = tw_p;
    a.tws = tws_p;
    a.rev = rev_p;
    a.inv = inv_p;
    return a;
  }
};

// Geometry of the split-Nyquist layout for one logical shape.
struct Layout {
  int d0, d1, d2;
  bool even;    // d2 even -> half-length real trick + Nyquist plane
  int h;        // length of the complex transform run on the last axis (d2/2 or d2)
  int C;        // complex bins per row kept in the main array (d2/2, or (d2+1)/2 for odd d2)
  int RP;       // real row pitch in floats (d2, or d2+1 for odd d2) == 2*C
  size_t rows;  // d0*d1
  size_t real_floats() const { return rows * (size_t)RP; }  // floats in the main array
  size_t nyq_cplx() const { return even ? rows : 0; }       // complex elements in the Nyquist plane
  size_t logical() const { return rows * (size_t)d2; }
  // algorithmic bytes "B" of SURVEY.md 8d: the reference's in-place r2c footprint
  size_t B() const { return 4 * rows * 2 * (size_t)(d2 / 2 + 1); }

  Layout(int a, int b, int c) : d0(a), d1(b), d2(c) {
    if (a < 1 || b < 1 || c < 1) throw std::invalid_argument("mvn: dims must be >= 1");
    even = (c % 2 == 0);
    h = even ? c / 2 : c;
    C = even ? c / 2 : (c + 1) / 2;
    RP = 2 * C;
    rows = (size_t)a * (size_t)b;
  }
};

}  // namespace mvn
