// mvn_fft_core.hpp -- radix butterflies and LDS-resident FFT stages (host/device).
//
// The 3-D transforms of the RL loop (reference: cufftExecR2C/C2R in place,
// inc/cufft_utils.cuh:41-75; FFTW twin inc/fft_utils.h:55-104) are built from per-axis
// passes.  Every pass stages a tile of T lines in LDS as  buf[pos * TP + line]  and runs
// the radix stages below on it.  Two stage orders are provided:
//
//   DIF (decimation in frequency): natural order in  -> digit-reversed out
//                                  (position p holds X[rev[p]])
//   DIT (decimation in time)     : digit-reversed in -> natural order out
//
// so a forward pass (DIF) followed by an inverse pass (DIT) never needs an explicit
// permutation: it is folded into the global-memory row addressing (rev / inv tables of the
// plan).  Radices 2,3,4,5,7,8 are in-place register butterflies; any other prime factor is an
// out-of-place O(r^2) stage that ping-pongs to a second LDS buffer.
//
// The same code is compiled for the host (MVN_HOST_EMU) to check the index math on a CPU-only
// box; there a "workgroup" is one thread and barriers are no-ops.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__) && !defined(MVN_HOST_EMU)
#include <hip/hip_runtime.h>
#define MVN_HD __host__ __device__ __forceinline__
#define MVN_D __device__ __forceinline__
typedef float2 cfloat;
#define MVN_SYNC() __syncthreads()
#else
#define MVN_HD inline
#define MVN_D inline
struct cfloat {
  float x, y;
};
#define MVN_SYNC() \
  do {             \
  } while (0)
#endif

#define MVN_MAX_STAGES 16

// Per-axis plan as the kernels see it (tables live in device memory; see mvn_plan.hpp).
struct AxisPlan {
  int n;        // transform length (number of rows of a tile)
  int nfft;     // length the radix stages run on: n, or the chirp-z length m >= 2n-1 (bluestein)
  int bluestein;  // 1: n has a prime factor the radix stages should not take on (> 31); the
                  // length-n DFT is then computed as a length-nfft cyclic convolution with a chirp
  int nstages;  // number of radix stages
  int generic;  // 1 if some radix is not one of {2,3,4,5,7,8} (needs the second LDS buffer)
  int radix[MVN_MAX_STAGES];  // DIF order, outermost first
  int M[MVN_MAX_STAGES];      // M[s] = n / (radix[0]*...*radix[s]) = butterfly input stride
  unsigned Mmul[MVN_MAX_STAGES];  // ceil(2^32 / M[s]): x / M[s] == umulhi(x, Mmul[s]) (mvn_fastdiv)
  unsigned nmul;                  // same for n
  const cfloat* tw;           // tw[j] = exp(-2 pi i j / n)
  const cfloat* tws;          // stage-ordered twiddles of the fixed-length kernels (mvn_fixed.hpp)
  const cfloat* chirp;        // bluestein: chirp[j] = exp(-i pi j^2 / n), j < n
  const cfloat* bhat;         // bluestein: FFT_nfft(conj chirp, wrapped) / nfft, in position order
  const int* rev;             // rev[p] = k : after DIF, position p holds X[k]
  const int* inv;             // inv[k] = p
};

// x / d for x < 2^17, 2 <= d <= 2^15 with mul = ceil(2^32 / d) (exact: x * (mul - 2^32/d) < 2^32/d);
// d == 1 is special-cased.  Every numerator in the pass kernels is < n*T <= 20480.
MVN_HD unsigned mvn_fastdiv(unsigned x, unsigned d, unsigned mul) {
#if defined(__HIP_DEVICE_COMPILE__)
  return d == 1 ? x : __umulhi(x, mul);
#else
  return d == 1 ? x : (unsigned)(((unsigned long long)x * mul) >> 32);
#endif
}
inline unsigned mvn_fastdiv_mul(unsigned d) {
  return d <= 1 ? 0u : (unsigned)((0x100000000ull + d - 1) / d);
}

MVN_HD cfloat cmake(float x, float y) {
  cfloat r;
  r.x = x;
  r.y = y;
  return r;
}
MVN_HD cfloat cadd(cfloat a, cfloat b) { return cmake(a.x + b.x, a.y + b.y); }
MVN_HD cfloat csub(cfloat a, cfloat b) { return cmake(a.x - b.x, a.y - b.y); }
MVN_HD cfloat cmul(cfloat a, cfloat b) {
  return cmake(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
MVN_HD cfloat cconj(cfloat a) { return cmake(a.x, -a.y); }
MVN_HD cfloat cscale(cfloat a, float s) { return cmake(a.x * s, a.y * s); }
// multiply by (SIGN * i): forward transforms use SIGN = -1 (i.e. * -i)
template <int SIGN>
MVN_HD cfloat cmul_si(cfloat a) {
  return SIGN < 0 ? cmake(a.y, -a.x) : cmake(-a.y, a.x);
}
// twiddle for direction SIGN from the forward table entry
template <int SIGN>
MVN_HD cfloat twdir(cfloat w) {
  return SIGN < 0 ? w : cconj(w);
}

// ---------------------------------------------------------------------------------------------
// register butterflies: a[0..R) <- DFT_R(a) with kernel exp(SIGN * 2 pi i jk / R)
// ---------------------------------------------------------------------------------------------
template <int SIGN>
MVN_HD void dft2(cfloat* a) {
  cfloat t = a[0];
  a[0] = cadd(t, a[1]);
  a[1] = csub(t, a[1]);
}

template <int SIGN>
MVN_HD void dft3(cfloat* a) {
  const float s3 = 0.86602540378443864676f;
  cfloat t1 = cadd(a[1], a[2]);
  cfloat m1 = cmake(a[0].x - 0.5f * t1.x, a[0].y - 0.5f * t1.y);
  cfloat m2 = cmul_si<SIGN>(cscale(csub(a[1], a[2]), s3));
  a[0] = cadd(a[0], t1);
  a[1] = cadd(m1, m2);
  a[2] = csub(m1, m2);
}

template <int SIGN>
MVN_HD void dft4(cfloat* a) {
  cfloat t0 = cadd(a[0], a[2]);
  cfloat t1 = csub(a[0], a[2]);
  cfloat t2 = cadd(a[1], a[3]);
  cfloat t3 = cmul_si<SIGN>(csub(a[1], a[3]));
  a[0] = cadd(t0, t2);
  a[1] = cadd(t1, t3);
  a[2] = csub(t0, t2);
  a[3] = csub(t1, t3);
}

template <int SIGN>
MVN_HD void dft5(cfloat* a) {
  const float c1 = 0.30901699437494742410f;   // cos(2pi/5)
  const float c2 = -0.80901699437494742410f;  // cos(4pi/5)
  const float s1 = 0.95105651629515357212f;   // sin(2pi/5)
  const float s2 = 0.58778525229247312917f;   // sin(4pi/5)
  cfloat t1 = cadd(a[1], a[4]), t2 = cadd(a[2], a[3]);
  cfloat t3 = csub(a[1], a[4]), t4 = csub(a[2], a[3]);
  cfloat b1 = cmake(a[0].x + c1 * t1.x + c2 * t2.x, a[0].y + c1 * t1.y + c2 * t2.y);
  cfloat b2 = cmake(a[0].x + c2 * t1.x + c1 * t2.x, a[0].y + c2 * t1.y + c1 * t2.y);
  cfloat d1 = cmul_si<SIGN>(cmake(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y));
  cfloat d2 = cmul_si<SIGN>(cmake(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y));
  a[0] = cadd(a[0], cadd(t1, t2));
  a[1] = cadd(b1, d1);
  a[4] = csub(b1, d1);
  a[2] = cadd(b2, d2);
  a[3] = csub(b2, d2);
}

template <int SIGN>
MVN_HD void dft7(cfloat* a) {
  const float c1 = 0.62348980185873353053f;   // cos(2pi/7)
  const float c2 = -0.22252093395631440429f;  // cos(4pi/7)
  const float c3 = -0.90096886790241912624f;  // cos(6pi/7)
  const float s1 = 0.78183148246802980871f;   // sin(2pi/7)
  const float s2 = 0.97492791218182360702f;   // sin(4pi/7)
  const float s3 = 0.43388373911755812048f;   // sin(6pi/7)
  cfloat t1 = cadd(a[1], a[6]), t2 = cadd(a[2], a[5]), t3 = cadd(a[3], a[4]);
  cfloat u1 = csub(a[1], a[6]), u2 = csub(a[2], a[5]), u3 = csub(a[3], a[4]);
  cfloat b1 = cmake(a[0].x + c1 * t1.x + c2 * t2.x + c3 * t3.x,
                    a[0].y + c1 * t1.y + c2 * t2.y + c3 * t3.y);
  cfloat b2 = cmake(a[0].x + c2 * t1.x + c3 * t2.x + c1 * t3.x,
                    a[0].y + c2 * t1.y + c3 * t2.y + c1 * t3.y);
  cfloat b3 = cmake(a[0].x + c3 * t1.x + c1 * t2.x + c2 * t3.x,
                    a[0].y + c3 * t1.y + c1 * t2.y + c2 * t3.y);
  cfloat d1 = cmul_si<SIGN>(cmake(s1 * u1.x + s2 * u2.x + s3 * u3.x,
                                  s1 * u1.y + s2 * u2.y + s3 * u3.y));
  cfloat d2 = cmul_si<SIGN>(cmake(s2 * u1.x - s3 * u2.x - s1 * u3.x,
                                  s2 * u1.y - s3 * u2.y - s1 * u3.y));
  cfloat d3 = cmul_si<SIGN>(cmake(s3 * u1.x - s1 * u2.x + s2 * u3.x,
                                  s3 * u1.y - s1 * u2.y + s2 * u3.y));
  a[0] = cadd(cadd(a[0], t1), cadd(t2, t3));
  a[1] = cadd(b1, d1);
  a[6] = csub(b1, d1);
  a[2] = cadd(b2, d2);
  a[5] = csub(b2, d2);
  a[3] = cadd(b3, d3);
  a[4] = csub(b3, d3);
}

template <int SIGN>
MVN_HD void dft8(cfloat* a) {
  const float r = 0.70710678118654752440f;
  cfloat e[4] = {a[0], a[2], a[4], a[6]};
  cfloat o[4] = {a[1], a[3], a[5], a[7]};
  dft4<SIGN>(e);
  dft4<SIGN>(o);
  // o[k] *= exp(SIGN * 2 pi i k / 8)
  cfloat o1 = SIGN < 0 ? cmake(r * (o[1].x + o[1].y), r * (o[1].y - o[1].x))
                       : cmake(r * (o[1].x - o[1].y), r * (o[1].y + o[1].x));
  cfloat o2 = cmul_si<SIGN>(o[2]);
  cfloat o3 = SIGN < 0 ? cmake(r * (o[3].y - o[3].x), -r * (o[3].x + o[3].y))
                       : cmake(-r * (o[3].x + o[3].y), r * (o[3].x - o[3].y));
  a[0] = cadd(e[0], o[0]);
  a[4] = csub(e[0], o[0]);
  a[1] = cadd(e[1], o1);
  a[5] = csub(e[1], o1);
  a[2] = cadd(e[2], o2);
  a[6] = csub(e[2], o2);
  a[3] = cadd(e[3], o3);
  a[7] = csub(e[3], o3);
}

template <int R, int SIGN>
MVN_HD void dftR(cfloat* a) {
  if (R == 2) dft2<SIGN>(a);
  if (R == 3) dft3<SIGN>(a);
  if (R == 4) dft4<SIGN>(a);
  if (R == 5) dft5<SIGN>(a);
  if (R == 7) dft7<SIGN>(a);
  if (R == 8) dft8<SIGN>(a);
}

MVN_HD bool mvn_inline_radix(int r) {
  return r == 2 || r == 3 || r == 4 || r == 5 || r == 7 || r == 8;
}

// ---------------------------------------------------------------------------------------------
// one in-place stage over a tile of T lines.  Work item w -> (line c = w % T, butterfly b).
// twstep = n / (R * M): tw[j2 * k * twstep] = exp(-2 pi i j2 k / (R M)).
// ---------------------------------------------------------------------------------------------
template <int R, int SIGN, bool DIF, int T>
MVN_HD void stage_inplace(cfloat* buf, int TP, int n, int M, unsigned Mmul, const cfloat* tw,
                          int tid, int nthreads) {
  const int nwork = (n / R) * T;
  const int twstep = n / (R * M);
  const int stride = M * TP;
  for (int w = tid; w < nwork; w += nthreads) {
    const int b = w / T;  // T is a compile-time power of two: shift / mask
    const int c = w % T;
    const int blk = (int)mvn_fastdiv((unsigned)b, (unsigned)M, Mmul);
    const int j2 = b - blk * M;
    cfloat* p = buf + (blk * R * M + j2) * TP + c;
    cfloat a[R];
#pragma unroll
    for (int j = 0; j < R; ++j) a[j] = p[j * stride];
    if (!DIF && M > 1) {
#pragma unroll
      for (int k = 1; k < R; ++k) a[k] = cmul(a[k], twdir<SIGN>(tw[j2 * k * twstep]));
    }
    dftR<R, SIGN>(a);
    if (DIF && M > 1) {
#pragma unroll
      for (int k = 1; k < R; ++k) a[k] = cmul(a[k], twdir<SIGN>(tw[j2 * k * twstep]));
    }
#pragma unroll
    for (int j = 0; j < R; ++j) p[j * stride] = a[j];
  }
}

// out-of-place O(R^2) stage for any radix; one work item per output element.
template <int SIGN, bool DIF, int T>
MVN_HD void stage_generic(const cfloat* in, cfloat* out, int TP, int n, int R, int M,
                          const cfloat* tw, int tid, int nthreads) {
  const int nwork = n * T;
  const int twstep = n / (R * M);  // step of exp(-2 pi i /(R M)) in the length-n table
  const int rstep = n / R;         // step of exp(-2 pi i / R)
  for (int w = tid; w < nwork; w += nthreads) {
    const int pos = w / T;
    const int c = w - pos * T;
    const int blk = pos / (R * M);
    const int rem = pos - blk * R * M;
    const int k = rem / M;  // output index within the butterfly
    const int j2 = rem - k * M;
    const cfloat* p = in + (blk * R * M + j2) * TP + c;
    float accx = 0.f, accy = 0.f;
    for (int j = 0; j < R; ++j) {
      // DIF: sum_j a_j w_R^{jk}, twiddle applied after; DIT: inputs pre-twiddled by w_L^{j2 j}
      long e = (long)((j * k) % R) * rstep;
      if (!DIF) e += (long)j2 * j * twstep;
      cfloat wv = twdir<SIGN>(tw[(int)(e % n)]);
      cfloat a = p[j * M * TP];
      accx += a.x * wv.x - a.y * wv.y;
      accy += a.x * wv.y + a.y * wv.x;
    }
    cfloat r = cmake(accx, accy);
    if (DIF) r = cmul(r, twdir<SIGN>(tw[j2 * k * twstep]));
    out[pos * TP + c] = r;
  }
}

template <int SIGN, bool DIF, int T>
MVN_HD void stage_dispatch(cfloat*& buf, cfloat*& alt, int TP, const AxisPlan& pl, int s,
                           const cfloat* tw, int tid, int nthreads) {
  const int R = pl.radix[s];
  const int M = pl.M[s];
  const unsigned mm = pl.Mmul[s];
  switch (R) {
    case 2: stage_inplace<2, SIGN, DIF, T>(buf, TP, pl.nfft, M, mm, tw, tid, nthreads); break;
    case 3: stage_inplace<3, SIGN, DIF, T>(buf, TP, pl.nfft, M, mm, tw, tid, nthreads); break;
    case 4: stage_inplace<4, SIGN, DIF, T>(buf, TP, pl.nfft, M, mm, tw, tid, nthreads); break;
    case 5: stage_inplace<5, SIGN, DIF, T>(buf, TP, pl.nfft, M, mm, tw, tid, nthreads); break;
    case 7: stage_inplace<7, SIGN, DIF, T>(buf, TP, pl.nfft, M, mm, tw, tid, nthreads); break;
    case 8: stage_inplace<8, SIGN, DIF, T>(buf, TP, pl.nfft, M, mm, tw, tid, nthreads); break;
    default: {
      stage_generic<SIGN, DIF, T>(buf, alt, TP, pl.nfft, R, M, tw, tid, nthreads);
      cfloat* t = buf;
      buf = alt;
      alt = t;
    }
  }
}

// Forward-order stages: natural in -> position p holds X[rev[p]].  `buf` is updated to point at
// the buffer that holds the result (it flips to `alt` once per generic stage).  The caller must
// have synchronised the tile before the call; the tile is synchronised on return.  `tw` may
// point to an LDS copy of the plan's twiddle table.
template <int SIGN, int T>
MVN_HD void lds_radix_dif(cfloat*& buf, cfloat*& alt, int TP, const AxisPlan& pl, const cfloat* tw,
                          int tid, int nthreads) {
  for (int s = 0; s < pl.nstages; ++s) {
    stage_dispatch<SIGN, true, T>(buf, alt, TP, pl, s, tw, tid, nthreads);
    MVN_SYNC();
  }
}

// Reverse-order stages: position p holds x[rev[p]] on entry -> natural order out.
template <int SIGN, int T>
MVN_HD void lds_radix_dit(cfloat*& buf, cfloat*& alt, int TP, const AxisPlan& pl, const cfloat* tw,
                          int tid, int nthreads) {
  for (int s = pl.nstages - 1; s >= 0; --s) {
    stage_dispatch<SIGN, false, T>(buf, alt, TP, pl, s, tw, tid, nthreads);
    MVN_SYNC();
  }
}

// Bluestein (chirp-z) transform of the n rows of a tile whose LDS buffer has room for nfft rows:
//   X[k] = c[k] * sum_j (x[j] c[j]) b[k-j],  c[j] = exp(-i pi j^2/n),  b = conj(c)
// evaluated as a cyclic convolution of length nfft >= 2n-1 with the radix stages (forward
// decimation-in-frequency, multiply by the pre-transformed chirp in position order, inverse
// decimation-in-time).  SIGN = +1 runs the same tables on conjugated data.  Input and output are
// both in natural order: bluestein axes have identity rev/inv tables.
template <int SIGN, int T>
MVN_HD void lds_bluestein(cfloat*& buf, cfloat*& alt, int TP, const AxisPlan& pl, const cfloat* tw,
                          int tid, int nthreads) {
  const int n = pl.n, m = pl.nfft;
  for (int w = tid; w < m * T; w += nthreads) {
    const int j = w / T, c = w % T;
    cfloat v = cmake(0.f, 0.f);
    if (j < n) {
      v = buf[j * TP + c];
      if (SIGN > 0) v = cconj(v);
      v = cmul(v, pl.chirp[j]);
    }
    buf[j * TP + c] = v;
  }
  MVN_SYNC();
  lds_radix_dif<-1, T>(buf, alt, TP, pl, tw, tid, nthreads);
  for (int w = tid; w < m * T; w += nthreads) {
    const int p = w / T, c = w % T;
    buf[p * TP + c] = cmul(buf[p * TP + c], pl.bhat[p]);
  }
  MVN_SYNC();
  lds_radix_dit<+1, T>(buf, alt, TP, pl, tw, tid, nthreads);
  for (int w = tid; w < n * T; w += nthreads) {
    const int k = w / T, c = w % T;
    cfloat v = cmul(buf[k * TP + c], pl.chirp[k]);
    if (SIGN > 0) v = cconj(v);
    buf[k * TP + c] = v;
  }
  MVN_SYNC();
}

// The transforms the pass bodies call: radix stages, or the chirp-z route for awkward lengths.
template <int SIGN, int T>
MVN_HD void lds_fft_dif(cfloat*& buf, cfloat*& alt, int TP, const AxisPlan& pl, const cfloat* tw,
                        int tid, int nthreads) {
  if (pl.bluestein)
    lds_bluestein<SIGN, T>(buf, alt, TP, pl, tw, tid, nthreads);
  else
    lds_radix_dif<SIGN, T>(buf, alt, TP, pl, tw, tid, nthreads);
}
template <int SIGN, int T>
MVN_HD void lds_fft_dit(cfloat*& buf, cfloat*& alt, int TP, const AxisPlan& pl, const cfloat* tw,
                        int tid, int nthreads) {
  if (pl.bluestein)
    lds_bluestein<SIGN, T>(buf, alt, TP, pl, tw, tid, nthreads);
  else
    lds_radix_dit<SIGN, T>(buf, alt, TP, pl, tw, tid, nthreads);
}

// copy the plan's twiddle table into LDS (stage loops then read it with broadcast ds_reads
// instead of going through the vector memory pipe)
MVN_HD const cfloat* lds_stage_twiddles(cfloat* dst, const AxisPlan& pl, int tid, int nthreads) {
  for (int j = tid; j < pl.nfft; j += nthreads) dst[j] = pl.tw[j];
  return dst;
}
