// mvn_pass_bodies.hpp -- workgroup bodies of the FFT pass kernels and the RL pointwise math.
//
// One body = what one workgroup does for one tile; `tid`/`nthreads` enumerate the work items
// of each phase, phases are separated by MVN_SYNC().  On the device the bodies are wrapped by
// __global__ kernels (mvn_kernels.hip); on a CPU-only box host_emu.cpp runs them with one
// "thread" per workgroup to validate the index math against numpy.
//
// Reference semantics restated here:
//   - un-normalised r2c / c2r over the last axis + c2c over the other two
//     (inc/cufft_utils.cuh:41-75, inc/fft_utils.h:55-104)
//   - spectrum multiply  F <- F*G  (inc/cuda_kernels.cuh:213-242; the 1/N scale is folded into G)
//   - quotient           I <- view * float(1.0/I)          (inc/cpu_kernels.h:19-26)
//   - update             psi <- w*(next-psi)+psi, next from the clamp/Tikhonov chain
//                                                           (inc/cpu_kernels.h:28-90)
#pragma once

#include <math.h>

#include "mvn_fft_core.hpp"

// The RL pointwise math mirrors the reference's rounding step by step: no FMA contraction there
// (the FFT butterflies may contract freely).  gcc has no per-function switch; the emulation
// build passes -ffp-contract=off globally instead.
#if defined(__clang__)
#define MVN_FP_EXACT _Pragma("clang fp contract(off)")
#else
#define MVN_FP_EXACT
#endif

enum MvnEpilogue {
  MVN_EPI_STORE = 0,   // out = x * scale
  MVN_EPI_DIVIDE = 1,  // out = view * float(1.0 / (x*scale))
  MVN_EPI_UPDATE = 2,  // psi = w * (next(psi, x*scale) - psi) + psi
  MVN_EPI_DELTA = 3    // delta (+)= w * (next(psi, x*scale) - psi)     (simultaneous mode)
};

struct EpilogueParams {
  int mode;
  float scale;          // applied to the raw inverse-transform output first
  const float* view;    // DIVIDE
  float* psi;           // UPDATE (in/out), DELTA (in)
  const float* weights; // UPDATE / DELTA
  float* delta;         // DELTA
  int accumulate;       // DELTA: 0 -> store, 1 -> add
  double lambda;        // > 0 selects the Tikhonov branch
  float lambda_inv;     // float(1.f / lambda)   (inc/cpu_kernels.h:71)
  float min_value;
  int guard_zero_view;  // DIVIDE: a view voxel that is exactly 0 gives quotient 0 even where the
                        // blurred estimate is 0 (0 * 1/0 = NaN otherwise); only set for the
                        // good-size zero-padding mode, whose extra zeros lie beyond the PSF's reach
  // The convolution this pass ends ran its dim0 leg as a direct convolution (mvn_dim0_direct.hpp), which stores
  // `poison_epoch` into *poison when its input held a non-finite value: the whole volume then has to come out
  // NaN, as it does through an FFT along dim0 (inc/cpu_convolve.h:256-268).  No such leg: a zero word and the
  // epoch 0xffffffff (Plan3D::no_poison) - the pointer is never null when a pass is launched.
  const unsigned* poison;
  unsigned poison_epoch;
};

// Once per workgroup, before the first epilogue: a reported non-finite input turns the scale every raw output is
// multiplied with first into NaN - and with it every voxel (DIVIDE: view * 1 / NaN; UPDATE / DELTA: the clamp
// chain maps NaN to minValue) - at no cost per element.
MVN_HD void mvn_arm_poison(EpilogueParams& e) {
  // (no branch, and selected as an integer: the word, the epoch and the scale are the same for every lane, and an integer
  // select of scalars stays in a scalar register where a float select would move the scale into a vector one)
  unsigned bits;
  __builtin_memcpy(&bits, &e.scale, sizeof(bits));
  bits = *e.poison == e.poison_epoch ? 0x7fc00000u : bits;
  __builtin_memcpy(&e.scale, &bits, sizeof(bits));
}

// quotient with the optional guard above
MVN_HD float mvn_quotient_g(float view, float blurred, int guard);

// inc/cpu_kernels.h:22-25: TransferT temp = 1. / out; out = in * temp, i.e. a DOUBLE divide
// rounded to float.  Double carries 53 >= 2*24+2 bits, so that double rounding is innocuous and
// float(1.0 / double(x)) is exactly the correctly rounded single-precision quotient 1.0f / x
// (checked on 1e8 random bit patterns in tests/test_oracle_kernels.py).  hipcc divides floats
// correctly rounded by default (-fhip-fp32-correctly-rounded-divide-sqrt), at a fraction of the
// cost of the f64 sequence.
MVN_HD float mvn_quotient(float view, float blurred) {
  MVN_FP_EXACT
  float t = 1.0f / blurred;
  return view * t;
}

MVN_HD float mvn_quotient_g(float view, float blurred, int guard) {
  return (guard && view == 0.f) ? 0.f : mvn_quotient(view, blurred);
}

// sqrt of a double >= 1 (the Tikhonov argument 1 + 2 lambda v with v > 0).  On the device this is
// the compiler's own correctly rounded f64 expansion (rsq seed, two coupled Goldschmidt steps,
// two residual corrections) minus its rescaling of tiny inputs and its zero / infinity fix-up,
// neither of which can trigger for arguments >= 1: same bits, a third fewer instructions in the
// VALU-bound update pass.  +inf comes out as NaN instead of +inf; the clamp that follows maps
// both to minValue (inc/cpu_kernels.h:82-83).
MVN_HD double mvn_sqrt_ge1(double x) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MVN_HOST_EMU)
  const double y = __builtin_amdgcn_rsq(x);
  const double h0 = y * 0.5, s0 = x * y;
  const double r0 = __builtin_fma(-h0, s0, 0.5);
  const double s1 = __builtin_fma(s0, r0, s0);
  const double h1 = __builtin_fma(h0, r0, h0);
  const double d0 = __builtin_fma(-s1, s1, x);
  const double s2 = __builtin_fma(d0, h1, s1);
  const double d1 = __builtin_fma(-s2, s2, x);
  return __builtin_fma(d1, h1, s2);
#else
  return sqrt(x);
#endif
}

// clamp / regularise chain of inc/cpu_kernels.h:40-49 (lambda == 0) and :75-86 (lambda > 0)
MVN_HD float mvn_next_value(float last, float integral, double lambda, float lambda_inv,
                            float min_value) {
  MVN_FP_EXACT
  float value = last * integral;
  // the regularised value is formed for every lane and selected afterwards: no divergent branch
  // around the f64 chain, so the chains of the 16 values a thread owns interleave (for
  // value <= 0 or NaN it is garbage that the select drops)
  float reg = value;
  if (lambda > 0.) reg = (float)((double)lambda_inv * (mvn_sqrt_ge1(1. + 2. * lambda * (double)value) - 1.));
  value = value > 0.f ? reg : min_value;
  float next;
  if (isnan(value) || isinf(value))
    next = min_value;
  else
    next = value > min_value ? value : min_value;
  return next;
}

// Legacy single-step update of iterate_fft_tikhonov (inc/cuda_kernels.cuh:162-193): lambda arrives
// as float, the regularised value is formed in double and divided (not multiplied by a
// reciprocal), there is no NaN/Inf test beyond the comparisons, and the weight blends against the
// NEW value: w * (max(min, t) - t) + t.
MVN_HD float mvn_legacy_tikhonov_value(float image, float integral, float weight, float lambda_f,
                                       float min_value) {
  MVN_FP_EXACT
  float t = image * integral;
  if (t > 0.f)
    t = (float)((sqrt(1.0 + 2.0 * (double)lambda_f * (double)t) - 1.) / (double)lambda_f);
  else
    t = min_value;
  float nv = (min_value > t) ? min_value : t;
  return weight * (nv - t) + t;
}

MVN_HD void mvn_epilogue(const EpilogueParams& e, float* out, long i, float x) {
  MVN_FP_EXACT
  x *= e.scale;
  switch (e.mode) {
    case MVN_EPI_STORE: out[i] = x; break;
    case MVN_EPI_DIVIDE: out[i] = mvn_quotient_g(e.view[i], x, e.guard_zero_view); break;
    case MVN_EPI_UPDATE: {
      float last = e.psi[i];
      float next = mvn_next_value(last, x, e.lambda, e.lambda_inv, e.min_value);
      e.psi[i] = e.weights[i] * (next - last) + last;  // inc/cpu_kernels.h:51-52
    } break;
    case MVN_EPI_DELTA: {
      float last = e.psi[i];
      float next = mvn_next_value(last, x, e.lambda, e.lambda_inv, e.min_value);
      float d = e.weights[i] * (next - last);
      e.delta[i] = e.accumulate ? e.delta[i] + d : d;
    } break;
  }
}

// Pair form used by the even-d2 c2r pass: elements i, i+1 (i even, so 8-byte aligned) come out
// of one complex LDS word.  Operands are fetched separately from the arithmetic so that the
// fetch can be issued a whole transform ahead of its use; the (uniform) mode branch sits outside
// the unrolled loops so that every branch is straight-line code with all its loads in flight.
template <int U>
MVN_HD void mvn_epilogue_fetch_batch(const EpilogueParams& e, const long* idx, cfloat* a, cfloat* b) {
  if (e.mode == MVN_EPI_DIVIDE) {
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] = *reinterpret_cast<const cfloat*>(e.view + idx[u]);
  } else if (e.mode == MVN_EPI_UPDATE || e.mode == MVN_EPI_DELTA) {
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] = *reinterpret_cast<const cfloat*>(e.psi + idx[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) b[u] = *reinterpret_cast<const cfloat*>(e.weights + idx[u]);
  }
}

MVN_HD void mvn_epilogue_pair(const EpilogueParams& e, float* out, long i, cfloat z, cfloat a,
                              cfloat b) {
  MVN_FP_EXACT
  const float x0 = z.x * e.scale, x1 = z.y * e.scale;
  switch (e.mode) {
    case MVN_EPI_STORE: *reinterpret_cast<cfloat*>(out + i) = cmake(x0, x1); break;
    case MVN_EPI_DIVIDE:
      *reinterpret_cast<cfloat*>(out + i) = cmake(mvn_quotient_g(a.x, x0, e.guard_zero_view),
                                                  mvn_quotient_g(a.y, x1, e.guard_zero_view));
      break;
    case MVN_EPI_UPDATE: {
      const float n0 = mvn_next_value(a.x, x0, e.lambda, e.lambda_inv, e.min_value);
      const float n1 = mvn_next_value(a.y, x1, e.lambda, e.lambda_inv, e.min_value);
      *reinterpret_cast<cfloat*>(e.psi + i) = cmake(b.x * (n0 - a.x) + a.x, b.y * (n1 - a.y) + a.y);
    } break;
    case MVN_EPI_DELTA: {
      const float n0 = mvn_next_value(a.x, x0, e.lambda, e.lambda_inv, e.min_value);
      const float n1 = mvn_next_value(a.y, x1, e.lambda, e.lambda_inv, e.min_value);
      cfloat d = cmake(b.x * (n0 - a.x), b.y * (n1 - a.y));
      if (e.accumulate) {
        const cfloat old = *reinterpret_cast<const cfloat*>(e.delta + i);
        d = cmake(old.x + d.x, old.y + d.y);
      }
      *reinterpret_cast<cfloat*>(e.delta + i) = d;
    } break;
  }
}

MVN_HD float mvn_blend(float w, float next, float last) {
  MVN_FP_EXACT
  return w * (next - last) + last;  // inc/cpu_kernels.h:51-52
}

// Pair epilogue of the fused c2r + pointwise + r2c pass: hands the two results back as the packed
// input z[j] = (y[2j], y[2j+1]) of the next forward transform; UPDATE also writes psi.
MVN_HD cfloat mvn_epilogue_pair_value(int mode, const EpilogueParams& e, long i, cfloat z, cfloat a,
                                      cfloat b) {
  MVN_FP_EXACT
  const float x0 = z.x * e.scale, x1 = z.y * e.scale;
  if (mode == MVN_EPI_DIVIDE)
    return cmake(mvn_quotient_g(a.x, x0, e.guard_zero_view), mvn_quotient_g(a.y, x1, e.guard_zero_view));
  if (mode == MVN_EPI_UPDATE) {
    const float n0 = mvn_next_value(a.x, x0, e.lambda, e.lambda_inv, e.min_value);
    const float n1 = mvn_next_value(a.y, x1, e.lambda, e.lambda_inv, e.min_value);
    const cfloat y = cmake(mvn_blend(b.x, n0, a.x), mvn_blend(b.y, n1, a.y));
    *reinterpret_cast<cfloat*>(e.psi + i) = y;
    return y;
  }
  return cmake(x0, x1);
}

// the same with the mode as a template constant (fixed-length kernels)
template <int EPI>
MVN_HD void mvn_epilogue_pair_t(const EpilogueParams& e, float* out, long i, cfloat z, cfloat a,
                                cfloat b) {
  EpilogueParams ee = e;
  ee.mode = EPI;
  mvn_epilogue_pair(ee, out, i, z, a, b);
}

#define MVN_ROWS_U 8

// ---------------------------------------------------------------------------------------------
// last-axis passes (contiguous rows).  A tile is T rows; LDS holds them transposed,
// lds[pos * TP + row], TP odd to keep the transposing accesses conflict-free.
// ---------------------------------------------------------------------------------------------
struct RowsParams {
  AxisPlan ax;        // length h (= d2/2 for even d2, d2 for odd)
  const cfloat* twr;  // d2-th roots of unity (even d2: real<->half-complex post-processing)
  int d2, h, C, RP;   // see mvn::Layout
  long rows;          // d0*d1
  int T, TP;
  long lds_alt;       // offset (in cfloat) of the second LDS buffer, 0 if none
  long lds_tw;        // offset (in cfloat) of the LDS twiddle copy
  unsigned hmul, Cmul;  // mvn_fastdiv multipliers for h and C
  int fixed;            // 1: launch the compile-time specialised kernel for this h (mvn_fixed.hpp)
  // r2c: real rows in (pitch RP floats) -> complex rows out (pitch C) + Nyquist plane
  const float* in_real;
  cfloat* out_cplx;
  cfloat* out_nyq;
  // c2r: complex rows + Nyquist plane in -> real rows out through the epilogue
  const cfloat* in_cplx;
  const cfloat* in_nyq;
  float* out_real;
  EpilogueParams epi;
  // 1: the Nyquist bin of a row (real, like its DC bin) is carried in the IMAGINARY part of the DC bin and
  // there is no Nyquist plane (in_nyq / out_nyq unused): the dim1 passes then transform DC + i Nyquist as one
  // complex column and the direct dim0 leg separates the two by the k1 <-> -k1 symmetry (mvn_dim0_direct.hpp)
  int nyq_packed;
  // 1: the half-spectrum (in_cplx / out_cplx) is in the LINE layout of the fused middle pass (mvn_mid_fused.hpp),
  // [plane][position][row of the plane] with lines_d1 rows per plane; row r of the launch is row row_base + r of the
  // volume and in_cplx / out_cplx point at the volume's first element.  Fixed kernels only, nyq_packed only.
  int lines, lines_d1;
  long row_base;
};

// forward stages + real<->complex step + store of a tile that already sits in LDS as the packed
// rows z[j] (shared by the plain r2c pass and the fused c2r + pointwise + r2c pass)
template <int T>
MVN_HD void rows_r2c_even_tail(const RowsParams& P, long r0, cfloat*& buf, cfloat*& alt,
                               const cfloat* tw, int tid, int nthreads) {
  const int h = P.h, TP = P.TP;
  lds_fft_dif<-1, T>(buf, alt, TP, P.ax, tw, tid, nthreads);
  const int npairs = h / 2 + 1;
  for (int w = tid; w < npairs * T; w += nthreads) {
    const int k = w / T, rho = w - k * T;
    const int m = (h - k) % h;
    const int pk = P.ax.inv[k], pm = P.ax.inv[m];
    const cfloat zk = buf[pk * TP + rho];
    if (k == 0) {
      if (P.nyq_packed) {
        buf[pk * TP + rho] = cmake(zk.x + zk.y, zk.x - zk.y);
      } else {
        buf[pk * TP + rho] = cmake(zk.x + zk.y, 0.f);
        if (r0 + rho < P.rows) P.out_nyq[r0 + rho] = cmake(zk.x - zk.y, 0.f);
      }
    } else {
      const cfloat zm = buf[pm * TP + rho];
      const cfloat E = cmake(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
      const cfloat D = cmake(0.5f * (zk.x - zm.x), 0.5f * (zk.y + zm.y));
      const cfloat G = cmul(P.twr[k], D);
      buf[pk * TP + rho] = cadd(E, cmul_si<-1>(G));
      if (m != k) buf[pm * TP + rho] = cadd(cconj(E), cmul_si<-1>(cconj(G)));
    }
  }
  MVN_SYNC();
  for (int w = tid; w < T * h; w += nthreads) {
    // spectral rows are kept in position (digit-reversed) order: bin k sits at column inv[k]
    const int rho = (int)mvn_fastdiv((unsigned)w, (unsigned)h, P.hmul), p = w - rho * h;
    const long row = r0 + rho;
    if (row < P.rows) P.out_cplx[row * P.C + p] = buf[p * TP + rho];
  }
}

// real -> half-complex, even d2: z[j] = x[2j] + i x[2j+1], Z = FFT_h(z), then
// X[k] = E - i w^k D,  E = (Z[k] + conj Z[h-k])/2,  D = (Z[k] - conj Z[h-k])/2,  w = exp(-2 pi i/d2)
template <int T>
MVN_HD void rows_r2c_even_body(const RowsParams& P, long tile, int tid, int nthreads, cfloat* lds) {
  const int h = P.h, TP = P.TP;
  const long r0 = tile * T;
  cfloat* buf = lds;
  cfloat* alt = lds + P.lds_alt;
  const cfloat* tw = lds_stage_twiddles(lds + P.lds_tw, P.ax, tid, nthreads);
  constexpr int U = MVN_ROWS_U;
  const int total = T * h;
  const long last_row = P.rows - 1;
  for (int w0 = tid; w0 < total; w0 += U * nthreads) {
    cfloat v[U];
    // every load is unconditional (clamped address) so that all U are in flight together
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int w = w0 + u * nthreads;
      w = w < total ? w : total - 1;
      const int rho = (int)mvn_fastdiv((unsigned)w, (unsigned)h, P.hmul), j = w - rho * h;
      long row = r0 + rho;
      row = row < last_row ? row : last_row;
      v[u] = reinterpret_cast<const cfloat*>(P.in_real + row * P.RP)[j];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int w = w0 + u * nthreads;
      if (w < total) {
        const int rho = (int)mvn_fastdiv((unsigned)w, (unsigned)h, P.hmul), j = w - rho * h;
        buf[j * TP + rho] = (r0 + rho <= last_row) ? v[u] : cmake(0.f, 0.f);
      }
    }
  }
  MVN_SYNC();
  rows_r2c_even_tail<T>(P, r0, buf, alt, tw, tid, nthreads);
}

// half-complex -> real, even d2: Z[k] = E + i O, E = X[k] + conj X[h-k],
// O = (X[k] - conj X[h-k]) exp(+2 pi i k/d2); z = IFFT_h(Z); x[2j] = Re z[j], x[2j+1] = Im z[j]
// KEEP = true is the fused pass: the epilogue results stay in LDS and run straight through the
// forward half (rows_r2c_even_tail), writing the half-spectrum of the result over the input.
template <int T, bool KEEP = false>
MVN_HD void rows_c2r_even_body(const RowsParams& P, long tile, int tid, int nthreads, cfloat* lds) {
  // (a copy of the epilogue's few fields, not of P: the run-time radix tables in P are indexed dynamically and a
  // private copy of the whole struct would live in scratch memory)
  EpilogueParams epi = P.epi;
  mvn_arm_poison(epi);
  constexpr int U = MVN_ROWS_U;
  const int h = P.h, TP = P.TP;
  const long r0 = tile * T;
  const int total = T * h;
  const bool single = total <= U * nthreads;
  cfloat* buf = lds;
  cfloat* alt = lds + P.lds_alt;
  const cfloat* tw = lds_stage_twiddles(lds + P.lds_tw, P.ax, tid, nthreads);
  cfloat ea[U], eb[U];  // epilogue operands, fetched a whole transform ahead when `single`
#pragma unroll
  for (int u = 0; u < U; ++u) {
    ea[u] = cmake(0.f, 0.f);
    eb[u] = cmake(0.f, 0.f);
  }
  const long last_row = P.rows - 1;
  for (int w0 = tid; w0 < total; w0 += U * nthreads) {
    cfloat v[U];
    long idx[U];
    // unconditional, clamped loads: all U (and, if `single`, the epilogue operands) in flight
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int w = w0 + u * nthreads;
      w = w < total ? w : total - 1;
      const int rho = (int)mvn_fastdiv((unsigned)w, (unsigned)h, P.hmul), k = w - rho * h;
      long row = r0 + rho;
      row = row < last_row ? row : last_row;
      v[u] = P.in_cplx[row * P.C + k];
      idx[u] = row * P.RP + 2 * k;
    }
    if (single) mvn_epilogue_fetch_batch<U>(epi, idx, ea, eb);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int w = w0 + u * nthreads;
      if (w < total) {
        const int rho = (int)mvn_fastdiv((unsigned)w, (unsigned)h, P.hmul), p = w - rho * h;
        buf[p * TP + rho] = (r0 + rho <= last_row) ? v[u] : cmake(0.f, 0.f);
      }
    }
  }
  MVN_SYNC();
  const int npairs = h / 2 + 1;
  for (int w = tid; w < npairs * T; w += nthreads) {
    const int k = w / T, rho = w - k * T;
    const int m = (h - k) % h;
    const int pk = P.ax.inv[k], pm = P.ax.inv[m];
    const cfloat xk = buf[pk * TP + rho];
    if (k == 0) {
      // imaginary parts of the DC and Nyquist bins are ignored, as FFTW's c2r does
      float xh = 0.f;
      if (P.nyq_packed)
        xh = xk.y;
      else if (r0 + rho < P.rows)
        xh = P.in_nyq[r0 + rho].x;
      buf[pk * TP + rho] = cmake(xk.x + xh, xk.x - xh);
    } else {
      const cfloat xm = buf[pm * TP + rho];
      const cfloat E = cmake(xk.x + xm.x, xk.y - xm.y);
      const cfloat Dk = cmake(xk.x - xm.x, xk.y + xm.y);
      const cfloat O = cmul(Dk, cconj(P.twr[k]));
      buf[pk * TP + rho] = cadd(E, cmul_si<+1>(O));
      if (m != k) buf[pm * TP + rho] = cadd(cconj(E), cmul_si<+1>(cconj(O)));
    }
  }
  MVN_SYNC();
  lds_fft_dit<+1, T>(buf, alt, TP, P.ax, tw, tid, nthreads);
  for (int w0 = tid; w0 < total; w0 += U * nthreads) {
    if (!single) {
      long idx[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        int w = w0 + u * nthreads;
        w = w < total ? w : total - 1;
        const int rho = (int)mvn_fastdiv((unsigned)w, (unsigned)h, P.hmul), j = w - rho * h;
        long row = r0 + rho;
        row = row < last_row ? row : last_row;
        idx[u] = row * P.RP + 2 * j;
      }
      mvn_epilogue_fetch_batch<U>(epi, idx, ea, eb);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int w = w0 + u * nthreads;
      const int rho = (int)mvn_fastdiv((unsigned)w, (unsigned)h, P.hmul), j = w - rho * h;
      const long row = r0 + rho;
      if (w < total && row <= last_row) {
        if (KEEP)
          buf[j * TP + rho] = mvn_epilogue_pair_value(epi.mode, epi, row * P.RP + 2 * j,
                                                      buf[j * TP + rho], ea[u], eb[u]);
        else
          mvn_epilogue_pair(epi, P.out_real, row * P.RP + 2 * j, buf[j * TP + rho], ea[u], eb[u]);
      }
    }
  }
  if (KEEP) {
    MVN_SYNC();
    rows_r2c_even_tail<T>(P, r0, buf, alt, tw, tid, nthreads);
  }
}

// odd d2: plain complex transform of the real row, first C = (d2+1)/2 bins kept
template <int T>
MVN_HD void rows_r2c_odd_body(const RowsParams& P, long tile, int tid, int nthreads, cfloat* lds) {
  const int n = P.h, TP = P.TP;
  const long r0 = tile * T;
  cfloat* buf = lds;
  cfloat* alt = lds + P.lds_alt;
  const cfloat* tw = lds_stage_twiddles(lds + P.lds_tw, P.ax, tid, nthreads);
  for (int w = tid; w < T * n; w += nthreads) {
    const int rho = (int)mvn_fastdiv((unsigned)w, (unsigned)n, P.hmul), j = w - rho * n;
    const long row = r0 + rho;
    float v = 0.f;
    if (row < P.rows) v = P.in_real[row * P.RP + j];
    buf[j * TP + rho] = cmake(v, 0.f);
  }
  MVN_SYNC();
  lds_fft_dif<-1, T>(buf, alt, TP, P.ax, tw, tid, nthreads);
  for (int w = tid; w < T * P.C; w += nthreads) {
    const int rho = (int)mvn_fastdiv((unsigned)w, (unsigned)P.C, P.Cmul), k = w - rho * P.C;
    const long row = r0 + rho;
    if (row < P.rows) P.out_cplx[row * P.C + k] = buf[P.ax.inv[k] * TP + rho];
  }
}

template <int T>
MVN_HD void rows_c2r_odd_body(const RowsParams& P, long tile, int tid, int nthreads, cfloat* lds) {
  // (a copy of the epilogue's few fields, not of P: the run-time radix tables in P are indexed dynamically and a
  // private copy of the whole struct would live in scratch memory)
  EpilogueParams epi = P.epi;
  mvn_arm_poison(epi);
  const int n = P.h, TP = P.TP;
  const long r0 = tile * T;
  cfloat* buf = lds;
  cfloat* alt = lds + P.lds_alt;
  const cfloat* tw = lds_stage_twiddles(lds + P.lds_tw, P.ax, tid, nthreads);
  for (int w = tid; w < T * n; w += nthreads) {
    const int rho = (int)mvn_fastdiv((unsigned)w, (unsigned)n, P.hmul), k = w - rho * n;
    const long row = r0 + rho;
    cfloat v = cmake(0.f, 0.f);
    if (row < P.rows) {
      if (k < P.C) {
        v = P.in_cplx[row * P.C + k];
        if (k == 0) v.y = 0.f;
      } else {
        v = cconj(P.in_cplx[row * P.C + (n - k)]);
      }
    }
    buf[P.ax.inv[k] * TP + rho] = v;
  }
  MVN_SYNC();
  lds_fft_dit<+1, T>(buf, alt, TP, P.ax, tw, tid, nthreads);
  for (int w = tid; w < T * n; w += nthreads) {
    const int rho = (int)mvn_fastdiv((unsigned)w, (unsigned)n, P.hmul), j = w - rho * n;
    const long row = r0 + rho;
    if (row < P.rows) mvn_epilogue(epi, P.out_real, row * P.RP + j, buf[j * TP + rho].x);
  }
}

// ---------------------------------------------------------------------------------------------
// strided-axis passes: lines run along an axis with element stride `estride`, a tile is T
// neighbouring lines (column stride `cstride`, 1 for the main array so a tile row is one
// contiguous T*8-byte segment).  lds[pos * TP + col].
// ---------------------------------------------------------------------------------------------
enum MvnStridedMode { MVN_ST_FWD = 0, MVN_ST_INV = 1, MVN_ST_FWD_MUL_INV = 2 };

struct StridedParams {
  AxisPlan ax;
  cfloat* data;        // destination (and source, unless `src` is set)
  const cfloat* src;   // optional separate source with the same addressing (out-of-place pass)
  const cfloat* spec;  // FWD_MUL_INV: pre-scaled PSF spectrum, same addressing as data ...
  int spec_tiled;      // ... or (fixed kernels only) tile-contiguous: [tile][row][T columns], so that a
                       // tile's operands are ONE contiguous stream of n * T * 8 bytes instead of n row
                       // segments n-1 pages apart
  long ostride;        // between outer slabs
  long estride;        // between elements of a line
  long cstride;        // between neighbouring lines of a tile
  int ncols;           // lines per outer slab
  int tiles_per_outer;
  int T, TP;
  long lds_alt;
  long lds_tw;  // offset (in cfloat) of the LDS twiddle copy
  int is_nyq;   // launch works on the Nyquist plane (profiling tag only)
  int fixed;    // 1: launch the compile-time specialised kernel for this length (mvn_fixed.hpp)
  long nblocks; // fixed kernels: tiles of the launch (a workgroup walks over several of them)
};

// Loads are issued in batches of U per thread BEFORE any of them is consumed, so a tile's HBM
// latency is paid once, not once per element.  When the whole tile is a single batch
// (n*T <= U*nthreads, e.g. 512 x 16 on 512 threads) the fused mode also fetches its PSF-spectrum
// operands up front and keeps them in registers across the forward transform.
#define MVN_STRIDED_U 8

template <int MODE, int T, bool COMP = false>
MVN_HD void strided_body(const StridedParams& P, long block, int tid, int nthreads, cfloat* lds) {
  constexpr int U = MVN_STRIDED_U;
  const int n = P.ax.n, TP = P.TP;
  const long o = block / P.tiles_per_outer;
  const int t = (int)(block - o * P.tiles_per_outer);
  const int c0 = t * T;
  const int ncol = (P.ncols - c0) < T ? (P.ncols - c0) : T;
  const long base = o * P.ostride + (long)c0 * P.cstride;
  const int total = n * T;
  const bool single = total <= U * nthreads;
  const cfloat* in = P.src ? P.src : P.data;
  cfloat* buf = lds;
  cfloat* alt = lds + P.lds_alt;
  const cfloat* tw = lds_stage_twiddles(lds + P.lds_tw, P.ax, tid, nthreads);
  cfloat g[U];
  for (int w0 = tid; w0 < total; w0 += U * nthreads) {
    cfloat v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int w = w0 + u * nthreads;
      w = w < total ? w : total - 1;
      int j = w / T, c = w % T;
      c = c < ncol ? c : ncol - 1;  // clamped: the load itself is unconditional
      v[u] = in[base + (long)j * P.estride + (long)c * P.cstride];
    }
    if (MODE == MVN_ST_FWD_MUL_INV && single) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        int w = w0 + u * nthreads;
        w = w < total ? w : total - 1;
        int p = w / T, c = w % T;
        c = c < ncol ? c : ncol - 1;
        g[u] = P.spec[base + (long)p * P.estride + (long)c * P.cstride];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int w = w0 + u * nthreads;
      if (w < total) {
        const int j = w / T, c = w % T;
        buf[j * TP + c] = (c < ncol) ? v[u] : cmake(0.f, 0.f);
      }
    }
  }
  MVN_SYNC();
  if (MODE == MVN_ST_INV) {
    lds_fft_dit<+1, T, COMP>(buf, alt, TP, P.ax, tw, tid, nthreads);
  } else {
    lds_fft_dif<-1, T, COMP>(buf, alt, TP, P.ax, tw, tid, nthreads);
    if (MODE == MVN_ST_FWD_MUL_INV) {
      if (single) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int w = tid + u * nthreads;
          if (w < total) {
            const int p = w / T, c = w % T;
            buf[p * TP + c] = cmul(buf[p * TP + c], g[u]);
          }
        }
      } else {
        for (int w0 = tid; w0 < total; w0 += U * nthreads) {
#pragma unroll
          for (int u = 0; u < U; ++u) {
            int w = w0 + u * nthreads;
            w = w < total ? w : total - 1;
            int p = w / T, c = w % T;
            c = c < ncol ? c : ncol - 1;
            g[u] = P.spec[base + (long)p * P.estride + (long)c * P.cstride];
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int w = w0 + u * nthreads;
            if (w < total) {
              const int p = w / T, c = w % T;
              buf[p * TP + c] = cmul(buf[p * TP + c], g[u]);
            }
          }
        }
      }
      MVN_SYNC();
      lds_fft_dit<+1, T, COMP>(buf, alt, TP, P.ax, tw, tid, nthreads);
    }
  }
  for (int w = tid; w < total; w += nthreads) {
    const int p = w / T, c = w % T;
    if (c < ncol) P.data[base + (long)p * P.estride + (long)c * P.cstride] = buf[p * TP + c];
  }
}

// ---------------------------------------------------------------------------------------------
// PSF placement: kernel voxel (z,y,x) -> ((z-k0/2) mod D0, (y-k1/2) mod D1, (x-k2/2) mod D2),
// integer k/2, so the centre voxel lands on the origin (inc/padd_utils.h:11-40).
// ---------------------------------------------------------------------------------------------
MVN_HD void mvn_scatter_psf_item(const float* kernel, int k0, int k1, int k2, float* target,
                                 int D0, int D1, int D2, long pitch, float scale, long i) {
  const int x = (int)(i % k2);
  const int y = (int)((i / k2) % k1);
  const int z = (int)(i / ((long)k2 * k1));
  int ix = x - k2 / 2, iy = y - k1 / 2, iz = z - k0 / 2;
  if (ix < 0) ix += D2;
  if (iy < 0) iy += D1;
  if (iz < 0) iz += D0;
  target[((long)iz * D1 + iy) * pitch + ix] = kernel[i] * scale;
}
