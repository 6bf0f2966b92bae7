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
// a native two-float vector: complex values live in 64-bit register pairs, which is what the
// packed f32 instructions of gfx950 operate on (see the complex helpers below)
typedef float cfloat __attribute__((ext_vector_type(2)));
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
// ---------------------------------------------------------------------------------------------
// complex arithmetic.  gfx950 executes two f32 operations per lane in one packed instruction
// (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 on 64-bit register pairs); every source operand can
// feed either of its halves to either result half (op_sel / op_sel_hi) and be negated per half
// (neg_lo / neg_hi).  With those modifiers a complex add, an add of +-i times a value and an add of a
// conjugate are ONE instruction each and a complex product is two, about half the instruction
// count of the scalar forms.  hipcc folds whole-vector negations into the modifiers but not the
// swap-and-negate-one-half patterns (it emits v_mov + v_xor for them), hence the inline forms on
// the device; the host emulation (and MVN_NO_PACKED builds, for A/B runs) use the scalar forms.
// Measured on MI355X at 512^3 (same box, tools/ab_variants.sh): strided passes 0.209 -> 0.198 ms
// (forward) and 0.192 -> 0.179 ms (inverse, 6.0 TB/s); the other passes within +-1.5 %.
// ---------------------------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MVN_HOST_EMU) && !defined(MVN_NO_PACKED)
#define MVN_PACKED 1
#define MVN_PK2(r, op, a, b, mods) asm(op " %0, %1, %2 " mods : "=v"(r) : "v"(a), "v"(b))
#define MVN_PK3(r, op, a, b, c, mods) asm(op " %0, %1, %2, %3 " mods : "=v"(r) : "v"(a), "v"(b), "v"(c))
MVN_HD cfloat cadd(cfloat a, cfloat b) { return a + b; }
MVN_HD cfloat csub(cfloat a, cfloat b) { return a - b; }
// Complex products stay in scalar form (two multiplies, two fused multiply-adds, all visible to the
// instruction scheduler): as two dependent inline packed instructions (MVN_PK_MUL_ASM) the LDS-staged
// fused dim0 pass lost 3 % at 512^3 and no pass gained.
#ifndef MVN_PK_MUL_ASM
MVN_HD cfloat cmul(cfloat a, cfloat b) { return cmake(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
MVN_HD cfloat cmulc(cfloat a, cfloat b) { return cmake(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }
#else
MVN_HD cfloat cmul(cfloat a, cfloat w) {
  cfloat t, r;
  MVN_PK2(t, "v_pk_mul_f32", a, w, "op_sel_hi:[0,1]");                                   // (a.x w.x, a.x w.y)
  MVN_PK3(r, "v_pk_fma_f32", a, w, t, "op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]");  // + (-a.y w.y, a.y w.x)
  return r;
}
// a * conj(w)
MVN_HD cfloat cmulc(cfloat a, cfloat w) {
  cfloat t, r;
  MVN_PK2(t, "v_pk_mul_f32", a, w, "op_sel_hi:[0,1] neg_hi:[0,1]");       // (a.x w.x, -a.x w.y)
  MVN_PK3(r, "v_pk_fma_f32", a, w, t, "op_sel:[1,1,0] op_sel_hi:[1,0,1]");  // + (a.y w.y, a.y w.x)
  return r;
}
#endif
MVN_HD cfloat cconj(cfloat a) { return cmake(a.x, -a.y); }
MVN_HD cfloat cscale(cfloat a, float s) { return a * cmake(s, s); }
// a * s + c with a real factor
MVN_HD cfloat cfma_s(cfloat a, float s, cfloat c) { return __builtin_elementwise_fma(a, cmake(s, s), c); }
// a + SIGN i b
template <int SIGN>
MVN_HD cfloat cadd_i(cfloat a, cfloat b) {
  cfloat r;
  if (SIGN > 0)
    MVN_PK2(r, "v_pk_add_f32", a, b, "op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]");  // (a.x - b.y, a.y + b.x)
  else
    MVN_PK2(r, "v_pk_add_f32", a, b, "op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]");  // (a.x + b.y, a.y - b.x)
  return r;
}
// a + conj(b), a - conj(b)
MVN_HD cfloat cadd_c(cfloat a, cfloat b) {
  cfloat r;
  MVN_PK2(r, "v_pk_add_f32", a, b, "neg_hi:[0,1]");
  return r;
}
MVN_HD cfloat csub_c(cfloat a, cfloat b) {
  cfloat r;
  MVN_PK2(r, "v_pk_add_f32", a, b, "neg_lo:[0,1]");
  return r;
}
// conj(a) + SIGN i conj(b)
template <int SIGN>
MVN_HD cfloat cconj_add_i(cfloat a, cfloat b) {
  cfloat r;
  if (SIGN > 0)
    MVN_PK2(r, "v_pk_add_f32", a, b, "op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[1,0]");              // (a.x + b.y, b.x - a.y)
  else
    MVN_PK2(r, "v_pk_add_f32", a, b, "op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[1,1]");  // (a.x - b.y, -a.y - b.x)
  return r;
}
#else
MVN_HD cfloat cadd(cfloat a, cfloat b) { return cmake(a.x + b.x, a.y + b.y); }
MVN_HD cfloat csub(cfloat a, cfloat b) { return cmake(a.x - b.x, a.y - b.y); }
MVN_HD cfloat cmul(cfloat a, cfloat b) {
  return cmake(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
MVN_HD cfloat cmulc(cfloat a, cfloat b) {
  return cmake(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
MVN_HD cfloat cconj(cfloat a) { return cmake(a.x, -a.y); }
MVN_HD cfloat cscale(cfloat a, float s) { return cmake(a.x * s, a.y * s); }
MVN_HD cfloat cfma_s(cfloat a, float s, cfloat c) { return cmake(a.x * s + c.x, a.y * s + c.y); }
template <int SIGN>
MVN_HD cfloat cadd_i(cfloat a, cfloat b) {
  return SIGN > 0 ? cmake(a.x - b.y, a.y + b.x) : cmake(a.x + b.y, a.y - b.x);
}
MVN_HD cfloat cadd_c(cfloat a, cfloat b) { return cmake(a.x + b.x, a.y - b.y); }
MVN_HD cfloat csub_c(cfloat a, cfloat b) { return cmake(a.x - b.x, a.y + b.y); }
template <int SIGN>
MVN_HD cfloat cconj_add_i(cfloat a, cfloat b) {
  return SIGN > 0 ? cmake(a.x + b.y, b.x - a.y) : cmake(a.x - b.y, -a.y - b.x);
}
#endif
// a - SIGN i b
template <int SIGN>
MVN_HD cfloat csub_i(cfloat a, cfloat b) {
  return cadd_i<-SIGN>(a, b);
}
// a times the twiddle of direction SIGN from the forward table entry w (SIGN > 0: its conjugate)
template <int SIGN>
MVN_HD cfloat cmul_dir(cfloat a, cfloat w) {
  return SIGN < 0 ? cmul(a, w) : cmulc(a, w);
}
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
  cfloat m1 = cfma_s(t1, -0.5f, a[0]);
  cfloat m2 = cscale(csub(a[1], a[2]), s3);
  a[0] = cadd(a[0], t1);
  a[1] = cadd_i<SIGN>(m1, m2);
  a[2] = csub_i<SIGN>(m1, m2);
}

template <int SIGN>
MVN_HD void dft4(cfloat* a) {
  cfloat t0 = cadd(a[0], a[2]);
  cfloat t1 = csub(a[0], a[2]);
  cfloat t2 = cadd(a[1], a[3]);
  cfloat t3 = csub(a[1], a[3]);
  a[0] = cadd(t0, t2);
  a[1] = cadd_i<SIGN>(t1, t3);
  a[2] = csub(t0, t2);
  a[3] = csub_i<SIGN>(t1, t3);
}

template <int SIGN>
MVN_HD void dft5(cfloat* a) {
  const float c1 = 0.30901699437494742410f;   // cos(2pi/5)
  const float c2 = -0.80901699437494742410f;  // cos(4pi/5)
  const float s1 = 0.95105651629515357212f;   // sin(2pi/5)
  const float s2 = 0.58778525229247312917f;   // sin(4pi/5)
  cfloat t1 = cadd(a[1], a[4]), t2 = cadd(a[2], a[3]);
  cfloat t3 = csub(a[1], a[4]), t4 = csub(a[2], a[3]);
  cfloat b1 = cfma_s(t2, c2, cfma_s(t1, c1, a[0]));
  cfloat b2 = cfma_s(t2, c1, cfma_s(t1, c2, a[0]));
  cfloat d1 = cfma_s(t4, s2, cscale(t3, s1));
  cfloat d2 = cfma_s(t4, -s1, cscale(t3, s2));
  a[0] = cadd(a[0], cadd(t1, t2));
  a[1] = cadd_i<SIGN>(b1, d1);
  a[4] = csub_i<SIGN>(b1, d1);
  a[2] = cadd_i<SIGN>(b2, d2);
  a[3] = csub_i<SIGN>(b2, d2);
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
  cfloat b1 = cfma_s(t3, c3, cfma_s(t2, c2, cfma_s(t1, c1, a[0])));
  cfloat b2 = cfma_s(t3, c1, cfma_s(t2, c3, cfma_s(t1, c2, a[0])));
  cfloat b3 = cfma_s(t3, c2, cfma_s(t2, c1, cfma_s(t1, c3, a[0])));
  cfloat d1 = cfma_s(u3, s3, cfma_s(u2, s2, cscale(u1, s1)));
  cfloat d2 = cfma_s(u3, -s1, cfma_s(u2, -s3, cscale(u1, s2)));
  cfloat d3 = cfma_s(u3, s2, cfma_s(u2, -s1, cscale(u1, s3)));
  a[0] = cadd(cadd(a[0], t1), cadd(t2, t3));
  a[1] = cadd_i<SIGN>(b1, d1);
  a[6] = csub_i<SIGN>(b1, d1);
  a[2] = cadd_i<SIGN>(b2, d2);
  a[5] = csub_i<SIGN>(b2, d2);
  a[3] = cadd_i<SIGN>(b3, d3);
  a[4] = csub_i<SIGN>(b3, d3);
}

template <int SIGN>
MVN_HD void dft8(cfloat* a) {
  const float r = 0.70710678118654752440f;
  cfloat e[4] = {a[0], a[2], a[4], a[6]};
  cfloat o[4] = {a[1], a[3], a[5], a[7]};
  dft4<SIGN>(e);
  dft4<SIGN>(o);
  // o[k] *= exp(SIGN * 2 pi i k / 8); for k = 1, 3 that is +-r (o -+ i o), folded into the sums
  //   k = 1: r (o[1] + SIGN i o[1]),   k = 3: -r (o[3] - SIGN i o[3])
  const cfloat q1 = cadd_i<SIGN>(o[1], o[1]);
  const cfloat q3 = csub_i<SIGN>(o[3], o[3]);
  a[0] = cadd(e[0], o[0]);
  a[4] = csub(e[0], o[0]);
  a[1] = cfma_s(q1, r, e[1]);
  a[5] = cfma_s(q1, -r, e[1]);
  a[2] = cadd_i<SIGN>(e[2], o[2]);
  a[6] = csub_i<SIGN>(e[2], o[2]);
  a[3] = cfma_s(q3, -r, e[3]);
  a[7] = cfma_s(q3, r, e[3]);
}

// radix 9 = 3 x 3 (Cooley-Tukey inside the butterfly): 576 = 8 * 8 * 9, the padded extent of a
// 512-block with a 31-tap PSF, runs three stages instead of four (8, 8, 3, 3).
//   b[j][k1] = DFT3 over m of a[j + 3 m];  b[j][k1] *= w9^(j k1);  X[k1 + 3 k2] = DFT3 over j of b[j][k1]
template <int SIGN>
MVN_HD void dft9(cfloat* a) {
  const cfloat w1 = cmake(0.76604444311897803520f, (SIGN < 0 ? -1.f : 1.f) * 0.64278760968653932632f);   // w9^1
  const cfloat w2 = cmake(0.17364817766693034885f, (SIGN < 0 ? -1.f : 1.f) * 0.98480775301220805937f);   // w9^2
  const cfloat w4 = cmake(-0.93969262078590838405f, (SIGN < 0 ? -1.f : 1.f) * 0.34202014332566873304f);  // w9^4
  cfloat b0[3] = {a[0], a[3], a[6]};
  cfloat b1[3] = {a[1], a[4], a[7]};
  cfloat b2[3] = {a[2], a[5], a[8]};
  dft3<SIGN>(b0);
  dft3<SIGN>(b1);
  dft3<SIGN>(b2);
  b1[1] = cmul(b1[1], w1);
  b1[2] = cmul(b1[2], w2);
  b2[1] = cmul(b2[1], w2);
  b2[2] = cmul(b2[2], w4);
#pragma unroll
  for (int k1 = 0; k1 < 3; ++k1) {
    cfloat c[3] = {b0[k1], b1[k1], b2[k1]};
    dft3<SIGN>(c);
    a[k1] = c[0];
    a[k1 + 3] = c[1];
    a[k1 + 6] = c[2];
  }
}

template <int R, int SIGN>
MVN_HD void dftR(cfloat* a);

// composite radices with coprime factors R1 * R2 = 6, 10, 12, 15 by the prime-factor (Good-Thomas)
// index maps: no twiddles inside the butterfly,
//   n = (R2 n1 + R1 n2) mod R,  k = (e1 k1 + e2 k2) mod R,  e1 = 1 mod R1 and 0 mod R2, e2 = 0 mod R1 and 1 mod R2,
//   X[k] = DFT_R2 over n2 of ( DFT_R1 over n1 of a[n] ).
// One stage of radix 12 instead of two of radix 4 and 3 is one LDS round trip less per direction
// (384 = 8*8*6, 640 = 8*8*10, 768 = 8*8*12, 960 = 8*8*15: three stages each).
constexpr int mvn_crt_unit(int r1, int r2) {  // e with e = 1 mod r1, e = 0 mod r2
  for (int e = 0; e < r1 * r2; e += r2)
    if (e % r1 == 1 % r1) return e;
  return 0;
}
template <int R1, int R2, int SIGN>
MVN_HD void dft_pfa(cfloat* a) {
  constexpr int R = R1 * R2, E1 = mvn_crt_unit(R1, R2), E2 = mvn_crt_unit(R2, R1);
  cfloat b[R2][R1];
#pragma unroll
  for (int n2 = 0; n2 < R2; ++n2) {
#pragma unroll
    for (int n1 = 0; n1 < R1; ++n1) b[n2][n1] = a[(R2 * n1 + R1 * n2) % R];
    dftR<R1, SIGN>(b[n2]);  // -> k1
  }
#pragma unroll
  for (int k1 = 0; k1 < R1; ++k1) {
    cfloat c[R2];
#pragma unroll
    for (int n2 = 0; n2 < R2; ++n2) c[n2] = b[n2][k1];
    dftR<R2, SIGN>(c);  // -> k2
#pragma unroll
    for (int k2 = 0; k2 < R2; ++k2) a[(E1 * k1 + E2 * k2) % R] = c[k2];
  }
}

template <int R, int SIGN>
MVN_HD void dftR(cfloat* a) {
  if (R == 2) dft2<SIGN>(a);
  if (R == 3) dft3<SIGN>(a);
  if (R == 4) dft4<SIGN>(a);
  if (R == 5) dft5<SIGN>(a);
  if (R == 7) dft7<SIGN>(a);
  if (R == 8) dft8<SIGN>(a);
  if (R == 9) dft9<SIGN>(a);
  if (R == 6) dft_pfa<2, 3, SIGN>(a);
  if (R == 10) dft_pfa<2, 5, SIGN>(a);
  if (R == 12) dft_pfa<4, 3, SIGN>(a);
  if (R == 15) dft_pfa<3, 5, SIGN>(a);
}

MVN_HD bool mvn_inline_radix(int r) {
  return r == 2 || r == 3 || r == 4 || r == 5 || r == 6 || r == 7 || r == 8 || r == 9 || r == 10 || r == 12 || r == 15;
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
      for (int k = 1; k < R; ++k) a[k] = cmul_dir<SIGN>(a[k], tw[j2 * k * twstep]);
    }
    dftR<R, SIGN>(a);
    if (DIF && M > 1) {
#pragma unroll
      for (int k = 1; k < R; ++k) a[k] = cmul_dir<SIGN>(a[k], tw[j2 * k * twstep]);
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

// COMP: the launch may meet the composite radices 6, 10, 12, 15 (only the fixed-length schedules
// contain them: the Nyquist-plane launches that accompany the fixed kernels).  The main run-time-
// radix kernels are compiled without those cases: their register allocation is the maximum over
// all cases, and radix 15 cost the 560^3 view update 15 %.
template <int SIGN, bool DIF, int T, bool COMP = false>
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
    case 9: stage_inplace<9, SIGN, DIF, T>(buf, TP, pl.nfft, M, mm, tw, tid, nthreads); break;
    case 6: if constexpr (COMP) { stage_inplace<6, SIGN, DIF, T>(buf, TP, pl.nfft, M, mm, tw, tid, nthreads); break; }
    case 10: if constexpr (COMP) { stage_inplace<10, SIGN, DIF, T>(buf, TP, pl.nfft, M, mm, tw, tid, nthreads); break; }
    case 12: if constexpr (COMP) { stage_inplace<12, SIGN, DIF, T>(buf, TP, pl.nfft, M, mm, tw, tid, nthreads); break; }
    case 15: if constexpr (COMP) { stage_inplace<15, SIGN, DIF, T>(buf, TP, pl.nfft, M, mm, tw, tid, nthreads); break; }
    // (without COMP these fall through to the out-of-place stage, which is never reached:
    // schedules with composite radices are only built for axes that run fixed kernels)
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
template <int SIGN, int T, bool COMP = false>
MVN_HD void lds_radix_dif(cfloat*& buf, cfloat*& alt, int TP, const AxisPlan& pl, const cfloat* tw,
                          int tid, int nthreads) {
  for (int s = 0; s < pl.nstages; ++s) {
    stage_dispatch<SIGN, true, T, COMP>(buf, alt, TP, pl, s, tw, tid, nthreads);
    MVN_SYNC();
  }
}

// Reverse-order stages: position p holds x[rev[p]] on entry -> natural order out.
template <int SIGN, int T, bool COMP = false>
MVN_HD void lds_radix_dit(cfloat*& buf, cfloat*& alt, int TP, const AxisPlan& pl, const cfloat* tw,
                          int tid, int nthreads) {
  for (int s = pl.nstages - 1; s >= 0; --s) {
    stage_dispatch<SIGN, false, T, COMP>(buf, alt, TP, pl, s, tw, tid, nthreads);
    MVN_SYNC();
  }
}

// Bluestein (chirp-z) transform of the n rows of a tile whose LDS buffer has room for nfft rows:
//   X[k] = c[k] * sum_j (x[j] c[j]) b[k-j],  c[j] = exp(-i pi j^2/n),  b = conj(c)
// evaluated as a cyclic convolution of length nfft >= 2n-1 with the radix stages (forward
// decimation-in-frequency, multiply by the pre-transformed chirp in position order, inverse
// decimation-in-time).  SIGN = +1 runs the same tables on conjugated data.  Input and output are
// both in natural order: bluestein axes have identity rev/inv tables.
template <int SIGN, int T, bool COMP = false>
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
  lds_radix_dif<-1, T, COMP>(buf, alt, TP, pl, tw, tid, nthreads);
  for (int w = tid; w < m * T; w += nthreads) {
    const int p = w / T, c = w % T;
    buf[p * TP + c] = cmul(buf[p * TP + c], pl.bhat[p]);
  }
  MVN_SYNC();
  lds_radix_dit<+1, T, COMP>(buf, alt, TP, pl, tw, tid, nthreads);
  for (int w = tid; w < n * T; w += nthreads) {
    const int k = w / T, c = w % T;
    cfloat v = cmul(buf[k * TP + c], pl.chirp[k]);
    if (SIGN > 0) v = cconj(v);
    buf[k * TP + c] = v;
  }
  MVN_SYNC();
}

// The transforms the pass bodies call: radix stages, or the chirp-z route for awkward lengths.
template <int SIGN, int T, bool COMP = false>
MVN_HD void lds_fft_dif(cfloat*& buf, cfloat*& alt, int TP, const AxisPlan& pl, const cfloat* tw,
                        int tid, int nthreads) {
  if (pl.bluestein)
    lds_bluestein<SIGN, T, COMP>(buf, alt, TP, pl, tw, tid, nthreads);
  else
    lds_radix_dif<SIGN, T, COMP>(buf, alt, TP, pl, tw, tid, nthreads);
}
template <int SIGN, int T, bool COMP = false>
MVN_HD void lds_fft_dit(cfloat*& buf, cfloat*& alt, int TP, const AxisPlan& pl, const cfloat* tw,
                        int tid, int nthreads) {
  if (pl.bluestein)
    lds_bluestein<SIGN, T, COMP>(buf, alt, TP, pl, tw, tid, nthreads);
  else
    lds_radix_dit<SIGN, T, COMP>(buf, alt, TP, pl, tw, tid, nthreads);
}

// copy the plan's twiddle table into LDS (stage loops then read it with broadcast ds_reads
// instead of going through the vector memory pipe)
MVN_HD const cfloat* lds_stage_twiddles(cfloat* dst, const AxisPlan& pl, int tid, int nthreads) {
  for (int j = tid; j < pl.nfft; j += nthreads) dst[j] = pl.tw[j];
  return dst;
}
