// mvn_mid_fused.hpp -- the three middle passes of a convolution as ONE pass over the volume (round 4).
//
// The reference convolves by 3-D FFT x PSF spectrum x inverse 3-D FFT (inc/gpu_convolve.cuh:113-142,
// inc/cpu_convolve.h:217-291).  Between the two last-axis passes this library ran three passes over the
// half-spectrum: dim1 forward, the dim0 leg as a direct K-tap convolution (mvn_dim0_direct.hpp), dim1 inverse -
// 6 volumes of HBM traffic per convolution, 0.62 ms at 512^3, none of them above 0.6 of the HBM roofline and
// none with anything left to tune (profiles/r04_dim0_direct.md).  The three are one pass here: 2 volumes.
//
// What makes that possible is the layout of the half-spectrum BETWEEN the last-axis passes ("line layout"):
//     spec[z][c][y]      z < d0 planes,  c < H columns (last-axis positions),  y < N1 rows of the plane
// i.e. the lines along dim1 are contiguous (the row-major layout [z][y][c] keeps them C bins apart).  A workgroup
// takes ONE column c and walks along dim0: a line (z, c, .) of N1 = 512 bins comes in, is transformed along dim1
// (one line per wave, 8 bins per lane, radix 8 x 8 x 8 with two wave-private LDS exchanges and no workgroup barrier), every bin goes through
// its K-tap filter along dim0 - the filter state (K taps, K inputs) lives in the registers of the ONE work item
// that owns the bin for the whole walk -, the line is transformed back and stored.  31 lines of a column are the
// window of the direct convolution: 124 KB, which is what the registers of one CU hold beside the code's own
// needs - so a column is a CU's work (256 columns, 256 CUs at d2 = 512), and the last-axis passes write / read
// the line layout (16 rows of a tile are 128 contiguous bytes of 16 lines; mvn_fixed.hpp, FxRowsCfg<H>::LINES).
//
// Bins of a line are kept in the order the forward transform leaves them in a wave's registers
//     q = 64 j + 8 k + a   <->   frequency f = k + 8 a + 64 j        (k, a, j < 8)
// the taps are stored in the same order (they go through the same forward code, MF_TAPS), and the inverse
// transform undoes it: no permutation anywhere.
//
// Pipeline of a workgroup (8 waves, batches of 8 lines = 8 consecutive planes, one line per wave), iteration i, on four
// line buffers in the LDS (FW[2]: transformed lines, the filter's input; OU[2]: the filter's output):
//     wave w:  transform back + store of line w of batch i - 1         (OU[(i - 1) & 1], line w: wave-private)
//              forward transform of line w of batch i + 1 -> LDS        (FW[(i + 1) & 1], line w: wave-private)
//              global loads of line w of batch i + 2 -> registers
//     all:     filter step on the 8 lines of batch i, FW[i & 1] -> OU[i & 1]   (work item q <-> bin q of every line)
//     ONE workgroup barrier
// The two transforms of a wave are independent and run as stage pairs; the filter's lines are dealt out between a
// pair's LDS requests and its arithmetic (mf_body).  The window is indexed with compile-time register numbers: KW = K
// rounded up to a multiple of 8 slots, the filter code exists KW / 8 times per line (batch i uses variant i mod KW / 8),
// selected by a workgroup-uniform branch.
#pragma once

#include <stdexcept>

#include "mvn_dim0_direct.hpp"
#include "mvn_wave_rows.hpp"

// A wave phase here ends where lanes of ONE wave hand data to each other through the LDS: what has to be kept is
// the order of the wave's LDS instructions (they execute in order), nothing else - in particular the global loads
// requested ahead must stay in flight across it (the all-address-space fence of MVN_WPHASE waits for them).
#ifndef MF_LOCAL_FENCE
#define MF_LOCAL_FENCE 1
#endif
#if MF_LOCAL_FENCE
#define MF_FENCE() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront", "local")
#else
#define MF_FENCE() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront")
#endif

#if defined(__HIPCC__) && !defined(MVN_HOST_EMU)
#define MF_WPHASE(ctx, ...)                                         \
  {                                                                 \
    const int tid = (ctx).tid;                                      \
    auto& r = (ctx).regs;                                           \
    (void)r;                                                        \
    __VA_ARGS__;                                                    \
  }                                                                 \
  MF_FENCE();                                                       \
  __builtin_amdgcn_wave_barrier();
#else
#define MF_WPHASE(ctx, ...) MVN_PHASE(ctx, __VA_ARGS__)
#endif

#define MF_N1 512         // line length (dim1 extent)
#define MF_NT 512         // work items: one per bin of a line
#define MF_LINES 8        // lines per batch: one per wave
#define MF_PITCH 576      // cfloats per LDS line: 8 blocks of 64 + 8 spare (the exchanges' bank spreading)
#define MF_BUF (MF_LINES * MF_PITCH)
#define MF_TWL 512          // the lanes' twiddles in the LDS (mf_build_twiddles)
#define MF_LDS_CFLOATS (4 * MF_BUF + MF_TWL)

struct MidFusedParams {
  const cfloat* in;    // [d0][H][MF_N1]
  cfloat* out;         // same layout; must not alias `in` (pieces read planes their neighbours write)
  const cfloat* taps;  // [kd][H][MF_N1], bins in the kernel's own order: tap j lives in plane (j - h + kd) % kd (the PSF
                       // scattered cyclically around plane 0, as for Dim0DirectParams)
  int kd;              // planes of the tap array, >= k
  const cfloat* tw;    // exp(-2 pi i j / MF_N1), j < MF_N1 (AxisPlan::tw of the dim1 axis)
  int d0, H;
  int k, h;            // taps, k / 2;   out[z] = sum_j tap[j] in[(z + h - j) mod d0]
  int seg;             // output planes per piece; 0: whole columns (one workgroup per column)
  int mode;            // MF_CONV, or MF_TAPS: forward transform only, bins stored in the kernel's order (PSF prep)
  int packed;          // column 0 holds DC + i Nyquist of every row (RowsParams::nyq_packed), in the volume and in the
                       // taps: see "packed DC column" below
  // zcount > 0: only the output planes [zbeg, zbeg + zcount) are produced, every column as one piece (a slab whose
  // first and last h planes are halo planes: zbeg = h, zcount = d0 - 2 h, and the walk never wraps); 0: all d0 planes
  int zbeg, zcount;
  // non-finite inputs: as Dim0DirectParams (poison_peers: the other slabs' words)
  unsigned* poison;
  unsigned poison_epoch;
  int n_peers;
  unsigned* const* poison_peers;
};
enum { MF_CONV = 0, MF_TAPS = 1 };
enum { MF_DC_NONE = 0, MF_DC_LOW = 1, MF_DC_HIGH = 2, MF_DC_SELF = 3 };  // a work item's bin in the packed DC column

constexpr int mf_slots(int K) { return (K + 7) / 8 * 8; }

#ifndef MF_TW0_AHEAD  // (see mf_tw0_ahead)
#define MF_TW0_AHEAD 1
#endif
template <int K>
struct MfRegs {
  cfloat tap[K];
  cfloat w[mf_slots(K)];
  cfloat xr[8];   // the wave's line of the batch after next, as loaded (element l + 64 m in xr[m])
  cfloat t[8];    // a transform stage's values between its reads and its writes
  cfloat bad;
  cfloat t2[8];   // ... of the forward transform that runs beside a transform back
  // the filter's input of the next line (and, DC column, its partner bin), requested a line ahead.  Two of each, lines
  // taking turns: the request for line C + 1 is issued in front of line C's arithmetic, OUTSIDE the branches that select
  // the line's variant (see MF_FLINE in mf_body)
  cfloat xn[2], xp[2];
  int dcmode, qm, cpar;  // packed DC column (see below): the bin's kind, its partner bin, the lane's partner offset
};

inline long mf_pieces(const MidFusedParams& P) { return (P.seg > 0 && P.zcount <= 0) ? (P.d0 + P.seg - 1) / P.seg : 1; }
inline long mf_blocks(const MidFusedParams& P) { return P.mode == MF_TAPS ? (long)P.H * ((P.d0 + MF_LINES - 1) / MF_LINES) : (long)P.H * mf_pieces(P); }

// the job of workgroup `block`: column c, first output plane z0, output planes nout
MVN_HD void mf_job(const MidFusedParams& P, long block, int& c, int& z0, int& nout) {
  const long piece = block / P.H;
  c = (int)(block - piece * P.H);
  if (P.zcount > 0) {
    z0 = P.zbeg;
    nout = P.zcount;
  } else if (P.seg > 0) {
    z0 = (int)(piece * P.seg);
    nout = P.d0 - z0 < P.seg ? P.d0 - z0 : P.seg;
  } else {
    z0 = 0;
    nout = P.d0;
  }
}

MVN_HD int mf_tap_plane(const MidFusedParams& P, int j) {
  const int p = j - P.h;
  return p < 0 ? p + P.kd : p;
}

inline void mf_check(const MidFusedParams& P) {
  const bool taps_mode = P.mode == MF_TAPS;
  if (!P.in || !P.out || !P.tw || P.d0 < 1 || P.H < 1 || (long)P.d0 * P.H * MF_N1 >= (1L << 31) * 4 || P.seg < 0)
    throw std::invalid_argument("mvn: fused middle pass: bad arguments");
  if (P.zcount < 0 || P.zbeg < 0 || P.zbeg + P.zcount > P.d0 || P.n_peers < 0 || P.n_peers > MVN_D0_MAX_PEERS ||
      (P.n_peers > 0 && !P.poison_peers))
    throw std::invalid_argument("mvn: fused middle pass: bad plane range or peers");
  if (!taps_mode && (!P.taps || P.in == P.out || P.k < 1 || mvn_dim0_taps_template(P.k) > 31 || P.h != P.k / 2 || P.kd < P.k))
    throw std::invalid_argument("mvn: fused middle pass called outside its range");
}

// plane that step n of the walk reads: the walk starts h planes before its first output (cyclically)
MVN_HD int mf_in_plane(const MidFusedParams& P, int z0, int n) {
  int z = (z0 - P.h + n) % P.d0;
  return z < 0 ? z + P.d0 : z;
}

template <int K>
MVN_HD void mf_fetch(const MidFusedParams& P, MfRegs<K>& r, int c, int z0, int batch, int nsteps, int tid) {
  const int wv = tid >> 6, l = tid & 63;
  const int n = batch * MF_LINES + wv;
  if (n >= nsteps) return;
  const cfloat* src = P.in + ((long)mf_in_plane(P, z0, n) * P.H + c) * MF_N1 + l;
#if defined(MVN_EXPERIMENTS) && defined(MF_EXP_NOLOAD)  // timing experiment (probe builds only, WRONG results): what the loads cost
  if (batch > 1) return;
#endif
#pragma unroll
  for (int m = 0; m < 8; ++m) r.xr[m] = src[64 * m];
}

template <int K>
MVN_HD void mf_setup(const MidFusedParams& P, MfRegs<K>& r, int c, int z0, int nsteps, int tid) {
  r.bad = cmake(0.f, 0.f);
  r.dcmode = MF_DC_NONE;
  r.qm = r.cpar = 0;
#pragma unroll
  for (int j = 0; j < K; ++j) r.tap[j] = j < P.k ? P.taps[((long)mf_tap_plane(P, j) * P.H + c) * MF_N1 + tid] : cmake(0.f, 0.f);
#pragma unroll
  for (int s = 0; s < mf_slots(K); ++s) r.w[s] = cmake(0.f, 0.f);
  mf_fetch<K>(P, r, c, z0, 0, nsteps, tid);
}

// twiddle products of the transforms here: two packed instructions each (the scalar form of cmul is four; the LDS-
// staged FFT passes of mvn_fixed.hpp lost time with the packed form, this kernel is short of vector issue slots)
#ifndef MF_PK_TWIDDLE
#define MF_PK_TWIDDLE 1
#endif
MVN_HD cfloat mf_cmul(cfloat a, cfloat w) {
#if defined(MVN_PACKED) && MF_PK_TWIDDLE
  cfloat t, r;
  MVN_PK2(t, "v_pk_mul_f32", a, w, "op_sel_hi:[0,1]");                                   // (a.x w.x, a.x w.y)
  MVN_PK3(r, "v_pk_fma_f32", a, w, t, "op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]");  // + (-a.y w.y, a.y w.x)
  return r;
#else
  return cmul(a, w);
#endif
}
MVN_HD cfloat mf_cmulc(cfloat a, cfloat w) {  // a * conj(w)
#if defined(MVN_PACKED) && MF_PK_TWIDDLE
  cfloat t, r;
  MVN_PK2(t, "v_pk_mul_f32", a, w, "op_sel_hi:[0,1] neg_hi:[0,1]");       // (a.x w.x, -a.x w.y)
  MVN_PK3(r, "v_pk_fma_f32", a, w, t, "op_sel:[1,1,0] op_sel_hi:[1,0,1]");  // + (a.y w.y, a.y w.x)
  return r;
#else
  return cmulc(a, w);
#endif
}

// The lanes' twiddles live in the LDS (they would be 28 registers next to the filter's 126): stage 0, lane l:
// exp(-2 pi i l m / 512) at [m - 1][l]; stage 1: exp(-2 pi i (l & 7) m / 64) at 448 + [m - 1][l & 7].  (As 16-byte
// pairs, four reads per stage instead of seven: +4 % on the kernel - tools/mf_ab.sh, profiles/r04_mid_fused.md.)
MVN_HD void mf_build_twiddles(const MidFusedParams& P, cfloat* twl, int tid) {
  if (tid < 448) {
    const int m = tid / 64 + 1, l = tid & 63;
    twl[tid] = P.tw[l * m];
  } else if (tid < 448 + 56) {
    const int e = tid - 448, m = e / 8 + 1, j = e & 7;
    twl[tid] = P.tw[8 * j * m];
  }
}
// (fetched one by one where a stage multiplies: all seven up front cost 4 - 20 % of the kernel, as 16-byte pairs 4 %)
MVN_HD cfloat mf_tw0(const cfloat* twl, int l, int m) { return twl[(m - 1) * 64 + l]; }
MVN_HD cfloat mf_tw1(const cfloat* twl, int l, int m) { return twl[448 + (m - 1) * 8 + (l & 7)]; }

// the transforms' 8-point butterflies (one place for two timing experiments, probe builds only, WRONG results:
// MF_EXP_NO_DFT leaves them out - what do they cost? -, MF_EXP_MFMA_DUMMY issues beside each the sixteen
// v_mfma_f32_16x16x4_f32 a 16 x 16 real matrix form of it would take - does the matrix pipe run beside the rest?)
template <int SIGN>
MVN_HD void mf_dft8(cfloat* a, cfloat& sink) {
  (void)sink;
#if !(defined(MVN_EXPERIMENTS) && defined(MF_EXP_NO_DFT))
  dftR<8, SIGN>(a);
#endif
#if defined(MVN_EXPERIMENTS) && defined(MF_EXP_MFMA_DUMMY) && defined(__HIPCC__) && !defined(MVN_HOST_EMU)
  typedef float mf_f4 __attribute__((ext_vector_type(4)));
  mf_f4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
  const float x = sink.x, y = sink.y;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, c3, 0, 0, 0);
  }
  sink.x += (c0.x + c1.y) + (c2.z + c3.w);
#endif
}

// ---- forward transform of a wave's line: registers -> LDS line, bins in the order q = 64 j + 8 k + a ----
template <int K>
MVN_HD void mf_fwd0(MfRegs<K>& r, cfloat* buf, const cfloat* twl, int tid) {
  const int wv = tid >> 6, l = tid & 63;
  cfloat* line = buf + wv * MF_PITCH;
  cfloat a[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) a[m] = r.xr[m];
  mf_dft8<-1>(a, r.bad);
#if MF_TW0_AHEAD
  (void)twl;
#pragma unroll
  for (int m = 1; m < 8; ++m) a[m] = mf_cmul(a[m], r.t2[m]);
#else
#pragma unroll
  for (int m = 1; m < 8; ++m) a[m] = mf_cmul(a[m], mf_tw0(twl, l, m));
#endif
#pragma unroll
  for (int m = 0; m < 8; ++m) line[72 * m + l] = a[m];  // sub-line m (frequencies m + 8 .), element l
}
// MF_TW0_AHEAD: the first forward stage is the first thing a wave does behind the batch's barrier, and it waited there
// for its seven twiddles behind every other wave's LDS requests.  They are requested in front of the barrier instead,
// into r.t2 - which holds nothing between the last forward stage of a batch and the second one of the next.
template <int K>
MVN_HD void mf_tw0_ahead(MfRegs<K>& r, const cfloat* twl, int tid) {
#if MF_TW0_AHEAD
  const int l = tid & 63;
#pragma unroll
  for (int m = 1; m < 8; ++m) r.t2[m] = mf_tw0(twl, l, m);
#endif
}
// Every exchange stage is a pair: `_a` reads the lane's inputs from the line and computes into r.t, `_b` writes
// r.t to the places the NEXT stage reads.  On the device the two run back to back (a wave's LDS instructions
// execute in order: every lane has read before any lane writes); the test-only host emulation, which runs a phase
// lane after lane, puts a phase boundary between them.
template <int K>
MVN_HD void mf_fwd1_r(MfRegs<K>& r, const cfloat* buf, int tid) {
  const int wv = tid >> 6, l = tid & 63;
  const cfloat* p = buf + wv * MF_PITCH + 72 * (l >> 3) + (l & 7);  // sub-line l >> 3, elements (l & 7) + 8 a
#pragma unroll
  for (int m = 0; m < 8; ++m) r.t[m] = p[8 * m];
}
template <int K>
MVN_HD void mf_fwd1_c(MfRegs<K>& r, const cfloat* twl, int tid) {
  const int l = tid & 63;
  (void)l;
  mf_dft8<-1>(r.t, r.bad);
#pragma unroll
  for (int m = 1; m < 8; ++m) r.t[m] = mf_cmul(r.t[m], mf_tw1(twl, l, m));
}
// (the same stages on the second register set: in the loop the forward transform of a wave's line runs beside
// the transform back of its other line)
template <int K>
MVN_HD void mf_fwd1_r2(MfRegs<K>& r, const cfloat* buf, int tid) {
  const int wv = tid >> 6, l = tid & 63;
  const cfloat* p = buf + wv * MF_PITCH + 72 * (l >> 3) + (l & 7);
#pragma unroll
  for (int m = 0; m < 8; ++m) r.t2[m] = p[8 * m];
}
template <int K>
MVN_HD void mf_fwd1_c2(MfRegs<K>& r, const cfloat* twl, int tid) {
  const int l = tid & 63;
  (void)l;
  mf_dft8<-1>(r.t2, r.bad);
#pragma unroll
  for (int m = 1; m < 8; ++m) r.t2[m] = mf_cmul(r.t2[m], mf_tw1(twl, l, m));
}
template <int K>
MVN_HD void mf_fwd1_b2(const MfRegs<K>& r, cfloat* buf, int tid) {
#if defined(MVN_EXPERIMENTS) && defined(MF_EXP_NO_S2)  // timing experiment (probe builds only, WRONG results): the exchange between stages 1 and 2 left out
  if (tid >= 0) return;
#endif
  const int wv = tid >> 6, l = tid & 63;
  cfloat* p = buf + wv * MF_PITCH + 72 * (l >> 3) + (l & 7);
#pragma unroll
  for (int m = 0; m < 8; ++m) p[9 * m] = r.t2[m];
}
template <int K>
MVN_HD void mf_fwd2_r2(MfRegs<K>& r, const cfloat* buf, int tid) {
#if defined(MVN_EXPERIMENTS) && defined(MF_EXP_NO_S2)  // timing experiment (probe builds only, WRONG results): the exchange between stages 1 and 2 left out
  if (tid >= 0) return;
#endif
  const int wv = tid >> 6, l = tid & 63;
  const cfloat* p = buf + wv * MF_PITCH + 72 * (l >> 3) + 9 * (l & 7);
#pragma unroll
  for (int m = 0; m < 8; ++m) r.t2[m] = p[m];
}
template <int K>
MVN_HD void mf_fwd2_b2(const MfRegs<K>& r, cfloat* buf, int tid) {
  const int wv = tid >> 6, l = tid & 63;
  cfloat* line = buf + wv * MF_PITCH;
#pragma unroll
  for (int m = 0; m < 8; ++m) line[64 * m + l] = r.t2[m];
}
template <int K>
MVN_HD void mf_fwd1_a(MfRegs<K>& r, const cfloat* buf, const cfloat* twl, int tid) {
  mf_fwd1_r<K>(r, buf, tid);
  mf_fwd1_c<K>(r, twl, tid);
}
template <int K>
MVN_HD void mf_fwd1_b(const MfRegs<K>& r, cfloat* buf, int tid) {
  const int wv = tid >> 6, l = tid & 63;
  cfloat* p = buf + wv * MF_PITCH + 72 * (l >> 3) + (l & 7);
#pragma unroll
  for (int m = 0; m < 8; ++m) p[9 * m] = r.t[m];  // [a' = m][j = l & 7] with pitch 9: both sides conflict-free
}
template <int K>
MVN_HD void mf_fwd2_r(MfRegs<K>& r, const cfloat* buf, int tid) {
  const int wv = tid >> 6, l = tid & 63;
  const cfloat* p = buf + wv * MF_PITCH + 72 * (l >> 3) + 9 * (l & 7);  // lane (k, a) = (l >> 3, l & 7): its 8 values j
#pragma unroll
  for (int m = 0; m < 8; ++m) r.t[m] = p[m];
}
template <int K>
MVN_HD void mf_fwd2_a(MfRegs<K>& r, const cfloat* buf, int tid) {
  mf_fwd2_r<K>(r, buf, tid);
  mf_dft8<-1>(r.t, r.bad);
}
template <int K>
MVN_HD void mf_fwd2_b(const MfRegs<K>& r, cfloat* buf, int tid) {
  const int wv = tid >> 6, l = tid & 63;
  cfloat* line = buf + wv * MF_PITCH;
#pragma unroll
  for (int m = 0; m < 8; ++m) line[64 * m + l] = r.t[m];  // bin q = 64 j + l
}

// ---- inverse transform of a wave's line: LDS line (bins in order q) -> registers in natural order ----
template <int K>
MVN_HD void mf_inv2_r(MfRegs<K>& r, const cfloat* buf, int tid) {
  const int wv = tid >> 6, l = tid & 63;
  const cfloat* line = buf + wv * MF_PITCH;
#pragma unroll
  for (int m = 0; m < 8; ++m) r.t[m] = line[64 * m + l];
}
template <int K>
MVN_HD void mf_inv2_a(MfRegs<K>& r, const cfloat* buf, int tid) {
  mf_inv2_r<K>(r, buf, tid);
  mf_dft8<+1>(r.t, r.bad);
}
template <int K>
MVN_HD void mf_inv2_b(const MfRegs<K>& r, cfloat* buf, int tid) {
#if defined(MVN_EXPERIMENTS) && defined(MF_EXP_NO_S2)  // timing experiment (probe builds only, WRONG results): the exchange between stages 1 and 2 left out
  if (tid >= 0) return;
#endif
  const int wv = tid >> 6, l = tid & 63;
  cfloat* p = buf + wv * MF_PITCH + 72 * (l >> 3) + 9 * (l & 7);
#pragma unroll
  for (int m = 0; m < 8; ++m) p[m] = r.t[m];
}
template <int K>
MVN_HD void mf_inv1_r(MfRegs<K>& r, const cfloat* buf, int tid) {
#if defined(MVN_EXPERIMENTS) && defined(MF_EXP_NO_S2)  // timing experiment (probe builds only, WRONG results): the exchange between stages 1 and 2 left out
  if (tid >= 0) return;
#endif
  const int wv = tid >> 6, l = tid & 63;
  const cfloat* p = buf + wv * MF_PITCH + 72 * (l >> 3) + (l & 7);
#pragma unroll
  for (int m = 0; m < 8; ++m) r.t[m] = p[9 * m];
}
template <int K>
MVN_HD void mf_inv1_c(MfRegs<K>& r, const cfloat* twl, int tid) {
  const int l = tid & 63;
  (void)l;
#pragma unroll
  for (int m = 1; m < 8; ++m) r.t[m] = mf_cmulc(r.t[m], mf_tw1(twl, l, m));
  mf_dft8<+1>(r.t, r.bad);
}
template <int K>
MVN_HD void mf_inv1_a(MfRegs<K>& r, const cfloat* buf, const cfloat* twl, int tid) {
  mf_inv1_r<K>(r, buf, tid);
  mf_inv1_c<K>(r, twl, tid);
}
template <int K>
MVN_HD void mf_inv1_b(const MfRegs<K>& r, cfloat* buf, int tid) {
  const int wv = tid >> 6, l = tid & 63;
  cfloat* p = buf + wv * MF_PITCH + 72 * (l >> 3) + (l & 7);
#pragma unroll
  for (int m = 0; m < 8; ++m) p[8 * m] = r.t[m];
}
template <int K>
MVN_HD void mf_inv0_r(MfRegs<K>& r, const cfloat* buf, int tid) {
  const int wv = tid >> 6, l = tid & 63;
  const cfloat* line = buf + wv * MF_PITCH;
#pragma unroll
  for (int m = 0; m < 8; ++m) r.t[m] = line[72 * m + l];
}
template <int K>
MVN_HD void mf_inv0_c_store(MfRegs<K>& r, const cfloat* twl, cfloat* dst, int tid) {
  const int l = tid & 63;
#pragma unroll
  for (int m = 1; m < 8; ++m) r.t[m] = mf_cmulc(r.t[m], mf_tw0(twl, l, m));
  mf_dft8<+1>(r.t, r.bad);
#if defined(MVN_EXPERIMENTS) && defined(MF_EXP_NOSTORE)  // timing experiment (probe builds only, WRONG results): what the stores cost
  if (r.t[0].x == 12345.678f)
#endif
#pragma unroll
  for (int m = 0; m < 8; ++m) dst[l + 64 * m] = r.t[m];
}
template <int K>
MVN_HD void mf_inv0_store(MfRegs<K>& r, const cfloat* buf, const cfloat* twl, cfloat* dst, int tid) {
  mf_inv0_r<K>(r, buf, tid);
  mf_inv0_c_store<K>(r, twl, dst, tid);
}

// line w of batch `batch` is output m = 8 batch + w - (K - 1) of the walk (the first K - 1 steps only fill the window)
template <int K>
MVN_HD void mf_store_line(const MidFusedParams& P, MfRegs<K>& r, const cfloat* buf, const cfloat* twl, int c, int z0,
                          int nout, int batch, int tid) {
  const int m = batch * MF_LINES + (tid >> 6) - (K - 1);
  if (m < 0 || m >= nout) return;
  int z = z0 + m;
  z = z >= P.d0 ? z - P.d0 : z;
  mf_inv0_store<K>(r, buf, twl, P.out + ((long)z * P.H + c) * MF_N1, tid);
}
// (the same in two halves, for the interleaved schedule: the reads are always done, the rest only for an output line)
template <int K>
MVN_HD void mf_store_line_c(const MidFusedParams& P, MfRegs<K>& r, const cfloat* twl, int c, int z0, int nout, int batch,
                            int tid) {
  const int m = batch * MF_LINES + (tid >> 6) - (K - 1);
  if (m < 0 || m >= nout) return;
  int z = z0 + m;
  z = z >= P.d0 ? z - P.d0 : z;
  mf_inv0_c_store<K>(r, twl, P.out + ((long)z * P.H + c) * MF_N1, tid);
}

// ---------------------------------------------------------------------------------------------------------
// Packed DC column.  Column 0 of the volume holds P[y] = x0[y] + i xh[y]: the DC and the Nyquist bin of row y,
// both real (RowsParams::nyq_packed); after the transform along the line P[f] = X0[f] + i XH[f] with X0, XH
// Hermitian in f.  The two need different taps (T0, TH; the taps' own column 0 holds T0 + i TH the same way).
// In the workgroup of column 0 the work item of bin f (0 < f < 256) filters X0[f] with T0[f] and the work item of
// bin -f filters XH[f] with TH[f]:
//     X0[f] = (P[f] + conj P[-f]) / 2,    XH[f] = -i (P[f] - conj P[-f]) / 2
// - each reads its partner's bin of the line next to its own (the filter step is out of place, so nobody has
// written there yet) -, and the wave that transforms the line back puts them together again as it reads the line:
//     R[f] = R0[f] + i RH[f],    R[-f] = conj R0[f] + i conj RH[f].
// One more LDS read per bin on either side; no pass of its own, no store.
// f = 0 and f = 256 pair with themselves: X0 and XH are real there, as are the taps, so P[f] IS the pair
// (X0[f], XH[f]) and the packed tap IS (T0[f], TH[f]); the work items of those two bins filter the two halves
// separately - with the two chains every work item keeps anyway (s1 runs on the real parts, s2 on the imaginary
// parts: the result is (s1.x, s2.y) instead of the complex combination).
// ---------------------------------------------------------------------------------------------------------
MVN_HD int mf_bin_of(int f) { return 64 * (f >> 6) + 8 * (f & 7) + ((f >> 3) & 7); }
MVN_HD int mf_freq_of(int q) { return ((q >> 3) & 7) + 8 * (q & 7) + 64 * (q >> 6); }
// Hermitian parts of a pair: a = P[f], b = P[-f]  ->  X0[f], XH[f]
MVN_HD cfloat mf_herm0(cfloat a, cfloat b) { return cmake(0.5f * (a.x + b.x), 0.5f * (a.y - b.y)); }
MVN_HD cfloat mf_hermh(cfloat a, cfloat b) { return cmake(0.5f * (a.y + b.y), 0.5f * (b.x - a.x)); }


template <int K>
MVN_HD void mf_setup_dc(const MidFusedParams& P, MfRegs<K>& r, int tid) {
  const int f = mf_freq_of(tid), fm = (MF_N1 - f) & (MF_N1 - 1);
  r.qm = mf_bin_of(fm);
  r.dcmode = f == fm ? MF_DC_SELF : (f < MF_N1 / 2 ? MF_DC_LOW : MF_DC_HIGH);
  // the wave that transforms a line back holds bins 64 m + l: frequency (l >> 3) + 8 (l & 7) + 64 m, whose partner
  // 512 - f sits in bin 64 (7 - m) + cpar (lane 0: 64 (8 - m), its own bin for m = 0)
  {
    const int l = tid & 63, g = (l >> 3) + 8 * (l & 7);
    r.cpar = g == 0 ? 64 : 8 * ((64 - g) & 7) + ((64 - g) >> 3);
  }
  const cfloat* tp = P.taps + r.qm;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const cfloat a = r.tap[j], b = j < P.k ? tp[(long)mf_tap_plane(P, j) * P.H * MF_N1] : cmake(0.f, 0.f);
    // bins 1 .. 255 filter X0 with T0[f]; bins 257 .. 511 filter XH[-f] with TH[-f] (the pair seen from the other side)
    r.tap[j] = f == fm ? a : (f < MF_N1 / 2 ? mf_herm0(a, b) : mf_hermh(b, a));
  }
}

// ---- the filter step, line by line: work item q <-> bin q of every line, `in` -> `out` ----
// U = (batch mod KW / 8): step n = 8 batch + C fills slot 8 U + C; tap j multiplies the input of step n - j.
// FILL: the first K - 1 steps of a walk only fill the window.  The input of a line is requested one line ahead
// (r.xn, in the DC column r.xp as well).
// The request for line C + 1 is issued in front of line C, in code every variant of the line shares (mf_fread_ahead),
// into one of two alternating registers.  Issued inside the variants (behind line C's last multiply-add), each
// variant's read was a different instruction and the value a merge of them: the compiler read into a scratch register
// and waited for it at the end of every variant - a whole LDS latency per line, 8 per batch.
//
// MF_ALWAYS_TRANSFORM (0: the form before, kept for same-box comparisons): see mf_body.
#ifndef MF_ALWAYS_TRANSFORM
#define MF_ALWAYS_TRANSFORM 1
#endif
// Issue priorities within a batch (device only).  A SIMD's two waves - w and w + 4 of the workgroup - are served oldest
// first: wave w runs ahead (good: its transform stages wait for the LDS while wave w + 4 still multiplies), reaches the
// batch's barrier ~3000 cycles early, and wave w + 4 finishes alone at the pace of one wave (shader-clock stamps,
// profiles/r04_mid_fused.md).  MF_PRIO_MODE 3 keeps the hardware's order up to point MF_PRIO_SWITCH of the batch (the
// numbers of the MF_STAMP points) and raises the younger wave from there: both reach the barrier together.  0: none.
// (Measured and removed: the younger wave always ahead, turns line by line, every wave raised inside its transform
// stages; the forward stage in front of the batch's other requests; the loads issued later in the batch; stage B's
// arithmetic of both lines as one block on shared twiddles - profiles/r04_mid_fused.md.)
#ifndef MF_PRIO_MODE
#define MF_PRIO_MODE 3
#endif
#ifndef MF_PRIO_SWITCH
#define MF_PRIO_SWITCH 5
#endif
template <int K>
MVN_HD void mf_fread(MfRegs<K>& r, const cfloat* in, int c, bool dc, int tid) {
  r.xn[c & 1] = in[c * MF_PITCH + tid];
  if (dc) r.xp[c & 1] = in[c * MF_PITCH + r.qm];
}
template <int K>
MVN_HD cfloat mf_filter_input(MfRegs<K>& r, int c, bool dc) {
  cfloat x = r.xn[c & 1];
  if (dc) {
    const cfloat xp = r.xp[c & 1];
    const cfloat lo = mf_herm0(x, xp), hi = mf_hermh(xp, x);
    x = r.dcmode == MF_DC_LOW ? lo : (r.dcmode == MF_DC_HIGH ? hi : x);
  }
  return x;
}
template <int K, int U, int C, bool FILL>
MVN_HD void mf_filter_line(MfRegs<K>& r, const cfloat* in, cfloat* out, bool dc, int tid) {
  constexpr int KW = mf_slots(K), S = 8 * U + C;
  if (FILL) {
    r.w[S] = mf_filter_input<K>(r, C, dc);
    return;
  }
  // OLDEST taps first: they multiply values that have sat in the registers for up to K steps.  The line's own input
  // x - requested one line ahead, and behind it the reads of the transform stage this line is dealt out behind - is
  // not touched before the last multiply-adds: taken first, the wave waited for all those reads at the line's first
  // instruction.  (Slot S, which x replaces, is read by no tap j >= 1.)
  // FOUR chains (taps j mod 4), two accumulators each: an instruction and the next one of its chain are eight apart.
  // With two chains the compiler put a wait state (`s_nop`, an issue slot of its own: 4 cycles) between every pair of
  // multiply-adds - it asks for three instructions between an inline instruction and a reader of its result.
  constexpr int NC = K < 4 ? K : 4;
  cfloat a1[NC], a2[NC];
#pragma unroll
  for (int j = K - 1; j >= 1; --j) {
    const int ch = j % NC;
    const bool first = j + NC > K - 1;  // the largest tap index of its chain
    if (first)
      mvn_cmul2(a1[ch], a2[ch], r.w[(S - j + 2 * KW) % KW], r.tap[j]);
    else
      mvn_cmac2(a1[ch], a2[ch], r.w[(S - j + 2 * KW) % KW], r.tap[j]);
  }
  const cfloat x = mf_filter_input<K>(r, C, dc);
  if (K <= NC)  // (chain 0 holds tap 0 alone)
    mvn_cmul2(a1[0], a2[0], x, r.tap[0]);
  else
    mvn_cmac2(a1[0], a2[0], x, r.tap[0]);
  cfloat s1 = a1[0], s2 = a2[0];
#pragma unroll
  for (int ch = 1; ch < NC; ++ch) {
    s1 = cadd(s1, a1[ch]);
    s2 = cadd(s2, a2[ch]);
  }
  r.w[S] = x;
  r.bad = mvn_dim0_track(r.bad, x);
  cfloat o = cadd_i<+1>(s1, s2);
  if (dc) o = r.dcmode == MF_DC_SELF ? cmake(s1.x, s2.y) : o;
  out[C * MF_PITCH + tid] = o;
}
template <int K, int C>
MVN_HD void mf_fread_ahead(MfRegs<K>& r, const cfloat* in, bool dc, int tid) {
  if (C + 1 < MF_LINES) mf_fread<K>(r, in, C + 1, dc, tid);
}
template <int K, int C, int U>
MVN_HD void mf_fline_dispatch(MfRegs<K>& r, const cfloat* in, cfloat* out, int u, bool fill, bool dc, int tid) {
  if (u == U) {
    if (fill)
      mf_filter_line<K, U, C, true>(r, in, out, dc, tid);
    else
      mf_filter_line<K, U, C, false>(r, in, out, dc, tid);
    return;
  }
  if constexpr (U + 1 < mf_slots(K) / 8) mf_fline_dispatch<K, C, U + 1>(r, in, out, u, fill, dc, tid);
}
// (a FILL batch tracks nothing: the planes it reads are tracked by the piece that outputs them.  A step that
// outputs tracks its newest input in[z + h]: over the pieces of a column every plane exactly once.)

// the first stage of the transform back reads the wave's line of filter outputs; in the DC column it reads the
// partners of its bins as well and puts the two Hermitian halves together
template <int K>
MVN_HD void mf_inv2_fix_dc(MfRegs<K>& r, const cfloat* buf, int tid) {  // behind mf_inv2_r: r.t holds the wave's own bins
  const int wv = tid >> 6, l = tid & 63;
  const cfloat* line = buf + wv * MF_PITCH;
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const cfloat o = r.t[m];
    const cfloat p = line[(64 * (7 - m) + r.cpar) & (MF_N1 - 1)];
    // m < 4: frequencies below 256 (own = R0, partner = RH); m > 4: above (own = RH, partner = R0); m = 4: above,
    // but lane 0 (f = 256) - like lane 0 of m = 0 (f = 0) - holds a pair that is complete as it is
    const cfloat low = cmake(o.x - p.y, o.y + p.x), high = cmake(p.x + o.y, o.x - p.y);
    cfloat v = m < 4 ? low : high;
    if (m == 0 || m == 4) v = l == 0 ? o : v;
    r.t[m] = v;
  }
}
template <int K>
MVN_HD void mf_inv2_r_dc(MfRegs<K>& r, const cfloat* buf, int tid) {
  mf_inv2_r<K>(r, buf, tid);
  mf_inv2_fix_dc<K>(r, buf, tid);
}

template <int K>
MVN_HD void mf_report(const MidFusedParams& P, const MfRegs<K>& r) {
  if (r.bad.x != 0.f || r.bad.y != 0.f) {  // (NaN != 0)
    if (P.poison) *P.poison = P.poison_epoch;
    for (int i = 0; i < P.n_peers; ++i) *P.poison_peers[i] = P.poison_epoch;
  }
}

template <int K, typename Ctx>
MVN_HD void mf_body(const MidFusedParams& P, long block, cfloat* lds, Ctx& ctx) {
  constexpr int NT_ = MF_NT;
  (void)NT_;
  int c, z0, nout;
  mf_job(P, block, c, z0, nout);
  const int nsteps = nout + K - 1;
  const int nb = (nsteps + MF_LINES - 1) / MF_LINES;
  const bool dc = P.packed && c == 0;
  cfloat* twl = lds + 4 * MF_BUF;
  MVN_PHASE_NOSYNC(ctx, (mf_setup<K>(P, r, c, z0, nsteps, tid), mf_build_twiddles(P, twl, tid)));
  if (dc) {
    MVN_PHASE_NOSYNC(ctx, (mf_setup_dc<K>(P, r, tid)));
  }
  MVN_PHASE(ctx, (void)0);
  MVN_PHASE_NOSYNC(ctx, (mf_tw0_ahead<K>(r, twl, tid)));
  // Four line buffers: the forward transforms of batch i + 1 go to FW[(i + 1) & 1], the filter step of batch i reads
  // FW[i & 1] and writes OU[i & 1], the transforms back of batch i - 1 read OU[(i - 1) & 1].  A wave's forward and
  // backward transform of an iteration are then independent of each other (two lines, two register sets): their
  // stages run in pairs (A, B, C), the LDS reads of a pair are requested together, and the lines of the filter step
  // - arithmetic on other buffers - are dealt out between a pair's requests and its arithmetic.
  // Orders that were tried and lost (profiles/r04_mid_fused.md): the filter step as one block in front of / behind the
  // transforms (+3 .. +18 %, and the loop that selects the order per wave spills), the forward transform one stage
  // behind the transform back (+8 %: a fourth stage in the batch's chain), the two halves of the CU's waves a third
  // of a batch out of step (+6 %), a stepped schedule read from a per-wave table (290 spills).
#if defined(__HIPCC__) && !defined(MVN_HOST_EMU) && MF_PRIO_MODE == 3
  const int wave_young = mvn_uniform((ctx.tid >> 8) & 1);
#endif
  for (int i = -1; i <= nb; ++i) {
    const cfloat* fin = lds + (i & 1) * MF_BUF;
    cfloat* fout = lds + (2 + (i & 1)) * MF_BUF;
    cfloat* fwd = lds + ((i + 1) & 1) * MF_BUF;
    cfloat* inv = lds + (2 + ((i + 1) & 1)) * MF_BUF;
    // MF_ALWAYS_TRANSFORM: the transform stages run in EVERY pass of the loop - in the two passes at either end on lines
    // nobody wrote (their results go nowhere: mf_store_line_c and mf_fetch test the line's number themselves).  Behind
    // a test each stage's LDS reads were followed, inside the test's block, by a wait and copies into the registers the
    // next block expects: the latency the lines of the filter step were dealt out to cover was paid on the spot.
#if MF_ALWAYS_TRANSFORM
    constexpr bool T1 = true, T2 = true;
#else
    const bool T1 = i >= 1, T2 = i + 1 < nb;
#endif
    const bool F = i >= 0 && i < nb;
    const bool fill = i * MF_LINES + MF_LINES - 1 < K - 1;
    const int u = F ? i % (mf_slots(K) / 8) : 0;
#if defined(MVN_EXPERIMENTS) && defined(MF_STAMPS) && defined(__HIPCC__) && !defined(MVN_HOST_EMU)
    // timing experiment (probe builds only): shader-clock stamps of every batch, kept in scalar registers; those of one
    // batch of one workgroup go, per wave, to P.poison_peers' place (a debug buffer of 8 waves x 32 stamps the probe
    // passes there) at the batch's end.  (A stamp is a scalar memory read: the next wait for LDS data behind it is a wait
    // for ALL the wave's LDS requests.)
    unsigned long long mf_stamps[14];
#define MF_STAMP(n) mf_stamps[n] = __builtin_amdgcn_s_memtime();
#define MF_STAMP_FLUSH()                                                                                          \
  if (block == 100 && i == 40 && (ctx.tid & 63) == 0) {                                                           \
    unsigned long long* dst_ = reinterpret_cast<unsigned long long*>(const_cast<unsigned**>(P.poison_peers)) +    \
                               (ctx.tid >> 6) * 32;                                                               \
    for (int n_ = 0; n_ < 14; ++n_) dst_[n_] = mf_stamps[n_];                                                     \
  }
#define MF_STAMP_VMWAIT() __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) alone
#else
#define MF_STAMP(n)
#define MF_STAMP_FLUSH()
#define MF_STAMP_VMWAIT()
#endif
#if defined(__HIPCC__) && !defined(MVN_HOST_EMU) && MF_PRIO_MODE == 3
#define MF_PRIO_AT(n)                                         \
  if ((n) == 0)                                               \
    __builtin_amdgcn_s_setprio(0);                            \
  else if ((n) == MF_PRIO_SWITCH && wave_young != 0)          \
    __builtin_amdgcn_s_setprio(2);
#else
#define MF_PRIO_AT(n)
#endif
#define MF_FLINE(C)                                                                \
  if (F) {                                                                         \
    MF_WPHASE(ctx, (mf_fread_ahead<K, C>(r, fin, dc, tid), mf_fline_dispatch<K, C, 0>(r, fin, fout, u, fill, dc, tid)));  \
  }
    MF_STAMP(0)
    MF_PRIO_AT(0)
    MF_STAMP_VMWAIT()
    MF_STAMP(12)
    // stage A: first stage back (reads the line) | first forward stage (reads the registers loaded ahead)
    if (F) {
      MF_WPHASE(ctx, (mf_fread<K>(r, fin, 0, dc, tid)));
    }
    if (T1) {
      MF_WPHASE(ctx, (mf_inv2_r<K>(r, inv, tid)));
      if (dc) {
        MF_WPHASE(ctx, (mf_inv2_fix_dc<K>(r, inv, tid)));
      }
    }
    if (T2) {
      MF_WPHASE(ctx, (mf_fwd0<K>(r, fwd, twl, tid), mf_fetch<K>(P, r, c, z0, i + 2, nsteps, tid)));
    }
    MF_STAMP(1)
    MF_PRIO_AT(1)
    MF_FLINE(0)
    MF_STAMP(2)
    MF_PRIO_AT(2)
    if (T1) {
      MF_WPHASE(ctx, (mf_dft8<+1>(r.t, r.bad), mf_inv2_b<K>(r, inv, tid)));
    }
    MF_STAMP(3)
    MF_PRIO_AT(3)
    // stage B
    if (T1) {
      MF_WPHASE(ctx, (mf_inv1_r<K>(r, inv, tid)));
    }
    if (T2) {
      MF_WPHASE(ctx, (mf_fwd1_r2<K>(r, fwd, tid)));
    }
    MF_STAMP(4)
    MF_PRIO_AT(4)
    MF_FLINE(1)
    MF_FLINE(2)
    MF_STAMP(5)
    MF_PRIO_AT(5)
    if (T1) {
      MF_WPHASE(ctx, (mf_inv1_c<K>(r, twl, tid), mf_inv1_b<K>(r, inv, tid)));
    }
    if (T2) {
      MF_WPHASE(ctx, (mf_fwd1_c2<K>(r, twl, tid), mf_fwd1_b2<K>(r, fwd, tid)));
    }
    MF_STAMP(6)
    MF_PRIO_AT(6)
    // stage C
    if (T1) {
      MF_WPHASE(ctx, (mf_inv0_r<K>(r, inv, tid)));
    }
    if (T2) {
      MF_WPHASE(ctx, (mf_fwd2_r2<K>(r, fwd, tid)));
    }
    MF_STAMP(7)
    MF_PRIO_AT(7)
    MF_FLINE(3)
    MF_FLINE(4)
    MF_STAMP(8)
    MF_PRIO_AT(8)
    if (T1) {
      MF_WPHASE(ctx, (mf_store_line_c<K>(P, r, twl, c, z0, nout, i - 1, tid)));
    }
    if (T2) {
      MF_WPHASE(ctx, (mf_dft8<-1>(r.t2, r.bad), mf_fwd2_b2<K>(r, fwd, tid)));
    }
    MF_STAMP(9)
    MF_PRIO_AT(9)
    MF_FLINE(5)
    MF_FLINE(6)
    MF_FLINE(7)
    MF_STAMP(10)
    MF_PRIO_AT(10)
#undef MF_FLINE
    MF_WPHASE(ctx, (mf_tw0_ahead<K>(r, twl, tid)));
    MVN_PHASE(ctx, (void)0);
    MF_STAMP(11)
    MF_PRIO_AT(11)
    MF_STAMP_FLUSH()
  }
  MVN_PHASE_NOSYNC(ctx, (mf_report<K>(P, r)));
}

// PSF preparation: lines [z][c][.] of a small stack (the PSF's planes after the last-axis pass) -> forward
// transform along dim1, bins stored in the order the filter keeps them.  One batch of 8 planes of one column per
// workgroup.
MVN_HD void mf_taps_load(const MidFusedParams& P, MfRegs<1>& r, long batch, int c, int tid) {
  const int l = tid & 63, z = (int)batch * MF_LINES + (tid >> 6);
  if (z >= P.d0) return;
  const cfloat* src = P.in + ((long)z * P.H + c) * MF_N1 + l;
#pragma unroll
  for (int m = 0; m < 8; ++m) r.xr[m] = src[64 * m];
}
MVN_HD void mf_taps_store(const MidFusedParams& P, const cfloat* lds, long batch, int c, int tid) {
  const int l = tid & 63, wv = tid >> 6, z = (int)batch * MF_LINES + wv;
  if (z >= P.d0) return;
  cfloat* dst = P.out + ((long)z * P.H + c) * MF_N1 + l;
#pragma unroll
  for (int m = 0; m < 8; ++m) dst[64 * m] = lds[wv * MF_PITCH + 64 * m + l];
}
template <typename Ctx>
MVN_HD void mf_taps_body(const MidFusedParams& P, long block, cfloat* lds, Ctx& ctx) {
  constexpr int NT_ = MF_NT;
  (void)NT_;
  const long batch = block / P.H;
  const int c = (int)(block - batch * P.H);
  cfloat* twl = lds + 4 * MF_BUF;
  MVN_PHASE(ctx, (mf_taps_load(P, r, batch, c, tid), mf_build_twiddles(P, twl, tid)));
  MF_WPHASE(ctx, (mf_tw0_ahead<1>(r, twl, tid)));
  MF_WPHASE(ctx, (mf_fwd0<1>(r, lds, twl, tid)));
  MF_WPHASE(ctx, (mf_fwd1_a<1>(r, lds, twl, tid)));
  MF_WPHASE(ctx, (mf_fwd1_b<1>(r, lds, tid)));
  MF_WPHASE(ctx, (mf_fwd2_a<1>(r, lds, tid)));
  MF_WPHASE(ctx, (mf_fwd2_b<1>(r, lds, tid)));
  MVN_PHASE_NOSYNC(ctx, (mf_taps_store(P, lds, batch, c, tid)));
}
