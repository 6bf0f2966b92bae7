// mvn_wave_rows.hpp -- last-axis passes for d2 = 512 (H = 256 complex bins per row) in which a ROW
// never leaves its (half) wavefront: no workgroup barrier, half the LDS round trips of the tiled
// kernels of mvn_fixed.hpp.
//
// Why (round 2, PMC + A/B on MI355X): the tiled fused pass kx_rows_c2r_r2c<256> keeps the LDS busy
// for ~55 % of its run time (8 store + 9 load sweeps over the tile, stores at ~80 B/clk/CU) with 30 %
// of those cycles lost to bank conflicts of the transposed tile, behind 9 workgroup barriers;
// dropping two of the round trips (timing experiment) took 11 % off the pass.  Here:
//   * a half-wave (32 lanes) owns one row, a lane owns 8 of its 256 bins in every phase; rows are
//     exchanged between phases through a private 2.3 KB LDS row buffer -- 4 exchanges per fused
//     pass, all conflict-free but the first / last (see wr_f) -- and the only synchronisation is the
//     in-order execution of one wave's LDS instructions;
//   * the real <-> half-complex step needs bins k and H-k together: the lane assignment of the
//     first / last phase puts both into the SAME lane (a lane holds one group of 4 neighbouring
//     positions and the group of their partners), so that step and the radix-4 stage next to it run
//     in registers right behind the global loads / in front of the global stores;
//   * a lane's twiddles of the real <-> complex step do not depend on the row and stay in registers.
// Radix plan 256 = 8 * 8 * 4 (as fx_radix): position p = 32 d0 + 4 d1 + d2 holds bin
// k = d0 + 8 d1 + 64 d2 (spectra are kept in position order).
//
// Phases of the fused c2r + pointwise + r2c pass (reference: cufftExecC2R + pointwise kernel +
// cufftExecR2C, inc/gpu_convolve.cuh:140-141 + inc/cuda_kernels.cuh:14-112 + inc/gpu_convolve.cuh:121):
//   A  global -> registers, half-complex -> complex step, inverse radix-4 stage      -> LDS
//   B  inverse radix-8 stage (M = 4)                                           LDS -> LDS
//   C  inverse radix-8 stage (M = 32) -> pointwise epilogue -> forward radix-8 stage   LDS -> LDS
//   D  forward radix-8 stage (M = 4)                                           LDS -> LDS
//   E  forward radix-4 stage, complex -> half-complex step, registers -> global
// The plain r2c pass is C' (real rows in), D, E; the plain c2r pass A, B, C'' (epilogue stores).
#pragma once

#include "mvn_fixed.hpp"

struct WrCfg {
  static constexpr int H = 256;        // complex bins per row (d2 = 512)
  static constexpr int LPR = 32;       // lanes per row
  static constexpr int WAVES = 4;      // waves per workgroup
  static constexpr int NT = 64 * WAVES;
  static constexpr int RB = 288;       // cfloats per LDS row buffer: 256 + 4 spare per 32 positions
  static constexpr int TW = fx_twsize(H);  // stage-ordered twiddle table (stage 0: 32 rows, stage 1: 4 rows)
  static constexpr int PWT = 4 * LPR;   // the lanes' pair twiddles, [j][lane] (update forms, see wr_pw)
  static constexpr int lds_cfloats = TW + WAVES * 2 * RB + PWT;
  static_assert(fx_radix(H, 0) == 8 && fx_radix(H, 1) == 8 && fx_radix(H, 2) == 4 && fx_nstages(H) == 3,
                "wave-row kernels are written for 256 = 8 * 8 * 4");
};

enum MvnWaveRowsMode { MVN_WR_R2C = 0, MVN_WR_C2R = 1, MVN_WR_C2R_R2C = 2 };

// LDS index of position p of a row: 4 spare entries after every 32 positions.  Phases B - D then
// touch 64 different banks with every half-wave access (lanes = neighbouring positions, blocks of
// 32 positions 8 banks apart); phases A / E (a lane owns 4 neighbouring positions) use 16-byte
// accesses that are 2-way conflicted.
MVN_HD int wr_f(int p) { return p + 4 * (p >> 5); }

// groups (4 neighbouring positions each) of lane t in phases A / E: a group and the group of its
// partners under k <-> H - k.  Group g = 8 d0 + d1 holds bins d0 + 8 d1 + 64 j (j = 0..3).
//   g >= 8 (d0 >= 1):  partner group 71 - g, bin j <-> 3 - j
//   g = 1..7 (d0 = 0): partner group 8 - g,  bin j <-> 3 - j   (g = 4 is its own partner)
//   g = 0:             bins 0 (pairs with the Nyquist bin), 64 <-> 192, 128 (its own partner)
// Lane 0 takes the two self-paired groups 0 and 4, lanes 1..3 the pairs (t, 8 - t), lanes 4..31 the
// pairs (t + 4, 67 - t).
MVN_HD void wr_groups(int t, int& ga, int& gb) {
  if (t == 0) {
    ga = 0;
    gb = 4;
  } else if (t < 4) {
    ga = t;
    gb = 8 - t;
  } else {
    ga = t + 4;
    gb = 67 - t;
  }
}

struct WrRegs {
  cfloat z[8];   // the lane's 8 bins of its row
  cfloat ea[8];  // epilogue operands of the 16 reals the lane finishes in phase C
  cfloat eb[8];
  cfloat pw[4];  // exp(-2 pi i k / d2) of the lane's (k, H - k) pairs
  qfloat nx[4];  // the lane's two groups of the NEXT row of its half-wave, requested one row ahead
};

// exp(-2 pi i k / d2) for 0 < k < H from the table of k <= H/2 (P.twr)
MVN_HD cfloat wr_root(const cfloat* twr, int k) {
  if (k <= WrCfg::H / 2) return twr[k];
  const cfloat w = twr[WrCfg::H - k];
  return cmake(-w.x, w.y);  // exp(-2 pi i (H - k') / 2H) = -conj(exp(-2 pi i k' / 2H))
}

// The stage twiddle tables in the LDS, TRANSPOSED: entry k of butterfly row j sits at [k][j], so the
// lanes of a half-wave (which own rows j = t, or j = t & 3 in the middle stages) read neighbouring
// 8-byte words.  In the row-major order of the tiled kernels lane t's row starts 64 bytes behind
// lane t - 1's and 16 lanes share two bank groups: measured (SQ_LDS_BANK_CONFLICT) 58 % of the
// pass's LDS cycles were conflict cycles, nearly all of them these reads.
MVN_HD void wr_copy_tables(cfloat* dst, const cfloat* src, int tid) {
  constexpr int S1 = fx_twoff(WrCfg::H, 1);  // 32 rows of 8 (stage 0), then 4 rows of 8 (stage 1)
  static_assert(S1 == 256 && WrCfg::TW == 288, "twiddle table of 256 = 8 * 8 * 4");
  // neighbouring lanes write neighbouring LDS words (the gather is on the global side, where it
  // costs nothing: the table is 2.3 KB of L2-resident data); the other way round the stores of a
  // half-wave hit one bank group, which showed in the conflict counters once a launch had 16x more
  // workgroups
  for (int d = tid; d < WrCfg::TW; d += WrCfg::NT) {
    const int i = d < S1 ? (d & 31) * 8 + (d >> 5) : S1 + ((d - S1) & 3) * 8 + ((d - S1) >> 2);
    dst[d] = src[i];
  }
}
// twiddles of butterfly row j of stage 0 / stage 1 (entry 0 is 1 and never read)
MVN_HD void wr_tw0(const cfloat* tws, int j, cfloat* tw) {
#pragma unroll
  for (int k = 1; k < 8; ++k) tw[k] = tws[k * 32 + j];
}
MVN_HD void wr_tw1(const cfloat* tws, int j, cfloat* tw) {
#pragma unroll
  for (int k = 1; k < 8; ++k) tw[k] = tws[fx_twoff(WrCfg::H, 1) + k * 4 + j];
}

MVN_HD cfloat* wr_pwt(cfloat* rows) { return rows + WrCfg::WAVES * 2 * WrCfg::RB; }

// once per kernel: the lane's pair twiddles
MVN_HD void wr_setup(const RowsParams& P, WrRegs& r, cfloat* rows, int tid) {
  const int t = tid & 31;
  if (t == 0) {
    r.pw[0] = wr_root(P.twr, 64);  // group 0: bins 64 <-> 192
    r.pw[1] = wr_root(P.twr, 32);  // group 4: bins 32 <-> 224
    r.pw[2] = wr_root(P.twr, 96);  //          bins 96 <-> 160
    r.pw[3] = cmake(1.f, 0.f);
  } else {
    int ga, gb;
    wr_groups(t, ga, gb);
    const int k0 = (ga >> 3) + 8 * (ga & 7);
#pragma unroll
    for (int j = 0; j < 4; ++j) r.pw[j] = wr_root(P.twr, k0 + 64 * j);
  }
  if (tid < WrCfg::LPR) {  // the same for every half-wave: one copy per workgroup (wr_pw)
#pragma unroll
    for (int j = 0; j < 4; ++j) wr_pwt(rows)[j * WrCfg::LPR + t] = r.pw[j];
  }
}

// The update forms sit at the 128-register limit of four waves per SIMD: they re-read the lane's four
// pair twiddles from a 1 KB LDS table in phases A and E instead of keeping them in 8 registers.
template <int EPI>
constexpr bool wr_pw_in_lds() {
  return EPI == MVN_EPI_UPDATE || EPI == MVN_EPI_DELTA;
}
template <int EPI>
MVN_HD void wr_pw(const WrRegs& r, const cfloat* rows, int t, cfloat* pw) {
  const cfloat* tab = rows + WrCfg::WAVES * 2 * WrCfg::RB;
#pragma unroll
  for (int j = 0; j < 4; ++j) pw[j] = wr_pw_in_lds<EPI>() ? tab[j * WrCfg::LPR + t] : r.pw[j];
}

// half-complex -> complex step on one pair: X[k], X[H-k] -> Z[k], Z[H-k]
//   E = X[k] + conj X[H-k],  O = (X[k] - conj X[H-k]) exp(+2 pi i k / d2)
//   Z[k] = E + i O,  Z[H-k] = conj(E) + i conj(O)
MVN_HD void wr_pre_pair(cfloat& xk, cfloat& xm, cfloat w) {
  const cfloat E = cadd_c(xk, xm);
  const cfloat O = cmulc(csub_c(xk, xm), w);
  xk = cadd_i<+1>(E, O);
  xm = cconj_add_i<+1>(E, O);
}

// complex -> half-complex step on one pair: Z[k], Z[H-k] -> X[k], X[H-k]
//   E = (Z[k] + conj Z[H-k]) / 2,  G = exp(-2 pi i k / d2) (Z[k] - conj Z[H-k]) / 2
//   X[k] = E - i G,  X[H-k] = conj(E) - i conj(G)
MVN_HD void wr_post_pair(cfloat& zk, cfloat& zm, cfloat w) {
  const cfloat E = cscale(cadd_c(zk, zm), 0.5f);
  const cfloat G = cmul(cscale(csub_c(zk, zm), 0.5f), w);
  zk = cadd_i<-1>(E, G);
  zm = cconj_add_i<-1>(E, G);
}

// row of lane `tid` in sweep `it` of workgroup `block` (two rows per wave: one per half-wave);
// consecutive waves take consecutive row pairs
#ifndef MVN_PROBE_TILE
#define MVN_PROBE_TILE(t) (t)  // timing probes only (mvn_kernels.hip, -DMVN_PROBE)
#endif
MVN_HD long wr_row(long block, long nblocks, long it, int tid) {
  const long pair = (it * nblocks + block) * WrCfg::WAVES + (tid >> 6);
  return 2 * (long)MVN_PROBE_TILE(pair) + ((tid >> 5) & 1);
}

// component-wise select (a select of the aggregates goes through a stack slot)
MVN_HD qfloat wr_pick(int c, qfloat a, qfloat b) {
  return qmake(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w);
}

MVN_HD cfloat* wr_buf(cfloat* rows, int tid) { return rows + (tid >> 5) * WrCfg::RB; }

// ---- phase A: spectral row in, half-complex -> complex, inverse radix-4 stage -> LDS -----------
// request the lane's two groups of spectral row `row` (phase A consumes them one sweep later, so the
// loads are in flight during phases B - E of the row before)
MVN_HD void wr_fetch_row(const RowsParams& P, long row, WrRegs& r, int tid) {
  if (row >= P.rows) return;
  int ga, gb;
  wr_groups(tid & 31, ga, gb);
  const qfloat* src = reinterpret_cast<const qfloat*>(P.in_cplx + row * P.C);
  r.nx[0] = src[2 * ga];
  r.nx[1] = src[2 * ga + 1];
  r.nx[2] = src[2 * gb];
  r.nx[3] = src[2 * gb + 1];
}

// the spectral row is requested one sweep ahead where the registers allow it: the divide and the
// plain forms (the update forms hold psi and the weights as well and would spill)
template <int EPI>
constexpr bool wr_prefetch() {
  return EPI == MVN_EPI_DIVIDE || EPI == MVN_EPI_STORE;
}

// The operands of the pointwise step (view, or psi and weights) of a row are requested one SWEEP
// ahead as well: right behind the pointwise step of the row before, whose operand registers have
// just become free (phase C).  Requested in phase A of the row itself (MVN_WR_EPI_AHEAD=0) they are
// one LDS stage ahead of their use, far less than a loaded HBM round trip, and every wave of the
// SIMD waits for them in phase C.
#ifndef MVN_WR_EPI_AHEAD
#define MVN_WR_EPI_AHEAD 1
#endif
// The fused update forms cannot hold the next spectral row across phases A - C next to psi, the
// weights and the f64 chains; they request it at the start of phase D (MVN_WR_NX_IN_D=0: at the
// start of phase A of the row itself, consumed at once).
#ifndef MVN_WR_NX_IN_D
#define MVN_WR_NX_IN_D 0
#endif
template <int MODE, int EPI>
constexpr bool wr_nx_in_d() {
  return MVN_WR_NX_IN_D && MODE == MVN_WR_C2R_R2C && !wr_prefetch<EPI>();
}

template <int EPI>
MVN_HD void wr_fetch_epi(const RowsParams& P, long row, WrRegs& r, int tid) {
  if (row >= P.rows) return;
  const int t = tid & 31;  // the lane finishes reals 2 (t + 32 jo), 2 (t + 32 jo) + 1 in phase C
  if (EPI != MVN_EPI_STORE) {
    const float* pa = (EPI == MVN_EPI_DIVIDE ? P.epi.view : P.epi.psi) + row * P.RP + 2 * t;
#pragma unroll
    for (int jo = 0; jo < 8; ++jo) r.ea[jo] = *reinterpret_cast<const cfloat*>(pa + 64 * jo);
  }
  if (EPI == MVN_EPI_UPDATE || EPI == MVN_EPI_DELTA) {
    const float* pb = P.epi.weights + row * P.RP + 2 * t;
#pragma unroll
    for (int jo = 0; jo < 8; ++jo) r.eb[jo] = *reinterpret_cast<const cfloat*>(pb + 64 * jo);
  }
}

template <int MODE, int EPI>
MVN_HD void wr_phase_a(const RowsParams& P, long row, long next_row, cfloat* rows, WrRegs& r, int tid) {
  if (row >= P.rows) return;
  const int t = tid & 31;
  int ga, gb;
  wr_groups(t, ga, gb);
  if (!wr_prefetch<EPI>() && !wr_nx_in_d<MODE, EPI>()) wr_fetch_row(P, row, r, tid);
  const qfloat a0 = r.nx[0], a1 = r.nx[1], b0 = r.nx[2], b1 = r.nx[3];
  cfloat za[4] = {cmake(a0.x, a0.y), cmake(a0.z, a0.w), cmake(a1.x, a1.y), cmake(a1.z, a1.w)};
  cfloat zb[4] = {cmake(b0.x, b0.y), cmake(b0.z, b0.w), cmake(b1.x, b1.y), cmake(b1.z, b1.w)};
  cfloat pw[4];
  wr_pw<EPI>(r, rows, t, pw);
  if (t == 0) {
    // imaginary parts of the DC and Nyquist bins are ignored, as FFTW's c2r does
    const float xh = P.nyq_packed ? za[0].y : P.in_nyq[row].x;
    za[0] = cmake(za[0].x + xh, za[0].x - xh);
    wr_pre_pair(za[1], za[3], pw[0]);
    za[2] = cmake(2.f * za[2].x, -2.f * za[2].y);  // bin H/2 pairs with itself: Z = 2 conj(X)
    wr_pre_pair(zb[0], zb[3], pw[1]);
    wr_pre_pair(zb[1], zb[2], pw[2]);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) wr_pre_pair(za[j], zb[3 - j], pw[j]);
  }
  dftR<4, +1>(za);  // last stage (M = 1): no twiddles
  dftR<4, +1>(zb);
  cfloat* buf = wr_buf(rows, tid);
  qfloat* da = reinterpret_cast<qfloat*>(buf + wr_f(4 * ga));
  qfloat* db = reinterpret_cast<qfloat*>(buf + wr_f(4 * gb));
  // a lane's two 16-byte halves go out in an order that depends on the lane (lanes 4..7 of every
  // eight start with the upper half): the eight lanes one ds_write_b128 serves together then hit
  // eight different 16-byte bank groups; both halves in program order are 2-way conflicted.  The
  // fences keep the selects of one group from overlapping the other group's registers and the
  // global loads below from being hoisted over both (that spilled).
  const int hi = (t >> 2) & 1;
  {
    const qfloat u = qmake(za[0].x, za[0].y, za[1].x, za[1].y), v = qmake(za[2].x, za[2].y, za[3].x, za[3].y);
    da[hi] = wr_pick(hi, v, u);
    da[hi ^ 1] = wr_pick(hi, u, v);
  }
  {
    const qfloat u = qmake(zb[0].x, zb[0].y, zb[1].x, zb[1].y), v = qmake(zb[2].x, zb[2].y, zb[3].x, zb[3].y);
    db[hi] = wr_pick(hi, v, u);
    db[hi ^ 1] = wr_pick(hi, u, v);
  }
  MVN_SCHED_FENCE();
  if (wr_prefetch<EPI>()) wr_fetch_row(P, next_row, r, tid);
  if (!MVN_WR_EPI_AHEAD) wr_fetch_epi<EPI>(P, row, r, tid);
}

// ---- phases B / D: the radix-8 stage with M = 4 through the LDS ---------------------------------
template <int SIGN>
MVN_HD void wr_phase_mid(long row, long nrows, cfloat* rows, const cfloat* tws, int tid) {
  if (row >= nrows) return;
  const int t = tid & 31;
  const int blk = t >> 2, j2 = t & 3;
  cfloat* p = wr_buf(rows, tid) + 36 * blk + j2;  // wr_f(32 blk + j2 + 4 j) = 36 blk + j2 + 4 j
  cfloat a[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = p[4 * j];
  cfloat tw[8];
  wr_tw1(tws, j2, tw);
  if (SIGN > 0) {  // inverse: decimation in time, twiddles first
#pragma unroll
    for (int k = 1; k < 8; ++k) a[k] = cmulc(a[k], tw[k]);
  }
  dftR<8, SIGN>(a);
  if (SIGN < 0) {
#pragma unroll
    for (int k = 1; k < 8; ++k) a[k] = cmul(a[k], tw[k]);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) p[4 * j] = a[j];
}

// ---- phase C: outermost stage (M = 32) both ways around the pointwise step ----------------------
// MODE R2C: real row in -> forward stage.  C2R: inverse stage -> epilogue stores.  C2R_R2C: both.
template <int MODE, int EPI>
MVN_HD void wr_phase_c(const RowsParams& P, long row, long next_row, cfloat* rows, const cfloat* tws,
                       WrRegs& r, int tid) {
  if (row >= P.rows) return;
  const int t = tid & 31;
  cfloat* p = wr_buf(rows, tid) + t;  // wr_f(t + 32 j) = t + 36 j
  cfloat a[8];
  cfloat tw[8];
  wr_tw0(tws, t, tw);
  const long i0 = row * P.RP + 2 * t;  // the lane's reals: i0 + 64 jo, i0 + 64 jo + 1
  if (MODE == MVN_WR_R2C) {
    const float* src = P.in_real + i0;
#pragma unroll
    for (int jo = 0; jo < 8; ++jo) a[jo] = *reinterpret_cast<const cfloat*>(src + 64 * jo);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = p[36 * j];
#pragma unroll
    for (int k = 1; k < 8; ++k) a[k] = cmulc(a[k], tw[k]);
    dftR<8, +1>(a);  // a[jo] = z[t + 32 jo] = (x[2 j], x[2 j + 1])
    if (MODE == MVN_WR_C2R) {
#pragma unroll
      for (int jo = 0; jo < 8; ++jo) {
        mvn_epilogue_pair_t<EPI>(P.epi, P.out_real, i0 + 64 * jo, a[jo], r.ea[jo], r.eb[jo]);
        if (jo & 1) MVN_SCHED_FENCE();  // four values at a time: the f64 chains of all 16 would not fit
      }
      if (MVN_WR_EPI_AHEAD) wr_fetch_epi<EPI>(P, next_row, r, tid);
      return;
    }
#pragma unroll
    for (int jo = 0; jo < 8; ++jo) {
      a[jo] = fx_epilogue_pair_value<EPI>(P.epi, i0 + 64 * jo, a[jo], r.ea[jo], r.eb[jo]);
      if (jo & 1) MVN_SCHED_FENCE();
    }
    if (MVN_WR_EPI_AHEAD) wr_fetch_epi<EPI>(P, next_row, r, tid);
  }
  dftR<8, -1>(a);
  if (MODE == MVN_WR_C2R_R2C) {
    // the twiddle row is read again rather than kept across the pointwise step (16 registers
    // next to its operands and the f64 chains of the update); the offset is made opaque so that
    // the two reads are not merged
    int off = t;
    MVN_JIT_ADDRESS(off);
    wr_tw0(tws, off, tw);
  }
#pragma unroll
  for (int k = 1; k < 8; ++k) a[k] = cmul(a[k], tw[k]);
#pragma unroll
  for (int j = 0; j < 8; ++j) p[36 * j] = a[j];
}

// ---- phase E: forward radix-4 stage, complex -> half-complex, spectral row out -------------------
template <int EPI>
MVN_HD void wr_phase_e(const RowsParams& P, long row, const cfloat* rows, const WrRegs& r, int tid) {
  if (row >= P.rows) return;
  const int t = tid & 31;
  int ga, gb;
  wr_groups(t, ga, gb);
  const cfloat* buf = rows + (tid >> 5) * WrCfg::RB;
  const qfloat* sa = reinterpret_cast<const qfloat*>(buf + wr_f(4 * ga));
  const qfloat* sb = reinterpret_cast<const qfloat*>(buf + wr_f(4 * gb));
  const qfloat a0 = sa[0], a1 = sa[1], b0 = sb[0], b1 = sb[1];
  cfloat za[4] = {cmake(a0.x, a0.y), cmake(a0.z, a0.w), cmake(a1.x, a1.y), cmake(a1.z, a1.w)};
  cfloat zb[4] = {cmake(b0.x, b0.y), cmake(b0.z, b0.w), cmake(b1.x, b1.y), cmake(b1.z, b1.w)};
  dftR<4, -1>(za);
  dftR<4, -1>(zb);
  cfloat pw[4];
  wr_pw<EPI>(r, rows, t, pw);
  if (t == 0) {
    const cfloat z0 = za[0];
    if (P.nyq_packed) {
      za[0] = cmake(z0.x + z0.y, z0.x - z0.y);            // DC + i Nyquist (RowsParams::nyq_packed)
    } else {
      za[0] = cmake(z0.x + z0.y, 0.f);                    // DC
      P.out_nyq[row] = cmake(z0.x - z0.y, 0.f);           // Nyquist, kept in its own plane
    }
    wr_post_pair(za[1], za[3], pw[0]);
    za[2] = cconj(za[2]);                                 // bin H/2 pairs with itself
    wr_post_pair(zb[0], zb[3], pw[1]);
    wr_post_pair(zb[1], zb[2], pw[2]);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) wr_post_pair(za[j], zb[3 - j], pw[j]);
  }
  qfloat* dst = reinterpret_cast<qfloat*>(P.out_cplx + row * P.C);
  dst[2 * ga] = qmake(za[0].x, za[0].y, za[1].x, za[1].y);
  dst[2 * ga + 1] = qmake(za[2].x, za[2].y, za[3].x, za[3].y);
  dst[2 * gb] = qmake(zb[0].x, zb[0].y, zb[1].x, zb[1].y);
  dst[2 * gb + 1] = qmake(zb[2].x, zb[2].y, zb[3].x, zb[3].y);
}

// A phase of the wave-row kernels ends at a point where lanes of ONE wave hand data to each other
// through the LDS.  A wave's LDS instructions execute in order, so no barrier instruction is
// needed on the device; the test-only host emulation runs the phase for every lane in turn.
#if defined(__HIPCC__) && !defined(MVN_HOST_EMU)
#define MVN_WPHASE(ctx, ...)                              \
  {                                                       \
    const int tid = (ctx).tid;                            \
    auto& r = (ctx).regs;                                 \
    (void)r;                                              \
    __VA_ARGS__;                                          \
  }                                                       \
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  \
  __builtin_amdgcn_wave_barrier();
#else
#define MVN_WPHASE(ctx, ...) MVN_PHASE(ctx, __VA_ARGS__)
#endif

// workgroup `block` of `nblocks`: sweeps over the row pairs (it nblocks + block) WAVES + wave
template <int MODE, int EPI, typename Ctx>
MVN_HD void wr_rows_body(const RowsParams& P, long block, long nblocks, cfloat* lds, Ctx& ctx) {
  constexpr int NT_ = WrCfg::NT;
  (void)NT_;
  cfloat* tws = lds;
  cfloat* rows = lds + WrCfg::TW;
  MVN_PHASE(ctx, (wr_copy_tables(tws, P.ax.tws, tid), wr_setup(P, r, rows, tid),
                  (MODE != MVN_WR_R2C && (wr_prefetch<EPI>() || wr_nx_in_d<MODE, EPI>()))
                      ? wr_fetch_row(P, wr_row(block, nblocks, 0, tid), r, tid)
                      : (void)0,
                  (MODE != MVN_WR_R2C && MVN_WR_EPI_AHEAD) ? wr_fetch_epi<EPI>(P, wr_row(block, nblocks, 0, tid), r, tid)
                                                           : (void)0));
  const long pairs = (P.rows + 1) / 2;
  const long per_sweep = nblocks * WrCfg::WAVES;
  const long sweeps = (pairs + per_sweep - 1) / per_sweep;
  for (long it = 0; it < sweeps; ++it) {
    if (MODE != MVN_WR_R2C) {
      MVN_WPHASE(ctx, (wr_phase_a<MODE, EPI>(P, wr_row(block, nblocks, it, tid),
                                             wr_row(block, nblocks, it + 1, tid), rows, r, tid)));
      MVN_WPHASE(ctx, (wr_phase_mid<+1>(wr_row(block, nblocks, it, tid), P.rows, rows, tws, tid)));
    }
    MVN_WPHASE(ctx, (wr_phase_c<MODE, EPI>(P, wr_row(block, nblocks, it, tid), wr_row(block, nblocks, it + 1, tid),
                                           rows, tws, r, tid)));
    if (MODE != MVN_WR_C2R) {
      MVN_WPHASE(ctx, ((wr_nx_in_d<MODE, EPI>() && MVN_WR_NX_IN_D == 1)
                           ? wr_fetch_row(P, wr_row(block, nblocks, it + 1, tid), r, tid)
                           : (void)0,
                       wr_phase_mid<-1>(wr_row(block, nblocks, it, tid), P.rows, rows, tws, tid)));
      MVN_WPHASE(ctx, ((wr_nx_in_d<MODE, EPI>() && MVN_WR_NX_IN_D == 2)
                           ? wr_fetch_row(P, wr_row(block, nblocks, it + 1, tid), r, tid)
                           : (void)0,
                       wr_phase_e<EPI>(P, wr_row(block, nblocks, it, tid), rows, r, tid)));
    }
  }
}
