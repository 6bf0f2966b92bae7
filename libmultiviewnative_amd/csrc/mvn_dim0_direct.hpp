// mvn_dim0_direct.hpp -- the dim0 leg of a convolution as a DIRECT cyclic convolution (round 3).
//
// The reference convolves by 3-D FFT x PSF spectrum x inverse 3-D FFT (inc/gpu_convolve.cuh:113-142,
// inc/cpu_convolve.h:217-291).  A PSF has K = kernel_dims[0] planes along dim0 (31 in BASELINE.json)
// against d0 = 512 of the volume.  In the spectral domain of dims 1 and 2 - where the volume is after
// the forward last-axis and dim1 passes - the 3-D cyclic convolution is, for every bin b of the
// (d1, C) plane, a 1-D cyclic convolution along dim0 with the K values the 2-D transformed PSF has
// there:
//
//     out[z][b] = sum_{j=0}^{K-1} tap[j][b] * in[(z + h - j) mod d0][b],      h = K / 2
//
// (tap j = PSF plane j, whose voxel h sits on the origin: inc/padd_utils.h:11-40).  One pass over the
// volume, 2 K packed fused multiply-adds per bin, instead of forward dim0 FFT x PSF x inverse dim0
// FFT - and, what counts on MI355X (profiles/r03_mem_counters.md: a pass costs 0.16 ms per READ
// volume at 512^3, its writes ride along), it reads the volume plus K / d0 of a volume of taps (6 %)
// where the fused FFT pass reads the volume plus a whole PSF spectrum: 0.20 - 0.22 ms instead of
// 0.32 ms at K = 31 (tools/dim0_direct_probe.hip), and the PSF costs 2 K / d0 volumes of HBM per
// view instead of 2.
//
// A work item owns ONE bin and walks along dim0 with its K taps and the K + PF most recent / next
// input values in registers (PF planes are requested ahead of their first use).  The walk is
// unrolled K + PF times so that every register index is a compile-time constant:
//   x_j = in[z + h - j] (j = -PF .. K-1) sits in slot (j + PF - u) mod (K + PF) at unrolled step u.
#pragma once

#include "mvn_fft_core.hpp"

struct Dim0DirectParams {
  const cfloat* in;    // [d0][plane]
  cfloat* out;         // [d0][plane]; must not alias `in`
  const cfloat* taps;  // [kd][plane]: tap j lives in plane (j - h + kd) % kd (the PSF scattered cyclically
                       // into kd >= k + 1 planes, then transformed along dims 1 and 2)
  int d0;              // planes of the volume; d0 >= K + MVN_D0_PF
  int k;               // PSF planes (taps); the kernel is instantiated for K = k | 1 (a zero tap appended)
  int kd;              // planes of the tap array
  int h;               // k / 2
  long plane;          // bins per plane
  int stagger;         // workgroup w starts its (cyclic) walk at plane (w * stagger) mod d0: 0 = all at plane 0
  int seg1;            // 0: a work item walks its whole column; > 0: the columns of the main array are cut into
                       // pieces of seg1 output planes, one work item each (planes with few bins: 256^3 has 33 k
                       // columns for 1024 wave slots), at the price of K + 3 window planes read again per piece
  // The Nyquist planes of the split layout ride in the same launch: bins [plane, plane + plane2) are the
  // bins of a second set of arrays with plane2 bins per plane (plane2 = 0: none).  A separate two-workgroup
  // launch on the side stream ran as long as the whole main launch (every work item walks all of dim0) and
  // slowed it by a third (0.30 against 0.22 ms at 512^3, K = 31).
  const cfloat* in2;
  cfloat* out2;
  const cfloat* taps2;
  long plane2;
  int seg2;  // the second set's columns are cut into pieces of seg2 output planes, one work item each: a column
             // walked by ONE work item takes as long as the whole launch, and this set has only plane2 of them
  // Packed Nyquist (RowsParams::nyq_packed): column 0 of the main array holds DC + i Nyquist of every row, both
  // real before the dim1 transform, so after it  P[k1] = X0[k1] + i XH[k1]  with X0, XH Hermitian in k1.  The
  // two need different taps (taps at column 0, taps2): workgroups beyond the main ones take one pair
  // (k1, -k1) of column 0 each and compute, with U = (T0 + TH) / 2, V = (T0 - TH) / 2 at k1,
  //     out[k1] = sum_j U_j a_j + V_j conj(b_j),   out[-k1] = conj( sum_j U_j conj(b_j) + V_j a_j )
  // (a = P[k1], b = P[-k1] along dim0).  The main work items leave column 0 alone.  No Nyquist plane, no
  // launches of its own for it, no second stream.
  int packed;
  int C, d1;        // bins per row, rows per plane (plane = d1 * C)
  const int* inv1;  // inv1[k1] = row that holds bin k1 along dim1 (position order, AxisPlan::inv)
};

#define MVN_D0_PF 4         // planes requested ahead
#define MVN_D0_MAX_TAPS 33  // largest instantiated K

// K the direct kernel is instantiated for: k itself if odd, else k + 1 (one zero tap)
inline int mvn_dim0_taps_template(int k) { return k | 1; }
inline bool mvn_dim0_direct_possible(int k, int d0) {
  const int K = mvn_dim0_taps_template(k);
  return k >= 1 && K <= MVN_D0_MAX_TAPS && d0 >= K + MVN_D0_PF;
}

// acc + a * w (complex): two packed fused multiply-adds on the device
MVN_HD cfloat mvn_cmac(cfloat acc, cfloat a, cfloat w) {
#if defined(MVN_PACKED)
  cfloat t, r;
  MVN_PK3(t, "v_pk_fma_f32", a, w, acc, "op_sel_hi:[0,1,1]");                                // (a.x w.x, a.x w.y) + acc
  MVN_PK3(r, "v_pk_fma_f32", a, w, t, "op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]");  // + (-a.y w.y, a.y w.x)
  return r;
#else
  return cmake(acc.x + a.x * w.x - a.y * w.y, acc.y + a.x * w.y + a.y * w.x);
#endif
}

// pairs (k1, -k1) of the packed DC column
inline int mvn_dim0_pairs(int d1) { return d1 / 2 + 1; }
// LDS of a DC-pair workgroup (two columns of d0 values + two sets of k taps) and the longest dim0 it allows:
// 64 KB of dynamic LDS, 4062 planes at 33 taps.  Longer volumes keep the separate Nyquist plane.
inline size_t mvn_dim0_dc_lds_bytes(int d0, int k) { return sizeof(cfloat) * (2 * (size_t)d0 + 2 * (size_t)k); }
inline bool mvn_dim0_packed_possible(int d0) { return mvn_dim0_dc_lds_bytes(d0, MVN_D0_MAX_TAPS) <= 64 * 1024; }

// One pair per workgroup, two phases around a workgroup barrier.  lds: 2 * d0 + 2 * k cfloats.
MVN_HD void mvn_dim0_dc_load(const Dim0DirectParams& P, int pair, cfloat* lds, int tid, int nthreads) {
  const int k1 = pair, k1m = (P.d1 - pair) % P.d1;
  const long ra = (long)P.inv1[k1] * P.C, rb = (long)P.inv1[k1m] * P.C;
  cfloat* a = lds;
  cfloat* b = lds + P.d0;
  cfloat* U = lds + 2 * (long)P.d0;
  cfloat* V = U + P.k;
  for (int z = tid; z < P.d0; z += nthreads) {
    a[z] = P.in[(long)z * P.plane + ra];
    b[z] = P.in[(long)z * P.plane + rb];
  }
  for (int j = tid; j < P.k; j += nthreads) {
    int p = j - P.h;
    p = p < 0 ? p + P.kd : p;
    const cfloat t0 = P.taps[(long)p * P.plane + ra];
    const cfloat th = P.taps2[(long)p * P.d1 + P.inv1[k1]];
    U[j] = cmake(0.5f * (t0.x + th.x), 0.5f * (t0.y + th.y));
    V[j] = cmake(0.5f * (t0.x - th.x), 0.5f * (t0.y - th.y));
  }
}
MVN_HD void mvn_dim0_dc_compute(const Dim0DirectParams& P, int pair, const cfloat* lds, int tid, int nthreads) {
  const int k1 = pair, k1m = (P.d1 - pair) % P.d1;
  const long ra = (long)P.inv1[k1] * P.C, rb = (long)P.inv1[k1m] * P.C;
  const cfloat* a = lds;
  const cfloat* b = lds + P.d0;
  const cfloat* U = lds + 2 * (long)P.d0;
  const cfloat* V = U + P.k;
  for (int z = tid; z < P.d0; z += nthreads) {
    cfloat s1 = cmake(0.f, 0.f), s2 = cmake(0.f, 0.f);
    int zi = z + P.h;
    zi = zi >= P.d0 ? zi - P.d0 : zi;
    for (int j = 0; j < P.k; ++j) {
      const cfloat av = a[zi], bc = cconj(b[zi]);
      s1 = mvn_cmac(mvn_cmac(s1, av, U[j]), bc, V[j]);
      s2 = mvn_cmac(mvn_cmac(s2, bc, U[j]), av, V[j]);
      zi = zi == 0 ? P.d0 - 1 : zi - 1;
    }
    P.out[(long)z * P.plane + ra] = s1;
    if (rb != ra) P.out[(long)z * P.plane + rb] = cconj(s2);
  }
}

// work item `g` of a launch: the arrays it belongs to, the bin it owns there and the output planes
// [z0, z0 + nout) it produces (cyclically)
MVN_HD bool mvn_dim0_select(const Dim0DirectParams& P, long g, int wg_start, Dim0DirectParams& Q, long& b, int& z0,
                            int& nout) {
  Q = P;
  const int seg1 = P.seg1 > 0 ? P.seg1 : P.d0;
  const long main_items = P.plane * ((P.d0 + seg1 - 1) / seg1);
  if (g < main_items) {
    const long piece = g / P.plane;
    b = g - piece * P.plane;
    if (P.packed && b % P.C == 0) return false;  // the packed DC column belongs to the pair workgroups
    z0 = P.seg1 > 0 ? (int)(piece * seg1) : wg_start;
    nout = P.seg1 > 0 ? (P.d0 - z0 < seg1 ? P.d0 - z0 : seg1) : P.d0;
    return true;
  }
  if (P.plane2 <= 0 || P.packed) return false;
  const long g2 = g - main_items;
  const int seg = P.seg2 > 0 ? P.seg2 : P.d0;
  const long piece = g2 / P.plane2;
  b = g2 - piece * P.plane2;
  z0 = (int)(piece * seg);
  if (z0 >= P.d0) return false;
  nout = P.d0 - z0 < seg ? P.d0 - z0 : seg;
  Q.in = P.in2;
  Q.out = P.out2;
  Q.taps = P.taps2;
  Q.plane = P.plane2;
  return true;
}
// work items of a launch
inline long mvn_dim0_items(const Dim0DirectParams& P) {
  const int seg1 = P.seg1 > 0 ? P.seg1 : P.d0;
  const long main_items = P.plane * ((P.d0 + seg1 - 1) / seg1);
  if (P.packed) return main_items;  // + mvn_dim0_pairs(P.d1) workgroups, see the launchers
  const int seg = P.seg2 > 0 ? P.seg2 : P.d0;
  return main_items + (P.plane2 > 0 ? P.plane2 * ((P.d0 + seg - 1) / seg) : 0);
}
// Piece length for the main columns of a (d0, plane) volume: as many pieces as bring a launch to `want` work
// items, but none shorter than 2 K + 8 planes (every piece reads its K + 3 window planes again); 0 = whole
// columns.  mvn_dim0_items_for() = the work items that gives.
inline int mvn_dim0_piece_len(int k, int d0, long plane, long want) {
  if (plane >= want || plane < 1) return 0;
  const int min_len = 2 * mvn_dim0_taps_template(k) + 8;
  const long need = (want + plane - 1) / plane;
  const long maxp = d0 / min_len;
  const long pieces = need < maxp ? need : maxp;
  if (pieces < 2) return 0;
  return (int)((d0 + pieces - 1) / pieces);
}
inline long mvn_dim0_items_for(int k, int d0, long plane, long want) {
  const int len = mvn_dim0_piece_len(k, d0, plane, want);
  return len > 0 ? plane * ((d0 + len - 1) / len) : plane;
}

template <int K, int PF>
struct Dim0Window {
  cfloat w[K + PF];
  cfloat tap[K];
  cfloat bad;  // sum of (x - x) over every input of the column: 0, or NaN once a non-finite value came by
};

// Non-finite values.  The RL loop lets NaN / Inf flow (0 / 0 and x / 0 quotients, inc/cpu_kernels.h:19-26)
// and an FFT-based convolution turns ONE such voxel into a volume of NaN, which the update then clamps to
// minValue everywhere (inc/cpu_kernels.h:76-80) - the reference's, and the oracle's, behaviour.  The last-axis
// and dim1 transforms have already spread the voxel over its whole plane when this pass runs, so every column
// meets a non-finite input: a column that did rewrites ALL its outputs as NaN, and the volume comes out as
// the FFT leg would have left it.
MVN_HD cfloat mvn_dim0_track(cfloat bad, cfloat v) { return cadd(bad, csub(v, v)); }

template <int K, int PF, int U>
MVN_HD void mvn_dim0_step(Dim0Window<K, PF>& r, const cfloat* __restrict__ in, cfloat* __restrict__ out, long plane,
                          int d0, int nout, int nn, int& z, int& znew) {
  constexpr int KW = K + PF;
  if (nn + U >= nout) return;  // nout outputs in all, the walk is cyclic
  cfloat acc = cmake(0.f, 0.f);
#pragma unroll
  for (int j = 0; j < K; ++j) acc = mvn_cmac(acc, r.w[(j + PF - U + KW) % KW], r.tap[j]);
  out[(long)z * plane] = acc;
  // x_0 = in[z + h] runs over every plane of the column once: tracked HERE, where it has long arrived
  // (at its load the check would stall on the request that was just issued)
  r.bad = mvn_dim0_track(r.bad, r.w[(PF - U + KW) % KW]);
  // the oldest value x_{K-1} leaves; in[z + h + PF + 1], x_{-PF} of the next step, takes its slot
  r.w[(K - 1 + PF - U + KW) % KW] = in[(long)znew * plane];
  znew = znew + 1 == d0 ? 0 : znew + 1;
  z = z + 1 == d0 ? 0 : z + 1;
  if constexpr (U + 1 < KW) mvn_dim0_step<K, PF, U + 1>(r, in, out, plane, d0, nout, nn, z, znew);
}

// outputs [z0, z0 + nout) (cyclically) of bin b
template <int K, int PF>
MVN_HD void mvn_dim0_direct_column(const Dim0DirectParams& P, long b, int z0, int nout) {
  constexpr int KW = K + PF;
  Dim0Window<K, PF> r;
  r.bad = cmake(0.f, 0.f);
#pragma unroll
  for (int j = 0; j < K; ++j) {
    int p = j - P.h;
    p = p < 0 ? p + P.kd : p;
    r.tap[j] = j < P.k ? P.taps[(long)p * P.plane + b] : cmake(0.f, 0.f);
  }
  // at u = 0 (z = z0): slot s holds x_{s - PF} = in[z0 + h + PF - s]
#pragma unroll
  for (int s = 0; s < KW; ++s) {
    int z = z0 + P.h + PF - s;
    z = z < 0 ? z + P.d0 : (z >= P.d0 ? z - P.d0 : z);
    r.w[s] = P.in[(long)z * P.plane + b];
  }
  int znew = z0 + P.h + PF + 1;
  if (znew >= P.d0) znew -= P.d0;
  int zout = z0;
  // (in and out never alias: the leg is out of place)
  const cfloat* __restrict__ in = P.in + b;
  cfloat* __restrict__ out = P.out + b;
  for (int nn = 0; nn < nout; nn += KW) mvn_dim0_step<K, PF, 0>(r, in, out, P.plane, P.d0, nout, nn, zout, znew);
  if (r.bad.x != 0.f || r.bad.y != 0.f) {  // (NaN != 0): see mvn_dim0_track
    const float q = r.bad.x != 0.f ? r.bad.x : r.bad.y;
    // (whole columns - the main array's - thereby turn NaN entirely, which is what makes the volume come out
    // all NaN after the passes that follow; a PIECE of a column rewrites its own outputs only)
    int z = z0;
    for (int n = 0; n < nout; ++n) {
      P.out[(long)z * P.plane + b] = cmake(q, q);
      z = z + 1 == P.d0 ? 0 : z + 1;
    }
  }
}

// the instantiated tap counts, for the launch switches of both backends
#define MVN_D0_TAP_COUNTS(X) X(1) X(3) X(5) X(7) X(9) X(11) X(13) X(15) X(17) X(19) X(21) X(23) X(25) X(27) X(29) X(31) X(33)
