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

#define MVN_D0_MAX_PEERS 7  // further slabs of one volume a launch reports non-finite inputs to

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
  // Non-finite inputs (see mvn_dim0_track): a work item that met one stores `poison_epoch` into *poison, and the
  // last-axis pass that ends this convolution (EpilogueParams::poison) then emits NaN for EVERY voxel - what an FFT
  // along dim0 would have made of it.  The epoch is the engine's count of direct legs, so the word is never cleared.
  // A device that runs one slab of a volume reports to every slab's word: poison_peers = a device table of
  // n_peers (<= MVN_D0_MAX_PEERS) further words - the other devices', written through peer access.
  unsigned* poison;
  unsigned poison_epoch;
  int n_peers;
  unsigned* const* poison_peers;
  // zcount > 0: only the output planes [zbeg, zbeg + zcount) are produced (a slab whose first and last planes are
  // halo planes computes its own planes only - and those that do not depend on the halos before they arrive);
  // the inputs are read cyclically over all d0 planes as ever.  0: all d0 planes.
  int zbeg, zcount;
};

#ifndef MVN_D0_PF
#define MVN_D0_PF 4         // planes requested ahead
#endif
#ifndef MVN_D0_CHAINS
#define MVN_D0_CHAINS 2     // independent pairs of multiply-add chains per output (1 or 2)
#endif
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

// The K complex multiply-adds of an output as TWO independent chains of packed fused multiply-adds,
//     s1 += (a.x w.x, a.x w.y),   s2 += (a.y w.x, a.y w.y),   out = (s1.x - s2.y, s1.y + s2.x),
// each with one half of the input broadcast as its first operand.  Same instruction count as the single chain of
// round 3 (acc += a.x w, then acc += i a.y w), but consecutive instructions never depend on each other: behind an
// inline instruction the compiler places a wait state in front of every instruction that reads its result, 2 K of
// them per output in the single chain.  (Left to the compiler's own packed multiply-adds the unrolled walk is
// re-scheduled across steps and needs 255 registers, or spills when bounded: the inline forms also pin the order.)
MVN_HD void mvn_cmac2(cfloat& s1, cfloat& s2, cfloat a, cfloat w) {
#if defined(MVN_PACKED)
  MVN_PK3(s1, "v_pk_fma_f32", a, w, s1, "op_sel_hi:[0,1,1]");
  MVN_PK3(s2, "v_pk_fma_f32", a, w, s2, "op_sel:[1,0,0] op_sel_hi:[1,1,1]");
#else
  s1 = cmake(s1.x + a.x * w.x, s1.y + a.x * w.y);
  s2 = cmake(s2.x + a.y * w.x, s2.y + a.y * w.y);
#endif
}

// the first tap of a pair of chains: products instead of multiply-adds onto a zeroed register pair
MVN_HD void mvn_cmul2(cfloat& s1, cfloat& s2, cfloat a, cfloat w) {
#if defined(MVN_PACKED)
  MVN_PK2(s1, "v_pk_mul_f32", a, w, "op_sel_hi:[0,1]");
  MVN_PK2(s2, "v_pk_mul_f32", a, w, "op_sel:[1,0] op_sel_hi:[1,1]");
#else
  s1 = cmake(a.x * w.x, a.x * w.y);
  s2 = cmake(a.y * w.x, a.y * w.y);
#endif
}

// Non-finite values.  The RL loop lets NaN / Inf flow (0 / 0 and x / 0 quotients, inc/cpu_kernels.h:19-26)
// and an FFT-based convolution turns ONE such voxel into a volume of NaN (inc/cpu_convolve.h:256-268), which
// the update then clamps to minValue everywhere (inc/cpu_kernels.h:40-47,76-83) - the reference's, and the
// oracle's, behaviour.  A direct convolution along dim0 would only spoil the K planes around the voxel.  The
// last-axis and dim1 transforms have already spread the voxel over its whole plane when this pass runs, so
// every column - and every piece of a column whose tracked planes include that plane - meets a non-finite
// input: it REPORTS it (mvn_dim0_report), and the last-axis pass that ends the convolution reads the report
// once per workgroup and turns every voxel into NaN (mvn_arm_poison, mvn_pass_bodies.hpp).  Whole columns,
// pieces, the Nyquist pieces and the packed DC-pair workgroups therefore all end in the same volume, the one
// the FFT leg would have left.
// bad += 0 * v, one packed fused multiply-add: 0 * finite = (+-)0, 0 * Inf = 0 * NaN = NaN; bad != 0 afterwards
// exactly when a non-finite value came by
MVN_HD cfloat mvn_dim0_track(cfloat bad, cfloat v) {
#if defined(MVN_PACKED)
  cfloat r;
  asm("v_pk_fma_f32 %0, %1, 0, %2" : "=v"(r) : "v"(v), "v"(bad));
  return r;
#else
  return cmake(bad.x + 0.f * v.x, bad.y + 0.f * v.y);
#endif
}
MVN_HD void mvn_dim0_report(const Dim0DirectParams& P) {
  if (P.poison) *P.poison = P.poison_epoch;  // (same value from every work item that gets here: a plain store)
  for (int i = 0; i < P.n_peers; ++i) *P.poison_peers[i] = P.poison_epoch;
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
  cfloat bad = cmake(0.f, 0.f);
  // tracked like the other work items' planes: in[z + h] for every output plane z of the launch (a ranged launch
  // may run before the halo planes of the input have arrived: what they still hold must not be looked at)
  const int zlo = P.zcount > 0 ? P.zbeg : 0, nplanes = P.zcount > 0 ? P.zcount : P.d0;
  for (int z = tid; z < P.d0; z += nthreads) {
    const cfloat va = P.in[(long)z * P.plane + ra], vb = P.in[(long)z * P.plane + rb];
    a[z] = va;
    b[z] = vb;
    int rel = z - zlo - P.h;
    rel = rel < 0 ? rel + P.d0 : rel;
    if (rel < nplanes) bad = mvn_dim0_track(mvn_dim0_track(bad, va), vb);
  }
  if (bad.x != 0.f || bad.y != 0.f) mvn_dim0_report(P);
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
  const int zlo = P.zcount > 0 ? P.zbeg : 0, zhi = P.zcount > 0 ? P.zbeg + P.zcount : P.d0;
  for (int z = zlo + tid; z < zhi; z += nthreads) {
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

// Workgroups of a launch (MVN_D0_WG work items each).  A workgroup takes MVN_D0_WG neighbouring bins of ONE piece of
// ONE set of arrays, so that the planes it walks - and with them every address computation of the walk - are the
// same for all its lanes: they live in scalar registers and cost no vector instructions (round 4: with per-lane
// planes the 64-bit multiplies of the addresses took more vector cycles than the K complex multiply-adds).
// Order: the main array piece by piece (a plane's bins in address order), then the second set likewise.
#define MVN_D0_WG 256
struct Dim0Blocks {
  long nb1, nb2;       // workgroups per piece (bins of a plane / MVN_D0_WG, rounded up)
  int seg1, seg2;      // output planes per piece
  long main_blocks;    // nb1 * pieces of the main array
  long blocks;         // all of them
};
MVN_HD Dim0Blocks mvn_dim0_blocks(const Dim0DirectParams& P) {
  Dim0Blocks B;
  const int planes = P.zcount > 0 ? P.zcount : P.d0;  // output planes of the launch
  B.seg1 = P.seg1 > 0 ? P.seg1 : planes;
  B.seg2 = P.seg2 > 0 ? P.seg2 : planes;
  B.nb1 = (P.plane + MVN_D0_WG - 1) / MVN_D0_WG;
  B.nb2 = (P.packed || P.plane2 <= 0) ? 0 : (P.plane2 + MVN_D0_WG - 1) / MVN_D0_WG;
  B.main_blocks = B.nb1 * ((planes + B.seg1 - 1) / B.seg1);
  B.blocks = B.main_blocks + B.nb2 * ((planes + B.seg2 - 1) / B.seg2);  // (+ mvn_dim0_pairs(d1) workgroups when packed)
  return B;
}
// workgroup `block`, lane `tid`: the arrays (Q), the bin b the lane owns there and the output planes
// [z0, z0 + nout) (cyclically) of the workgroup; false: the lane has nothing to do
MVN_HD bool mvn_dim0_job(const Dim0DirectParams& P, long block, int tid, Dim0DirectParams& Q, long& b, int& z0,
                         int& nout) {
  const Dim0Blocks B = mvn_dim0_blocks(P);
  Q = P;
  const bool ranged = P.zcount > 0;
  const int zlo = ranged ? P.zbeg : 0, zhi = ranged ? P.zbeg + P.zcount : P.d0;
  if (block < B.main_blocks) {
    const long piece = block / B.nb1;
    b = (block - piece * B.nb1) * MVN_D0_WG + tid;
    // whole columns: workgroups start their cyclic walk at different planes (all of them on one plane at a time
    // keep the whole chip on 1 MB in and 1 MB out)
    if (P.seg1 > 0 || ranged) {
      z0 = zlo + (int)(piece * B.seg1);
      nout = zhi - z0 < B.seg1 ? zhi - z0 : B.seg1;
    } else {
      z0 = P.stagger > 0 ? (int)((block * P.stagger) % P.d0) : 0;
      nout = P.d0;
    }
    if (b >= P.plane) return false;
    return !(P.packed && b % P.C == 0);  // the packed DC column belongs to the pair workgroups
  }
  if (B.nb2 == 0) return false;
  const long block2 = block - B.main_blocks;
  const long piece = block2 / B.nb2;
  b = (block2 - piece * B.nb2) * MVN_D0_WG + tid;
  z0 = zlo + (int)(piece * B.seg2);
  nout = zhi - z0 < B.seg2 ? zhi - z0 : B.seg2;
  Q.in = P.in2;
  Q.out = P.out2;
  Q.taps = P.taps2;
  Q.plane = P.plane2;
  return z0 < zhi && b < P.plane2;
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
  cfloat bad;  // sum of (x - x) over the tracked inputs: 0, or NaN once a non-finite value came by
};


// The walk of a workgroup: the plane being written (z) and the plane being requested (znew) with their addresses,
// all the same for every lane and advanced by additions (scalar registers, no multiplies inside the walk); what
// varies per lane is the 32-bit bin index b alone.
struct Dim0Walk {
  const cfloat* in0;   // plane 0 of the input
  cfloat* out0;        // plane 0 of the output
  const cfloat* pin;   // plane znew of the input
  cfloat* pout;        // plane z of the output
  long plane;
  unsigned bytes;      // of a plane
  int z, znew, d0;
};
// (the wraps happen twice per walk: kept as rarely taken BRANCHES - the empty inline statement stops the compiler
// from turning them into four selects per pointer and step, on the critical path of a wave that has only one or
// two others to hide behind)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MVN_HOST_EMU)
#define MVN_D0_KEEP_BRANCH() asm volatile("")
#else
#define MVN_D0_KEEP_BRANCH() (void)0
#endif
MVN_HD void mvn_dim0_advance(Dim0Walk& w) {
  ++w.znew;
  w.pin += w.plane;
  if (__builtin_expect(w.znew == w.d0, 0)) {
    MVN_D0_KEEP_BRANCH();
    w.znew = 0;
    w.pin = w.in0;
  }
  ++w.z;
  w.pout += w.plane;
  if (__builtin_expect(w.z == w.d0, 0)) {
    MVN_D0_KEEP_BRANCH();
    w.z = 0;
    w.pout = w.out0;
  }
}

// Element at byte offset `boff` (32 bits, the lane's part) of the plane at `p` (the workgroup's part).  On the
// device as a BUFFER access - the plane's address in a scalar resource descriptor, the lane's offset in one vector
// register - so that walking from plane to plane is two scalar additions and no vector instruction at all (a flat
// 64-bit address per lane would be rebuilt with vector adds and selects at every step).  `bytes` = extent of a
// plane: lanes beyond it read 0 / write nothing.
// values that are the same in every lane of a workgroup, said so to the compiler (they then live in scalar registers)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MVN_HOST_EMU)
MVN_HD int mvn_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
MVN_HD long mvn_uniform(long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long)v & 0xffffffffu));
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long)v >> 32));
  return (long)(((unsigned long)hi << 32) | lo);
}
#else
MVN_HD int mvn_uniform(int v) { return v; }
MVN_HD long mvn_uniform(long v) { return v; }
#endif
template <typename T>
MVN_HD T* mvn_uniform(T* p) { return reinterpret_cast<T*>(mvn_uniform(reinterpret_cast<long>(p))); }

#if defined(__HIP_DEVICE_COMPILE__) && !defined(MVN_HOST_EMU)
MVN_HD cfloat mvn_dim0_ld(const cfloat* p, unsigned boff, unsigned bytes) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000);
  return __builtin_amdgcn_raw_buffer_load_b64(r, (int)boff, 0, 0);
}
MVN_HD void mvn_dim0_st(cfloat* p, unsigned boff, unsigned bytes, cfloat v) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b64(v, r, (int)boff, 0, 0);
}
#else
MVN_HD cfloat mvn_dim0_ld(const cfloat* p, unsigned boff, unsigned) {
  return *reinterpret_cast<const cfloat*>(reinterpret_cast<const char*>(p) + boff);
}
MVN_HD void mvn_dim0_st(cfloat* p, unsigned boff, unsigned, cfloat v) {
  *reinterpret_cast<cfloat*>(reinterpret_cast<char*>(p) + boff) = v;
}
#endif

template <int K, int PF, int U>
MVN_HD void mvn_dim0_step(Dim0Window<K, PF>& r, Dim0Walk& w, unsigned b, int nout, int nn) {
  constexpr int KW = K + PF;
  if (nn + U >= nout) return;  // nout outputs in all, the walk is cyclic
  cfloat s1, s2;
#if defined(MVN_EXPERIMENTS) && defined(MVN_D0_EXP_TAPS)
  s1 = s2 = cmake(0.f, 0.f);
  // timing experiment (variant builds only, WRONG results): the walk with only the first few multiply-adds
#pragma unroll
  for (int j = 0; j < (K < MVN_D0_EXP_TAPS ? K : MVN_D0_EXP_TAPS); ++j) mvn_cmac2(s1, s2, r.w[(j + PF - U + KW) % KW], r.tap[j]);
#elif MVN_D0_CHAINS == 2
  // even and odd taps in chains of their own: four instructions between a multiply-add and the next one of its
  // chain - the compiler asks for two between inline instructions and fills what is missing with wait states
  mvn_cmul2(s1, s2, r.w[(PF - U + KW) % KW], r.tap[0]);
  if constexpr (K > 1) {
    cfloat t1, t2;
    mvn_cmul2(t1, t2, r.w[(1 + PF - U + KW) % KW], r.tap[1]);
#pragma unroll
    for (int j = 2; j + 1 < K; j += 2) {
      mvn_cmac2(s1, s2, r.w[(j + PF - U + KW) % KW], r.tap[j]);
      mvn_cmac2(t1, t2, r.w[(j + 1 + PF - U + KW) % KW], r.tap[j + 1]);
    }
    if (K % 2) mvn_cmac2(s1, s2, r.w[(K - 1 + PF - U + KW) % KW], r.tap[K - 1]);
    s1 = cadd(s1, t1);
    s2 = cadd(s2, t2);
  }
#else
  mvn_cmul2(s1, s2, r.w[(PF - U + KW) % KW], r.tap[0]);
#pragma unroll
  for (int j = 1; j < K; ++j) mvn_cmac2(s1, s2, r.w[(j + PF - U + KW) % KW], r.tap[j]);
#endif
  mvn_dim0_st(w.pout, b, w.bytes, cadd_i<+1>(s1, s2));  // (s1.x - s2.y, s1.y + s2.x)
  // x_0 = in[z + h] runs over every plane of the column once: tracked HERE, where it has long arrived
  // (at its load the check would stall on the request that was just issued)
  r.bad = mvn_dim0_track(r.bad, r.w[(PF - U + KW) % KW]);
  // the oldest value x_{K-1} leaves; in[z + h + PF + 1], x_{-PF} of the next step, takes its slot
  r.w[(K - 1 + PF - U + KW) % KW] = mvn_dim0_ld(w.pin, b, w.bytes);
  mvn_dim0_advance(w);
  if constexpr (U + 1 < KW) mvn_dim0_step<K, PF, U + 1>(r, w, b, nout, nn);
}

// outputs [z0, z0 + nout) (cyclically) of bin b_ (< 2^28: checked at the launch)
template <int K, int PF>
MVN_HD void mvn_dim0_direct_column(const Dim0DirectParams& P0, long b_, int z0_, int nout_) {
  // everything but b_ is the workgroup's (mvn_dim0_job)
  Dim0DirectParams P = P0;
  P.in = mvn_uniform(P0.in);
  P.out = mvn_uniform(P0.out);
  P.taps = mvn_uniform(P0.taps);
  P.plane = mvn_uniform(P0.plane);
  const int z0 = mvn_uniform(z0_), nout = mvn_uniform(nout_);
  const unsigned b = (unsigned)b_ * (unsigned)sizeof(cfloat);  // byte offset inside a plane
  const unsigned bytes = (unsigned)P.plane * (unsigned)sizeof(cfloat);
  constexpr int KW = K + PF;
  Dim0Window<K, PF> r;
  r.bad = cmake(0.f, 0.f);
#pragma unroll
  for (int j = 0; j < K; ++j) {
    int p = j - P.h;
    p = p < 0 ? p + P.kd : p;
    r.tap[j] = j < P.k ? mvn_dim0_ld(P.taps + (long)p * P.plane, b, bytes) : cmake(0.f, 0.f);
  }
  // at u = 0 (z = z0): slot s holds x_{s - PF} = in[z0 + h + PF - s]
#pragma unroll
  for (int s = 0; s < KW; ++s) {
    int z = z0 + P.h + PF - s;
    z = z < 0 ? z + P.d0 : (z >= P.d0 ? z - P.d0 : z);
    r.w[s] = mvn_dim0_ld(P.in + (long)z * P.plane, b, bytes);
  }
  Dim0Walk w;
  w.in0 = P.in;
  w.out0 = P.out;  // (in and out never alias: the leg is out of place)
  w.plane = P.plane;
  w.bytes = bytes;
  w.d0 = P.d0;
  w.znew = z0 + P.h + PF + 1;
  if (w.znew >= P.d0) w.znew -= P.d0;
  w.z = z0;
  w.pin = P.in + (long)w.znew * P.plane;
  w.pout = P.out + (long)w.z * P.plane;
  for (int nn = 0; nn < nout; nn += KW) mvn_dim0_step<K, PF, 0>(r, w, b, nout, nn);
  // (NaN != 0): the tracked planes of this work item - in[z + h] for each of its outputs z, so the pieces of a
  // column between them see every plane exactly once - held a non-finite value
  if (r.bad.x != 0.f || r.bad.y != 0.f) mvn_dim0_report(P);
}

// the instantiated tap counts, for the launch switches of both backends
#define MVN_D0_TAP_COUNTS(X) X(1) X(3) X(5) X(7) X(9) X(11) X(13) X(15) X(17) X(19) X(21) X(23) X(25) X(27) X(29) X(31) X(33)
