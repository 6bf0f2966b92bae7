// mvn_fixed.hpp -- compile-time specialised pass bodies for the line lengths 2^a 3^b 5^c 7^d listed
// in mvn_fixed_geom.hpp (powers of two 64..1024 / 2048 and the common mixed-radix sizes).
//
// The generic bodies of mvn_pass_bodies.hpp take the radix schedule at run time; PMC counters on
// MI355X showed them VALU-bound (index arithmetic, multiply-high divisions, per-element address
// math) with 8-way LDS bank conflicts on the permuted rows of the last-axis passes.  Here every
// length, radix, stride and tile shape is a template constant, so
//   - LDS accesses of a butterfly use immediate offsets from one base address,
//   - digit reversal is a few shifts/masks instead of a table load,
//   - twiddles come from a stage-ordered LDS table (one 16-byte-aligned row per butterfly),
//   - global memory is touched in 16-byte units (two complex bins / four reals per lane),
//   - the transposing last-axis tiles insert one spare LDS row every 32 rows, which spreads the
//     digit-reversed row accesses over the banks.
// A body is a sequence of PHASES separated by workgroup barriers.  On the device a phase is
// straight-line code for `threadIdx.x`; the test-only host emulation runs each phase for every
// thread id in turn with per-thread "register" state kept in an array, i.e. with the real
// thread-to-data mapping.
#pragma once

#include <type_traits>

#include "mvn_pass_bodies.hpp"

#if defined(__HIPCC__) && !defined(MVN_HOST_EMU)
typedef float4 qfloat;
#else
struct alignas(16) qfloat {
  float x, y, z, w;
};
#endif

MVN_HD qfloat qmake(float x, float y, float z, float w) {
  qfloat q;
  q.x = x;
  q.y = y;
  q.z = z;
  q.w = w;
  return q;
}

// ---------------------------------------------------------------------------------------------
// phase plumbing
// ---------------------------------------------------------------------------------------------
#if defined(__HIPCC__) && !defined(MVN_HOST_EMU)
template <typename Regs, int NT>
struct FxCtx {
  int tid;
  Regs regs;
};
#define MVN_PHASE(ctx, ...)      \
  {                              \
    const int tid = (ctx).tid;   \
    auto& r = (ctx).regs;        \
    (void)r;                     \
    __VA_ARGS__;                 \
  }                              \
  __syncthreads();
#define MVN_PHASE_NOSYNC(ctx, ...) \
  {                                \
    const int tid = (ctx).tid;     \
    auto& r = (ctx).regs;          \
    (void)r;                       \
    __VA_ARGS__;                   \
  }
// At the top of a tile loop: makes the thread id opaque to the optimiser, so that the per-thread
// address arithmetic of every phase is redone per tile (a handful of integer instructions)
// instead of being hoisted out of the loop and parked in registers for its whole length.
#define MVN_TILE_LOOP_TOP(ctx) asm volatile("" : "+v"((ctx).tid))
// Between independent work items of one thread: stops the scheduler from interleaving them all for
// ILP, which multiplies the temporaries of a butterfly by the item count (and spills).
#define MVN_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// Keeps a walking element offset opaque so that its successors are computed one add at a time,
// right before use, instead of all up front (40 64-bit addresses are 80 registers).  Applied to an
// integer, not to the pointer: an opaque pointer loses its address space (flat instead of global
// loads).
#define MVN_JIT_ADDRESS(p) asm volatile("" : "+v"(p))
#else
template <typename Regs, int NT>
struct FxCtx {
  Regs regs[NT];
};
#define MVN_PHASE_NOSYNC(ctx, ...) MVN_PHASE(ctx, __VA_ARGS__)
#define MVN_JIT_ADDRESS(p) (void)0
#define MVN_TILE_LOOP_TOP(ctx) (void)0
#define MVN_SCHED_FENCE() (void)0
#define MVN_PHASE(ctx, ...)                  \
  for (int tid = 0; tid < NT_; ++tid) {      \
    auto& r = (ctx).regs[tid];               \
    (void)r;                                 \
    __VA_ARGS__;                             \
  }
#endif

// ---------------------------------------------------------------------------------------------
// compile-time plan of a length 2^a 3^b 5^c 7^d: same radix order as AxisPlanHost::factorize
// ---------------------------------------------------------------------------------------------
constexpr bool fx_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

constexpr bool fx_smooth(int n) {  // only the factors the register butterflies cover
  if (n < 1) return false;
  for (int p : {2, 3, 5, 7})
    while (n % p == 0) n /= p;
  return n == 1;
}

constexpr int fx_nstages(int n) {
  int c = 0;
  for (int r : {8, 12, 4, 6, 10, 2, 15, 9, 3, 5, 7})
    while (n % r == 0) { n /= r; ++c; }
  return c;
}

constexpr int fx_radix(int n, int s) {
  int c = 0;
  for (int r : {8, 12, 4, 6, 10, 2, 15, 9, 3, 5, 7})
    while (n % r == 0) { if (c == s) return r; n /= r; ++c; }
  return 1;
}

// twiddle rows are padded to an even number of entries so that they stay 16-byte aligned
constexpr int fx_rs(int r) { return r + (r & 1); }

constexpr int fx_M(int n, int s) {  // butterfly input stride of stage s
  int m = n;
  for (int t = 0; t <= s; ++t) m /= fx_radix(n, t);
  return m;
}

constexpr int fx_W(int n, int s) {  // weight of digit s in the natural index
  int w = 1;
  for (int t = 0; t < s; ++t) w *= fx_radix(n, t);
  return w;
}

// stage-ordered twiddle table: for every stage with M > 1 a block of M rows of fx_rs(R) entries,
// row j2 = { exp(-2 pi i j2 k / (R M)) : k = 0..R-1 } (+ one pad entry for odd R)
constexpr int fx_twoff(int n, int s) {
  int off = 0;
  for (int t = 0; t < s; ++t)
    if (fx_M(n, t) > 1) off += fx_M(n, t) * fx_rs(fx_radix(n, t));
  return off;
}
constexpr int fx_twsize(int n) { return fx_twoff(n, fx_nstages(n)); }

template <int N>
MVN_HD int fx_rev(int p) {  // position -> natural index after DIF
  int k = 0;
#pragma unroll
  for (int s = 0; s < fx_nstages(N); ++s) k += ((p / fx_M(N, s)) % fx_radix(N, s)) * fx_W(N, s);
  return k;
}

template <int N>
MVN_HD int fx_inv(int k) {  // natural index -> position
  int p = 0;
#pragma unroll
  for (int s = 0; s < fx_nstages(N); ++s) p += ((k / fx_W(N, s)) % fx_radix(N, s)) * fx_M(N, s);
  return p;
}

// LDS row of logical row p; PAD inserts one spare row after every 32
template <bool PAD>
MVN_HD int fx_row(int p) {
  return PAD ? p + (p >> 5) : p;
}
constexpr int fx_rows_alloc(int n, bool pad) { return pad ? n + (n >> 5) : n; }
// row distance between inputs 0 and j of one butterfly (compile-time)
template <bool PAD, int R, int M>
constexpr int fx_rowoff(int j) {
  return j * M + ((PAD && R * M >= 32) ? ((j * M) >> 5) : 0);
}

// ---------------------------------------------------------------------------------------------
// one radix stage, fully unrolled for NT threads
// ---------------------------------------------------------------------------------------------
// twiddle row j2 of stage S out of a stage-ordered table: rows of fx_rs(R) entries, or (TWT: the LDS copies of
// the walking last-axis kernels, fx_rows_tables) 16-byte pairs {k, k + 1} stored [k / 2][j2], so that lanes
// with neighbouring j2 read neighbouring 16-byte slots.  Row-wise the lanes of a ds_read_b128 group sit 64
// bytes apart on 4 slots: a 4-way bank conflict on every twiddle read of these kernels, whose lanes run along
// j2 (52 - 61 % of the LDS cycles of the fused pass at H = 960 / 288 were conflict cycles and the LDS was busy
// 81 % of the pass; profiles/r03_rows_lds.md).  The strided kernels' lanes run along the columns of one row
// and read a twiddle row as a broadcast: they keep the row-wise table.
template <int N, int S, bool TWT>
MVN_HD void fx_tw_fetch(const cfloat* tws, int j2, cfloat* tw) {
  constexpr int R = fx_radix(N, S), M = fx_M(N, S), Q = fx_rs(R) / 2;
  const qfloat* t4 = reinterpret_cast<const qfloat*>(tws + fx_twoff(N, S));
#pragma unroll
  for (int k = 0; k < Q; ++k) {
    const qfloat t = TWT ? t4[k * M + j2] : t4[j2 * Q + k];
    tw[2 * k] = cmake(t.x, t.y);
    tw[2 * k + 1] = cmake(t.z, t.w);
  }
}

template <int N, int T, int TP, bool PAD, int NT, int S, int SIGN, bool DIF, bool TWT = false>
MVN_HD void fx_stage(cfloat* buf, const cfloat* tws, int tid) {
  constexpr int R = fx_radix(N, S), M = fx_M(N, S);
  constexpr int nwork = (N / R) * T;
  constexpr int iters = (nwork + NT - 1) / NT;
#pragma unroll
  for (int it = 0; it < iters; ++it) {
    const int w = tid + it * NT;
    if (nwork % NT != 0 && w >= nwork) break;
    // Walking last-axis kernels with 2 or 4 tile rows (TWT, TP = 3 or 5): lanes run along the butterflies of
    // one tile row - their words sit TP apart, every bank pair once per lane group - instead of along the tile
    // rows of one butterfly (2-way conflicts between neighbouring butterflies).  8-row tiles keep the
    // row-fastest order: 8 neighbouring words per butterfly are already conflict-free and share a twiddle row.
    // (576^3: -2.3 % per view update, 1920-long rows: fused divide -3 %, 384^3 with 8 rows: +2 % and left alone;
    // profiles/r03_rows_lds.md)
    constexpr bool JF = TWT && T <= 4;
    const int b = JF ? w % (N / R) : w / T, c = JF ? w / (N / R) : w % T;
    const int blk = b / M, j2 = b % M;
    cfloat* p = buf + fx_row<PAD>(blk * R * M + j2) * TP + c;
    cfloat a[R];
#pragma unroll
    for (int j = 0; j < R; ++j) a[j] = p[fx_rowoff<PAD, R, M>(j) * TP];
    cfloat tw[fx_rs(R)];
    if (M > 1) fx_tw_fetch<N, S, TWT>(tws, j2, tw);
    if (!DIF && M > 1) {
#pragma unroll
      for (int k = 1; k < R; ++k) a[k] = cmul_dir<SIGN>(a[k], tw[k]);
    }
    dftR<R, SIGN>(a);
    if (DIF && M > 1) {
#pragma unroll
      for (int k = 1; k < R; ++k) a[k] = cmul_dir<SIGN>(a[k], tw[k]);
    }
#pragma unroll
    for (int j = 0; j < R; ++j) p[fx_rowoff<PAD, R, M>(j) * TP] = a[j];
  }
}

// stages LO..ns-1 of a transform as phases (DIF runs them upwards from LO, DIT downwards to LO)
template <int N, int T, int TP, bool PAD, int NT, int SIGN, bool DIF, int S, int LO, bool TWT, typename Ctx>
struct FxStages {
  static MVN_HD void run(cfloat* buf, const cfloat* tws, Ctx& ctx) {
    constexpr int NT_ = NT;
    (void)NT_;
    MVN_PHASE(ctx, (fx_stage<N, T, TP, PAD, NT, S, SIGN, DIF, TWT>(buf, tws, tid)));
    constexpr int next = DIF ? S + 1 : S - 1;
    if constexpr (next >= LO && next < fx_nstages(N))
      FxStages<N, T, TP, PAD, NT, SIGN, DIF, next, LO, TWT, Ctx>::run(buf, tws, ctx);
  }
};

template <int N, int T, int TP, bool PAD, int NT, int SIGN, int LO = 0, bool TWT = false, typename Ctx>
MVN_HD void fx_dif(cfloat* buf, const cfloat* tws, Ctx& ctx) {
  if constexpr (LO < fx_nstages(N))
    FxStages<N, T, TP, PAD, NT, SIGN, true, LO, LO, TWT, Ctx>::run(buf, tws, ctx);
}
template <int N, int T, int TP, bool PAD, int NT, int SIGN, int LO = 0, bool TWT = false, typename Ctx>
MVN_HD void fx_dit(cfloat* buf, const cfloat* tws, Ctx& ctx) {
  if constexpr (LO < fx_nstages(N))
    FxStages<N, T, TP, PAD, NT, SIGN, false, fx_nstages(N) - 1, LO, TWT, Ctx>::run(buf, tws, ctx);
}

template <int NT>
MVN_HD void fx_copy_table(cfloat* dst, const cfloat* src, int count, int tid) {
  for (int i = tid; i < count; i += NT) dst[i] = src[i];
}

// largest multiple of 64 that is <= min(cap, full) and divides a, b and c (0 if none)
constexpr int fx_pick_nt(int cap, int full, int a, int b, int c) {
  for (int nt = cap; nt >= 64; nt -= 64)
    if (nt <= full && a % nt == 0 && b % nt == 0 && c % nt == 0) return nt;
  return 0;
}

// ---------------------------------------------------------------------------------------------
// strided-axis pass, full tiles of T neighbouring bins (cstride == 1, ncols % T == 0).
//
// A thread owns one 16-byte chunk (two neighbouring bins) of a tile row, so every butterfly is
// done on two columns at once with 16-byte LDS accesses and one shared twiddle row.  The outer
// stages never touch the LDS:
//   stage 0 (radix 8, input stride M0 = N/8) sits right behind the global loads of a forward
//   transform / right before the global stores of an inverse one (thread <-> rows j2 + k M0),
//   the last stage (M = 1) sits right before the stores of a forward transform / behind the loads
//   of an inverse one (thread <-> rows b R + k); in the fused FWD*PSF*INV pass the last forward
//   stage, the multiplication by the PSF spectrum and the first inverse stage are one register
//   sequence.
// That leaves ns-1 LDS round trips per transform (2 for N = 512; the fused pass 4 instead of 8).
// ---------------------------------------------------------------------------------------------
#ifndef MVN_FX_ST_NT_TARGET
#define MVN_FX_ST_NT_TARGET 512
#endif
#ifndef MVN_FX_ST_MAX_WAVES
#define MVN_FX_ST_MAX_WAVES 4
#endif
// stage-0 butterflies per thread: the fewest that bring the workgroup to <= target threads
constexpr int fx_st_it0(int nt1, int target) {
  for (int it = 1; it <= 8; ++it)
    if (nt1 % it == 0 && (nt1 / it) % 64 == 0 && nt1 / it <= target) return it;
  return 1;
}

template <int N>
struct FxStridedCfg {
  // 128-byte row segments wherever the tile fits the LDS at all (two workgroups per CU up to
  // N = 576, one beyond; measured: 64-byte segments cost 15-30 % of the pass at N = 640..1024)
  static constexpr int T = N <= 1024 ? 16 : 8;
  static constexpr int TP = T;
  static constexpr int CH = T / 2;    // 16-byte chunks per tile row
  static constexpr int TPQ = TP / 2;  // row pitch in 16-byte units
  static constexpr int NS = fx_nstages(N);
  static constexpr int R0 = fx_radix(N, 0), M0 = fx_M(N, 0);
  static constexpr int RL = fx_radix(N, NS - 1);  // radix of the last stage (M = 1)
  // threads: NT1 = M0 * CH would give every thread one stage-0 butterfly (on two columns); IT0 of
  // them per thread keep the workgroup at or below MVN_FX_ST_NT_TARGET threads, which leaves the
  // registers for the tile fetched ahead (see fx_strided_body)
  // Lengths whose stage-0 work items do not fill whole waves (96, 160, 288: the padded extents of 64-,
  // 128- and 256-blocks) round the workgroup up; the idle threads of stage 0 are guarded (RAGGED).
  static constexpr int NT1 = M0 * CH;
  static constexpr int IT0 = fx_st_it0(NT1, MVN_FX_ST_NT_TARGET);
  static constexpr int NT = ((NT1 / IT0 + 63) / 64) * 64;
  static constexpr bool RAGGED = NT * IT0 != NT1;
  static constexpr int NWL = (N / RL) * CH;  // work items of the last stage
  static constexpr int ITL = (NWL + NT - 1) / NT;
  static constexpr int lds_cfloats = N * TP + fx_twsize(N);
  // waves per SIMD the register allocation should leave room for: what the LDS admits, at most 4
  static constexpr int WG_PER_CU = (160 * 1024) / (int)(sizeof(cfloat) * lds_cfloats) > 0 ? (160 * 1024) / (int)(sizeof(cfloat) * lds_cfloats) : 1;
  static constexpr int WAVES_WANTED = (WG_PER_CU * (NT / 64) + 3) / 4;
  static constexpr int WAVES = WAVES_WANTED > MVN_FX_ST_MAX_WAVES ? MVN_FX_ST_MAX_WAVES : WAVES_WANTED;
  static_assert(fx_smooth(N) && N >= 64 && N <= 2048 && N % 32 == 0, "unsupported fixed length");
  static_assert(R0 == 8 && NS >= 2 && M0 > 1 && fx_M(N, NS - 1) == 1, "unexpected radix plan");
  static_assert(NT % 64 == 0 && NT <= 1024 && (!RAGGED || IT0 == 1), "workgroup size");
  static_assert(sizeof(cfloat) * lds_cfloats <= 160 * 1024, "tile does not fit the LDS");
};

template <int N>
struct FxStridedRegs {
  static constexpr int NF = 8 * FxStridedCfg<N>::IT0, NL = FxStridedCfg<N>::ITL * FxStridedCfg<N>::RL;
  qfloat a[NF > NL ? NF : NL];  // the tile rows of this thread, fetched one tile ahead
  qfloat g[NL];                 // PSF-spectrum operands (fused pass), fetched one tile ahead
};

// one radix-R butterfly on both complex halves of R 16-byte registers
template <int R, int SIGN>
MVN_HD void fx_dft_q(qfloat* a) {
  cfloat lo[R], hi[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    lo[k] = cmake(a[k].x, a[k].y);
    hi[k] = cmake(a[k].z, a[k].w);
  }
  dftR<R, SIGN>(lo);
  dftR<R, SIGN>(hi);
#pragma unroll
  for (int k = 0; k < R; ++k) a[k] = qmake(lo[k].x, lo[k].y, hi[k].x, hi[k].y);
}

MVN_HD qfloat fx_qmul_c(qfloat a, cfloat w) {  // both halves times one complex factor
  const cfloat lo = cmul(cmake(a.x, a.y), w), hi = cmul(cmake(a.z, a.w), w);
  return qmake(lo.x, lo.y, hi.x, hi.y);
}

template <int SIGN>
MVN_HD qfloat fx_qmul_dir(qfloat a, cfloat w) {  // both halves times the twiddle of direction SIGN
  const cfloat lo = cmul_dir<SIGN>(cmake(a.x, a.y), w), hi = cmul_dir<SIGN>(cmake(a.z, a.w), w);
  return qmake(lo.x, lo.y, hi.x, hi.y);
}

MVN_HD qfloat fx_qmul_q(qfloat a, qfloat g) {  // half-wise complex product
  const cfloat lo = cmul(cmake(a.x, a.y), cmake(g.x, g.y)), hi = cmul(cmake(a.z, a.w), cmake(g.z, g.w));
  return qmake(lo.x, lo.y, hi.x, hi.y);
}

template <int R>
MVN_HD void fx_tw_row(const cfloat* row, cfloat* tw) {  // fx_rs(R) entries, 16-byte aligned
  const qfloat* t4 = reinterpret_cast<const qfloat*>(row);
#pragma unroll
  for (int k = 0; k < fx_rs(R) / 2; ++k) {
    const qfloat t = t4[k];
    tw[2 * k] = cmake(t.x, t.y);
    tw[2 * k + 1] = cmake(t.z, t.w);
  }
}

// an inner stage (1 <= S <= ns-2, so M > 1) through the LDS
template <int N, int NT, int S, int SIGN, bool DIF>
MVN_HD void fx_stage_q(cfloat* bufc, const cfloat* tws, int tid) {
  typedef FxStridedCfg<N> C;
  constexpr int R = fx_radix(N, S), M = fx_M(N, S), CH = C::CH, TPQ = C::TPQ;
  constexpr int nwork = (N / R) * CH;
  constexpr int iters = (nwork + NT - 1) / NT;
  qfloat* buf = reinterpret_cast<qfloat*>(bufc);
#pragma unroll
  for (int it = 0; it < iters; ++it) {
    const int w = tid + it * NT;
    if (nwork % NT != 0 && w >= nwork) break;
    const int b = w / CH, q = w % CH;
    const int blk = b / M, j2 = b % M;
    qfloat* p = buf + (blk * R * M + j2) * TPQ + q;
    qfloat a[R];
#pragma unroll
    for (int j = 0; j < R; ++j) a[j] = p[j * M * TPQ];
    cfloat tw[fx_rs(R)];
    fx_tw_row<R>(tws + fx_twoff(N, S) + j2 * fx_rs(R), tw);
    if (!DIF) {
#pragma unroll
      for (int k = 1; k < R; ++k) a[k] = fx_qmul_dir<SIGN>(a[k], tw[k]);
    }
    fx_dft_q<R, SIGN>(a);
    if (DIF) {
#pragma unroll
      for (int k = 1; k < R; ++k) a[k] = fx_qmul_dir<SIGN>(a[k], tw[k]);
    }
#pragma unroll
    for (int j = 0; j < R; ++j) p[j * M * TPQ] = a[j];
  }
}

// inner stages S = FROM, FROM +- 1, ..., TO as phases (DIF upwards, DIT downwards)
template <int N, int NT, int SIGN, bool DIF, int S, int TO, typename Ctx>
struct FxStagesQ {
  static MVN_HD void run(cfloat* buf, const cfloat* tws, Ctx& ctx) {
    constexpr int NT_ = NT;
    (void)NT_;
    if constexpr (DIF ? (S <= TO) : (S >= TO)) {
      MVN_PHASE(ctx, (fx_stage_q<N, NT, S, SIGN, DIF>(buf, tws, tid)));
      FxStagesQ<N, NT, SIGN, DIF, DIF ? S + 1 : S - 1, TO, Ctx>::run(buf, tws, ctx);
    }
  }
};

// phase functions (plain functions so that loop pragmas are honoured; MVN_PHASE only calls them)
//
// A workgroup walks over several tiles (block, block + step, ...).  The global loads of the NEXT
// tile are issued as soon as the registers that receive them are free -- right after stage 0 /
// the first inverse stage has been written to the LDS -- so they are in flight during the LDS
// stages and the stores of the current tile (fused pass: see fx_st_first).

#ifndef MVN_PROBE_BLOCK
#define MVN_PROBE_BLOCK(b) (b)  // timing probes only (mvn_kernels.hip, -DMVN_PROBE): re-use a few tiles
#endif
template <int N>
MVN_HD long fx_st_base(const StridedParams& P, long block) {
  block = MVN_PROBE_BLOCK(block);
  // launches have far fewer than 2^31 tiles (checked by the launcher): 32-bit division
  const unsigned o = (unsigned)block / (unsigned)P.tiles_per_outer;
  const unsigned t = (unsigned)block - o * (unsigned)P.tiles_per_outer;
  return (long)o * P.ostride + (long)t * FxStridedCfg<N>::T;
}

// where tile `block` finds its PSF-spectrum operands, and the distance between their rows
template <int N>
MVN_HD const cfloat* fx_spec_tile(const StridedParams& P, long block, long base) {
  return P.spec_tiled ? P.spec + (long)MVN_PROBE_BLOCK(block) * ((long)N * FxStridedCfg<N>::T) : P.spec + base;
}
template <int N>
MVN_HD long fx_spec_estride(const StridedParams& P) {
  return P.spec_tiled ? (long)FxStridedCfg<N>::T : P.estride;
}

// tile rows with the mapping of stage 0: work item w = tid + it NT <-> rows j2 + k M0
template <int N>
MVN_HD void fx_st_fetch_first(const StridedParams& P, long base, FxStridedRegs<N>& r, int tid) {
  typedef FxStridedCfg<N> C;
  const cfloat* src0 = (P.src ? P.src : P.data) + base;
  const long rstep = (long)C::M0 * P.estride;
#pragma unroll
  for (int it = 0; it < C::IT0; ++it) {
    const int w = tid + it * C::NT;
    if (C::RAGGED && w >= C::NT1) break;
    const int q = w % C::CH, j2 = w / C::CH;
    const cfloat* src = src0 + (long)j2 * P.estride + 2 * q;
#pragma unroll
    for (int k = 0; k < 8; ++k) r.a[it * 8 + k] = *reinterpret_cast<const qfloat*>(src + k * rstep);
  }
}

// rows of `from` with the mapping of the last stage: thread <-> rows b R + k (clamped work items
// are loaded but never used)
template <int N>
MVN_HD void fx_st_fetch_last_at(const cfloat* tile, long estride, qfloat* dst, int tid) {
  typedef FxStridedCfg<N> C;
#pragma unroll
  for (int it = 0; it < C::ITL; ++it) {
    int w = tid + it * C::NT;
    if (C::NWL % C::NT != 0 && w >= C::NWL) w = tid % C::NWL;  // clamped to a valid item: loaded, never used
    const int b = w / C::CH, q = w % C::CH;
    const cfloat* src = tile + (long)(b * C::RL) * estride + 2 * q;
#pragma unroll
    for (int k = 0; k < C::RL; ++k)
      dst[it * C::RL + k] = *reinterpret_cast<const qfloat*>(src + k * estride);
  }
}
template <int N>
MVN_HD void fx_st_fetch_last(const cfloat* from, const StridedParams& P, long base, qfloat* dst,
                             int tid) {
  fx_st_fetch_last_at<N>(from + base, P.estride, dst, tid);
}

// before the first tile: the twiddle table into the LDS, first tile's rows (and PSF operands)
// requested
template <int N, int MODE>
MVN_HD void fx_st_prologue(const StridedParams& P, long base, cfloat* tws, FxStridedRegs<N>& r,
                           int tid) {
  typedef FxStridedCfg<N> C;
  if (MODE == MVN_ST_INV)
    fx_st_fetch_last<N>(P.src ? P.src : P.data, P, base, r.a, tid);
  else
    fx_st_fetch_first<N>(P, base, r, tid);
  fx_copy_table<C::NT>(tws, P.ax.tws, fx_twsize(N), tid);
}

// forward entry: stage 0 in registers, then the next tile's rows are requested.  The fused pass
// instead requests this tile's PSF operands first (they land during stage 0 and the inner
// forward stages) and leaves the next tile's rows to the middle phase, when those registers are
// free again: both sets ahead of time at once do not fit 128 registers.
template <int N, int MODE>
MVN_HD void fx_st_first(const StridedParams& P, long block, long base, long next_base, bool has_next,
                        cfloat* buf, const cfloat* tws, FxStridedRegs<N>& r, int tid) {
  typedef FxStridedCfg<N> C;
  if (MODE == MVN_ST_FWD_MUL_INV)
    fx_st_fetch_last_at<N>(fx_spec_tile<N>(P, block, base), fx_spec_estride<N>(P), r.g, tid);
#pragma unroll
  for (int it = 0; it < C::IT0; ++it) {
    const int w = tid + it * C::NT;
    if (C::RAGGED && w >= C::NT1) break;
    const int q = w % C::CH, j2 = w / C::CH;
    qfloat a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = r.a[it * 8 + k];
    cfloat tw[8];
    fx_tw_row<8>(tws + fx_twoff(N, 0) + j2 * fx_rs(8), tw);
    fx_dft_q<8, -1>(a);
#pragma unroll
    for (int k = 1; k < 8; ++k) a[k] = fx_qmul_dir<-1>(a[k], tw[k]);
    qfloat* d = reinterpret_cast<qfloat*>(buf) + j2 * C::TPQ + q;
#pragma unroll
    for (int k = 0; k < 8; ++k) d[k * C::M0 * C::TPQ] = a[k];
  }
  if (MODE != MVN_ST_FWD_MUL_INV && has_next) fx_st_fetch_first<N>(P, next_base, r, tid);
}

// inverse entry: the last stage (M = 1, no twiddles) in registers
template <int N>
MVN_HD void fx_st_last_in(const StridedParams& P, long next_base, bool has_next, cfloat* buf,
                          FxStridedRegs<N>& r, int tid) {
  typedef FxStridedCfg<N> C;
  constexpr int CH = C::CH, RL = C::RL, ITL = C::ITL;
#pragma unroll
  for (int it = 0; it < ITL; ++it) {
    const int w = tid + it * C::NT;
    if (C::NWL % C::NT != 0 && w >= C::NWL) break;
    const int b = w / CH, q = w % CH;
    qfloat a[RL];
#pragma unroll
    for (int k = 0; k < RL; ++k) a[k] = r.a[it * RL + k];
    fx_dft_q<RL, +1>(a);
    qfloat* d = reinterpret_cast<qfloat*>(buf) + (b * RL) * C::TPQ + q;
#pragma unroll
    for (int k = 0; k < RL; ++k) d[k * C::TPQ] = a[k];
  }
  if (has_next) fx_st_fetch_last<N>(P.src ? P.src : P.data, P, next_base, r.a, tid);
}

// forward exit: last stage in registers, stored straight to global memory
template <int N>
MVN_HD void fx_st_last_store(const StridedParams& P, long base, const cfloat* buf, int tid) {
  typedef FxStridedCfg<N> C;
  constexpr int CH = C::CH, RL = C::RL, ITL = C::ITL;
#pragma unroll
  for (int it = 0; it < ITL; ++it) {
    const int w = tid + it * C::NT;
    if (C::NWL % C::NT != 0 && w >= C::NWL) break;
    const int b = w / CH, q = w % CH;
    const qfloat* s = reinterpret_cast<const qfloat*>(buf) + (b * RL) * C::TPQ + q;
    qfloat a[RL];
#pragma unroll
    for (int k = 0; k < RL; ++k) a[k] = s[k * C::TPQ];
    fx_dft_q<RL, -1>(a);
    cfloat* dst = P.data + base + (long)(b * RL) * P.estride + 2 * q;
#pragma unroll
    for (int k = 0; k < RL; ++k) *reinterpret_cast<qfloat*>(dst + k * P.estride) = a[k];
  }
}

// fused pass, middle: last forward stage, times the PSF spectrum (stored in the digit-reversed
// row order the forward transform produces), first inverse stage; then the next tile's rows
// are requested
template <int N>
MVN_HD void fx_st_last_mul_last(const StridedParams& P, long next_base, bool has_next, cfloat* buf,
                                FxStridedRegs<N>& r, int tid) {
  typedef FxStridedCfg<N> C;
  constexpr int CH = C::CH, RL = C::RL, ITL = C::ITL;
#pragma unroll
  for (int it = 0; it < ITL; ++it) {
    const int w = tid + it * C::NT;
    if (C::NWL % C::NT != 0 && w >= C::NWL) break;
    const int b = w / CH, q = w % CH;
    qfloat* s = reinterpret_cast<qfloat*>(buf) + (b * RL) * C::TPQ + q;
    qfloat a[RL];
#pragma unroll
    for (int k = 0; k < RL; ++k) a[k] = s[k * C::TPQ];
    fx_dft_q<RL, -1>(a);
#pragma unroll
    for (int k = 0; k < RL; ++k) a[k] = fx_qmul_q(a[k], r.g[it * RL + k]);
    fx_dft_q<RL, +1>(a);
#pragma unroll
    for (int k = 0; k < RL; ++k) s[k * C::TPQ] = a[k];
  }
  if (has_next) fx_st_fetch_first<N>(P, next_base, r, tid);
}

// inverse exit: stage 0 (twiddles first, decimation in time) in registers, stored to global
template <int N>
MVN_HD void fx_st_stage0_store(const StridedParams& P, long base, const cfloat* buf,
                               const cfloat* tws, int tid) {
  typedef FxStridedCfg<N> C;
  constexpr int M0 = C::M0, CH = C::CH;
  const long rstep = (long)M0 * P.estride;
#pragma unroll
  for (int it = 0; it < C::IT0; ++it) {
    const int w = tid + it * C::NT;
    if (C::RAGGED && w >= C::NT1) break;
    const int q = w % CH, j2 = w / CH;
    const qfloat* s = reinterpret_cast<const qfloat*>(buf) + j2 * C::TPQ + q;
    qfloat a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = s[k * M0 * C::TPQ];
    cfloat tw[8];
    fx_tw_row<8>(tws + fx_twoff(N, 0) + j2 * fx_rs(8), tw);
#pragma unroll
    for (int k = 1; k < 8; ++k) a[k] = fx_qmul_dir<+1>(a[k], tw[k]);
    fx_dft_q<8, +1>(a);
    cfloat* dst = P.data + base + (long)j2 * P.estride + 2 * q;
#pragma unroll
    for (int k = 0; k < 8; ++k) *reinterpret_cast<qfloat*>(dst + k * rstep) = a[k];
  }
}

// tiles first, first + step, ... < total.  WALK = false: exactly one tile per workgroup (the launch
// has one workgroup per tile), nothing is fetched ahead -- the fused pass then holds only its rows
// and its PSF operands in registers.
template <int N, int MODE, typename Ctx, bool WALK = true>
MVN_HD void fx_strided_body(const StridedParams& P, long first, long total, long step, cfloat* lds,
                            Ctx& ctx) {
  typedef FxStridedCfg<N> C;
  constexpr int NT = C::NT, NT_ = C::NT, NS = C::NS;
  (void)NT_;
  cfloat* buf = lds;
  cfloat* tws = lds + N * C::TP;
  if (first >= total) return;
  MVN_PHASE(ctx, (fx_st_prologue<N, MODE>(P, fx_st_base<N>(P, first), tws, r, tid)));
  for (long block = first; block < (WALK ? total : first + 1); block += step) {
    const long base = fx_st_base<N>(P, block);
    const bool has_next = WALK && block + step < total;
    const long next_base = has_next ? fx_st_base<N>(P, block + step) : base;
    if (MODE == MVN_ST_INV) {
      MVN_PHASE(ctx, (fx_st_last_in<N>(P, next_base, has_next, buf, r, tid)));
      FxStagesQ<N, NT, +1, false, NS - 2, 1, Ctx>::run(buf, tws, ctx);
      MVN_PHASE(ctx, (fx_st_stage0_store<N>(P, base, buf, tws, tid)));
    } else {
      MVN_PHASE(ctx, (fx_st_first<N, MODE>(P, block, base, next_base, has_next, buf, tws, r, tid)));
      FxStagesQ<N, NT, -1, true, 1, NS - 2, Ctx>::run(buf, tws, ctx);
      if (MODE == MVN_ST_FWD) {
        MVN_PHASE(ctx, (fx_st_last_store<N>(P, base, buf, tid)));
      } else {
        MVN_PHASE(ctx, (fx_st_last_mul_last<N>(P, next_base, has_next, buf, r, tid)));
        FxStagesQ<N, NT, +1, false, NS - 2, 1, Ctx>::run(buf, tws, ctx);
        MVN_PHASE(ctx, (fx_st_stage0_store<N>(P, base, buf, tws, tid)));
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// LDS-staged variant of the fused FWD * PSF * INV pass, used where two or more workgroups share
// a CU (tiles of at most half the LDS).  Every stage goes through the LDS with one column per work
// item (fx_stage), which keeps the working set at ~50 registers next to the tile's PSF operands.
// The register-staged walking body above cannot hold next tile + PSF tile + working tile in the
// 128 registers that two workgroups per CU leave and spills (measured: 0.38 ms vs 0.33 ms at
// 512^3, and 5.4 ms vs 3.0 ms on the 320-long axis of 320 x 1920 x 1920); walking itself did not
// pay for this body either (0.38 ms vs 0.33 ms one workgroup per tile).
// ---------------------------------------------------------------------------------------------
template <int N>
struct FxFusedCfg {
  typedef FxStridedCfg<N> S;
#if defined(MVN_EXPERIMENTS) && defined(MVN_FX_NO_LDS_FUSED)
  static constexpr bool USE = false;  // experiment (variant builds only): register-staged body for every length
#else
  static constexpr bool USE = S::WG_PER_CU >= 2;
#endif
  static constexpr int T = S::T, TP = S::TP, CH = S::CH;
  // threads: whole tile rows per sweep (N * CH divisible by NT), not more than one radix-8
  // butterfly each; 512 unless that leaves more than eight 16-byte loads per thread
  static constexpr int FULL = N * T / 8 >= 64 ? N * T / 8 : 64;
  static constexpr int NT512 = fx_pick_nt(512, FULL, N * CH, N * CH, N * CH);
  // (N = 576 takes nine loads per thread on 512 threads: with the 768 threads that eight loads would
  // mean, the 104 registers of the pass leave ONE 12-wave workgroup per CU where the LDS holds two -
  // measured at 576^3: 0.634 ms with 768 threads, 0.498 ms with 512 threads and 32 bytes of scratch)
#ifndef MVN_FX_FUSED_MAX_U
#define MVN_FX_FUSED_MAX_U 9
#endif
#ifndef MVN_FX_FUSED_NT_CAP
#define MVN_FX_FUSED_NT_CAP 1024
#endif
  static constexpr int NT = (NT512 > 0 && N * CH / NT512 <= MVN_FX_FUSED_MAX_U) ? NT512 : fx_pick_nt(MVN_FX_FUSED_NT_CAP, FULL, N * CH, N * CH, N * CH);
  static constexpr int RPT = NT / CH;  // tile rows covered by one sweep of the workgroup
  static constexpr int U = N / RPT;    // 16-byte loads per thread
#ifndef MVN_FX_FUSED_MAX_WAVES
#define MVN_FX_FUSED_MAX_WAVES MVN_FX_ST_MAX_WAVES
#endif
  static constexpr int WAVES_WANTED = (S::WG_PER_CU * (NT / 64) + 3) / 4;
  static constexpr int WAVES = WAVES_WANTED > MVN_FX_FUSED_MAX_WAVES ? MVN_FX_FUSED_MAX_WAVES : WAVES_WANTED;
  static_assert(NT >= 64 && N % RPT == 0, "tile rows must divide");
};

template <int N>
struct FxFusedRegs {
  qfloat g[FxFusedCfg<N>::U];  // PSF-spectrum operands of the tile
};

// thread <-> 16-byte chunk q of rows jr + u RPT
template <int N>
MVN_HD void fx_fu_fetch(const cfloat* from, const StridedParams& P, long base, qfloat* dst, int tid) {
  typedef FxFusedCfg<N> C;
  const int q = tid % C::CH, jr = tid / C::CH;
  const cfloat* src = from + base + (long)jr * P.estride + 2 * q;
  const long rstep = (long)C::RPT * P.estride;
#pragma unroll
  for (int u = 0; u < C::U; ++u) dst[u] = *reinterpret_cast<const qfloat*>(src + u * rstep);
}

// tile entry: this tile's rows and its PSF operands (the spectrum is stored in the digit-reversed
// row order the forward transform produces, i.e. row for row what the LDS holds after it) are
// requested together; the rows go to the LDS, the operands stay in registers until the multiply
template <int N>
MVN_HD void fx_fu_top(const StridedParams& P, long block, long base, cfloat* buf, cfloat* tws,
                      FxFusedRegs<N>& r, int tid) {
  typedef FxFusedCfg<N> C;
  qfloat v[C::U];
  fx_fu_fetch<N>(P.src ? P.src : P.data, P, base, v, tid);
  {  // the tile's PSF operands: same rows and chunks, from the spectrum's own layout
    const int q = tid % C::CH, jr = tid / C::CH;
    const long es = fx_spec_estride<N>(P);
    const cfloat* src = fx_spec_tile<N>(P, block, base) + (long)jr * es + 2 * q;
    const long rstep = (long)C::RPT * es;
#pragma unroll
    for (int u = 0; u < C::U; ++u) r.g[u] = *reinterpret_cast<const qfloat*>(src + u * rstep);
  }
  fx_copy_table<C::NT>(tws, P.ax.tws, fx_twsize(N), tid);
  const int q = tid % C::CH, jr = tid / C::CH;
#pragma unroll
  for (int u = 0; u < C::U; ++u)
    *reinterpret_cast<qfloat*>(buf + (jr + u * C::RPT) * C::TP + 2 * q) = v[u];
}

template <int N>
MVN_HD void fx_fu_mul(cfloat* buf, FxFusedRegs<N>& r, int tid) {
  typedef FxFusedCfg<N> C;
  const int q = tid % C::CH, jr = tid / C::CH;
#pragma unroll
  for (int u = 0; u < C::U; ++u) {
    qfloat* d = reinterpret_cast<qfloat*>(buf + (jr + u * C::RPT) * C::TP + 2 * q);
    *d = fx_qmul_q(*d, r.g[u]);
  }
}

template <int N>
MVN_HD void fx_fu_store(const StridedParams& P, long base, const cfloat* buf, int tid) {
  typedef FxFusedCfg<N> C;
  const int q = tid % C::CH, jr = tid / C::CH;
  const long rstep = (long)C::RPT * P.estride;
  cfloat* dst = P.data + base + (long)jr * P.estride + 2 * q;
#pragma unroll
  for (int u = 0; u < C::U; ++u)
    *reinterpret_cast<qfloat*>(dst + u * rstep) =
        *reinterpret_cast<const qfloat*>(buf + (jr + u * C::RPT) * C::TP + 2 * q);
}

// one tile per workgroup (see mvn_kernels.hip: walking was slower for this body)
template <int N, typename Ctx>
MVN_HD void fx_fused_lds_body(const StridedParams& P, long block, cfloat* lds, Ctx& ctx) {
  typedef FxFusedCfg<N> C;
  constexpr int T = C::T, TP = C::TP, NT = C::NT, NT_ = C::NT;
  (void)NT_;
  cfloat* buf = lds;
  cfloat* tws = lds + N * TP;
  const long base = fx_st_base<N>(P, block);
  MVN_PHASE(ctx, (fx_fu_top<N>(P, block, base, buf, tws, r, tid)));
  fx_dif<N, T, TP, false, NT, -1>(buf, tws, ctx);
  MVN_PHASE(ctx, (fx_fu_mul<N>(buf, r, tid)));
  fx_dit<N, T, TP, false, NT, +1>(buf, tws, ctx);
  MVN_PHASE(ctx, (fx_fu_store<N>(P, base, buf, tid)));
}

// ---------------------------------------------------------------------------------------------
// Long lines (N > 1024): a 16-column tile (128-byte row segments) is twice what the LDS holds.
// After stage 0 (radix 8, in registers) a line is 8 independent sub-lines of M0 = N/8 rows, so the
// whole tile is kept in REGISTERS (8 * IT0 16-byte values per thread) and the inner stages run on
// a part of it at a time: NWIN windows of 8 / NWIN sub-lines, each through W = N / NWIN LDS rows.
//   [window r: loads + last stage -> LDS, inner stages, LDS -> registers] x NWIN -> stage 0 -> stores
// The 8-column form (64-byte segments, FxStridedCfg) ran these passes at 3.0-3.4 TB/s.
// Round 4: the windows are a parameter (MVN_FX_SPLIT_NWIN_LONG = 2 or 4 for N >= 1536).  With half-line windows the
// rows a thread holds between their loads and the last stage (r.pf) are 80 registers and the NEXT tile's first
// window cannot be requested before the finished tile's stores (256 VGPRs + scratch, round 3); the counters show the
// pass short of requests in flight (11.7 k device-wide against 20 k at 512^3, profiles/r04_mem_counters_1920.md).
// With QUARTER windows (40 registers) the next tile's first window is requested under the last window's stages and
// the stores, 237 - 247 VGPRs, no scratch - and the pass takes exactly as long (64 x 1920 x 1920, same box:
// forward 0.436 / 0.434 ms, inverse 0.396 / 0.396, profiles/r04_ab_split_windows_1920.txt): those loads were not what
// it waits for.  Half windows stay the default (fewer barriers).
// ---------------------------------------------------------------------------------------------
template <int N>
struct FxSplitCfg {
  static constexpr bool USE = N > 1024;
  static constexpr int T = 16, TP = 16, CH = 8, TPQ = 8;
  static constexpr int NS = fx_nstages(N);
  static constexpr int M0 = fx_M(N, 0);
  static constexpr int RL = fx_radix(N, NS - 1);
#ifndef MVN_FX_SPLIT_NWIN_LONG
#define MVN_FX_SPLIT_NWIN_LONG 2
#endif
  static constexpr int NWIN = N >= 1536 ? MVN_FX_SPLIT_NWIN_LONG : 2;  // windows per tile
  static constexpr int SUB = 8 / NWIN;                               // sub-lines per window
  static constexpr int W = N / NWIN;                                 // rows per window
  // the next tile's first window is requested under the last window's stages and the stores (quarter windows)
  static constexpr bool NEXT_AHEAD = NWIN > 2;
  static constexpr int NT1 = M0 * CH;
  // threads: 1280 runs best with 640 (two stage-0 items per thread; 320: +7 %), 1920 with 384 (five items; 640
  // and 960 threads: +2 - 6 %) - profiles/r03_rows_lds.md
  static constexpr int IT0 = fx_st_it0(NT1, N <= 1280 ? 640 : 512);
  static constexpr int NT = NT1 / IT0;
  static constexpr int NWL = (W / RL) * CH;  // last-stage work items per window
  static constexpr int ITL = (NWL + NT - 1) / NT;
  static constexpr int TW1 = 0;  // the whole stage-ordered table sits behind the window
  // the forward form (row-permuted accesses, a little more register pressure): round 1 measured a
  // win at 1280 (0.263 -> 0.246 ms) and a loss at 1920 (2.77 -> 2.95 ms, 172 bytes of scratch);
  // built without SLP vectorisation (packed f32 math costs registers and issue slots on gfx950)
  // the 1920 kernel keeps 28 bytes of scratch and wins too: 2.91 -> 2.37 ms on 320 x 1920 x 1920
  static constexpr bool FWD_DEFAULT = true;
  static constexpr int lds_cfloats = W * TP + (fx_twsize(N) - TW1);
  static_assert(!USE || (fx_radix(N, 0) == 8 && NS >= 3 && N % 128 == 0 && W % RL == 0 && 8 % NWIN == 0), "split plan");
  static_assert(!USE || (NT % 64 == 0 && NT * IT0 == NT1), "split workgroup size");
  static_assert(!USE || sizeof(cfloat) * lds_cfloats <= 160 * 1024, "window does not fit the LDS");
};

template <int N>
struct FxSplitRegs {
  qfloat a[8 * FxSplitCfg<N>::IT0];  // the whole tile: rows j2 + k M0 of IT0 work items
  // the rows of a window between their global loads and the last stage; window 1's are requested while window 0
  // is in its inner stages, where `a` is still empty
  qfloat pf[FxSplitCfg<N>::ITL * FxSplitCfg<N>::RL];
};

// an inner stage on one window (rows are window-relative; sub-lines never straddle windows)
template <int N, int S, int SIGN, bool DIF>
MVN_HD void fx_stage_q_win(cfloat* bufc, const cfloat* twl, int tid) {
  typedef FxSplitCfg<N> C;
  constexpr int R = fx_radix(N, S), M = fx_M(N, S), CH = C::CH, TPQ = C::TPQ, NT = C::NT;
  static_assert(C::M0 % (R * M) == 0, "a stage's blocks must tile a sub-line");
  constexpr int nwork = (C::W / R) * CH;
  constexpr int iters = (nwork + NT - 1) / NT;
  qfloat* buf = reinterpret_cast<qfloat*>(bufc);
#pragma unroll
  for (int it = 0; it < iters; ++it) {
    const int w = tid + it * NT;
    if (nwork % NT != 0 && w >= nwork) break;
    const int b = w / CH, q = w % CH;
    const int blk = b / M, j2 = b % M;
    qfloat* p = buf + (blk * R * M + j2) * TPQ + q;
    qfloat a[R];
#pragma unroll
    for (int j = 0; j < R; ++j) a[j] = p[j * M * TPQ];
    cfloat tw[fx_rs(R)];
    fx_tw_row<R>(twl + (fx_twoff(N, S) - C::TW1) + j2 * fx_rs(R), tw);
    if (!DIF) {
#pragma unroll
      for (int k = 1; k < R; ++k) a[k] = fx_qmul_dir<SIGN>(a[k], tw[k]);
    }
    fx_dft_q<R, SIGN>(a);
    if (DIF) {
#pragma unroll
      for (int k = 1; k < R; ++k) a[k] = fx_qmul_dir<SIGN>(a[k], tw[k]);
    }
#pragma unroll
    for (int j = 0; j < R; ++j) p[j * M * TPQ] = a[j];
    MVN_SCHED_FENCE();
  }
}

template <int N, int SIGN, bool DIF, int S, int TO, typename Ctx>
struct FxStagesQWin {
  static MVN_HD void run(cfloat* buf, const cfloat* twl, Ctx& ctx) {
    constexpr int NT_ = FxSplitCfg<N>::NT;
    (void)NT_;
    if constexpr (DIF ? (S <= TO) : (S >= TO)) {
      MVN_PHASE(ctx, (fx_stage_q_win<N, S, SIGN, DIF>(buf, twl, tid)));
      FxStagesQWin<N, SIGN, DIF, DIF ? S + 1 : S - 1, TO, Ctx>::run(buf, twl, ctx);
    }
  }
};

template <int N>
MVN_HD void fx_sp_tables(const StridedParams& P, cfloat* twl, int tid) {
  typedef FxSplitCfg<N> C;
  fx_copy_table<C::NT>(twl, P.ax.tws + C::TW1, fx_twsize(N) - C::TW1, tid);
}

// LDS -> registers of window WIN (sub-lines SUB WIN .. SUB WIN + SUB - 1); WIN is a template parameter so
// that every register index is a compile-time constant
template <int N, int WIN>
MVN_HD void fx_sp_collect(const cfloat* buf, FxSplitRegs<N>& r, int tid) {
  typedef FxSplitCfg<N> C;
#pragma unroll
  for (int it = 0; it < C::IT0; ++it) {
    const int w = tid + it * C::NT;
    const int q = w % C::CH, j2 = w / C::CH;
    const qfloat* d = reinterpret_cast<const qfloat*>(buf) + j2 * C::TPQ + q;
#pragma unroll
    for (int kk = 0; kk < C::SUB; ++kk) r.a[it * 8 + C::SUB * WIN + kk] = d[kk * C::M0 * C::TPQ];
  }
}

// entry of a window, in two halves.  fx_sp_fetch: the global loads with the last stage's mapping (positions
// win W + b RL + k) into r.pf, items [IT_LO, IT_HI) of the thread.  PERM: position p is fetched from row
// rev(p) -- the forward transform reads its natural-order input in digit-reversed order.  fx_sp_last_to_lds:
// that stage (M = 1, no twiddles) on r.pf, results to the LDS window.
// With one 140 KB workgroup per CU nothing else covers a window's load latency: the loads of window 1 are
// issued when window 0 has reached the LDS and fly during its inner stages, while the tile registers are still
// empty (1920-long lines: forward 0.434 - 0.456 -> 0.417 - 0.421 ms, inverse 0.406 - 0.418 -> 0.393 - 0.401 ms at
// 64 x 1920 x 1920).  Requesting the next tile's window 0 between the stores of the finished tile as well brought
// the 1920 kernels to 256 VGPRs + 88 - 132 bytes of scratch and lost 8 - 20 % (profiles/r03_rows_lds.md).
template <int N, bool PERM, int IT_LO, int IT_HI>
MVN_HD void fx_sp_fetch(const StridedParams& P, long base, int win, FxSplitRegs<N>& r, int tid) {
  typedef FxSplitCfg<N> C;
  constexpr int CH = C::CH, RL = C::RL;
  const cfloat* src0 = (P.src ? P.src : P.data) + base;
  // the last digit has weight W_last in the natural index: the RL rows of an item are equally
  // spaced either way
  const long kstep = PERM ? (long)fx_W(N, C::NS - 1) * P.estride : P.estride;
#pragma unroll
  for (int it = IT_LO; it < IT_HI; ++it) {
    int w = tid + it * C::NT;
    if (C::NWL % C::NT != 0 && w >= C::NWL) w = tid % C::NWL;  // clamped to a valid item: loaded, never used
    const int b = w / CH, q = w % CH;
    const int pos = win * C::W + b * RL;
    long off = (long)(PERM ? fx_rev<N>(pos) : pos) * P.estride + 2 * q;
#pragma unroll
    for (int k = 0; k < RL; ++k) {
      r.pf[it * RL + k] = *reinterpret_cast<const qfloat*>(src0 + off);
      off += kstep;
      MVN_JIT_ADDRESS(off);
    }
  }
}

template <int N, int SIGN>
MVN_HD void fx_sp_last_to_lds(cfloat* buf, FxSplitRegs<N>& r, int tid) {
  typedef FxSplitCfg<N> C;
  constexpr int CH = C::CH, RL = C::RL, ITL = C::ITL;
#pragma unroll
  for (int it = 0; it < ITL; ++it) {
    const int w = tid + it * C::NT;
    if (C::NWL % C::NT != 0 && w >= C::NWL) break;
    const int b = w / CH, q = w % CH;
    qfloat x[RL];
#pragma unroll
    for (int k = 0; k < RL; ++k) x[k] = r.pf[it * RL + k];
    fx_dft_q<RL, SIGN>(x);
    qfloat* d = reinterpret_cast<qfloat*>(buf) + (b * RL) * C::TPQ + q;
#pragma unroll
    for (int k = 0; k < RL; ++k) d[k * C::TPQ] = x[k];
    MVN_SCHED_FENCE();
  }
}

// exit: stage 0 (twiddles first) on the registers, stored to global memory.  PERM: natural index
// n = j2 + k M0 goes to row inv(n) -- the forward transform leaves its spectrum in position order.
template <int N, int SIGN, bool PERM>
MVN_HD void fx_sp_stage0_store(const StridedParams& P, long base, const cfloat* twl, FxSplitRegs<N>& r,
                               int tid) {
  typedef FxSplitCfg<N> C;
  const long rstep = (long)C::M0 * P.estride;
  cfloat* dst0 = P.data + base;
#pragma unroll
  for (int it = 0; it < C::IT0; ++it) {
    const int w = tid + it * C::NT;
    const int q = w % C::CH, j2 = w / C::CH;
    cfloat tw[8];
    fx_tw_row<8>(twl + fx_twoff(N, 0) + j2 * fx_rs(8), tw);
    qfloat a[8];  // a local copy: a pointer into r.a would keep the whole array in memory
    a[0] = r.a[it * 8];
#pragma unroll
    for (int k = 1; k < 8; ++k) a[k] = fx_qmul_dir<SIGN>(r.a[it * 8 + k], tw[k]);
    fx_dft_q<8, SIGN>(a);
    if (PERM) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        *reinterpret_cast<qfloat*>(dst0 + (long)fx_inv<N>(j2 + k * C::M0) * P.estride + 2 * q) = a[k];
    } else {
      long off = (long)j2 * P.estride + 2 * q;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        *reinterpret_cast<qfloat*>(dst0 + off) = a[k];
        off += rstep;
        MVN_JIT_ADDRESS(off);
      }
    }
    MVN_SCHED_FENCE();
  }
}

template <int N>
MVN_HD long fx_sp_base(const StridedParams& P, long block) {
  const unsigned o = (unsigned)block / (unsigned)P.tiles_per_outer;
  const unsigned t = (unsigned)block - o * (unsigned)P.tiles_per_outer;
  return (long)o * P.ostride + (long)t * FxSplitCfg<N>::T;
}

// The windows of a tile one after the other: last stage of window WIN (its rows are in r.pf) -> LDS, and in the same
// phase the request for the rows that come next - window WIN + 1 of this tile, or (quarter windows) window 0 of the
// workgroup's NEXT tile, which then fly during the inner stages, the collect, stage 0 and the stores.
template <int N, int SIGN, bool PERM, int WIN, typename Ctx>
struct FxSplitWindows {
  static MVN_HD void run(const StridedParams& P, long base, long next_base, bool has_next, cfloat* buf,
                         const cfloat* twl, Ctx& ctx) {
    typedef FxSplitCfg<N> C;
    constexpr int NT_ = C::NT, NS = C::NS;
    (void)NT_;
    if constexpr (WIN < C::NWIN) {
      if constexpr (WIN + 1 < C::NWIN) {
        MVN_PHASE(ctx, (fx_sp_last_to_lds<N, SIGN>(buf, r, tid), fx_sp_fetch<N, PERM, 0, C::ITL>(P, base, WIN + 1, r, tid)));
      } else if constexpr (C::NEXT_AHEAD) {
        MVN_PHASE(ctx, (fx_sp_last_to_lds<N, SIGN>(buf, r, tid),
                        has_next ? fx_sp_fetch<N, PERM, 0, C::ITL>(P, next_base, 0, r, tid) : (void)0));
      } else {
        MVN_PHASE(ctx, (fx_sp_last_to_lds<N, SIGN>(buf, r, tid)));
      }
      FxStagesQWin<N, SIGN, false, NS - 2, 1, Ctx>::run(buf, twl, ctx);
      MVN_PHASE(ctx, (fx_sp_collect<N, WIN>(buf, r, tid)));
      FxSplitWindows<N, SIGN, PERM, WIN + 1, Ctx>::run(P, base, next_base, has_next, buf, twl, ctx);
    }
  }
};

// tiles first, first + step, ... < total (MODE is FWD or INV).  Both directions are decimation in
// time -- the tile ACCUMULATES in registers window by window, so the registers fill up only at
// the end (a decimation-in-frequency forward form, with every row live from the start, spilled
// 320 bytes per thread and lost 45 %).  The forward transform gets its digit-reversed input and
// its position-ordered output by permuting whole rows in the global accesses, which costs
// nothing for 128-byte row segments.
template <int N, int MODE, typename Ctx>
MVN_HD void fx_strided_split_body(const StridedParams& P, long first, long total, long step,
                                  cfloat* lds, Ctx& ctx) {
  typedef FxSplitCfg<N> C;
  constexpr int NT_ = C::NT;
  constexpr int SIGN = MODE == MVN_ST_FWD ? -1 : +1;
  constexpr bool PERM = MODE == MVN_ST_FWD;
  (void)NT_;
  cfloat* buf = lds;
  cfloat* twl = lds + C::W * C::TP;
  if (first >= total) return;
  MVN_PHASE(ctx, (fx_sp_tables<N>(P, twl, tid)));
  if constexpr (C::NEXT_AHEAD) {
    MVN_PHASE_NOSYNC(ctx, (fx_sp_fetch<N, PERM, 0, C::ITL>(P, fx_sp_base<N>(P, first), 0, r, tid)));
  }
  for (long block = first; block < total; block += step) {
    MVN_TILE_LOOP_TOP(ctx);
    const long base = fx_sp_base<N>(P, block);
    if constexpr (!C::NEXT_AHEAD) {
      MVN_PHASE_NOSYNC(ctx, (fx_sp_fetch<N, PERM, 0, C::ITL>(P, base, 0, r, tid)));
    }
    const bool has_next = block + step < total;
    const long next_base = has_next ? fx_sp_base<N>(P, block + step) : base;
    FxSplitWindows<N, SIGN, PERM, 0, Ctx>::run(P, base, next_base, has_next, buf, twl, ctx);
    MVN_PHASE_NOSYNC(ctx, (fx_sp_stage0_store<N, SIGN, PERM>(P, base, twl, r, tid)));
  }
}

// which body, register block and workgroup size a (length, mode) pair uses
template <int N, int MODE>
struct FxStridedSel {
  // the fused pass where two or more workgroups share a CU: 0 = every stage through the LDS with
  // one column per work item (fx_fused_lds_body), 1 = outer stages in registers, one tile per
  // workgroup (fx_strided_body<.., WALK = false>: half the LDS traffic, 16-byte LDS accesses)
#ifndef MVN_FX_FUSED_VARIANT
#define MVN_FX_FUSED_VARIANT 0
#endif
  static constexpr bool ONE_TILE = MODE == MVN_ST_FWD_MUL_INV && FxFusedCfg<N>::USE;
  static constexpr bool LDS_FUSED = ONE_TILE && MVN_FX_FUSED_VARIANT == 0;
  static constexpr int NT = LDS_FUSED ? FxFusedCfg<N>::NT : FxStridedCfg<N>::NT;
  static constexpr int WAVES = LDS_FUSED ? FxFusedCfg<N>::WAVES : FxStridedCfg<N>::WAVES;
  typedef typename std::conditional<LDS_FUSED, FxFusedRegs<N>, FxStridedRegs<N>>::type Regs;
  typedef FxCtx<Regs, NT> Ctx;
  static MVN_HD void run(const StridedParams& P, long first, long total, long step, cfloat* lds,
                         Ctx& ctx) {
    if constexpr (LDS_FUSED) {
      // launched one workgroup per tile (step == grid size == total)
      for (long block = first; block < total; block += step) fx_fused_lds_body<N>(P, block, lds, ctx);
    } else if constexpr (ONE_TILE) {
      for (long block = first; block < total; block += step)
        fx_strided_body<N, MODE, Ctx, false>(P, block, total, step, lds, ctx);
    } else
      fx_strided_body<N, MODE>(P, first, total, step, lds, ctx);
  }
};

// ---------------------------------------------------------------------------------------------
// last-axis passes for even d2 = 2H, H a power of two; T rows per tile, transposed in LDS with
// an odd pitch and one spare row per 32 rows.  Requires rows % T == 0.
//
// Stage 0 of the transform (radix 8, stride M0 = H/8) is done in registers right next to the
// global accesses, with the bins of one row on neighbouring lanes ("row-fastest": slot w ->
// j2 = w % M0, row = w / M0), so that
//   r2c : global real row -> first forward stage -> LDS                (no separate load phase)
//   c2r : LDS -> last inverse stage -> pointwise epilogue -> global    (no separate store phase)
//   c2r+r2c fused: ... last inverse stage -> epilogue -> first forward stage ... all in registers
// The remaining stages use the column-fastest mapping of fx_stage.
// ---------------------------------------------------------------------------------------------
template <int H>
struct FxRowsCfg {
  static constexpr bool PAD = fx_pow2(H);  // the spare-row trick relies on power-of-two block sizes
#ifndef MVN_FX_ROWS_T
#define MVN_FX_ROWS_T 16
#endif
  // Two tile classes.  TILED: one 16-row tile per workgroup (8 rows above H = 512), tables rebuilt
  // per tile.  WALKING: small tiles of 2 - 8 rows, several small workgroups per CU that WALK over
  // the tiles of the launch so that the tables (up to 16 KB) are built once per workgroup.  Which
  // one a length uses is measured (tools/rows_tune.sh on MI355X, fused divide + fused update of
  // d2^3 cubes, round 2):
  //   H = 96, 128, 256                tiled wins (0.31 vs 0.27..0.33 at 320^3 is the crossover)
  //   H = 160, 288, 320, 384, 512     walking, 4 rows: -12 % (320^3), -29 % (576^3: 2.50 -> 1.77 ms;
  //                                   its 9-wave workgroups of 576 threads were the worst fit),
  //                                   -9 % (640^3), -9 % (768^3), -13 % (1024^3)
  //   H = 192                         walking, 8 rows: -18 % (384^3)
  //   H = 48, 80, 144                 walking, 8 rows (d2 = 96, 160, 288: no whole-wave tiled geometry)
  //   H = 480                         walking, 2 rows (2 % ahead of 4 rows at 960^3)
  //   H >= 640                        walking, 2 rows (a 16-row tile does not fit at all, the 8-row
  //                                   tile leaves ONE workgroup per CU: 3.0 / 3.7 TB/s at 320 x 1920 x
  //                                   1920 in round 1)
  // The plain r2c pass alone would prefer the tiled class for the middle lengths (+1..48 %); it runs
  // once per call.  MVN_FX_ROWS_SMALL_T > 0 forces the walking class with that many rows for every
  // H >= MVN_FX_ROWS_SMALL_MIN (tuning builds).
#ifndef MVN_FX_ROWS_SMALL_MIN
#define MVN_FX_ROWS_SMALL_MIN 65
#endif
#ifndef MVN_FX_ROWS_SMALL_T
#define MVN_FX_ROWS_SMALL_T 0
#endif
  static constexpr int WALK_T = (MVN_FX_ROWS_SMALL_T > 0 && H >= MVN_FX_ROWS_SMALL_MIN) ? MVN_FX_ROWS_SMALL_T
                                : (H >= 640 || H == 480)                   ? 2
                                : (H == 192 || H == 144 || H == 80 || H == 48) ? 8
                                : (H >= 160 && !(fx_pow2(H) && H < 512))   ? 4
                                                                           : 0;
  static constexpr bool SMALL = WALK_T > 0;
  static constexpr bool WALK = SMALL;
  // walking kernels transpose the stage twiddles when they copy them to the LDS (fx_tw_fetch)
  static constexpr bool TWT = WALK;
  static constexpr int T = SMALL ? WALK_T : (H > 512 ? 8 : ((fx_pow2(H) && H >= 128) ? MVN_FX_ROWS_T : 16));
  static constexpr int TP = T + 1;
  static constexpr int QR = H / 2;  // 16-byte chunks per spectral row (2 complex bins each)
  static constexpr int R0 = fx_radix(H, 0);
  static constexpr int M0 = fx_M(H, 0);
  // threads: one stage-0 butterfly each where possible (up to 1024 per workgroup), dividing the
  // 16-byte row chunks, the stage-0 butterflies and the bin pairs evenly
#ifndef MVN_FX_ROWS_NT_CAP
#define MVN_FX_ROWS_NT_CAP 1024
#endif
  // (small tiles: one stage-0 butterfly per thread, rounded up to whole waves; the last threads
  // of a sweep then idle -- every per-thread loop below is guarded where the counts do not divide)
  static constexpr int NT = SMALL ? ((M0 * T + 63) / 64) * 64
                                  : fx_pick_nt(MVN_FX_ROWS_NT_CAP, H * T / 8 >= 64 ? H * T / 8 : 64, T * QR, M0 * T, H / 2 * T);
  static constexpr int NTD = NT > 0 ? NT : 1;
  static constexpr int U = (T * QR + NTD - 1) / NTD;   // 16-byte spectral loads/stores per thread
  static constexpr int IT0 = (M0 * T + NTD - 1) / NTD;  // stage-0 butterflies per thread
  static constexpr bool EXACT = (T * QR) % NTD == 0 && (M0 * T) % NTD == 0 && (H / 2 * T) % NTD == 0;
  static constexpr int ROWS = fx_rows_alloc(H, PAD);
  static constexpr int TILE = (ROWS * TP + 1) & ~1;  // keeps the tables behind it 16-byte aligned
  // tile | stage twiddles | d2-th roots (H/2+1, padded to even) | pair table (H/2 entries of 2 ints)
  static constexpr int TWR = (H / 2 + 2) & ~1;
  static constexpr int lds_cfloats = TILE + fx_twsize(H) + TWR + H / 2;
  static constexpr int ITP = (H / 2 * T + NTD - 1) / NTD;  // bin pairs per thread in the real<->complex step
  static_assert(fx_smooth(H) && H % 2 == 0 && H >= 32 && H <= 1024, "unsupported fixed length");
  static_assert(NT >= 64 && R0 == 8, "stage 0: whole radix-8 butterflies per thread");
  static_assert(sizeof(cfloat) * lds_cfloats <= 160 * 1024, "tile does not fit the LDS");
};

template <int H>
struct FxRowsRegs {
  qfloat v[FxRowsCfg<H>::U];
  // epilogue operands of the elements this thread finishes in the last inverse stage
  cfloat ea[FxRowsCfg<H>::IT0][FxRowsCfg<H>::R0];
  cfloat eb[FxRowsCfg<H>::IT0][FxRowsCfg<H>::R0];
};

// LDS offsets (in cfloat, row * TP) of the two bins the real<->complex step combines: entry k
// (0 < k < H/2) = {bin k, bin H-k}; entry 0 = {bin 0, bin H/2} (the two self-paired bins).
// Built in LDS once per workgroup so that the per-pair index math is one 8-byte LDS read.
struct FxPair {
  int a, b;
};

template <int H>
MVN_HD void fx_build_pair_table(FxPair* tab, int tid) {
  typedef FxRowsCfg<H> C;
  for (int k = tid; k < H / 2; k += C::NT) {
    FxPair e;
    e.a = fx_row<C::PAD>(fx_inv<H>(k)) * C::TP;
    e.b = fx_row<C::PAD>(fx_inv<H>(k == 0 ? H / 2 : H - k)) * C::TP;
    tab[k] = e;
  }
}

// twiddles of one butterfly row: R entries, 16-byte aligned
template <int R>
MVN_HD void fx_load_tw_row(const cfloat* row, cfloat* tw) {
  const qfloat* t4 = reinterpret_cast<const qfloat*>(row);
#pragma unroll
  for (int k = 0; k < fx_rs(R) / 2; ++k) {
    const qfloat t = t4[k];
    tw[2 * k] = cmake(t.x, t.y);
    tw[2 * k + 1] = cmake(t.z, t.w);
  }
}

// the tables of a last-axis workgroup: stage twiddles, d2-th roots, pair offsets (built per tile by
// the one-tile kernels, once per workgroup by the walking ones)
template <int H>
MVN_HD void fx_rows_tables(const RowsParams& P, cfloat* tws, cfloat* twr, int tid) {
  typedef FxRowsCfg<H> C;
  if constexpr (C::TWT) {
    // stage blocks transposed to [k / 2][j2] 16-byte pairs, see fx_tw_fetch
    for (int st = 0; st < fx_nstages(H); ++st) {
      const int M = fx_M(H, st), Q = fx_rs(fx_radix(H, st)) / 2;
      if (M <= 1) continue;
      const qfloat* src = reinterpret_cast<const qfloat*>(P.ax.tws + fx_twoff(H, st));
      qfloat* dst = reinterpret_cast<qfloat*>(tws + fx_twoff(H, st));
      for (int i = tid; i < M * Q; i += C::NT) {
        const qfloat t = src[i];
        dst[(i % Q) * M + i / Q] = qmake(t.x, t.y, t.z, t.w);
      }
    }
  } else {
    fx_copy_table<C::NT>(tws, P.ax.tws, fx_twsize(H), tid);
  }
  fx_copy_table<C::NT>(twr, P.twr, H / 2 + 1, tid);
  fx_build_pair_table<H>(reinterpret_cast<FxPair*>(twr + C::TWR), tid);
}

// r2c phase A: global load -> first forward stage in registers -> LDS
template <int H>
MVN_HD void fx_r2c_load_stage0(const RowsParams& P, long r0, cfloat* buf, cfloat* tws, cfloat* twr,
                               int tid) {
  typedef FxRowsCfg<H> C;
  constexpr int TP = C::TP, NT = C::NT, R = C::R0, M = C::M0;
  cfloat a[C::IT0][R];
#pragma unroll
  for (int it = 0; it < C::IT0; ++it) {
    const int w = tid + it * NT;
    if (!C::EXACT && w >= M * C::T) break;
    const int j2 = w % M, rho = w / M;
    const cfloat* src = reinterpret_cast<const cfloat*>(P.in_real + (r0 + rho) * P.RP) + j2;
#pragma unroll
    for (int j = 0; j < R; ++j) a[it][j] = src[j * M];
  }
  if (!C::WALK) fx_rows_tables<H>(P, tws, twr, tid);
#pragma unroll
  for (int it = 0; it < C::IT0; ++it) {
    const int w = tid + it * NT;
    if (!C::EXACT && w >= M * C::T) break;
    const int j2 = w % M, rho = w / M;
    cfloat tw[fx_rs(R)];
    // stage 0 opens the table; a one-tile workgroup's LDS copy is not ready yet
    if constexpr (C::WALK)
      fx_tw_fetch<H, 0, C::TWT>(tws, j2, tw);
    else
      fx_load_tw_row<R>(P.ax.tws + j2 * fx_rs(R), tw);
    dftR<R, -1>(a[it]);
#pragma unroll
    for (int k = 1; k < R; ++k) a[it][k] = cmul(a[it][k], tw[k]);
    cfloat* p = buf + fx_row<C::PAD>(j2) * TP + rho;
#pragma unroll
    for (int k = 0; k < R; ++k) p[fx_rowoff<C::PAD, R, M>(k) * TP] = a[it][k];
  }
}

// real -> half-complex step on the packed transform Z (positions via the pair table):
// X[k] = E - i w^k D, X[H-k] = conj(E) - i conj(w^k D), E = (Z[k] + conj Z[H-k])/2, D = (Z[k] - conj Z[H-k])/2
template <int H>
MVN_HD void fx_r2c_post(const RowsParams& P, long r0, cfloat* buf, const cfloat* twr, int tid) {
  typedef FxRowsCfg<H> C;
  constexpr int T = C::T, NT = C::NT;
  const FxPair* tab = reinterpret_cast<const FxPair*>(twr + C::TWR);
#pragma unroll
  for (int it = 0; it < C::ITP; ++it) {
    const int i = tid + it * NT;
    if (!C::EXACT && i >= H / 2 * T) break;
    const int k = i / T, rho = i % T;
    const FxPair t = tab[k];
    cfloat* pa = buf + t.a + rho;
    cfloat* pb = buf + t.b + rho;
    const cfloat zk = *pa;
    const cfloat zm = *pb;
    if (k == 0) {
      if (P.nyq_packed) {
        *pa = cmake(zk.x + zk.y, zk.x - zk.y);             // DC + i Nyquist (RowsParams::nyq_packed)
      } else {
        *pa = cmake(zk.x + zk.y, 0.f);                     // DC
        P.out_nyq[r0 + rho] = cmake(zk.x - zk.y, 0.f);     // Nyquist, kept in its own plane
      }
      *pb = cconj(zm);                                     // bin H/2 pairs with itself
    } else {
      const cfloat E = cscale(cadd_c(zk, zm), 0.5f);
      const cfloat D = cscale(csub_c(zk, zm), 0.5f);
      const cfloat G = cmul(D, twr[k]);
      *pa = cadd_i<-1>(E, G);
      *pb = cconj_add_i<-1>(E, G);
    }
  }
}

// LINES: the half-spectrum is kept in the line layout of the fused middle pass (mvn_mid_fused.hpp),
// spec[plane][position][row of the plane]: the T rows of a tile are T neighbouring entries of H lines, one
// 8-byte access per lane, T lanes per line (128 contiguous bytes for 16-row tiles).  Row `r0` of the launch is row
// P.row_base + r0 of the volume; tiles never straddle planes (T divides the rows of a plane).
// The line-layout side in 16-byte accesses (two neighbouring rows per lane) or 8-byte ones (one row per lane), per
// epilogue: measured at 512^3 (tools/ab_libs.sh, profiles/r04_mid_fused.md) the fused divide gains 11 % with 16 bytes
// (0.337 -> 0.299 ms), the plain r2c pass 6 %, the fused update loses 1.4 % (0.495 -> 0.502: three more streams).
#ifndef MVN_FX_LINES_16B
#define MVN_FX_LINES_16B -1  // -1: per epilogue; 0 / 1: all passes 8 / 16 bytes (A/B builds)
#endif
template <int EPI>
constexpr bool fx_lines_wide() {
  return MVN_FX_LINES_16B < 0 ? (EPI != MVN_EPI_UPDATE && EPI != MVN_EPI_DELTA) : MVN_FX_LINES_16B != 0;
}
template <int H>
MVN_HD long fx_lines_base(const RowsParams& P, long r0) {
  const long R = P.row_base + r0;
  const long z = R / P.lines_d1;
  return z * (long)H * P.lines_d1 + (R - z * P.lines_d1);
}

template <int H, bool LINES = false, bool W16 = true>
MVN_HD void fx_r2c_store(const RowsParams& P, long r0, const cfloat* buf, int tid) {
  typedef FxRowsCfg<H> C;
  constexpr int TP = C::TP, NT = C::NT, U = C::U;
  if constexpr (LINES) {
    static_assert((C::T * H) % NT == 0, "line layout: whole sweeps");
    cfloat* dst = P.out_cplx + fx_lines_base<H>(P, r0);
    if constexpr (W16) {
      // two neighbouring rows per lane: one 16-byte global access, two 8-byte LDS accesses (the tile's pitch is odd)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = tid + u * NT;
        const int r2 = 2 * (e % (C::T / 2)), p = e / (C::T / 2);
        const cfloat a = buf[fx_row<C::PAD>(p) * TP + r2], b = buf[fx_row<C::PAD>(p) * TP + r2 + 1];
        *reinterpret_cast<qfloat*>(dst + (long)p * P.lines_d1 + r2) = qmake(a.x, a.y, b.x, b.y);
      }
    } else {
#pragma unroll
      for (int u = 0; u < 2 * U; ++u) {
        const int e = tid + u * NT;
        const int rho = e % C::T, p = e / C::T;
        dst[(long)p * P.lines_d1 + rho] = buf[fx_row<C::PAD>(p) * TP + rho];
      }
    }
    return;
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int e = tid + u * NT;
    if (!C::EXACT && e >= C::T * C::QR) break;
    const int rho = e / C::QR, kk = e % C::QR;
    // spectral rows are kept in position (digit-reversed) order: bin k sits at column inv(k)
    const cfloat a = buf[fx_row<C::PAD>(2 * kk) * TP + rho];
    const cfloat b = buf[fx_row<C::PAD>(2 * kk + 1) * TP + rho];
    reinterpret_cast<qfloat*>(P.out_cplx + (r0 + rho) * P.C)[kk] = qmake(a.x, a.y, b.x, b.y);
  }
}

template <int H, typename Ctx, bool LINES = false>
MVN_HD void fx_rows_r2c_body(const RowsParams& P, long tile, cfloat* lds, Ctx& ctx) {
  typedef FxRowsCfg<H> C;
  constexpr int T = C::T, TP = C::TP, NT = C::NT, NT_ = C::NT;
  (void)NT_;
  const long r0 = tile * T;
  cfloat* buf = lds;
  cfloat* tws = lds + C::TILE;
  cfloat* twr = tws + fx_twsize(H);
  MVN_PHASE(ctx, (fx_r2c_load_stage0<H>(P, r0, buf, tws, twr, tid)));
  fx_dif<H, T, TP, C::PAD, NT, -1, 1, C::TWT>(buf, tws, ctx);
  MVN_PHASE(ctx, (fx_r2c_post<H>(P, r0, buf, twr, tid)));
  MVN_PHASE(ctx, (fx_r2c_store<H, LINES, fx_lines_wide<MVN_EPI_STORE>()>(P, r0, buf, tid)));
}

// compile-time-mode form of the fused pass's pair epilogue: the kernels are instantiated per
// mode so that a mode's dead operand registers and branches vanish
template <int EPI>
MVN_HD cfloat fx_epilogue_pair_value(const EpilogueParams& e, long i, cfloat z, cfloat a, cfloat b) {
  return mvn_epilogue_pair_value(EPI, e, i, z, a, b);
}

// c2r phase 0: spectral rows -> LDS (position order both sides), plus the epilogue operands of the
// elements this thread will finish in the last inverse stage, fetched a whole transform ahead
template <int H, int EPI, bool LINES = false>
MVN_HD void fx_c2r_load(const RowsParams& P, long r0, cfloat* buf, cfloat* tws, cfloat* twr,
                        FxRowsRegs<H>& r, int tid) {
  typedef FxRowsCfg<H> C;
  constexpr int TP = C::TP, NT = C::NT, U = C::U;
  if constexpr (LINES) {
    // two 8-byte entries per register quad: sweep 2 u in .xy, sweep 2 u + 1 in .zw
    const cfloat* src = P.in_cplx + fx_lines_base<H>(P, r0);
    if constexpr (fx_lines_wide<EPI>()) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = tid + u * NT;
        r.v[u] = *reinterpret_cast<const qfloat*>(src + (long)(e / (C::T / 2)) * P.lines_d1 + 2 * (e % (C::T / 2)));
      }
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e0 = tid + 2 * u * NT, e1 = e0 + NT;
        const cfloat a = src[(long)(e0 / C::T) * P.lines_d1 + e0 % C::T];
        const cfloat b = src[(long)(e1 / C::T) * P.lines_d1 + e1 % C::T];
        r.v[u] = qmake(a.x, a.y, b.x, b.y);
      }
    }
  } else {
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int e = tid + u * NT;
    if (!C::EXACT && e >= C::T * C::QR) break;
    const int rho = e / C::QR, kk = e % C::QR;
    r.v[u] = reinterpret_cast<const qfloat*>(P.in_cplx + (r0 + rho) * P.C)[kk];
  }
  }
  constexpr int mode = EPI;
  const float* pa = mode == MVN_EPI_DIVIDE ? P.epi.view : P.epi.psi;
  if (mode != MVN_EPI_STORE) {
#pragma unroll
    for (int it = 0; it < C::IT0; ++it) {
      const int w = tid + it * NT;
      if (!C::EXACT && w >= C::M0 * C::T) break;
      const cfloat* src = reinterpret_cast<const cfloat*>(pa + (r0 + w / C::M0) * P.RP) + (w % C::M0);
#pragma unroll
      for (int jo = 0; jo < C::R0; ++jo) r.ea[it][jo] = src[jo * C::M0];
    }
  }
  if (mode == MVN_EPI_UPDATE || mode == MVN_EPI_DELTA) {
#pragma unroll
    for (int it = 0; it < C::IT0; ++it) {
      const int w = tid + it * NT;
      if (!C::EXACT && w >= C::M0 * C::T) break;
      const cfloat* src =
          reinterpret_cast<const cfloat*>(P.epi.weights + (r0 + w / C::M0) * P.RP) + (w % C::M0);
#pragma unroll
      for (int jo = 0; jo < C::R0; ++jo) r.eb[it][jo] = src[jo * C::M0];
    }
  }
  if (!C::WALK) fx_rows_tables<H>(P, tws, twr, tid);
  if constexpr (LINES) {
    if constexpr (fx_lines_wide<EPI>()) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = tid + u * NT;
        cfloat* q = buf + fx_row<C::PAD>(e / (C::T / 2)) * TP + 2 * (e % (C::T / 2));
        q[0] = cmake(r.v[u].x, r.v[u].y);
        q[1] = cmake(r.v[u].z, r.v[u].w);
      }
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e0 = tid + 2 * u * NT, e1 = e0 + NT;
        buf[fx_row<C::PAD>(e0 / C::T) * TP + e0 % C::T] = cmake(r.v[u].x, r.v[u].y);
        buf[fx_row<C::PAD>(e1 / C::T) * TP + e1 % C::T] = cmake(r.v[u].z, r.v[u].w);
      }
    }
    return;
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int e = tid + u * NT;
    if (!C::EXACT && e >= C::T * C::QR) break;
    const int rho = e / C::QR, kk = e % C::QR;
    buf[fx_row<C::PAD>(2 * kk) * TP + rho] = cmake(r.v[u].x, r.v[u].y);
    buf[fx_row<C::PAD>(2 * kk + 1) * TP + rho] = cmake(r.v[u].z, r.v[u].w);
  }
}

// half-complex -> real step: Z[k] = E + i O, Z[H-k] = conj(E) + i conj(O),
// E = X[k] + conj X[H-k], O = (X[k] - conj X[H-k]) exp(+2 pi i k / d2)
template <int H>
MVN_HD void fx_c2r_pre(const RowsParams& P, long r0, cfloat* buf, const cfloat* twr, int tid) {
  typedef FxRowsCfg<H> C;
  constexpr int T = C::T, NT = C::NT;
  const FxPair* tab = reinterpret_cast<const FxPair*>(twr + C::TWR);
#pragma unroll
  for (int it = 0; it < C::ITP; ++it) {
    const int i = tid + it * NT;
    if (!C::EXACT && i >= H / 2 * T) break;
    const int k = i / T, rho = i % T;
    const FxPair t = tab[k];
    cfloat* pa = buf + t.a + rho;
    cfloat* pb = buf + t.b + rho;
    const cfloat xk = *pa;
    const cfloat xm = *pb;
    if (k == 0) {
      // imaginary parts of the DC and Nyquist bins are ignored, as FFTW's c2r does
      const float xh = P.nyq_packed ? xk.y : P.in_nyq[r0 + rho].x;
      *pa = cmake(xk.x + xh, xk.x - xh);
      *pb = cmake(2.f * xm.x, -2.f * xm.y);  // bin H/2 pairs with itself: Z = 2 conj(X)
    } else {
      const cfloat E = cadd_c(xk, xm);
      const cfloat Dk = csub_c(xk, xm);
      const cfloat O = cmulc(Dk, twr[k]);
      *pa = cadd_i<+1>(E, O);
      *pb = cconj_add_i<+1>(E, O);
    }
  }
}

// c2r last phase: LDS -> last inverse stage in registers -> pointwise epilogue -> either the
// real rows in global memory (KEEP = false) or, for the fused pass, straight into the first
// forward stage of the next transform and back to LDS (KEEP = true)
template <int H, bool KEEP, int EPI>
MVN_HD void fx_c2r_stage0_epilogue(const RowsParams& P, long r0, cfloat* buf, const cfloat* tws,
                                   FxRowsRegs<H>& r, int tid) {
  typedef FxRowsCfg<H> C;
  constexpr int TP = C::TP, NT = C::NT, R = C::R0, M = C::M0;
#pragma unroll
  for (int it = 0; it < C::IT0; ++it) {
    const int w = tid + it * NT;
    if (!C::EXACT && w >= M * C::T) break;
    const int j2 = w % M, rho = w / M;
    cfloat* p = buf + fx_row<C::PAD>(j2) * TP + rho;
    cfloat a[R];
#pragma unroll
    for (int k = 0; k < R; ++k) a[k] = p[fx_rowoff<C::PAD, R, M>(k) * TP];
    cfloat tw[fx_rs(R)];
    fx_tw_fetch<H, 0, C::TWT>(tws, j2, tw);
#pragma unroll
    for (int k = 1; k < R; ++k) a[k] = cmulc(a[k], tw[k]);
    dftR<R, +1>(a);  // a[jo] = z[j2 + M*jo] = (x[2j], x[2j+1])
    const long i0 = (r0 + rho) * P.RP + 2 * j2;
    if (KEEP) {
#pragma unroll
      for (int jo = 0; jo < R; ++jo)
        a[jo] = fx_epilogue_pair_value<EPI>(P.epi, i0 + 2 * jo * M, a[jo], r.ea[it][jo], r.eb[it][jo]);
      dftR<R, -1>(a);
#pragma unroll
      for (int k = 1; k < R; ++k) a[k] = cmul(a[k], tw[k]);
#pragma unroll
      for (int k = 0; k < R; ++k) p[fx_rowoff<C::PAD, R, M>(k) * TP] = a[k];
    } else {
#pragma unroll
      for (int jo = 0; jo < R; ++jo)
        mvn_epilogue_pair_t<EPI>(P.epi, P.out_real, i0 + 2 * jo * M, a[jo], r.ea[it][jo], r.eb[it][jo]);
    }
  }
}

// c2r + pointwise step + r2c in ONE pass over the rows: reads a half-spectrum (and the operands
// of the pointwise step), writes the half-spectrum of the result in place (UPDATE also writes
// psi).  The reference runs cufftExecC2R, a pointwise kernel and cufftExecR2C here
// (inc/gpu_convolve.cuh:140-141 + inc/cuda_kernels.cuh:14-112 + inc/gpu_convolve.cuh:121).
template <int H, int EPI, typename Ctx, bool LINES = false>
MVN_HD void fx_rows_c2r_r2c_body(const RowsParams& P, long tile, cfloat* lds, Ctx& ctx) {
  typedef FxRowsCfg<H> C;
  constexpr int T = C::T, TP = C::TP, NT = C::NT, NT_ = C::NT;
  (void)NT_;
  const long r0 = tile * T;
  cfloat* buf = lds;
  cfloat* tws = lds + C::TILE;
  cfloat* twr = tws + fx_twsize(H);
  MVN_PHASE(ctx, (fx_c2r_load<H, EPI, LINES>(P, r0, buf, tws, twr, r, tid)));
#if !(defined(MVN_EXPERIMENTS) && defined(MVN_EXP_SKIP_PREPOST))  // timing experiment (variant builds only, WRONG results): what two LDS round trips cost
  MVN_PHASE(ctx, (fx_c2r_pre<H>(P, r0, buf, twr, tid)));
#endif
  fx_dit<H, T, TP, C::PAD, NT, +1, 1, C::TWT>(buf, tws, ctx);
  MVN_PHASE(ctx, (fx_c2r_stage0_epilogue<H, true, EPI>(P, r0, buf, tws, r, tid)));
  fx_dif<H, T, TP, C::PAD, NT, -1, 1, C::TWT>(buf, tws, ctx);
#if !(defined(MVN_EXPERIMENTS) && defined(MVN_EXP_SKIP_PREPOST))
  MVN_PHASE(ctx, (fx_r2c_post<H>(P, r0, buf, twr, tid)));
#endif
  MVN_PHASE(ctx, (fx_r2c_store<H, LINES, fx_lines_wide<EPI>()>(P, r0, buf, tid)));
}

template <int H, int EPI, typename Ctx, bool LINES = false>
MVN_HD void fx_rows_c2r_body(const RowsParams& P, long tile, cfloat* lds, Ctx& ctx) {
  typedef FxRowsCfg<H> C;
  constexpr int T = C::T, TP = C::TP, NT = C::NT, NT_ = C::NT;
  (void)NT_;
  const long r0 = tile * T;
  cfloat* buf = lds;
  cfloat* tws = lds + C::TILE;
  cfloat* twr = tws + fx_twsize(H);
  MVN_PHASE(ctx, (fx_c2r_load<H, EPI, LINES>(P, r0, buf, tws, twr, r, tid)));
  MVN_PHASE(ctx, (fx_c2r_pre<H>(P, r0, buf, twr, tid)));
  fx_dit<H, T, TP, C::PAD, NT, +1, 1, C::TWT>(buf, tws, ctx);
  MVN_PHASE(ctx, (fx_c2r_stage0_epilogue<H, false, EPI>(P, r0, buf, tws, r, tid)));
}

// One workgroup of a last-axis launch: its single tile, or (FxRowsCfg<H>::WALK) the tables once
// and then tiles block, block + nblocks, ...  KIND 0: r2c, 1: c2r, 2: c2r + pointwise + r2c.
template <int H, int KIND, int EPI, typename Ctx, bool LINES = false>
MVN_HD void fx_rows_tile(const RowsParams& P, long tile, cfloat* lds, Ctx& ctx) {
  if constexpr (KIND == 0)
    fx_rows_r2c_body<H, Ctx, LINES>(P, tile, lds, ctx);
  else if constexpr (KIND == 1)
    fx_rows_c2r_body<H, EPI, Ctx, LINES>(P, tile, lds, ctx);
  else
    fx_rows_c2r_r2c_body<H, EPI, Ctx, LINES>(P, tile, lds, ctx);
}

// which last-axis lengths have the line-layout forms (one 16-row tile per workgroup)
template <int H>
constexpr bool fx_rows_lines_ok() {
  return H == 256 && !FxRowsCfg<H>::WALK && FxRowsCfg<H>::T == 16 && FxRowsCfg<H>::EXACT;
}

template <int H, int KIND, int EPI, typename Ctx, bool LINES = false>
MVN_HD void fx_rows_run(const RowsParams& P, long block, long nblocks, cfloat* lds, Ctx& ctx) {
  if constexpr (LINES) {
    fx_rows_tile<H, KIND, EPI, Ctx, true>(P, block, lds, ctx);
    return;
  }
  typedef FxRowsCfg<H> C;
  if constexpr (C::WALK) {
    constexpr int NT_ = C::NT;
    (void)NT_;
    cfloat* tws = lds + C::TILE;
    cfloat* twr = tws + fx_twsize(H);
    MVN_PHASE(ctx, (fx_rows_tables<H>(P, tws, twr, tid)));
    const long ntiles = P.rows / C::T;
    for (long tile = block; tile < ntiles; tile += nblocks) fx_rows_tile<H, KIND, EPI>(P, tile, lds, ctx);
  } else {
    fx_rows_tile<H, KIND, EPI>(P, block, lds, ctx);
  }
}
