// mvn_fixed.hpp -- compile-time specialised pass bodies for power-of-two line lengths.
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
#else
template <typename Regs, int NT>
struct FxCtx {
  Regs regs[NT];
};
#define MVN_PHASE(ctx, ...)                  \
  for (int tid = 0; tid < NT_; ++tid) {      \
    auto& r = (ctx).regs[tid];               \
    (void)r;                                 \
    __VA_ARGS__;                             \
  }
#endif

// ---------------------------------------------------------------------------------------------
// compile-time plan of a power-of-two length: same radix order as AxisPlanHost::factorize
// ---------------------------------------------------------------------------------------------
constexpr bool fx_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

constexpr int fx_nstages(int n) {
  int c = 0;
  while (n % 8 == 0) { n /= 8; ++c; }
  while (n % 4 == 0) { n /= 4; ++c; }
  while (n % 2 == 0) { n /= 2; ++c; }
  return c;
}

constexpr int fx_radix(int n, int s) {
  int c = 0;
  while (n % 8 == 0) { if (c == s) return 8; n /= 8; ++c; }
  while (n % 4 == 0) { if (c == s) return 4; n /= 4; ++c; }
  while (n % 2 == 0) { if (c == s) return 2; n /= 2; ++c; }
  return 1;
}

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

// stage-ordered twiddle table: for every stage with M > 1 a block of M rows of R entries,
// row j2 = { exp(-2 pi i j2 k / (R M)) : k = 0..R-1 }
constexpr int fx_twoff(int n, int s) {
  int off = 0;
  for (int t = 0; t < s; ++t)
    if (fx_M(n, t) > 1) off += fx_M(n, t) * fx_radix(n, t);
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
template <int N, int T, int TP, bool PAD, int NT, int S, int SIGN, bool DIF>
MVN_HD void fx_stage(cfloat* buf, const cfloat* tws, int tid) {
  constexpr int R = fx_radix(N, S), M = fx_M(N, S);
  constexpr int nwork = (N / R) * T;
  constexpr int iters = (nwork + NT - 1) / NT;
#pragma unroll
  for (int it = 0; it < iters; ++it) {
    const int w = tid + it * NT;
    if (nwork % NT != 0 && w >= nwork) break;
    const int b = w / T, c = w % T;
    const int blk = b / M, j2 = b % M;
    cfloat* p = buf + fx_row<PAD>(blk * R * M + j2) * TP + c;
    cfloat a[R];
#pragma unroll
    for (int j = 0; j < R; ++j) a[j] = p[fx_rowoff<PAD, R, M>(j) * TP];
    cfloat tw[R];
    if (M > 1) {
      const qfloat* t4 = reinterpret_cast<const qfloat*>(tws + fx_twoff(N, S) + j2 * R);
#pragma unroll
      for (int k = 0; k < R / 2; ++k) {
        const qfloat t = t4[k];
        tw[2 * k] = cmake(t.x, t.y);
        tw[2 * k + 1] = cmake(t.z, t.w);
      }
    }
    if (!DIF && M > 1) {
#pragma unroll
      for (int k = 1; k < R; ++k) a[k] = cmul(a[k], twdir<SIGN>(tw[k]));
    }
    dftR<R, SIGN>(a);
    if (DIF && M > 1) {
#pragma unroll
      for (int k = 1; k < R; ++k) a[k] = cmul(a[k], twdir<SIGN>(tw[k]));
    }
#pragma unroll
    for (int j = 0; j < R; ++j) p[fx_rowoff<PAD, R, M>(j) * TP] = a[j];
  }
}

// all stages of a transform as phases (DIF: S = 0..ns-1, DIT: ns-1..0)
template <int N, int T, int TP, bool PAD, int NT, int SIGN, bool DIF, int S, typename Ctx>
struct FxStages {
  static MVN_HD void run(cfloat* buf, const cfloat* tws, Ctx& ctx) {
    constexpr int NT_ = NT;
    (void)NT_;
    MVN_PHASE(ctx, (fx_stage<N, T, TP, PAD, NT, S, SIGN, DIF>(buf, tws, tid)));
    constexpr int next = DIF ? S + 1 : S - 1;
    if constexpr (next >= 0 && next < fx_nstages(N))
      FxStages<N, T, TP, PAD, NT, SIGN, DIF, next, Ctx>::run(buf, tws, ctx);
  }
};

template <int N, int T, int TP, bool PAD, int NT, int SIGN, typename Ctx>
MVN_HD void fx_dif(cfloat* buf, const cfloat* tws, Ctx& ctx) {
  FxStages<N, T, TP, PAD, NT, SIGN, true, 0, Ctx>::run(buf, tws, ctx);
}
template <int N, int T, int TP, bool PAD, int NT, int SIGN, typename Ctx>
MVN_HD void fx_dit(cfloat* buf, const cfloat* tws, Ctx& ctx) {
  FxStages<N, T, TP, PAD, NT, SIGN, false, fx_nstages(N) - 1, Ctx>::run(buf, tws, ctx);
}

template <int NT>
MVN_HD void fx_copy_table(cfloat* dst, const cfloat* src, int count, int tid) {
  for (int i = tid; i < count; i += NT) dst[i] = src[i];
}

// ---------------------------------------------------------------------------------------------
// strided-axis pass, full tiles of T neighbouring bins (cstride == 1, ncols % T == 0)
// ---------------------------------------------------------------------------------------------
template <int N>
struct FxStridedCfg {
  static constexpr int T = N <= 512 ? 16 : 8;
  static constexpr int TP = T;
  static constexpr int CH = T / 2;  // 16-byte chunks per tile row
  static constexpr int NTfull = N * T / 8;
  static constexpr int NT = NTfull >= 512 ? 512 : (NTfull >= 64 ? NTfull : 64);
  static constexpr int RPT = NT / CH;      // tile rows covered by one sweep of the workgroup
  static constexpr int U = N / RPT;        // 16-byte loads per thread
  static constexpr int lds_cfloats = N * TP + fx_twsize(N);
  static_assert(fx_pow2(N) && N >= 64 && N <= 1024, "unsupported fixed length");
  static_assert(N % RPT == 0, "tile rows must divide");
};

template <int N>
struct FxStridedRegs {
  qfloat v[FxStridedCfg<N>::U];
  qfloat g[FxStridedCfg<N>::U];
};

// phase functions (plain functions so that loop pragmas are honoured; MVN_PHASE only calls them)
template <int N, int MODE>
MVN_HD void fx_st_load(const StridedParams& P, long base, cfloat* buf, cfloat* tws,
                       FxStridedRegs<N>& r, int tid) {
  typedef FxStridedCfg<N> C;
  constexpr int TP = C::TP, U = C::U;
  const long rstep = (long)C::RPT * P.estride;
  const int q = tid % C::CH, jr = tid / C::CH;
  const cfloat* src = (P.src ? P.src : P.data) + base + (long)jr * P.estride + 2 * q;
#pragma unroll
  for (int u = 0; u < U; ++u) r.v[u] = *reinterpret_cast<const qfloat*>(src + u * rstep);
  if (MODE == MVN_ST_FWD_MUL_INV) {
    // PSF-spectrum operands for the rows this thread multiplies later (the spectrum is stored
    // in the same digit-reversed row order the forward transform produces); fetched now, used
    // after the forward transform
    const cfloat* sp = P.spec + base + (long)jr * P.estride + 2 * q;
#pragma unroll
    for (int u = 0; u < U; ++u) r.g[u] = *reinterpret_cast<const qfloat*>(sp + u * rstep);
  }
  fx_copy_table<C::NT>(tws, P.ax.tws, fx_twsize(N), tid);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int j = jr + u * C::RPT;
    *reinterpret_cast<qfloat*>(buf + j * TP + 2 * q) = r.v[u];
  }
}

template <int N>
MVN_HD void fx_st_mul(cfloat* buf, FxStridedRegs<N>& r, int tid) {
  typedef FxStridedCfg<N> C;
  const int q = tid % C::CH, jr = tid / C::CH;
#pragma unroll
  for (int u = 0; u < C::U; ++u) {
    qfloat* d = reinterpret_cast<qfloat*>(buf + (jr + u * C::RPT) * C::TP + 2 * q);
    const qfloat a = *d;
    const qfloat g = r.g[u];
    *d = qmake(a.x * g.x - a.y * g.y, a.x * g.y + a.y * g.x, a.z * g.z - a.w * g.w,
               a.z * g.w + a.w * g.z);
  }
}

template <int N, int MODE>
MVN_HD void fx_st_store(const StridedParams& P, long base, const cfloat* buf, int tid) {
  typedef FxStridedCfg<N> C;
  const int q = tid % C::CH, jr = tid / C::CH;
  const long rstep = (long)C::RPT * P.estride;
  cfloat* dst = P.data + base + (long)jr * P.estride + 2 * q;
#pragma unroll
  for (int u = 0; u < C::U; ++u)
    *reinterpret_cast<qfloat*>(dst + u * rstep) =
        *reinterpret_cast<const qfloat*>(buf + (jr + u * C::RPT) * C::TP + 2 * q);
}

template <int N, int MODE, typename Ctx>
MVN_HD void fx_strided_body(const StridedParams& P, long block, cfloat* lds, Ctx& ctx) {
  typedef FxStridedCfg<N> C;
  constexpr int T = C::T, TP = C::TP, NT = C::NT, NT_ = C::NT;
  (void)NT_;
  const long o = block / P.tiles_per_outer;
  const int t = (int)(block - o * P.tiles_per_outer);
  const long base = o * P.ostride + (long)t * T;
  cfloat* buf = lds;
  cfloat* tws = lds + N * TP;
  MVN_PHASE(ctx, (fx_st_load<N, MODE>(P, base, buf, tws, r, tid)));
  if (MODE == MVN_ST_INV) {
    fx_dit<N, T, TP, false, NT, +1>(buf, tws, ctx);
  } else {
    fx_dif<N, T, TP, false, NT, -1>(buf, tws, ctx);
    if (MODE == MVN_ST_FWD_MUL_INV) {
      MVN_PHASE(ctx, (fx_st_mul<N>(buf, r, tid)));
      fx_dit<N, T, TP, false, NT, +1>(buf, tws, ctx);
    }
  }
  MVN_PHASE(ctx, (fx_st_store<N, MODE>(P, base, buf, tid)));
}

// ---------------------------------------------------------------------------------------------
// last-axis passes for even d2 = 2H, H a power of two; T rows per tile, transposed in LDS with
// an odd pitch and one spare row per 32 rows.  Requires rows % T == 0.
// ---------------------------------------------------------------------------------------------
#ifndef MVN_FX_ROWS_T
#define MVN_FX_ROWS_T 16
#endif
template <int H>
struct FxRowsCfg {
  static constexpr int T = H <= 512 ? MVN_FX_ROWS_T : 8;
  static constexpr int TP = T + 1;
  static constexpr int QR = H / 2;  // 16-byte chunks per row (2 complex bins = 4 reals each)
  static constexpr int NTfull = H * T / 8;
  static constexpr int NT = NTfull >= 512 ? 512 : (NTfull >= 64 ? NTfull : 64);
  static constexpr int U = T * QR / NT;  // 16-byte loads per thread
  static constexpr int ROWS = fx_rows_alloc(H, true);
  static constexpr int TILE = (ROWS * TP + 1) & ~1;  // keeps the tables behind it 16-byte aligned
  static constexpr int lds_cfloats = TILE + fx_twsize(H) + (H / 2 + 2);
  static_assert(fx_pow2(H) && H >= 32 && H <= 1024, "unsupported fixed length");
  static_assert((T * QR) % NT == 0, "tile must divide");
};

template <int H>
struct FxRowsRegs {
  qfloat v[FxRowsCfg<H>::U];
  qfloat ea[FxRowsCfg<H>::U];
  qfloat eb[FxRowsCfg<H>::U];
};

template <int H>
MVN_HD void fx_r2c_load(const RowsParams& P, long r0, cfloat* buf, cfloat* tws, cfloat* twr,
                        FxRowsRegs<H>& r, int tid) {
  typedef FxRowsCfg<H> C;
  constexpr int TP = C::TP, NT = C::NT, U = C::U;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int e = tid + u * NT;
    const int rho = e / C::QR, jj = e % C::QR;
    r.v[u] = reinterpret_cast<const qfloat*>(P.in_real + (r0 + rho) * P.RP)[jj];
  }
  fx_copy_table<NT>(tws, P.ax.tws, fx_twsize(H), tid);
  fx_copy_table<NT>(twr, P.twr, H / 2 + 1, tid);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int e = tid + u * NT;
    const int rho = e / C::QR, jj = e % C::QR;
    buf[fx_row<true>(2 * jj) * TP + rho] = cmake(r.v[u].x, r.v[u].y);
    buf[fx_row<true>(2 * jj + 1) * TP + rho] = cmake(r.v[u].z, r.v[u].w);
  }
}

template <int H>
MVN_HD void fx_r2c_post(const RowsParams& P, long r0, cfloat* buf, const cfloat* twr, int tid) {
  typedef FxRowsCfg<H> C;
  constexpr int T = C::T, TP = C::TP, NT = C::NT;
  constexpr int npairs = H / 2 + 1;
  for (int w = tid; w < npairs * T; w += NT) {
    const int k = w / T, rho = w % T;
    const int m = (H - k) & (H - 1);
    const int pk = fx_row<true>(fx_inv<H>(k)), pm = fx_row<true>(fx_inv<H>(m));
    const cfloat zk = buf[pk * TP + rho];
    if (k == 0) {
      buf[pk * TP + rho] = cmake(zk.x + zk.y, 0.f);
      P.out_nyq[r0 + rho] = cmake(zk.x - zk.y, 0.f);
    } else {
      const cfloat zm = buf[pm * TP + rho];
      const cfloat E = cmake(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
      const cfloat D = cmake(0.5f * (zk.x - zm.x), 0.5f * (zk.y + zm.y));
      const cfloat G = cmul(twr[k], D);
      buf[pk * TP + rho] = cadd(E, cmul_si<-1>(G));
      if (m != k) buf[pm * TP + rho] = cadd(cconj(E), cmul_si<-1>(cconj(G)));
    }
  }
}

template <int H>
MVN_HD void fx_r2c_store(const RowsParams& P, long r0, const cfloat* buf, int tid) {
  typedef FxRowsCfg<H> C;
  constexpr int TP = C::TP, NT = C::NT, U = C::U;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int e = tid + u * NT;
    const int rho = e / C::QR, kk = e % C::QR;
    // spectral rows are kept in position (digit-reversed) order: bin k sits at column inv(k)
    const cfloat a = buf[fx_row<true>(2 * kk) * TP + rho];
    const cfloat b = buf[fx_row<true>(2 * kk + 1) * TP + rho];
    reinterpret_cast<qfloat*>(P.out_cplx + (r0 + rho) * P.C)[kk] = qmake(a.x, a.y, b.x, b.y);
  }
}

template <int H, typename Ctx>
MVN_HD void fx_rows_r2c_body(const RowsParams& P, long tile, cfloat* lds, Ctx& ctx) {
  typedef FxRowsCfg<H> C;
  constexpr int T = C::T, TP = C::TP, NT = C::NT, NT_ = C::NT;
  (void)NT_;
  const long r0 = tile * T;
  cfloat* buf = lds;
  cfloat* tws = lds + C::TILE;
  cfloat* twr = tws + fx_twsize(H);
  MVN_PHASE(ctx, (fx_r2c_load<H>(P, r0, buf, tws, twr, r, tid)));
  fx_dif<H, T, TP, true, NT, -1>(buf, tws, ctx);
  MVN_PHASE(ctx, (fx_r2c_post<H>(P, r0, buf, twr, tid)));
  MVN_PHASE(ctx, (fx_r2c_store<H>(P, r0, buf, tid)));
}

// epilogue on four consecutive reals (one 16-byte unit); operands were fetched in phase 0
MVN_HD float fx_blend(float w, float next, float last) {
  MVN_FP_EXACT
  return w * (next - last) + last;
}
MVN_HD float fx_delta(float w, float next, float last) {
  MVN_FP_EXACT
  return w * (next - last);
}

// Same arithmetic, but hands the four results back instead of (DIVIDE) / in addition to (UPDATE)
// storing them: they are the input of the forward transform fused behind this pass.
MVN_HD qfloat fx_epilogue_quad_value(const EpilogueParams& e, long i, qfloat x, qfloat a, qfloat b) {
  MVN_FP_EXACT
  x = qmake(x.x * e.scale, x.y * e.scale, x.z * e.scale, x.w * e.scale);
  if (e.mode == MVN_EPI_DIVIDE)
    return qmake(mvn_quotient(a.x, x.x), mvn_quotient(a.y, x.y), mvn_quotient(a.z, x.z),
                 mvn_quotient(a.w, x.w));
  if (e.mode == MVN_EPI_UPDATE) {
    const float n0 = mvn_next_value(a.x, x.x, e.lambda, e.lambda_inv, e.min_value);
    const float n1 = mvn_next_value(a.y, x.y, e.lambda, e.lambda_inv, e.min_value);
    const float n2 = mvn_next_value(a.z, x.z, e.lambda, e.lambda_inv, e.min_value);
    const float n3 = mvn_next_value(a.w, x.w, e.lambda, e.lambda_inv, e.min_value);
    const qfloat y = qmake(fx_blend(b.x, n0, a.x), fx_blend(b.y, n1, a.y), fx_blend(b.z, n2, a.z),
                           fx_blend(b.w, n3, a.w));
    *reinterpret_cast<qfloat*>(e.psi + i) = y;
    return y;
  }
  return x;  // STORE: plain inverse followed by a forward transform
}

MVN_HD void fx_epilogue_quad(const EpilogueParams& e, float* out, long i, qfloat x, qfloat a,
                             qfloat b) {
  MVN_FP_EXACT
  x = qmake(x.x * e.scale, x.y * e.scale, x.z * e.scale, x.w * e.scale);
  switch (e.mode) {
    case MVN_EPI_STORE: *reinterpret_cast<qfloat*>(out + i) = x; break;
    case MVN_EPI_DIVIDE:
      *reinterpret_cast<qfloat*>(out + i) = qmake(mvn_quotient(a.x, x.x), mvn_quotient(a.y, x.y),
                                                  mvn_quotient(a.z, x.z), mvn_quotient(a.w, x.w));
      break;
    case MVN_EPI_UPDATE: {
      const float n0 = mvn_next_value(a.x, x.x, e.lambda, e.lambda_inv, e.min_value);
      const float n1 = mvn_next_value(a.y, x.y, e.lambda, e.lambda_inv, e.min_value);
      const float n2 = mvn_next_value(a.z, x.z, e.lambda, e.lambda_inv, e.min_value);
      const float n3 = mvn_next_value(a.w, x.w, e.lambda, e.lambda_inv, e.min_value);
      *reinterpret_cast<qfloat*>(e.psi + i) = qmake(fx_blend(b.x, n0, a.x), fx_blend(b.y, n1, a.y),
                                                    fx_blend(b.z, n2, a.z), fx_blend(b.w, n3, a.w));
    } break;
    case MVN_EPI_DELTA: {
      const float n0 = mvn_next_value(a.x, x.x, e.lambda, e.lambda_inv, e.min_value);
      const float n1 = mvn_next_value(a.y, x.y, e.lambda, e.lambda_inv, e.min_value);
      const float n2 = mvn_next_value(a.z, x.z, e.lambda, e.lambda_inv, e.min_value);
      const float n3 = mvn_next_value(a.w, x.w, e.lambda, e.lambda_inv, e.min_value);
      qfloat d = qmake(fx_delta(b.x, n0, a.x), fx_delta(b.y, n1, a.y), fx_delta(b.z, n2, a.z),
                       fx_delta(b.w, n3, a.w));
      if (e.accumulate) {
        const qfloat old = *reinterpret_cast<const qfloat*>(e.delta + i);
        d = qmake(old.x + d.x, old.y + d.y, old.z + d.z, old.w + d.w);
      }
      *reinterpret_cast<qfloat*>(e.delta + i) = d;
    } break;
  }
}

template <int H>
MVN_HD void fx_c2r_load(const RowsParams& P, long r0, cfloat* buf, cfloat* tws, cfloat* twr,
                        FxRowsRegs<H>& r, int tid) {
  typedef FxRowsCfg<H> C;
  constexpr int TP = C::TP, NT = C::NT, U = C::U;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int e = tid + u * NT;
    const int rho = e / C::QR, kk = e % C::QR;
    r.v[u] = reinterpret_cast<const qfloat*>(P.in_cplx + (r0 + rho) * P.C)[kk];
  }
  // epilogue operands, fetched a whole transform ahead of their use; the (uniform) mode branch
  // stays outside the unrolled loops so that each branch is straight-line code
  // Two independent guarded loops, each filling ONE array: if both arrays were written in
  // sibling branches LLVM sinks the stores into a common block with a selected address and the
  // register arrays fall back to scratch.
  const int mode = P.epi.mode;
  const float* pa = mode == MVN_EPI_DIVIDE ? P.epi.view : P.epi.psi;
  if (mode != MVN_EPI_STORE) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = tid + u * NT;
      r.ea[u] = *reinterpret_cast<const qfloat*>(pa + (r0 + e / C::QR) * P.RP + 4 * (e % C::QR));
    }
  }
  if (mode == MVN_EPI_UPDATE || mode == MVN_EPI_DELTA) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = tid + u * NT;
      r.eb[u] = *reinterpret_cast<const qfloat*>(P.epi.weights + (r0 + e / C::QR) * P.RP + 4 * (e % C::QR));
    }
  }
  fx_copy_table<NT>(tws, P.ax.tws, fx_twsize(H), tid);
  fx_copy_table<NT>(twr, P.twr, H / 2 + 1, tid);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int e = tid + u * NT;
    const int rho = e / C::QR, kk = e % C::QR;
    buf[fx_row<true>(2 * kk) * TP + rho] = cmake(r.v[u].x, r.v[u].y);
    buf[fx_row<true>(2 * kk + 1) * TP + rho] = cmake(r.v[u].z, r.v[u].w);
  }
}

template <int H>
MVN_HD void fx_c2r_pre(const RowsParams& P, long r0, cfloat* buf, const cfloat* twr, int tid) {
  typedef FxRowsCfg<H> C;
  constexpr int T = C::T, TP = C::TP, NT = C::NT;
  constexpr int npairs = H / 2 + 1;
  for (int w = tid; w < npairs * T; w += NT) {
    const int k = w / T, rho = w % T;
    const int m = (H - k) & (H - 1);
    const int pk = fx_row<true>(fx_inv<H>(k)), pm = fx_row<true>(fx_inv<H>(m));
    const cfloat xk = buf[pk * TP + rho];
    if (k == 0) {
      // imaginary parts of the DC and Nyquist bins are ignored, as FFTW's c2r does
      const float xh = P.in_nyq[r0 + rho].x;
      buf[pk * TP + rho] = cmake(xk.x + xh, xk.x - xh);
    } else {
      const cfloat xm = buf[pm * TP + rho];
      const cfloat E = cmake(xk.x + xm.x, xk.y - xm.y);
      const cfloat Dk = cmake(xk.x - xm.x, xk.y + xm.y);
      const cfloat O = cmul(Dk, cconj(twr[k]));
      buf[pk * TP + rho] = cadd(E, cmul_si<+1>(O));
      if (m != k) buf[pm * TP + rho] = cadd(cconj(E), cmul_si<+1>(cconj(O)));
    }
  }
}

template <int H>
MVN_HD void fx_c2r_epi(const RowsParams& P, long r0, const cfloat* buf, FxRowsRegs<H>& r, int tid) {
  typedef FxRowsCfg<H> C;
  constexpr int TP = C::TP, NT = C::NT, U = C::U;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int e = tid + u * NT;
    const int rho = e / C::QR, jj = e % C::QR;
    const cfloat z0 = buf[fx_row<true>(2 * jj) * TP + rho];
    const cfloat z1 = buf[fx_row<true>(2 * jj + 1) * TP + rho];
    fx_epilogue_quad(P.epi, P.out_real, (r0 + rho) * P.RP + 4 * jj, qmake(z0.x, z0.y, z1.x, z1.y),
                     r.ea[u], r.eb[u]);
  }
}

// fused form: the epilogue results stay in LDS as the packed input z[j] = y[2j] + i y[2j+1] of
// the NEXT convolution's forward last-axis transform
template <int H>
MVN_HD void fx_c2r_epi_keep(const RowsParams& P, long r0, cfloat* buf, FxRowsRegs<H>& r, int tid) {
  typedef FxRowsCfg<H> C;
  constexpr int TP = C::TP, NT = C::NT, U = C::U;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int e = tid + u * NT;
    const int rho = e / C::QR, jj = e % C::QR;
    cfloat* p0 = buf + fx_row<true>(2 * jj) * TP + rho;
    cfloat* p1 = buf + fx_row<true>(2 * jj + 1) * TP + rho;
    const cfloat z0 = *p0;
    const cfloat z1 = *p1;
    const qfloat y = fx_epilogue_quad_value(P.epi, (r0 + rho) * P.RP + 4 * jj,
                                            qmake(z0.x, z0.y, z1.x, z1.y), r.ea[u], r.eb[u]);
    *p0 = cmake(y.x, y.y);
    *p1 = cmake(y.z, y.w);
  }
}

// c2r + pointwise step + r2c in ONE pass over the rows: reads a half-spectrum (and the operands
// of the pointwise step), writes the half-spectrum of the result in place (UPDATE also writes
// psi).  The reference runs cufftExecC2R, a pointwise kernel and cufftExecR2C here
// (inc/gpu_convolve.cuh:140-141 + inc/cuda_kernels.cuh:14-112 + inc/gpu_convolve.cuh:121).
template <int H, typename Ctx>
MVN_HD void fx_rows_c2r_r2c_body(const RowsParams& P, long tile, cfloat* lds, Ctx& ctx) {
  typedef FxRowsCfg<H> C;
  constexpr int T = C::T, TP = C::TP, NT = C::NT, NT_ = C::NT;
  (void)NT_;
  const long r0 = tile * T;
  cfloat* buf = lds;
  cfloat* tws = lds + C::TILE;
  cfloat* twr = tws + fx_twsize(H);
  MVN_PHASE(ctx, (fx_c2r_load<H>(P, r0, buf, tws, twr, r, tid)));
  MVN_PHASE(ctx, (fx_c2r_pre<H>(P, r0, buf, twr, tid)));
  fx_dit<H, T, TP, true, NT, +1>(buf, tws, ctx);
  MVN_PHASE(ctx, (fx_c2r_epi_keep<H>(P, r0, buf, r, tid)));
  fx_dif<H, T, TP, true, NT, -1>(buf, tws, ctx);
  MVN_PHASE(ctx, (fx_r2c_post<H>(P, r0, buf, twr, tid)));
  MVN_PHASE(ctx, (fx_r2c_store<H>(P, r0, buf, tid)));
}

template <int H, typename Ctx>
MVN_HD void fx_rows_c2r_body(const RowsParams& P, long tile, cfloat* lds, Ctx& ctx) {
  typedef FxRowsCfg<H> C;
  constexpr int T = C::T, TP = C::TP, NT = C::NT, NT_ = C::NT;
  (void)NT_;
  const long r0 = tile * T;
  cfloat* buf = lds;
  cfloat* tws = lds + C::TILE;
  cfloat* twr = tws + fx_twsize(H);
  MVN_PHASE(ctx, (fx_c2r_load<H>(P, r0, buf, tws, twr, r, tid)));
  MVN_PHASE(ctx, (fx_c2r_pre<H>(P, r0, buf, twr, tid)));
  fx_dit<H, T, TP, true, NT, +1>(buf, tws, ctx);
  MVN_PHASE(ctx, (fx_c2r_epi<H>(P, r0, buf, r, tid)));
}
