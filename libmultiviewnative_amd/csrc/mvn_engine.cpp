// mvn_engine.cpp -- see mvn_engine.hpp
#include "mvn_engine.hpp"

#include <atomic>
#include <cstdlib>
#include <cstring>

#include "mvn_fixed_geom.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace mvn {

static const size_t kLdsSoftBudget = 80 * 1024;   // two workgroups per CU (160 KiB LDS)
static const size_t kLdsHardBudget = 160 * 1024;  // one workgroup may own the whole CU

const char* kernel_kind_name(int k) {
  static const char* names[KK_COUNT] = {"rows_r2c",    "rows_c2r",  "rows_fused_div",
                                        "rows_fused_upd", "axis1_fwd", "axis1_inv",
                                        "axis0_fused", "axis0_fwd",  "axis0_inv",
                                        "nyquist",     "other",     "axis0_direct", "mid_fused"};
  return (k >= 0 && k < KK_COUNT) ? names[k] : "?";
}

// ---------------------------------------------------------------------------------------------
// Profiler
// ---------------------------------------------------------------------------------------------
be::event_t Profiler::get_event() {
  if (!pool_.empty()) {
    be::event_t e = pool_.back();
    pool_.pop_back();
    return e;
  }
  return be::event_create();
}

void Profiler::begin(int kind, be::stream_t s) {
  if (!enabled) return;
  Rec r;
  r.kind = kind;
  r.a = get_event();
  r.b = get_event();
  be::event_record(r.a, s);
  recs_.push_back(r);
}

void Profiler::end(be::stream_t s) {
  if (!enabled || recs_.empty()) return;
  be::event_record(recs_.back().b, s);
}

void Profiler::collect() {
  for (size_t i = 0; i < recs_.size(); ++i) {
    be::event_sync(recs_[i].b);
    total_ms[recs_[i].kind] += be::event_elapsed_ms(recs_[i].a, recs_[i].b);
    count[recs_[i].kind] += 1;
    pool_.push_back(recs_[i].a);
    pool_.push_back(recs_[i].b);
  }
  recs_.clear();
}

void Profiler::reset() {
  collect();
  for (int k = 0; k < KK_COUNT; ++k) {
    total_ms[k] = 0;
    count[k] = 0;
  }
}

Profiler::~Profiler() {
  try {
    collect();
  } catch (...) {
  }
  for (size_t i = 0; i < pool_.size(); ++i) be::event_destroy(pool_[i]);
}

struct ProfScope {
  Profiler* p;
  be::stream_t s;
  ProfScope(Profiler* p_, int kind, be::stream_t s_) : p(p_), s(s_) {
    if (p) p->begin(kind, s);
  }
  ~ProfScope() {
    if (p) p->end(s);
  }
};

// ---------------------------------------------------------------------------------------------
// DevAxis / Plan3D
// ---------------------------------------------------------------------------------------------
DevAxis::DevAxis(int n, bool composite) : host(n, composite) {
  auto upload = [](const void* src, size_t bytes) -> void* {
    void* d = be::dmalloc(bytes);
    be::h2d(d, src, bytes, nullptr);
    return d;
  };
  tw = (cfloat*)upload(host.tw.data(), sizeof(cfloat) * host.tw.size());  // nfft entries
  rev = (int*)upload(host.rev.data(), sizeof(int) * host.rev.size());
  inv = (int*)upload(host.inv.data(), sizeof(int) * host.inv.size());
  if (!host.tws.empty()) tws = (cfloat*)upload(host.tws.data(), sizeof(cfloat) * host.tws.size());
  if (host.bluestein) {
    chirp = (cfloat*)upload(host.chirp.data(), sizeof(cfloat) * host.chirp.size());
    bhat = (cfloat*)upload(host.bhat.data(), sizeof(cfloat) * host.bhat.size());
  }
  be::stream_sync(nullptr);
  view = host.view(tw, rev, inv, tws, chirp, bhat);
}

DevAxis::~DevAxis() {
  be::dfree(tw);
  be::dfree(tws);
  be::dfree(rev);
  be::dfree(inv);
  be::dfree(chirp);
  be::dfree(bhat);
}

// tuning knobs for experiments (not part of the ABI): MVN_T_ROWS / MVN_T_AXIS / MVN_T_FUSED cap
// the tile width of the three pass families, MVN_THREADS fixes the workgroup size
static int env_int(const char* name, int dflt) {
  const char* v = std::getenv(name);
  return (v && *v) ? std::atoi(v) : dflt;
}

PassGeom Plan3D::pick_geom(int n, bool generic, bool rows, int max_t) {
  static const int cand[5] = {16, 8, 4, 2, 1};
  PassGeom g;
  bool found = false;
  for (int pass = 0; pass < 2 && !found; ++pass) {
    size_t budget = pass == 0 ? kLdsSoftBudget : kLdsHardBudget;
    if (pass == 0 && env_int("MVN_LDS_SOFT_KB", 0) > 0) budget = (size_t)env_int("MVN_LDS_SOFT_KB", 0) * 1024;
    for (int i = 0; i < 5; ++i) {
      const int T = cand[i];
      if (T > max_t) continue;
      const int TP = rows ? (T | 1) : T;  // odd pitch keeps the row transpose conflict-free
      const size_t one = (size_t)n * (size_t)TP * sizeof(cfloat);
      const size_t twb = (size_t)n * sizeof(cfloat);  // LDS copy of the twiddle table
      const size_t bytes = one * (generic ? 2 : 1) + twb;
      if (bytes <= budget) {
        g.T = T;
        g.TP = TP;
        g.lds_bytes = bytes;
        g.lds_alt = generic ? (long)n * TP : 0;
        g.lds_tw = (long)n * TP * (generic ? 2 : 1);
        found = true;
        break;
      }
    }
  }
  if (!found)
    throw std::invalid_argument("mvn: axis length " + std::to_string(n) +
                                " does not fit the single-pass LDS FFT (160 KiB)");
  // long strided axes: 32-byte row segments (T = 4) waste HBM bursts; one workgroup per CU with
  // 64-byte segments is the better trade (measured at n = 1920: 5.2 -> 4.1 ms per pass)
  if (!rows && g.T < 8 && max_t >= 8) {
    const size_t one = (size_t)n * 8 * sizeof(cfloat);
    const size_t bytes = one * (generic ? 2 : 1) + (size_t)n * sizeof(cfloat);
    if (bytes <= kLdsHardBudget - 4096) {
      g.T = g.TP = 8;
      g.lds_bytes = bytes;
      g.lds_alt = generic ? (long)n * 8 : 0;
      g.lds_tw = (long)n * 8 * (generic ? 2 : 1);
    }
  }
  // one radix-8 butterfly per thread and stage when the tile is big enough
  const long work = (long)n * g.T / 8;
  g.threads = work >= 512 ? 512 : (work >= 256 ? 256 : (work >= 128 ? 128 : 64));
  const int forced = env_int("MVN_THREADS", 0);
  if (forced >= 64 && forced <= 512 && forced % 64 == 0) g.threads = forced;
  return g;
}

// which axes run the fixed-length kernels: known from the extents alone, and needed before the
// axis plans are built (the fixed kernels have their own radix schedule, see AxisPlanHost::factorize)
static bool fixed_allowed() { return env_int("MVN_NO_FIXED", 0) == 0; }
static bool rows_fixed(const Layout& L) {
  int T = 0, threads = 0;
  size_t lds = 0;
  return fixed_allowed() && L.even && fixed_rows_geom(L.h, &T, &threads, &lds) && L.rows % (size_t)T == 0;
}
static bool ax1_fixed(const Layout& L) {
  int T = 0, threads = 0;
  size_t lds = 0;
  return fixed_allowed() && fixed_strided_geom(L.d1, &T, &threads, &lds) && L.C % T == 0;
}
static bool ax0_fixed(const Layout& L) {
  int T = 0, threads = 0;
  size_t lds = 0;
  return fixed_allowed() && fixed_strided_geom(L.d0, &T, &threads, &lds) && ((long)L.d1 * L.C) % T == 0;
}

Plan3D::Plan3D(int dev, int d0, int d1, int d2)
    : device(dev), L(d0, d1, d2), ax2(L.h, rows_fixed(L)), ax1(d1, ax1_fixed(L)), ax0(d0, ax0_fixed(L)) {
  // the word an epilogue that follows no direct dim0 leg compares with (EpilogueParams::poison is never null
  // on the device): zero, against the epoch 0xffffffff no leg ever has
  no_poison = (unsigned*)be::dmalloc(64);
  be::dzero(no_poison, 64, nullptr);
  be::stream_sync(nullptr);
  if (L.even) {
    std::vector<cfloat> roots((size_t)L.h / 2 + 1);
    for (size_t k = 0; k < roots.size(); ++k) {
      double a = -2.0 * M_PI * (double)k / (double)d2;
      roots[k].x = (float)std::cos(a);
      roots[k].y = (float)std::sin(a);
    }
    twr = (cfloat*)be::dmalloc(sizeof(cfloat) * roots.size());
    be::h2d(twr, roots.data(), sizeof(cfloat) * roots.size(), nullptr);
    be::stream_sync(nullptr);
  }
  // LDS rows per tile = length of the radix transform (the chirp-z length for bluestein axes)
  g_rows = pick_geom(ax2.host.nfft, ax2.host.generic, true, env_int("MVN_T_ROWS", 16));
  g_ax1 = pick_geom(ax1.host.nfft, ax1.host.generic, false, env_int("MVN_T_AXIS", 16));
  g_ax0 = pick_geom(ax0.host.nfft, ax0.host.generic, false, env_int("MVN_T_AXIS", 16));
  // the fused pass has its own geometry so that it can be tuned apart (measured on MI355X at
  // 512^3: T=16 0.46 ms, T=8 0.63 ms)
  g_ax0f = pick_geom(ax0.host.nfft, ax0.host.generic, false, env_int("MVN_T_FUSED", 16));
  g_nyq1 = g_ax1;
  g_nyq0 = g_ax0;
  g_nyq1_line = pick_geom(ax1.host.nfft, ax1.host.generic, false, 1);  // one line per workgroup (riders)
  // fixed-length fast path: full tiles, 16-byte aligned rows
  {
    int T = 0, threads = 0;
    size_t lds = 0;
    if (rows_fixed(L)) {
      fixed_rows_geom(L.h, &T, &threads, &lds);
      fx_rows = true;
      gx_rows.T = T;
      gx_rows.threads = threads;
      gx_rows.lds_bytes = lds;
    }
    if (ax1_fixed(L)) {
      fixed_strided_geom(d1, &T, &threads, &lds);
      fx_ax1 = true;
      gx_ax1.T = gx_ax1.TP = T;
      gx_ax1.threads = threads;
      gx_ax1.lds_bytes = lds;
    }
    if (ax0_fixed(L)) {
      fixed_strided_geom(d0, &T, &threads, &lds);
      fx_ax0 = true;
      gx_ax0.T = gx_ax0.TP = T;
      gx_ax0.threads = threads;
      gx_ax0.lds_bytes = lds;
    }
  }
}

Plan3D::~Plan3D() {
  be::dfree(twr);
  be::dfree(no_poison);
}

// a range of rows of a last-axis pass: whole tiles where the kernels need them
static long rows_in_range(const Plan3D& P, long row0, long nrows) {
  const long all = (long)P.L.rows;
  const long n = nrows < 0 ? all - row0 : nrows;
  if (row0 < 0 || n < 1 || row0 + n > all) throw std::out_of_range("mvn: row range of a last-axis pass");
  if (P.rows_need_full_tiles() && (row0 % P.rows_tile() || n % P.rows_tile()))
    throw std::logic_error("mvn: a row range of the fixed last-axis kernels must be whole tiles");
  return n;
}

// a last-axis pass on the line layout: the complex pointer stays at the volume's first element, the kernel gets the
// first row instead
static void lines_args(const Plan3D& P, RowsParams& p, long row0, const void* nyq, const void* real_side,
                       const void* cplx_side) {
  if (!P.lines_capable()) throw std::logic_error("mvn: this shape has no line-layout last-axis passes");
  if (nyq) throw std::logic_error("mvn: the line layout keeps the Nyquist bins in the DC column");
  if (real_side && real_side == cplx_side) throw std::logic_error("mvn: a plain last-axis pass on the line layout is out of place");
  p.lines = 1;
  p.lines_d1 = P.L.d1;
  p.row_base = row0;
  p.nyq_packed = 1;
}

void Plan3D::rows_r2c(const float* in_real, cfloat* out, cfloat* out_nyq, be::stream_t s,
                      Profiler* prof, long row0, long nrows, bool lines) const {
  const long R = rows_in_range(*this, row0, nrows);
  const void* real0 = in_real;
  in_real += row0 * L.RP;
  if (!lines) out += row0 * L.C;
  if (out_nyq) out_nyq += row0;
  RowsParams p;
  std::memset(&p, 0, sizeof(p));
  p.ax = ax2.view;
  p.twr = twr;
  p.d2 = L.d2;
  p.h = L.h;
  p.C = L.C;
  p.RP = L.RP;
  p.rows = R;
  p.T = g_rows.T;
  p.TP = g_rows.TP;
  p.lds_alt = g_rows.lds_alt;
  p.lds_tw = g_rows.lds_tw;
  p.hmul = mvn_fastdiv_mul((unsigned)L.h);
  p.Cmul = mvn_fastdiv_mul((unsigned)L.C);
  p.in_real = in_real;
  p.out_cplx = out;
  p.out_nyq = out_nyq;
  p.nyq_packed = (L.even && !out_nyq) ? 1 : 0;  // no Nyquist plane given: DC + i Nyquist in column 0
  if (lines) lines_args(*this, p, row0, out_nyq, real0, out);
  ProfScope ps(prof, KK_ROWS_R2C, s);
  if (fx_rows) {
    p.fixed = 1;
    p.T = gx_rows.T;
    be::launch_rows_r2c(p, true, R / p.T, gx_rows.threads, gx_rows.lds_bytes, s);
    return;
  }
  const long ntiles = (R + p.T - 1) / p.T;
  be::launch_rows_r2c(p, L.even, ntiles, g_rows.threads, g_rows.lds_bytes, s);
}

void Plan3D::rows_c2r(const cfloat* in, const cfloat* in_nyq, float* out_real,
                      const EpilogueParams& epi_all, be::stream_t s, Profiler* prof, long row0,
                      long nrows, bool lines) const {
  const long R = rows_in_range(*this, row0, nrows);
  const void* cplx0 = in;
  const void* real0 = out_real;
  if (!lines) in += row0 * L.C;
  if (in_nyq) in_nyq += row0;
  if (out_real) out_real += row0 * L.RP;
  EpilogueParams epi = epi_all;
  const long eoff = row0 * L.RP;
  if (epi.view) epi.view += eoff;
  if (epi.psi) epi.psi += eoff;
  if (epi.weights) epi.weights += eoff;
  if (epi.delta) epi.delta += eoff;
  RowsParams p;
  std::memset(&p, 0, sizeof(p));
  p.ax = ax2.view;
  p.twr = twr;
  p.d2 = L.d2;
  p.h = L.h;
  p.C = L.C;
  p.RP = L.RP;
  p.rows = R;
  p.T = g_rows.T;
  p.TP = g_rows.TP;
  p.lds_alt = g_rows.lds_alt;
  p.lds_tw = g_rows.lds_tw;
  p.hmul = mvn_fastdiv_mul((unsigned)L.h);
  p.Cmul = mvn_fastdiv_mul((unsigned)L.C);
  p.in_cplx = in;
  p.in_nyq = in_nyq;
  p.nyq_packed = (L.even && !in_nyq) ? 1 : 0;
  if (lines) lines_args(*this, p, row0, in_nyq, real0, cplx0);
  p.out_real = out_real;
  p.epi = epi;
  if (!p.epi.poison) {
    p.epi.poison = no_poison;
    p.epi.poison_epoch = 0xffffffffu;
  }
  ProfScope ps(prof, KK_ROWS_C2R, s);
  if (fx_rows) {
    p.fixed = 1;
    p.T = gx_rows.T;
    be::launch_rows_c2r(p, true, R / p.T, gx_rows.threads, gx_rows.lds_bytes, s);
    return;
  }
  const long ntiles = (R + p.T - 1) / p.T;
  be::launch_rows_c2r(p, L.even, ntiles, g_rows.threads, g_rows.lds_bytes, s);
}

void Plan3D::rows_c2r_r2c(cfloat* data, cfloat* nyq, const EpilogueParams& epi_all, be::stream_t s,
                          Profiler* prof, long row0, long nrows, bool lines) const {
  if (!L.even) throw std::logic_error("mvn: rows_c2r_r2c needs an even last extent");
  const long R = rows_in_range(*this, row0, nrows);
  if (!lines) data += row0 * L.C;
  if (nyq) nyq += row0;
  EpilogueParams epi = epi_all;
  const long eoff = row0 * L.RP;
  if (epi.view) epi.view += eoff;
  if (epi.psi) epi.psi += eoff;
  if (epi.weights) epi.weights += eoff;
  if (epi.delta) epi.delta += eoff;
  RowsParams p;
  std::memset(&p, 0, sizeof(p));
  p.ax = ax2.view;
  p.twr = twr;
  p.d2 = L.d2;
  p.h = L.h;
  p.C = L.C;
  p.RP = L.RP;
  p.rows = R;
  p.in_cplx = data;
  p.in_nyq = nyq;
  p.out_cplx = data;
  p.out_nyq = nyq;
  p.nyq_packed = nyq ? 0 : 1;
  if (lines) lines_args(*this, p, row0, nyq, nullptr, nullptr);
  p.epi = epi;
  if (!p.epi.poison) {
    p.epi.poison = no_poison;
    p.epi.poison_epoch = 0xffffffffu;
  }
  ProfScope ps(prof, epi.mode == MVN_EPI_UPDATE ? KK_ROWS_FUSED_UPD : KK_ROWS_FUSED, s);
  if (fx_rows) {
    p.fixed = 1;
    p.T = gx_rows.T;
    be::launch_rows_c2r_r2c(p, R / p.T, gx_rows.threads, gx_rows.lds_bytes, s);
    return;
  }
  p.T = g_rows.T;
  p.TP = g_rows.TP;
  p.lds_alt = g_rows.lds_alt;
  p.lds_tw = g_rows.lds_tw;
  p.hmul = mvn_fastdiv_mul((unsigned)L.h);
  p.Cmul = mvn_fastdiv_mul((unsigned)L.C);
  const long ntiles = (R + p.T - 1) / p.T;
  be::launch_rows_c2r_r2c(p, ntiles, g_rows.threads, g_rows.lds_bytes, s);
}

bool Plan3D::lines_capable() const {
  return fx_rows && L.even && L.h == 256 && L.C == 256 && L.d1 == MF_N1 && gx_rows.T == 16 && !ax1.host.bluestein &&
         ax1.host.nfft == MF_N1;
}

void Plan3D::mid_fused(const cfloat* in, cfloat* out, const cfloat* taps, int k, int kd, unsigned* poison,
                       unsigned poison_epoch, be::stream_t s, Profiler* prof, int zbeg, int zcount, int n_peers,
                       unsigned* const* peers) const {
  if (!lines_capable()) throw std::logic_error("mvn: this shape has no fused middle pass");
  MidFusedParams p;
  std::memset(&p, 0, sizeof(p));
  p.in = in;
  p.out = out;
  p.taps = taps;
  p.kd = kd;
  p.tw = ax1.view.tw;
  p.d0 = L.d0;
  p.H = L.C;
  p.k = k;
  p.h = k / 2;
  p.seg = 0;  // whole columns: C = 256 workgroups, one per CU of an MI355X
  p.mode = MF_CONV;
  p.packed = 1;
  p.poison = poison;
  p.poison_epoch = poison_epoch;
  p.zbeg = zbeg;
  p.zcount = zcount;
  p.n_peers = n_peers;
  p.poison_peers = peers;
  ProfScope ps(prof, KK_MID_FUSED, s);
  be::launch_mid_fused(p, s);
}

void Plan3D::taps_to_lines(cfloat* taps, be::stream_t s) const {
  if (!lines_capable()) throw std::logic_error("mvn: this shape has no fused middle pass");
  MidFusedParams p;
  std::memset(&p, 0, sizeof(p));
  p.in = taps;
  p.out = taps;
  p.tw = ax1.view.tw;
  p.d0 = L.d0;
  p.H = L.C;
  p.mode = MF_TAPS;
  be::launch_mid_fused(p, s);
}

static StridedParams make_strided(const DevAxis& ax, const PassGeom& g, cfloat* data,
                                  const cfloat* spec, long ostride, long estride, long cstride,
                                  int ncols) {
  StridedParams p;
  std::memset(&p, 0, sizeof(p));
  p.ax = ax.view;
  p.data = data;
  p.spec = spec;
  p.ostride = ostride;
  p.estride = estride;
  p.cstride = cstride;
  p.ncols = ncols;
  p.T = g.T;
  p.TP = g.TP;
  p.lds_alt = g.lds_alt;
  p.lds_tw = g.lds_tw;
  p.tiles_per_outer = (ncols + g.T - 1) / g.T;
  return p;
}

void SideStream::create() {
  s = be::stream_create();
  fork = be::event_create_sync();
  join = be::event_create_sync();
}

void SideStream::destroy() {
  if (s) be::stream_destroy(s);
  be::event_destroy(fork);
  be::event_destroy(join);
  s = nullptr;
  fork = join = nullptr;
}

void SideStream::fork_from(be::stream_t main) {
  be::event_record(fork, main);
  be::stream_wait_event(s, fork);
}

void SideStream::join_into(be::stream_t main) {
  be::event_record(join, s);
  be::stream_wait_event(main, join);
}

void Plan3D::axis1(int mode, cfloat* data, cfloat* nyq, be::stream_t s, Profiler* prof,
                   be::stream_t s_nyq, int z0, int nz) const {
  if (!s_nyq) s_nyq = s;
  if (nz < 0) nz = L.d0 - z0;
  if (z0 < 0 || nz < 1 || z0 + nz > L.d0) throw std::out_of_range("mvn: plane range of a dim1 pass");
  data += (size_t)z0 * L.d1 * L.C;
  if (nyq) nyq += (size_t)z0 * L.d1;
  const bool ride = nyq_rides() && L.even && nyq;
  {
    // main array [d0][d1][C]: lines along d1, tiles of neighbouring bins
    const PassGeom& g = fx_ax1 ? gx_ax1 : g_ax1;
    StridedParams p = make_strided(ax1, g, data, nullptr, (long)L.d1 * L.C, L.C, 1, L.C);
    p.fixed = fx_ax1 ? 1 : 0;
    ProfScope ps(prof, mode == MVN_ST_FWD ? KK_AXIS1_FWD : KK_AXIS1_INV, s);
    if (ride) {
      // the Nyquist plane [d0][d1] rides in the same launch: its nz lines (contiguous along d1), one per workgroup
      StridedParams r = make_strided(ax1, g_nyq1_line, nyq, nullptr, 0, 1, L.d1, nz);
      r.is_nyq = 1;
      be::launch_strided(mode, p, (long)nz * p.tiles_per_outer, g.threads, g.lds_bytes, s, &r, g_nyq1_line.lds_bytes);
    } else {
      be::launch_strided(mode, p, (long)nz * p.tiles_per_outer, g.threads, g.lds_bytes, s);
    }
  }
  if (L.even && nyq && !ride) {  // (no plane: the Nyquist bins are packed into column 0 of the main array)
    // Nyquist plane [d0][d1]: lines along d1 are contiguous, neighbouring lines d1 apart
    StridedParams p = make_strided(ax1, g_nyq1, nyq, nullptr, 0, 1, L.d1, nz);
    p.is_nyq = 1;
    ProfScope ps(s_nyq == s ? prof : nullptr, KK_NYQ, s_nyq);
    be::launch_strided(mode, p, p.tiles_per_outer, g_nyq1.threads, g_nyq1.lds_bytes, s_nyq);
  }
}

// The Nyquist plane's dim1 transforms ride in the main array's launches where those are the fixed-length walking
// kernels (MVN_NYQ_RIDE=0: launches of their own, on the stream the caller names, as in rounds 1 - 3)
bool Plan3D::nyq_rides() const {
  static const bool off = env_int("MVN_NYQ_RIDE", 1) == 0;  // A/B knob
  return fx_ax1 && !off;
}

bool Plan3D::tiles_spectra() const {
  static const bool off = env_int("MVN_NO_TILED_SPECTRA", 0) != 0;  // A/B knob
  return fx_ax0 && !off;
}

void Plan3D::retile_spectrum(const cfloat* natural, cfloat* tiled, be::stream_t s) const {
  const long cols = (long)L.d1 * L.C;
  const int T = gx_ax0.T;
  // tiled[(tile * d0 + row) * T + c] = natural[row * cols + tile * T + c], in floats for copy3d
  be::launch_copy3d((float*)tiled, 2L * T, 2L * T * L.d0, (const float*)natural, 2L * cols, 2L * T, 2 * T,
                    L.d0, (int)(cols / T), s);
}

void Plan3D::axis0(int mode, cfloat* data, cfloat* nyq, const cfloat* spec,
                   const cfloat* spec_nyq, be::stream_t s, Profiler* prof, be::stream_t s_nyq,
                   const cfloat* src, const cfloat* src_nyq, bool spec_tiled) const {
  if (!s_nyq) s_nyq = s;
  const int kind = mode == MVN_ST_FWD ? KK_AXIS0_FWD
                                      : (mode == MVN_ST_INV ? KK_AXIS0_INV : KK_AXIS0_FUSED);
  {
    // main array viewed as [d0][d1*C]: lines along d0, all (d1, bin) columns are contiguous
    const long cols = (long)L.d1 * L.C;
    if (cols > 0x7fffffffL) throw std::invalid_argument("mvn: d1*d2 too large");
    const PassGeom& g = fx_ax0 ? gx_ax0 : (mode == MVN_ST_FWD_MUL_INV ? g_ax0f : g_ax0);
    StridedParams p = make_strided(ax0, g, data, spec, 0, cols, 1, (int)cols);
    p.fixed = fx_ax0 ? 1 : 0;
    p.src = src;
    if (spec_tiled && !fx_ax0) throw std::logic_error("mvn: tile-contiguous spectra need the fixed dim0 kernels");
    p.spec_tiled = spec_tiled ? 1 : 0;
    ProfScope ps(prof, kind, s);
    be::launch_strided(mode, p, p.tiles_per_outer, g.threads, g.lds_bytes, s);
  }
  if (L.even) {
    StridedParams p = make_strided(ax0, g_nyq0, nyq, spec_nyq, 0, L.d1, 1, L.d1);
    p.is_nyq = 1;
    p.src = src_nyq;
    ProfScope ps(s_nyq == s ? prof : nullptr, KK_NYQ, s_nyq);
    be::launch_strided(mode, p, p.tiles_per_outer, g_nyq0.threads, g_nyq0.lds_bytes, s_nyq);
  }
}

void Plan3D::forward(float* vol, cfloat* nyq, be::stream_t s, Profiler* prof) const {
  rows_r2c(vol, (cfloat*)vol, nyq, s, prof);
  axis1(MVN_ST_FWD, (cfloat*)vol, nyq, s, prof);
  axis0(MVN_ST_FWD, (cfloat*)vol, nyq, nullptr, nullptr, s, prof);
}

void Plan3D::backward(float* vol, cfloat* nyq, float scale, be::stream_t s,
                      Profiler* prof) const {
  axis0(MVN_ST_INV, (cfloat*)vol, nyq, nullptr, nullptr, s, prof);
  axis1(MVN_ST_INV, (cfloat*)vol, nyq, s, prof);
  EpilogueParams e;
  std::memset(&e, 0, sizeof(e));
  e.mode = MVN_EPI_STORE;
  e.scale = scale;
  rows_c2r((const cfloat*)vol, nyq, vol, e, s, prof);
}

void Plan3D::middle_passes(cfloat* work, cfloat* work_nyq, const cfloat* spec,
                           const cfloat* spec_nyq, be::stream_t s, Profiler* prof,
                           SideStream* side, bool spec_tiled) const {
  be::stream_t sn = s;
  // (where the plane's dim1 transforms ride in the main launches, its dim0 launch runs in line between them)
  if (side && side->s && L.even && !nyq_rides()) {
    side->fork_from(s);  // the plane was written by the last-axis pass just enqueued on s
    sn = side->s;
  }
  axis1(MVN_ST_FWD, work, work_nyq, s, prof, sn);
  axis0(MVN_ST_FWD_MUL_INV, work, work_nyq, spec, spec_nyq, s, prof, sn, nullptr, nullptr, spec_tiled);
  axis1(MVN_ST_INV, work, work_nyq, s, prof, sn);
  if (sn != s) side->join_into(s);  // the next last-axis pass on s reads the plane
}

void Plan3D::convolve(const float* in_real, cfloat* work, cfloat* work_nyq, const cfloat* spec,
                      const cfloat* spec_nyq, float* out_real, const EpilogueParams& epi,
                      be::stream_t s, Profiler* prof) const {
  rows_r2c(in_real, work, work_nyq, s, prof);
  axis1(MVN_ST_FWD, work, work_nyq, s, prof);
  axis0(MVN_ST_FWD_MUL_INV, work, work_nyq, spec, spec_nyq, s, prof);
  axis1(MVN_ST_INV, work, work_nyq, s, prof);
  rows_c2r(work, work_nyq, out_real, epi, s, prof);
}

void Plan3D::psf_spectrum(const float* d_kernel, const int* kdims, float scale, float* spec_vol,
                          cfloat* spec_nyq, be::stream_t s) const {
  for (int i = 0; i < 3; ++i) {
    const int D = i == 0 ? L.d0 : (i == 1 ? L.d1 : L.d2);
    if (kdims[i] < 1 || kdims[i] > D)
      throw std::invalid_argument("mvn: kernel extent must be in [1, image extent]");
  }
  be::dzero(spec_vol, main_bytes(), s);
  be::launch_scatter_psf(d_kernel, kdims[0], kdims[1], kdims[2], spec_vol, L.d0, L.d1, L.d2, L.RP,
                         scale, s);
  forward(spec_vol, spec_nyq, s, nullptr);
}

// ---------------------------------------------------------------------------------------------
// PlanStore
// ---------------------------------------------------------------------------------------------
PlanStore& PlanStore::get() {
  static PlanStore* inst = new PlanStore();  // leaked on purpose: outlives static destructors
  return *inst;
}

static std::array<int, 4> plan_key(int device, const shape_t& s) {
  std::array<int, 4> k = {{device, s[0], s[1], s[2]}};
  return k;
}

std::shared_ptr<Plan3D> PlanStore::add(int device, const shape_t& shape) {
  std::lock_guard<std::mutex> lk(mu_);
  auto key = plan_key(device, shape);
  auto it = plans_.find(key);
  if (it != plans_.end()) return it->second;
  const int prev = be::get_device();
  be::set_device(device);
  std::shared_ptr<Plan3D> p(new Plan3D(device, shape[0], shape[1], shape[2]));
  be::set_device(prev);
  plans_[key] = p;
  return p;
}

bool PlanStore::has_key(int device, const shape_t& shape) {
  std::lock_guard<std::mutex> lk(mu_);
  return plans_.count(plan_key(device, shape)) != 0;
}

std::shared_ptr<Plan3D> PlanStore::lookup(int device, const shape_t& shape) {
  std::lock_guard<std::mutex> lk(mu_);
  auto it = plans_.find(plan_key(device, shape));
  if (it == plans_.end())
    throw std::runtime_error("mvn::PlanStore: no plan for " + std::to_string(shape[0]) + "x" +
                             std::to_string(shape[1]) + "x" + std::to_string(shape[2]));
  return it->second;
}

bool PlanStore::empty() {
  std::lock_guard<std::mutex> lk(mu_);
  return plans_.empty();
}

size_t PlanStore::size() {
  std::lock_guard<std::mutex> lk(mu_);
  return plans_.size();
}

void PlanStore::clear() {
  std::lock_guard<std::mutex> lk(mu_);
  plans_.clear();
}

// ---------------------------------------------------------------------------------------------
// Engine
// ---------------------------------------------------------------------------------------------
Engine::Engine(int device, const shape_t& dims, int num_views) : device_(device) {
  // zero views: a rank of a view-sharded run that got none (6 views on 8 GPUs) still keeps a
  // replica of psi and takes part in the all-reduce with a zero correction
  if (num_views < 0) throw std::invalid_argument("mvn: num_views must be >= 0");
  be::set_device(device_);
  plan_ = PlanStore::get().add(device_, dims);
  stream_ = be::stream_create();
  side_.create();
  const size_t mb = plan_->main_bytes();
  psi_ = (float*)be::dmalloc(mb);
  work_ = work_alloc_ = (float*)be::dmalloc(mb);
  be::dzero(psi_, mb, stream_);
  be::dzero(work_, mb, stream_);
  if (plan_->nyq_bytes()) work_nyq_ = (cfloat*)be::dmalloc(plan_->nyq_bytes());
  poison_own_ = (unsigned*)be::dmalloc(256);  // the word, and at byte 64 the table of the peers' words
  be::dzero(poison_own_, 256, stream_);
  poison_ = poison_own_;
  views_.resize((size_t)num_views);
  spec_tiled_ = plan_->tiles_spectra();
  // read per engine (A/B runs, tests).  MVN_DIM0_DIRECT_MAX: most PSF planes the direct dim0 leg takes on
  // (MVN_D0_MAX_TAPS = 33 are instantiated); measured at 512^3 x 6 views on MI355X (profiles/r03_dim0_direct.md)
  // the whole iteration is 6.7 / 4.1 % faster than with the fused FFT pass at 15 / 31 planes
  direct_enabled_ = env_int("MVN_DIM0_DIRECT", 1) != 0;
  lines_capable_ = plan_->lines_capable() && env_int("MVN_MID_FUSED", 1) != 0;
  lines_forced_ = env_int("MVN_MID_FUSED", 1) >= 2;
  direct_max_taps_ = env_int("MVN_DIM0_DIRECT_MAX", MVN_D0_MAX_TAPS);
  direct_min_plane_ = env_int("MVN_DIM0_DIRECT_MIN_PLANE", 131072);  // see direct_ok_for()
  direct_min_items_ = env_int("MVN_DIM0_DIRECT_MIN_ITEMS", 0);
  d0_stagger_ = env_int("MVN_D0_STAGGER", 64);
  packed_allowed_ = packed_layout_for(plan_->main_bytes()) && mvn_dim0_packed_possible(dims[0]);
  for (int d = 0; d < 3; ++d) host_dims_[d] = dims[d];
  be::stream_sync(stream_);
}

// Packed Nyquist layout (mvn_dim0_direct.hpp): MVN_NYQ_PACKED = 1 always / 0 never / unset: for volumes up to
// MVN_NYQ_PACKED_MAX_MB (default 256).  Small volumes are bound by launches and latency and the layout's single
// launch chain wins (one view update, round 4, against the split layout with its Nyquist lines riding in the dim1
// launches: 64^3 0.055 / 0.062 ms, 128^3 0.082 / 0.089, 256^3 0.246 / 0.262; 320^3 and 384^3 even).  At 512^3 it
// is 6 % SLOWER (2.24 / 2.11 ms): the DC column's 2 x 257 dim0 columns are 8-byte accesses one row apart in every
// plane, and the leg takes 0.27 instead of 0.23 ms (profiles/r04_layouts.md).
bool Engine::packed_layout_for(size_t volume_bytes) {
  const int sw = env_int("MVN_NYQ_PACKED", -1);
  const size_t max_bytes = (size_t)env_int("MVN_NYQ_PACKED_MAX_MB", 256) << 20;
  return sw > 0 || (sw < 0 && volume_bytes <= max_bytes);
}

Engine::~Engine() {
  try {
    be::set_device(device_);
    if (stream_) be::stream_sync(stream_);
  } catch (...) {
  }
  for (size_t v = 0; v < views_.size(); ++v) {
    be::dfree(views_[v].image);
    be::dfree(views_[v].weights);
    be::dfree(views_[v].spec1);
    be::dfree(views_[v].nyq1);
    be::dfree(views_[v].spec2);
    be::dfree(views_[v].nyq2);
    for (int i = 0; i < 2; ++i) {
      be::dfree(views_[v].taps[i]);
      be::dfree(views_[v].taps_nyq[i]);
      be::dfree(views_[v].taps_l[i]);
      be::dfree(views_[v].taps_scr[i]);
    }
  }
  be::dfree(psi_);
  // (work_ and work2_ swap roles at every direct leg: freed by their allocations, not by their roles)
  be::dfree(work_alloc_);
  be::dfree(work_nyq_);
  be::dfree(work2_alloc_);
  be::dfree(work2_nyq_);
  be::dfree(psi_spec_);
  be::dfree(psi_spec_nyq_);
  be::dfree(embed_scratch_);
  be::dfree(poison_own_);
  be::graph_destroy(sweep_graph_);
  if (!delta_external_) be::dfree(delta_);
  try {
    if (side_.s) be::stream_sync(side_.s);
  } catch (...) {
  }
  side_.destroy();
  try {
    if (upload_stream_) be::stream_sync(upload_stream_);
  } catch (...) {
  }
  for (size_t i = 0; i < stage_scratch_.size(); ++i) be::dfree(stage_scratch_[i]);
  for (size_t i = 0; i < staged_ev_.size(); ++i) be::event_destroy(staged_ev_[i]);
  if (upload_stream_) be::stream_destroy(upload_stream_);
  if (stream_) be::stream_destroy(stream_);
}

void Engine::set_embedding(const int dims[3], const int off[3]) {
  be::set_device(device_);
  const Layout& L = plan_->L;
  const int ext[3] = {L.d0, L.d1, L.d2};
  bool dense = true, same = true;
  for (int d = 0; d < 3; ++d) {
    if (dims[d] < 1 || off[d] < 0 || off[d] + dims[d] > ext[d])
      throw std::invalid_argument("mvn: embedded stack does not fit the engine volume");
    dense = dense && dims[d] == ext[d];
    same = same && host_dims_[d] == dims[d] && host_off_[d] == off[d];
  }
  if (same && embedded_ == !dense) return;
  be::stream_sync(stream_);
  // the interior moves: what used to be interior may now be padding and must read zero
  for (size_t v = 0; v < views_.size(); ++v) {
    if (views_[v].image) be::dzero(views_[v].image, plan_->main_bytes(), stream_);
    if (views_[v].weights) be::dzero(views_[v].weights, plan_->main_bytes(), stream_);
  }
  be::stream_sync(stream_);
  be::dfree(embed_scratch_);
  embed_scratch_ = nullptr;
  for (int d = 0; d < 3; ++d) {
    host_dims_[d] = dims[d];
    host_off_[d] = off[d];
  }
  embedded_ = !dense;
  if (embedded_) embed_scratch_ = (float*)be::dmalloc(host_floats() * sizeof(float));
}

void Engine::upload_volume(float* dst, const float* host, be::stream_t s) {
  const Layout& L = plan_->L;
  if (embedded_) {
    be::h2d(embed_scratch_, host, host_floats() * sizeof(float), s);
    be::launch_copy3d(dst + ((size_t)host_off_[0] * L.d1 + host_off_[1]) * L.RP + host_off_[2], L.RP,
                      (long)L.d1 * L.RP, embed_scratch_, host_dims_[2], (long)host_dims_[1] * host_dims_[2],
                      host_dims_[2], host_dims_[1], host_dims_[0], s);
    return;
  }
  if (L.RP == L.d2)
    be::h2d(dst, host, L.logical() * sizeof(float), s);
  else
    be::h2d_2d(dst, (size_t)L.RP * sizeof(float), host, (size_t)L.d2 * sizeof(float),
               (size_t)L.d2 * sizeof(float), L.rows, s);
}

void Engine::alloc_view(ViewSlot& s) {
  const size_t mb = plan_->main_bytes();
  if (s.image) return;
  s.image = (float*)be::dmalloc(mb);
  s.weights = (float*)be::dmalloc(mb);
  // the PSF buffers (3-D spectra or direct-form taps) are allocated by prepare_psf, which knows the form
  if (plan_->L.RP != plan_->L.d2 || embedded_) {  // odd d2 / embedded stacks: the padding must hold zeros
    be::dzero(s.image, mb, stream_);
    be::dzero(s.weights, mb, stream_);
    be::stream_sync(stream_);
  }
}

static std::atomic<long> g_psf_hits{0}, g_psf_misses{0};
long Engine::psf_cache_hits() { return g_psf_hits.load(); }
long Engine::psf_cache_misses() { return g_psf_misses.load(); }

// The spectra of a slot stay valid for as long as the engine (shape, scale and layout are fixed
// per engine): compare the incoming kernel with the host copy of the one they were made from.
// A bytewise comparison of ~100 kB, not a hash: a false match would silently change results.
// MVN_PSF_CACHE=0 always prepares.
bool Engine::psf_resident(ViewSlot& s, int i, const float* kernel, const int* kdims) {
  static const bool enabled = [] {
    const char* e = std::getenv("MVN_PSF_CACHE");
    return !(e && std::strcmp(e, "0") == 0);
  }();
  const size_t n = (size_t)kdims[0] * (size_t)kdims[1] * (size_t)kdims[2];
  const bool same = enabled && s.set && s.kcopy[i].size() == n && s.kdims[i][0] == kdims[0] &&
                    s.kdims[i][1] == kdims[1] && s.kdims[i][2] == kdims[2] &&
                    std::memcmp(s.kcopy[i].data(), kernel, n * sizeof(float)) == 0;
  if (same) {
    ++g_psf_hits;
    return true;
  }
  ++g_psf_misses;
  s.kcopy[i].assign(kernel, kernel + n);
  for (int d = 0; d < 3; ++d) s.kdims[i][d] = kdims[d];
  return false;
}

void Engine::make_spectrum(const float* d_kernel, const int* kdims, float scale, float* spec,
                           cfloat* nyq, float* scratch, be::stream_t s) {
  if (!spec_tiled_) {
    plan_->psf_spectrum(d_kernel, kdims, scale, spec, nyq, s);
    return;
  }
  plan_->psf_spectrum(d_kernel, kdims, scale, scratch, nyq, s);  // the Nyquist plane stays as it is
  plan_->retile_spectrum((const cfloat*)scratch, (cfloat*)spec, s);
}

// The dim0 leg runs as a direct convolution when the PSF has at most MVN_D0_MAX_TAPS planes, the volume is
// deep enough for the kernel's window, and the small plan of the tap arrays transforms dims 1 and 2
// exactly as the volume's plan does (same kernel family => same position order of the spectra).
bool Engine::direct_form(const int* kdims) {
  if (!direct_enabled_ || kdims[0] > direct_max_taps_ || !mvn_dim0_direct_possible(kdims[0], plan_->L.d0) ||
      mvn_dim0_items_for(kdims[0], plan_->L.d0, (long)plan_->L.d1 * plan_->L.C, direct_min_plane_) < direct_min_items_)
    return false;
  const int kd = ((kdims[0] + 1 + 15) / 16) * 16;
  const Plan3D* tp = taps_plan(kd);
  return tp->fx_rows == plan_->fx_rows && tp->fx_ax1 == plan_->fx_ax1 && tp->L.C == plan_->L.C &&
         tp->L.RP == plan_->L.RP && tp->L.even == plan_->L.even;
}

Plan3D* Engine::taps_plan(int kd) {
  auto it = taps_plans_.find(kd);
  if (it != taps_plans_.end()) return it->second.get();
  std::unique_ptr<Plan3D> p(new Plan3D(device_, kd, plan_->L.d1, plan_->L.d2));
  Plan3D* raw = p.get();
  taps_plans_[kd] = std::move(p);
  return raw;
}

void Engine::prepare_psf(ViewSlot& s, int i, const float* d_kernel, const int* kdims, float* scratch,
                         bool staging, be::stream_t st) {
  const Layout& L = plan_->L;
  for (int d = 0; d < 3; ++d) {
    const int D = d == 0 ? L.d0 : (d == 1 ? L.d1 : L.d2);
    if (kdims[d] < 1 || kdims[d] > D) throw std::invalid_argument("mvn: kernel extent must be in [1, image extent]");
  }
  ++graph_gen_;  // a captured sweep holds the PSF buffers, their form and depth
  if (direct_form(kdims)) {
    const int kd = ((kdims[0] + 1 + 15) / 16) * 16;
    Plan3D* tp = taps_plan(kd);
    if (s.tap_kd[i] != kd) {  // (re)allocate for this depth; the previous arrays may still be read by the stream
      be::stream_sync(st);
      be::stream_sync(stream_);
      be::dfree(s.taps[i]);
      be::dfree(s.taps_nyq[i]);
      s.taps[i] = nullptr;
      s.taps_nyq[i] = nullptr;
      s.taps[i] = (float*)be::dmalloc(tp->main_bytes());
      if (tp->nyq_bytes()) s.taps_nyq[i] = (cfloat*)be::dmalloc(tp->nyq_bytes());
      be::dfree(s.taps_l[i]);
      be::dfree(s.taps_scr[i]);
      s.taps_l[i] = s.taps_scr[i] = nullptr;
      s.tap_kd[i] = kd;
    }
    s.taps_l_ok[i] = false;
    s.tap_k[i] = 0;  // not valid until the launches below are enqueued
    // dims 1 and 2 are transformed (un-normalised, both ways), dim0 is not: 1 / (d1 d2)
    const float scale = (float)(1.0 / ((double)L.d1 * (double)L.d2));
    be::dzero(s.taps[i], tp->main_bytes(), st);
    be::launch_scatter_psf(d_kernel, kdims[0], kdims[1], kdims[2], s.taps[i], kd, L.d1, L.d2, tp->L.RP, scale, st);
    tp->rows_r2c(s.taps[i], (cfloat*)s.taps[i], s.taps_nyq[i], st, nullptr);
    tp->axis1(MVN_ST_FWD, (cfloat*)s.taps[i], s.taps_nyq[i], st, nullptr);
    if (lines_capable_ && tp->lines_capable() && mvn_dim0_taps_template(kdims[0]) <= 31) {
      // the same planes for the fused middle pass: line layout, Nyquist bins packed, its own bin order along dim1
      if (!s.taps_l[i]) {
        s.taps_l[i] = (float*)be::dmalloc(tp->main_bytes());
        s.taps_scr[i] = (float*)be::dmalloc(tp->main_bytes());
      }
      be::dzero(s.taps_scr[i], tp->main_bytes(), st);
      be::launch_scatter_psf(d_kernel, kdims[0], kdims[1], kdims[2], s.taps_scr[i], kd, L.d1, L.d2, tp->L.RP, scale, st);
      tp->rows_r2c(s.taps_scr[i], (cfloat*)s.taps_l[i], nullptr, st, nullptr, 0, -1, true);
      tp->taps_to_lines((cfloat*)s.taps_l[i], st);
      s.taps_l_ok[i] = true;
    }
    s.tap_k[i] = kdims[0];
    return;
  }
  // 3-D spectrum for the fused FFT pass
  float*& spec = i == 0 ? s.spec1 : s.spec2;
  cfloat*& nyq = i == 0 ? s.nyq1 : s.nyq2;
  if (!spec) {
    spec = (float*)be::dmalloc(plan_->main_bytes());
    if (plan_->nyq_bytes()) nyq = (cfloat*)be::dmalloc(plan_->nyq_bytes());
  }
  s.tap_k[i] = 0;
  s.taps_l_ok[i] = false;
  if (spec_tiled_ && !scratch) {
    if (!staging) throw std::logic_error("mvn: spectrum scratch missing");
    if (!stage_spec_scratch_) {  // the main thread iterates on the work volume meanwhile: own scratch, freed
                                 // with the other staging scratch once the upload stream has drained
      stage_spec_scratch_ = (float*)be::dmalloc(plan_->main_bytes());
      stage_scratch_.push_back(stage_spec_scratch_);
    }
    scratch = stage_spec_scratch_;
  }
  const float scale = (float)(1.0 / (double)L.logical());  // inc/cpu_convolve.h:271-274
  make_spectrum(d_kernel, kdims, scale, spec, nyq, scratch, st);
}

void Engine::ensure_work2() {
  if (work2_) return;
  size_t skew = 0;
#ifdef MVN_EXPERIMENTS  // variant builds only: the second work volume displaced against the first (DRAM bank / channel phase)
  skew = (size_t)env_int("MVN_WORK2_SKEW_KB", 0) * 1024;
#endif
  work2_alloc_ = (float*)be::dmalloc(plan_->main_bytes() + skew);
  work2_ = work2_alloc_ + skew / sizeof(float);
  if (plan_->nyq_bytes()) work2_nyq_ = (cfloat*)be::dmalloc(plan_->nyq_bytes());
}

// the dim0 leg with the direct form of kernel i: in -> out (never in place).  Packed layout: one launch, the
// DC + i Nyquist column separated inside it.  Split layout: sn == stream_: main array and Nyquist plane in ONE
// launch on the engine's stream; otherwise the Nyquist plane as a launch of its own on sn
void Engine::dim0_conv(const ViewSlot& s, int i, const cfloat* in, const cfloat* in_nyq, cfloat* out,
                       cfloat* out_nyq, Profiler* prof, be::stream_t sn, int zbeg, int zcount, bool first) {
  const Layout& L = plan_->L;
  if (zcount < 0 || zbeg < 0 || zbeg + zcount > L.d0) throw std::out_of_range("mvn: plane range of a dim0 leg");
  Dim0DirectParams p;
  std::memset(&p, 0, sizeof(p));
  p.in = in;
  p.out = out;
  p.taps = (const cfloat*)s.taps[i];
  p.d0 = L.d0;
  p.k = s.tap_k[i];
  p.kd = s.tap_kd[i];
  p.h = s.tap_k[i] / 2;
  p.plane = (long)L.d1 * L.C;
  p.stagger = d0_stagger_;
  p.zbeg = zbeg;
  p.zcount = zcount;
  p.seg1 = mvn_dim0_piece_len(p.k, zcount > 0 ? zcount : L.d0, p.plane, direct_min_plane_);
  p.C = L.C;
  p.d1 = L.d1;
  // non-finite inputs are reported under this leg's epoch (never 0, never a value the word may still hold); the
  // launches of one leg (a slab's interior planes, then those next to its halos) share it
  if (first && ++epoch_ == 0x7fffffffu) {
    be::dzero(poison_, sizeof(unsigned), stream_);
    epoch_ = 1;
  }
  armed_epoch_ = epoch_;
  p.poison = poison_;
  p.poison_epoch = epoch_;
  p.n_peers = (int)poison_peers_.size();
  p.poison_peers = (unsigned* const*)(poison_own_ + 16);
  if (packed_) {
    p.packed = 1;
    p.taps2 = s.taps_nyq[i];
    p.inv1 = plan_->ax1.view.inv;
    ProfScope ps(prof, KK_AXIS0_DIRECT, stream_);
    be::launch_dim0_direct(p, stream_);
    return;
  }
  p.in2 = L.even ? in_nyq : nullptr;
  p.out2 = L.even ? out_nyq : nullptr;
  p.taps2 = L.even ? s.taps_nyq[i] : nullptr;
  p.plane2 = L.even ? L.d1 : 0;
  p.seg2 = 16;
  if (sn != stream_ && L.even) {
    Dim0DirectParams q = p;
    q.plane = 0;  // the Nyquist plane alone, in pieces of 16 output planes
    be::launch_dim0_direct(q, sn);
    p.plane2 = 0;
  }
  ProfScope ps(prof, KK_AXIS0_DIRECT, stream_);
  be::launch_dim0_direct(p, stream_);
}

void Engine::arm(EpilogueParams& e) {
  e.poison = armed_epoch_ ? poison_ : nullptr;
  e.poison_epoch = armed_epoch_;
  armed_epoch_ = 0;
}

void Engine::bind_poison(unsigned* external) {
  be::set_device(device_);
  be::stream_sync(stream_);
  be::graph_destroy(sweep_graph_);  // a captured sweep holds the word's address
  sweep_graph_ = nullptr;
  poison_ = external ? external : poison_own_;
  be::dzero(poison_own_, sizeof(unsigned), stream_);
  be::stream_sync(stream_);
}

unsigned Engine::poison_get() {
  be::set_device(device_);
  unsigned v = 0;
  be::d2h(&v, poison_, sizeof(v), stream_);
  be::stream_sync(stream_);
  return v;
}

// word = max(word, value): epochs only grow, so the larger one is the later report
void Engine::poison_merge(unsigned value) {
  if (value > poison_get()) {
    be::h2d(poison_, &value, sizeof(value), stream_);
    be::stream_sync(stream_);  // (`value` is a stack variable)
  }
}

void Engine::add_poison_peer(unsigned* word) {
  if (!word) throw std::invalid_argument("mvn: null poison word");
  if ((int)poison_peers_.size() >= MVN_D0_MAX_PEERS) throw std::invalid_argument("mvn: too many poison peers");
  be::set_device(device_);
  be::stream_sync(stream_);
  poison_peers_.push_back(word);
  be::h2d(poison_own_ + 16, poison_peers_.data(), poison_peers_.size() * sizeof(unsigned*), stream_);
  be::stream_sync(stream_);
}

// every view's two kernels are held in the direct form: the Nyquist bins can ride in the DC column
bool Engine::all_direct() const {
  if (views_.empty()) return false;
  for (size_t v = 0; v < views_.size(); ++v)
    if (!views_[v].set || !views_[v].tap_k[0] || !views_[v].tap_k[1]) return false;
  return true;
}

void Engine::decide_layout() {
  const bool allowed = layout_override_ < 0 ? packed_allowed_ : (layout_override_ > 0 && mvn_dim0_packed_possible(plan_->L.d0));
  const bool want = allowed && plan_->L.even && plan_->nyq_bytes() > 0;
  const bool was = packed_;
  packed_ = want && (pipelined_ ? packed_hint_ : all_direct());
  if (was != packed_) ++graph_gen_;  // ... and the layout
  lines_ = false;  // (only the sequential sweep asks for the line layout: decide_lines())
}

bool Engine::would_be_direct(const int* kdims) { return direct_form(kdims); }

// A whole column walks d0 + K - 1 steps for d0 outputs (the first K - 1 fill the filter's window).  Measured per view
// update on (d0, 512, 512) volumes (tools/mid_fused_planes.py, profiles/r04_mid_fused.md): with 31-plane PSFs the two
// forms meet at 96 - 128 planes (48: +10 %, 256: -15 %, 512: -18 %), with 15 planes the fused pass is ahead from 32
// planes on (-2 %; 256: -24 %), with 5 planes always (-12 .. -30 %): three PSF depths of planes.
// MVN_MID_FUSED=2 takes the fused pass whenever the shape has it (tests).
bool Engine::lines_worth(int k0) const { return lines_forced_ || plan_->L.d0 >= 3 * k0; }

bool Engine::would_be_lines(const int* kdims) {
  return lines_capable_ && lines_worth(kdims[0]) && direct_form(kdims) && mvn_dim0_taps_template(kdims[0]) <= 31 &&
         taps_plan(((kdims[0] + 1 + 15) / 16) * 16)->lines_capable();
}

// The sequential sweep takes the fused middle pass when the shape has it, every kernel is held in the form it
// reads and nothing else lays claim to the layout of the work volumes (halo exchanges copy planes of the row-major
// spectrum).  Decided per iterate() call, behind decide_layout().
void Engine::decide_lines() {
  const bool was = lines_last_sweep_;
  bool all = lines_capable_ && !views_.empty();
  if (halo_fn_ || halo_planes_ > 0)  // slabs: the common decision of whoever drives them, on plane ranges
    all = all && halo_fn_ && lines_override_ == 1 && halo_ranged() && halo_nyq_aware_;
  if (all) {
    if (pipelined_) {
      all = lines_hint_;
    } else {
      for (size_t v = 0; v < views_.size() && all; ++v)
        all = views_[v].set && views_[v].tap_k[0] && views_[v].tap_k[1] && views_[v].taps_l_ok[0] && views_[v].taps_l_ok[1] &&
              lines_worth(views_[v].tap_k[0]) && lines_worth(views_[v].tap_k[1]);
    }
  }
  lines_ = lines_last_sweep_ = all;
  if (was != lines_) ++graph_gen_;  // (a captured sweep holds the form of its passes)
}

// the three middle passes of convolution i as ONE: work_ -> work2_ (line layout), the volumes swap roles
void Engine::mid_fused_conv(const ViewSlot& s, int i, Profiler* prof, int zbeg, int zcount) {
  if (!s.taps_l_ok[i] || !s.tap_k[i]) throw std::logic_error("mvn: fused middle pass with a kernel that is not in its form");
  ensure_work2();
  if (++epoch_ == 0x7fffffffu) {
    be::dzero(poison_, sizeof(unsigned), stream_);
    epoch_ = 1;
  }
  armed_epoch_ = epoch_;
  plan_->mid_fused((const cfloat*)work_, (cfloat*)work2_, (const cfloat*)s.taps_l[i], s.tap_k[i], s.tap_kd[i], poison_,
                   epoch_, stream_, prof, zbeg, zcount, (int)poison_peers_.size(), (unsigned* const*)(poison_own_ + 16));
  std::swap(work_, work2_);
  std::swap(work_nyq_, work2_nyq_);
}

// Is the direct dim0 leg to be used for PSFs of k0 planes on a (d0, d1, d2) volume?  Switches: MVN_DIM0_DIRECT,
// MVN_DIM0_DIRECT_MAX (deepest PSF, <= 33), MVN_DIM0_DIRECT_MIN_PLANE (work items a launch should have: columns
// are cut into pieces below that, default 131072) and MVN_DIM0_DIRECT_MIN_ITEMS (fewest work items for which the
// leg is used at all, default 0: it wins at every size measured, profiles/r03_shapes.txt - 32^3 0.098 -> 0.062 ms
// per view update, 128^3 0.132 -> 0.079, 256^3 0.311 -> 0.245, 512^3 2.31 -> 2.16, 1024^3 19.9 -> 17.4 - on small
// volumes because the packed Nyquist layout it allows removes 6 of 14 launches).
bool Engine::direct_ok_for(int k0, int d0, int d1, int d2) {
  const long plane = (long)d1 * (d2 % 2 == 0 ? d2 / 2 : (d2 + 1) / 2);
  return env_int("MVN_DIM0_DIRECT", 1) != 0 && k0 >= 1 && k0 <= env_int("MVN_DIM0_DIRECT_MAX", MVN_D0_MAX_TAPS) &&
         mvn_dim0_direct_possible(k0, d0) &&
         mvn_dim0_items_for(k0, d0, plane, (long)env_int("MVN_DIM0_DIRECT_MIN_PLANE", 131072)) >=
             (long)env_int("MVN_DIM0_DIRECT_MIN_ITEMS", 0);
}

void Engine::middle(const ViewSlot& s, int i, Profiler* prof, SideStream* side, const RowsProducer* produce) {
  const Plan3D& P = *plan_;
  if (lines_) {
    if (!halo_fn_) {
      if (produce && *produce) (*produce)(0, -1);
      mid_fused_conv(s, i, prof);
      return;
    }
    // Halo mode on the line layout: the first and last H planes of the slab are the neighbours'.  The last-axis pass
    // that produces this convolution's input runs on the own planes [H, d0 - H) - the H planes at either end of them
    // FIRST, then the call that lets the neighbours pull them (planes are contiguous in the line layout as they are
    // in the row-major one), then the interior beside the copies -, the second call waits for the halos, and ONE
    // middle pass walks every column from plane 0 to produce the own planes (it never wraps: its window fills on
    // the lower halo).  Same calls, same order as the three-pass form below.
    const int H = halo_planes_, own = P.L.d0 - 2 * H, view = (int)(&s - views_.data());
    const long d1 = P.L.d1;
    auto rows = [&](int z0, int nz) {
      if (produce && *produce) (*produce)((long)z0 * d1, (long)nz * d1);
    };
    auto call = [&](int what) {
      if (halo_drain_) be::stream_sync(stream_);
      halo_fn_(halo_user_, work_, view, what);
    };
    if (produce && boundary_first()) {
      rows(H, H);
      rows(own, H);
      call(i);
      rows(2 * H, own - 2 * H);
    } else {
      rows(H, own);
      call(i);
    }
    if (halo_split_) call(i + 4);
    mid_fused_conv(s, i, prof, H, own);
    if (halo_post_) call(i + 2);
    return;
  }
  // One launch chain on stream_ - dim1 forward, direct leg, dim1 inverse, three launches - wherever the Nyquist bins
  // need no launches of their own: packed into the DC column, or (split layout) riding in the dim1 launches and in
  // the leg's.
  const bool one_chain = packed_ || (s.tap_k[i] != 0 && (!P.L.even || P.nyq_rides()));
  if (one_chain) {
    if (!s.tap_k[i]) throw std::logic_error("mvn: packed Nyquist layout with a kernel that is not in the direct form");
    if (halo_fn_ && !packed_ && !halo_nyq_aware_)
      throw std::logic_error("mvn: this halo hook exchanges the main array only: it needs the packed Nyquist layout");
    // Halo mode: the first and last H planes of the volume are the neighbours'.  No pass computes them - the
    // last-axis and dim1 passes run on the own planes [H, d0 - H) only, the leg produces only those - and the leg
    // runs in two parts: the planes that do not depend on the halos first, so that the exchange (the hook's
    // business) can go on beside them, then - behind the hook's second call - the H planes next to either halo.
    const int H = halo_ranged() ? halo_planes_ : 0;
    const int own = P.L.d0 - 2 * H;
    const int view = (int)(&s - views_.data());
    const long d1 = P.L.d1;
    auto rows = [&](int z0, int nz) {  // the pass that produces the planes [z0, z0 + nz) of this convolution's input
      if (produce && *produce) (*produce)(H > 0 ? (long)z0 * d1 : 0, H > 0 ? (long)nz * d1 : -1);
    };
    if (produce && boundary_first()) {
      // the planes the neighbours need FIRST - last-axis pass and dim1 pass of the H planes at either end of the own
      // range -, then the call that lets them pull, then the interior: their copies run beside these two passes'
      // interior and the leg's, not beside the leg's alone
      rows(H, H);
      rows(own, H);
      P.axis1(MVN_ST_FWD, (cfloat*)work_, wn(), stream_, prof, stream_, H, H);
      P.axis1(MVN_ST_FWD, (cfloat*)work_, wn(), stream_, nullptr, stream_, own, H);
      if (halo_drain_) be::stream_sync(stream_);
      halo_fn_(halo_user_, work_, view, i);
      rows(2 * H, own - 2 * H);
      P.axis1(MVN_ST_FWD, (cfloat*)work_, wn(), stream_, nullptr, stream_, 2 * H, own - 2 * H);
    } else {
      rows(H, own);
      P.axis1(MVN_ST_FWD, (cfloat*)work_, wn(), stream_, prof, stream_, H, own);
      if (halo_fn_) {  // the neighbours' planes arrive in the halo planes of the leg's input
        if (halo_drain_) be::stream_sync(stream_);
        halo_fn_(halo_user_, work_, view, i);
      }
    }
    ensure_work2();
    const cfloat* in = (const cfloat*)work_;
    const cfloat* in_n = packed_ ? nullptr : work_nyq_;
    cfloat* out = (cfloat*)work2_;
    cfloat* out_n = packed_ ? nullptr : work2_nyq_;
    if (H == 0) {
      if (halo_fn_ && halo_split_) {  // (no ranges, e.g. rows that do not end on tile boundaries: one part)
        if (halo_drain_) be::stream_sync(stream_);
        halo_fn_(halo_user_, work_, view, i + 4);
      }
      dim0_conv(s, i, in, in_n, out, out_n, prof, stream_);
    } else if (halo_split_ && own > 2 * H) {
      dim0_conv(s, i, in, in_n, out, out_n, prof, stream_, 2 * H, own - 2 * H, true);
      if (halo_drain_) be::stream_sync(stream_);
      halo_fn_(halo_user_, work_, view, i + 4);  // the halo planes must be in place behind this call
      dim0_conv(s, i, in, in_n, out, out_n, nullptr, stream_, H, H, false);
      dim0_conv(s, i, in, in_n, out, out_n, nullptr, stream_, own, H, false);
    } else {
      if (halo_split_) {
        if (halo_drain_) be::stream_sync(stream_);
        halo_fn_(halo_user_, work_, view, i + 4);
      }
      dim0_conv(s, i, in, in_n, out, out_n, prof, stream_, H, own, true);
    }
    std::swap(work_, work2_);
    std::swap(work_nyq_, work2_nyq_);
    P.axis1(MVN_ST_INV, (cfloat*)work_, wn(), stream_, prof, stream_, H, own);
    if (halo_fn_ && halo_post_) {  // the slabs merge their reports of non-finite inputs under the dim1 pass
      if (halo_drain_) be::stream_sync(stream_);
      halo_fn_(halo_user_, work_, view, i + 2);
    }
    return;
  }
  if (halo_fn_)
    throw std::logic_error("mvn: halo mode needs every PSF in the direct form (<= 33 planes) and Nyquist bins that need no "
                           "launches of their own (packed, or riding in the fixed-length dim1 kernels)");
  if (produce && *produce) (*produce)(0, -1);  // (no plane ranges outside halo mode)
  if (!s.tap_k[i]) {
    P.middle_passes((cfloat*)work_, work_nyq_, (const cfloat*)(i == 0 ? s.spec1 : s.spec2), i == 0 ? s.nyq1 : s.nyq2,
                    stream_, prof, side, spec_tiled_);
    return;
  }
  // The folded dim0 leg needs two fork / join pairs per convolution (17 - 20 us of cross-queue latency each):
  // below MVN_D0_SIDE_MIN_MB (default 256) the Nyquist plane's two dim1 launches cost less in line
  // (256^3 per view update: 0.329 ms with the side stream, 0.312 with the fused FFT pass)
  static const size_t d0_side_min = (size_t)env_int("MVN_D0_SIDE_MIN_MB", 256) << 20;
  const bool use_side = side && side->s && P.L.even && P.main_bytes() > d0_side_min && !P.nyq_rides();
  be::stream_t sn = use_side ? side->s : stream_;
  // MVN_D0_NYQ_SIDE=1: the Nyquist plane's whole chain (dim1, dim0 leg, dim1) on the side stream, one fork
  // and one join per convolution; default: its dim0 leg rides in the main launch, the side stream joins
  // before it and forks again behind it
  static const bool nyq_side = env_int("MVN_D0_NYQ_SIDE", 0) != 0;
  if (use_side) side->fork_from(stream_);  // the plane was written by the last-axis pass just enqueued on stream_
  P.axis1(MVN_ST_FWD, (cfloat*)work_, work_nyq_, stream_, prof, sn);
  ensure_work2();
  if (use_side && nyq_side) {
    dim0_conv(s, i, (const cfloat*)work_, work_nyq_, (cfloat*)work2_, work2_nyq_, prof, sn);
  } else {
    if (use_side) side->join_into(stream_);
    dim0_conv(s, i, (const cfloat*)work_, work_nyq_, (cfloat*)work2_, work2_nyq_, prof, stream_);
    if (use_side) side->fork_from(stream_);
  }
  std::swap(work_, work2_);
  std::swap(work_nyq_, work2_nyq_);
  P.axis1(MVN_ST_INV, (cfloat*)work_, work_nyq_, stream_, prof, sn);
  if (use_side) side->join_into(stream_);  // the next last-axis pass on stream_ reads the plane
}

void Engine::set_view(int v, const float* image, const float* weights, const float* kernel1,
                      const int* k1dims, const float* kernel2, const int* k2dims) {
  if (v < 0 || v >= (int)views_.size()) throw std::out_of_range("mvn: view index");
  be::set_device(device_);
  ViewSlot& s = views_[(size_t)v];
  alloc_view(s);
  upload_volume(s.image, image, stream_);
  upload_volume(s.weights, weights, stream_);
  const float* ks[2] = {kernel1, kernel2};
  const int* kd[2] = {k1dims, k2dims};
  for (int i = 0; i < 2; ++i) {
    if (psf_resident(s, i, ks[i], kd[i])) continue;
    const size_t kb = sizeof(float) * (size_t)kd[i][0] * (size_t)kd[i][1] * (size_t)kd[i][2];
    float* dk = (float*)be::dmalloc(kb);
    be::h2d(dk, ks[i], kb, stream_);
    try {
      // nothing else runs on this engine during a blocking set_view: the work volume is the scratch
      work_has_psi_spectrum_ = false;
      prepare_psf(s, i, dk, kd[i], work_, false, stream_);
      if (s.tap_k[i]) ensure_work2();  // the direct leg's second work volume: allocated here, not inside a sweep
    } catch (...) {
      be::stream_sync(stream_);
      be::dfree(dk);
      s.kcopy[i].clear();  // the spectrum is in an unknown state
      throw;
    }
    be::stream_sync(stream_);
    be::dfree(dk);
  }
  s.set = true;
}

// ---- pipelined staging ------------------------------------------------------------------------
void Engine::reserve_views() {
  be::set_device(device_);
  for (size_t v = 0; v < views_.size(); ++v) alloc_view(views_[v]);
  if (direct_enabled_) ensure_work2();  // (not inside the first sweep: the direct dim0 leg is out of place)
  if (!upload_stream_) upload_stream_ = be::stream_create_upload();
  staged_ev_.resize(views_.size(), nullptr);
  for (size_t v = 0; v < views_.size(); ++v)
    if (!staged_ev_[v]) staged_ev_[v] = be::event_create_sync();
  staged_.assign(views_.size(), 0);
  pipelined_ = true;
}

void Engine::stage_view(int v, const float* image, const float* weights, const float* kernel1,
                        const int* k1dims, const float* kernel2, const int* k2dims) {
  be::set_device(device_);  // the HIP device is per host thread
  ViewSlot& s = views_[(size_t)v];
  upload_volume(s.image, image, upload_stream_);
  upload_volume(s.weights, weights, upload_stream_);
  const float* ks[2] = {kernel1, kernel2};
  const int* kd[2] = {k1dims, k2dims};
  for (int i = 0; i < 2; ++i) {
    if (psf_resident(s, i, ks[i], kd[i])) continue;
    const size_t kb = sizeof(float) * (size_t)kd[i][0] * (size_t)kd[i][1] * (size_t)kd[i][2];
    float* dk = (float*)be::dmalloc(kb);
    stage_scratch_.push_back(dk);  // freed in finish_staging(), after the stream has drained
    be::h2d(dk, ks[i], kb, upload_stream_);
    try {
      prepare_psf(s, i, dk, kd[i], nullptr, true, upload_stream_);
    } catch (...) {
      s.kcopy[i].clear();  // the resident form is in an unknown state: never a cache hit
      throw;
    }
  }
  be::event_record(staged_ev_[(size_t)v], upload_stream_);
  s.set = true;
  {
    std::lock_guard<std::mutex> lk(stage_mu_);
    staged_[(size_t)v] = 1;
  }
  stage_cv_.notify_all();
}

void Engine::staging_failed() {
  {
    std::lock_guard<std::mutex> lk(stage_mu_);
    for (size_t v = 0; v < staged_.size(); ++v)
      if (staged_[v] == 0) staged_[v] = -1;
  }
  stage_cv_.notify_all();
}

void Engine::finish_staging() {
  be::set_device(device_);
  if (upload_stream_) be::stream_sync(upload_stream_);
  for (size_t i = 0; i < stage_scratch_.size(); ++i) be::dfree(stage_scratch_[i]);
  stage_scratch_.clear();
  stage_spec_scratch_ = nullptr;
}

// main thread: block until the uploader has enqueued view v, then make the compute stream wait
// for the upload stream's event
void Engine::wait_staged(int v) {
  std::unique_lock<std::mutex> lk(stage_mu_);
  stage_cv_.wait(lk, [&] { return staged_[(size_t)v] != 0; });
  if (staged_[(size_t)v] < 0) throw std::runtime_error("mvn: staging of view " + std::to_string(v) + " failed");
  lk.unlock();
  be::stream_wait_event(stream_, staged_ev_[(size_t)v]);
}

void Engine::set_psi(const float* host) {
  be::set_device(device_);
  psi_spec_valid_ = false;
  if (embedded_) be::dzero(psi_, plan_->main_bytes(), stream_);
  upload_volume(psi_, host, stream_);
  be::stream_sync(stream_);
}

void Engine::get_psi(float* host) {
  be::set_device(device_);
  const Layout& L = plan_->L;
  if (embedded_) {
    be::launch_copy3d(embed_scratch_, host_dims_[2], (long)host_dims_[1] * host_dims_[2],
                      psi_ + ((size_t)host_off_[0] * L.d1 + host_off_[1]) * L.RP + host_off_[2], L.RP,
                      (long)L.d1 * L.RP, host_dims_[2], host_dims_[1], host_dims_[0], stream_);
    be::d2h(host, embed_scratch_, host_floats() * sizeof(float), stream_);
    be::stream_sync(stream_);
    return;
  }
  if (L.RP == L.d2)
    be::d2h(host, psi_, L.logical() * sizeof(float), stream_);
  else
    be::d2h_2d(host, (size_t)L.d2 * sizeof(float), psi_, (size_t)L.RP * sizeof(float),
               (size_t)L.d2 * sizeof(float), L.rows, stream_);
  be::stream_sync(stream_);
}

// one (view, iteration): the reference's steps 1-4, src/gpu_deconvolve_methods.cuh:491-532.
//   psi (*) kernel1 -> view / blurred -> (*) kernel2 -> psi update
// With a fixed-length plan the two last-axis passes that meet between the convolutions
// (c2r + divide, then r2c) run as ONE kernel, and so do the update pass and the forward
// last-axis pass of the NEXT (view, iteration) when `feed_next` is set: 8 full passes instead of 10.
void Engine::conv_pair(int v, double lambda, float min_value, int final_mode, int accumulate,
                       bool feed_next) {
  const ViewSlot& s = views_[(size_t)v];
  if (!s.set) throw std::runtime_error("mvn: view " + std::to_string(v) + " was never set");
  Profiler* prof = nullptr;
  if (prof_.enabled && (pair_counter_++ % (prof_.sample_every > 0 ? prof_.sample_every : 1)) == 0)
    prof = &prof_;
  const Plan3D& P = *plan_;
  static const bool no_fuse = env_int("MVN_NO_FUSE", 0) != 0;  // A/B knob for experiments
  const bool fuse = P.can_fuse_rows() && !no_fuse;

  EpilogueParams e1;
  std::memset(&e1, 0, sizeof(e1));
  e1.mode = MVN_EPI_DIVIDE;
  e1.scale = 1.f;  // 1/N already lives in the PSF spectrum
  e1.view = s.image;
  e1.guard_zero_view = quotient_guard_ ? 1 : 0;

  EpilogueParams e2;
  std::memset(&e2, 0, sizeof(e2));
  e2.mode = final_mode;
  e2.scale = 1.f;
  e2.psi = psi_;
  e2.weights = s.weights;
  e2.delta = delta_;
  e2.accumulate = accumulate;
  e2.lambda = lambda;
  e2.lambda_inv = lambda > 0 ? (float)(1.f / lambda) : 0.f;
  e2.min_value = min_value;

  // convolution 1: psi (*) kernel1
  // The Nyquist-plane launches run on a second stream under the full-volume passes - except for
  // small volumes, where the two event waits per convolution cost more than three more 4 us
  // launches in line (measured per view update: 64^3 0.095 -> 0.074 ms, 96^3 0.124 -> 0.105 ms,
  // 128^3 0.143 -> 0.134 ms in line; 256^3 0.308 -> 0.340 ms, so the switch sits in between).
  static const bool no_side = env_int("MVN_NO_SIDE_STREAM", 0) != 0;  // A/B knob
  static const size_t side_min_bytes = (size_t)env_int("MVN_SIDE_MIN_MB", 16) << 20;
  SideStream* side = (no_side || P.main_bytes() <= side_min_bytes) ? nullptr : &side_;
  // The last-axis pass that PRODUCES a convolution's input is handed to middle() as a function of a row range: in
  // halo mode with a split leg the planes next to the halos are produced (and dim1-transformed) FIRST, so that the
  // neighbours can pull them while this slab still works on its interior (middle()); otherwise it is called once.
  // (work_ / work_nyq_ are read when a producer is BUILT: the direct dim0 leg swaps the two work volumes, and a
  // producer runs before the next leg.)
  const Plan3D* Pp = plan_.get();
  be::stream_t st = stream_;
  RowsProducer p1;
  if (pending_rows_) {  // the previous view update's fused update + forward pass, deferred (boundary-first order)
    p1 = std::move(pending_rows_);
    pending_rows_ = nullptr;
  } else if (!work_has_psi_spectrum_) {
    const float* psi = psi_;
    cfloat* w = (cfloat*)work_;
    cfloat* wnq = wn();
    const bool ln = lines_;
    p1 = [=](long r0, long nr) { Pp->rows_r2c(psi, w, ln ? nullptr : wnq, st, prof, r0, nr, ln); };
  }
  work_has_psi_spectrum_ = false;
  middle(s, 0, prof, side, p1 ? &p1 : nullptr);
  arm(e1);
  // view / blurred, handed to convolution 2 as its last-axis spectrum
  {
    float* w = work_;
    cfloat* wnq = wn();
    const bool ln = lines_;
    RowsProducer p2 = [=](long r0, long nr) {
      if (ln) {
        Pp->rows_c2r_r2c((cfloat*)w, nullptr, e1, st, prof, r0, nr, true);
      } else if (fuse) {
        Pp->rows_c2r_r2c((cfloat*)w, wnq, e1, st, prof, r0, nr);
      } else {
        Pp->rows_c2r((const cfloat*)w, wnq, w, e1, st, prof, r0, nr);
        Pp->rows_r2c(w, (cfloat*)w, wnq, st, prof, r0, nr);
      }
    };
    // convolution 2: quotient (*) kernel2, then the psi update fused into the last pass
    middle(s, 1, prof, side, &p2);
  }
  arm(e2);
  // (halo mode: the rows of the own planes only)
  const long r0 = halo_ranged() ? (long)halo_planes_ * P.L.d1 : 0;
  const long nr = halo_ranged() ? (long)(P.L.d0 - 2 * halo_planes_) * P.L.d1 : -1;
  if (fuse && feed_next && final_mode == MVN_EPI_UPDATE) {
    float* w = work_;
    cfloat* wnq = wn();
    const bool ln = lines_;
    RowsProducer upd = [=](long a, long n) { Pp->rows_c2r_r2c((cfloat*)w, ln ? nullptr : wnq, e2, st, prof, a, n, ln); };
    if (boundary_first())
      pending_rows_ = std::move(upd);  // runs as the producer of the NEXT convolution, boundary planes first
    else
      upd(r0, nr);
    work_has_psi_spectrum_ = true;
  } else {
    P.rows_c2r((const cfloat*)work_, lines_ ? nullptr : wn(), psi_, e2, stream_, prof, r0, nr, lines_);
  }
}

// a deferred producer also WRITES psi (the update epilogue): whoever leaves the loop runs it
void Engine::flush_pending_rows() {
  if (!pending_rows_) return;
  const long d1 = plan_->L.d1;
  const bool rg = halo_ranged();
  pending_rows_(rg ? (long)halo_planes_ * d1 : 0, rg ? (long)(plan_->L.d0 - 2 * halo_planes_) * d1 : -1);
  pending_rows_ = nullptr;
}

// the leg runs in two parts and the slab has planes that do not depend on its halos: boundary planes first
bool Engine::boundary_first() const {
  return halo_fn_ && halo_split_ && halo_ranged() && plan_->L.d0 - 2 * halo_planes_ > 2 * halo_planes_;
}

// Sweeps 2 .. n-1 of a call are identical launch sequences (every view update starts from the
// last-axis transform the previous one left and leaves one itself).  Where a sweep is
// launch-bound -- a 64^3 view update is 14 launches of ~6 us kernels -- it can be captured once
// as a graph and replayed.  Measured on MI355X (tools/graph_probe.py): 64^3 0.091 -> 0.081 ms,
// 128^3 0.132 -> 0.120 ms, 256^3 0.308 -> 0.305 ms per view update, against ~11 ms for capture
// and instantiation, which only a long-lived engine (thousands of small view updates) earns
// back.  Hence opt-in: MVN_GRAPH=1 enables it, MVN_GRAPH_MAX_MB (default 160) bounds the volume.
void Engine::iterate(int iterations, double lambda, float min_value) {
  be::set_device(device_);
  work_has_psi_spectrum_ = false;  // psi may have been replaced since the last call
  pending_rows_ = nullptr;
  psi_spec_valid_ = false;
  decide_layout();
  decide_lines();
  const int V = (int)views_.size();
  static const bool graphs_on = env_int("MVN_GRAPH", 0) != 0 && be::graphs_supported();
  static const size_t graph_max_bytes = (size_t)env_int("MVN_GRAPH_MAX_MB", 160) << 20;
  static const bool no_fuse = env_int("MVN_NO_FUSE", 0) != 0;
  // (a captured sweep holds buffer addresses: the two work volumes must be back in their roles after it,
  // i.e. the sweep must contain an even number of direct dim0 legs)
  bool use_graph = graphs_on && !halo_fn_ && iterations >= 3 && !prof_.enabled && plan_->can_fuse_rows() &&
                   !no_fuse && plan_->main_bytes() <= graph_max_bytes;
  for (int it = 0; it < iterations; ++it) {
    if (use_graph && it == 1) {  // every view has been staged by now: its PSF forms are known
      int swaps = 0;
      for (int v = 0; v < V; ++v) swaps += (views_[(size_t)v].tap_k[0] != 0) + (views_[(size_t)v].tap_k[1] != 0);
      use_graph = swaps % 2 == 0;
    }
    if (use_graph && it >= 1 && it < iterations - 1) {
      if (sweep_graph_ && (graph_lambda_ != lambda || graph_min_ != min_value ||
                           graph_guard_ != quotient_guard_ || graph_captured_gen_ != graph_gen_ ||
                           graph_work_ != work_)) {
        be::graph_destroy(sweep_graph_);
        sweep_graph_ = nullptr;
      }
      if (!sweep_graph_) {
        // all views were used by sweep 0, so none is still being staged
        const float* work_at_capture = work_;
        be::capture_begin(stream_);
        try {
          // (a replay reports under the epochs of the capture: the word must not still hold one of them)
          be::dzero(poison_, sizeof(unsigned), stream_);
          for (int v = 0; v < V; ++v) conv_pair(v, lambda, min_value, MVN_EPI_UPDATE, 0, true);
        } catch (...) {
          try {
            be::graph_destroy(be::capture_end(stream_));
          } catch (...) {
          }
          throw;
        }
        sweep_graph_ = be::capture_end(stream_);
        graph_lambda_ = lambda;
        graph_min_ = min_value;
        graph_guard_ = quotient_guard_;
        graph_captured_gen_ = graph_gen_;
        graph_work_ = work_at_capture;  // (the two work volumes swap roles at every direct leg)
      }
      be::graph_launch(sweep_graph_, stream_);
      continue;
    }
    for (int v = 0; v < V; ++v) {
      if (pipelined_ && it == 0) wait_staged(v);  // the uploader thread may still be busy with v
      const bool last = (it == iterations - 1) && (v == V - 1);
      conv_pair(v, lambda, min_value, MVN_EPI_UPDATE, 0, !last);
    }
  }
  flush_pending_rows();
  work_has_psi_spectrum_ = false;
}

float* Engine::delta_ptr() {
  if (!delta_) {
    be::set_device(device_);
    delta_ = (float*)be::dmalloc(plan_->main_bytes());
    be::dzero(delta_, plan_->main_bytes(), stream_);
  }
  return delta_;
}

// the rows of the own planes start and end on tile boundaries of the last-axis passes
bool Engine::halo_ranged() const {
  if (!halo_fn_ || halo_planes_ < 1) return false;
  if (!plan_->rows_need_full_tiles()) return true;
  const long T = plan_->rows_tile(), d1 = plan_->L.d1;
  return ((long)halo_planes_ * d1) % T == 0 && ((long)(plan_->L.d0 - 2 * halo_planes_) * d1) % T == 0;
}

void Engine::set_halo_planes(int planes, bool split) {
  if (planes < 0 || 2 * planes >= plan_->L.d0) throw std::invalid_argument("mvn: halo planes must leave own planes");
  halo_planes_ = planes;
  halo_split_ = split && planes > 0;
}

void Engine::set_halo_hook(halo_fn_t fn, void* user, bool drain, bool post) {
  if (fn) {
    if (!plan_->L.even || !mvn_dim0_packed_possible(plan_->L.d0))
      throw std::invalid_argument("mvn: halo mode needs an even last extent and at most 4062 planes per rank");
    packed_allowed_ = true;  // (a hook that also exchanges the Nyquist plane says so: set_halo_nyq_aware)
  }
  halo_nyq_aware_ = false;
  halo_fn_ = fn;
  halo_user_ = user;
  halo_drain_ = drain;
  halo_post_ = post && fn;
  if (!fn) {
    halo_planes_ = 0;
    halo_split_ = false;
  }
}

void Engine::copy_planes(void* spectrum, int plane0, int nplanes, void* buffer, bool to_buffer, bool host_buffer,
                         bool wait) {
  be::set_device(device_);
  const Layout& L = plan_->L;
  if (!spectrum || !buffer || plane0 < 0 || nplanes < 1 || plane0 + nplanes > L.d0)
    throw std::invalid_argument("mvn: copy_planes out of range");
  const size_t pb = (size_t)L.d1 * (size_t)L.C * sizeof(cfloat);
  char* p = (char*)spectrum + (size_t)plane0 * pb;
  const size_t n = (size_t)nplanes * pb;
  if (host_buffer) {
    if (to_buffer)
      be::d2h(buffer, p, n, stream_);
    else
      be::h2d(p, buffer, n, stream_);
  } else if (to_buffer) {
    be::d2d(buffer, p, n, stream_);
  } else {
    be::d2d(p, buffer, n, stream_);
  }
  if (wait) be::stream_sync(stream_);
}

void Engine::bind_delta(float* external) {
  be::set_device(device_);
  be::stream_sync(stream_);
  if (!delta_external_) be::dfree(delta_);
  delta_ = external;
  delta_external_ = external != nullptr;
}

// Simultaneous (Jacobi) step.  psi is not touched while the local views are swept, so its
// forward last-axis and dim1 passes are done ONCE and every view's first convolution starts at
// the (out-of-place) dim0 pass: 7 full passes per view instead of 9.
void Engine::compute_delta(double lambda, float min_value) {
  compute_delta_head(lambda, min_value);
  compute_delta_chunk(0, 1);
}

// planes [z0, z0 + nz) of chunk c of n
void Engine::chunk_planes(int c, int n, int* z0, int* nz) const {
  const int d0 = plan_->L.d0;
  if (n < 1 || n > d0 || c < 0 || c >= n) throw std::out_of_range("mvn: delta chunk index");
  const int a = (int)((long)c * d0 / n), b = (int)((long)(c + 1) * d0 / n);
  *z0 = a;
  *nz = b - a;
}

// the largest chunk count <= wanted whose plane boundaries fall on tile boundaries of the
// last-axis passes (the fixed-length kernels work on whole tiles of T rows)
int Engine::delta_chunks(int wanted) const {
  const Layout& L = plan_->L;
  int n = wanted < 1 ? 1 : (wanted > L.d0 ? L.d0 : wanted);
  if (!plan_->rows_need_full_tiles()) return n;
  const int T = plan_->rows_tile();
  for (; n > 1; --n) {
    bool ok = true;
    for (int c = 1; c < n && ok; ++c) ok = (((long)c * L.d0 / n) * L.d1) % T == 0;
    if (ok) break;
  }
  return n;
}

void Engine::delta_chunk_range(int c, int n, size_t* first_float, size_t* n_floats) const {
  int z0 = 0, nz = 0;
  chunk_planes(c, n, &z0, &nz);
  const size_t plane = (size_t)plan_->L.d1 * (size_t)plan_->L.RP;
  *first_float = (size_t)z0 * plane;
  *n_floats = (size_t)nz * plane;
}

void Engine::compute_delta_head(double lambda, float min_value) {
  be::set_device(device_);
  delta_ptr();
  tail_pending_ = false;
  tail_prof_ = nullptr;
  work_has_psi_spectrum_ = false;
  const int V = (int)views_.size();
  if (V == 0) return;  // a rank without views contributes a zero correction (compute_delta_chunk)
  {
    const bool was = packed_;
    decide_layout();
    decide_lines();
    // the chunk-fed spectrum of psi is in the other layout
    if (was != packed_ || (psi_spec_valid_ && psi_spec_lines_ != lines_)) psi_spec_valid_ = false;
  }
  const Plan3D& P = *plan_;
  // event pairs around every launch cost ~3 % of a sweep: sample like iterate() does
  const int every = prof_.sample_every > 0 ? prof_.sample_every : 1;
  Profiler* prof = (prof_.enabled && (pair_counter_++ % every) == 0) ? &prof_ : nullptr;
  if (!psi_spec_) {
    psi_spec_ = (float*)be::dmalloc(P.main_bytes());
    if (P.nyq_bytes()) psi_spec_nyq_ = (cfloat*)be::dmalloc(P.nyq_bytes());
    psi_spec_valid_ = false;
  }
  static const bool no_fuse = env_int("MVN_NO_FUSE", 0) != 0;
  const bool fuse = P.can_fuse_rows() && !no_fuse;
  static const bool no_side = env_int("MVN_NO_SIDE_STREAM", 0) != 0;
  static const size_t side_min_bytes = (size_t)env_int("MVN_SIDE_MIN_MB", 16) << 20;
  const bool use_side = !no_side && side_.s && P.L.even && P.main_bytes() > side_min_bytes && !packed_ && !P.nyq_rides();
  be::stream_t sn = use_side ? side_.s : stream_;
  if (!psi_spec_valid_) {  // else: left there chunk by chunk by apply_delta_chunk(.., feed_next)
    if (lines_) {  // line layout: the fused middle pass transforms along dim1 itself
      P.rows_r2c(psi_, (cfloat*)psi_spec_, nullptr, stream_, prof, 0, -1, true);
    } else {
      P.rows_r2c(psi_, (cfloat*)psi_spec_, pn(), stream_, prof);
      if (use_side) side_.fork_from(stream_);
      P.axis1(MVN_ST_FWD, (cfloat*)psi_spec_, pn(), stream_, prof, sn);
    }
    psi_spec_lines_ = lines_;
  }
  psi_spec_valid_ = false;  // consumed by this step; psi changes when the correction is applied
  for (int v = 0; v < V; ++v) {
    const ViewSlot& s = views_[(size_t)v];
    if (!s.set) throw std::runtime_error("mvn: view " + std::to_string(v) + " was never set");
    prof = (prof_.enabled && (pair_counter_++ % every) == 0) ? &prof_ : nullptr;
    EpilogueParams e1;
    std::memset(&e1, 0, sizeof(e1));
    e1.mode = MVN_EPI_DIVIDE;
    e1.scale = 1.f;
    e1.view = s.image;
    e1.guard_zero_view = quotient_guard_ ? 1 : 0;
    EpilogueParams e2;
    std::memset(&e2, 0, sizeof(e2));
    e2.mode = MVN_EPI_DELTA;
    e2.scale = 1.f;
    e2.psi = psi_;
    e2.weights = s.weights;
    e2.delta = delta_;
    e2.accumulate = v == 0 ? 0 : 1;
    e2.lambda = lambda;
    e2.lambda_inv = lambda > 0 ? (float)(1.f / lambda) : 0.f;
    e2.min_value = min_value;
    if (lines_) {
      // line layout: psi's last-axis spectrum -> ONE middle pass -> fused divide -> ONE middle pass -> correction
      if (!s.taps_l_ok[0] || !s.tap_k[0]) throw std::logic_error("mvn: fused middle pass with a kernel that is not in its form");
      if (++epoch_ == 0x7fffffffu) {
        be::dzero(poison_, sizeof(unsigned), stream_);
        epoch_ = 1;
      }
      armed_epoch_ = epoch_;
      P.mid_fused((const cfloat*)psi_spec_, (cfloat*)work_, (const cfloat*)s.taps_l[0], s.tap_k[0], s.tap_kd[0], poison_,
                  epoch_, stream_, prof);
      arm(e1);
      P.rows_c2r_r2c((cfloat*)work_, nullptr, e1, stream_, prof, 0, -1, true);
      mid_fused_conv(s, 1, prof);
      arm(e2);
      if (v + 1 < V) {
        P.rows_c2r((const cfloat*)work_, nullptr, psi_, e2, stream_, prof, 0, -1, true);
      } else {
        tail_epi_ = e2;
        tail_prof_ = prof;
        tail_pending_ = true;
      }
      continue;
    }
    // convolution 1 from the shared spectrum of psi; the Nyquist-plane launches ride on the side
    // stream (forked per view: the previous view's last pass still reads work_nyq_, and the
    // chunk-fed spectrum of psi was written on the main stream)
    if (s.tap_k[0]) {
      // (the shared spectrum of psi may still be in the making on the side stream: v == 0)
      if (use_side && v == 0) side_.join_into(stream_);
      dim0_conv(s, 0, (const cfloat*)psi_spec_, pn(), (cfloat*)work_, wn(), prof, stream_);
      if (use_side) side_.fork_from(stream_);
    } else {
      if (use_side) side_.fork_from(stream_);
      P.axis0(MVN_ST_FWD_MUL_INV, (cfloat*)work_, wn(), (const cfloat*)s.spec1, s.nyq1, stream_, prof, sn,
              (const cfloat*)psi_spec_, pn(), spec_tiled_);
    }
    P.axis1(MVN_ST_INV, (cfloat*)work_, wn(), stream_, prof, sn);
    if (use_side) side_.join_into(stream_);
    arm(e1);
    if (fuse) {
      P.rows_c2r_r2c((cfloat*)work_, wn(), e1, stream_, prof);
    } else {
      P.rows_c2r((const cfloat*)work_, wn(), work_, e1, stream_, prof);
      P.rows_r2c(work_, (cfloat*)work_, wn(), stream_, prof);
    }
    middle(s, 1, prof, use_side ? &side_ : nullptr);
    arm(e2);
    if (v + 1 < V) {
      P.rows_c2r((const cfloat*)work_, wn(), psi_, e2, stream_, prof);
    } else {  // the last view's final pass is launched chunk by chunk (compute_delta_chunk)
      tail_epi_ = e2;
      tail_prof_ = prof;
      tail_pending_ = true;
    }
  }
}

void Engine::compute_delta_chunk(int c, int n) {
  be::set_device(device_);
  int z0 = 0, nz = 0;
  chunk_planes(c, n, &z0, &nz);
  const Layout& L = plan_->L;
  if (views_.empty()) {
    size_t off = 0, cnt = 0;
    delta_chunk_range(c, n, &off, &cnt);
    be::dzero(delta_ptr() + off, cnt * sizeof(float), stream_);
    return;
  }
  if (!tail_pending_) throw std::logic_error("mvn: compute_delta_chunk without compute_delta_head");
  plan_->rows_c2r((const cfloat*)work_, lines_ ? nullptr : wn(), psi_, tail_epi_, stream_, c == 0 ? tail_prof_ : nullptr,
                  (long)z0 * L.d1, (long)nz * L.d1, lines_);
  if (c == n - 1) tail_pending_ = false;
}

void Engine::apply_delta_chunk(int c, int n, bool feed_next) {
  be::set_device(device_);
  int z0 = 0, nz = 0;
  chunk_planes(c, n, &z0, &nz);
  size_t off = 0, cnt = 0;
  delta_chunk_range(c, n, &off, &cnt);
  const Plan3D& P = *plan_;
  work_has_psi_spectrum_ = false;
  psi_spec_valid_ = false;
  if (fed_n_ != n) {  // a new round of chunks (any order within a round)
    fed_n_ = n;
    fed_ = 0;
  }
  be::launch_axpy1(psi_ + off, delta_ptr() + off, cnt, stream_);
  if (feed_next && !views_.empty()) {
    if (!psi_spec_) {
      psi_spec_ = (float*)be::dmalloc(P.main_bytes());
      if (P.nyq_bytes()) psi_spec_nyq_ = (cfloat*)be::dmalloc(P.nyq_bytes());
    }
    if (lines_) {  // (the layout of the step this correction belongs to; the next head checks it against its own)
      P.rows_r2c(psi_, (cfloat*)psi_spec_, nullptr, stream_, nullptr, (long)z0 * P.L.d1, (long)nz * P.L.d1, true);
    } else {
      P.rows_r2c(psi_, (cfloat*)psi_spec_, pn(), stream_, nullptr, (long)z0 * P.L.d1,
                 (long)nz * P.L.d1);
      P.axis1(MVN_ST_FWD, (cfloat*)psi_spec_, pn(), stream_, nullptr, stream_, z0, nz);
    }
    psi_spec_lines_ = lines_;
    if (++fed_ == n) {  // every plane of the spectrum belongs to the new psi
      psi_spec_valid_ = true;
      fed_ = 0;
    }
  } else {
    fed_ = 0;
  }
}

void Engine::apply_delta() {
  be::set_device(device_);
  psi_spec_valid_ = false;
  be::launch_axpy1(psi_, delta_ptr(), plan_->L.real_floats(), stream_);
}

void Engine::sync() {
  be::set_device(device_);
  be::stream_sync(stream_);
  prof_.collect();
}

// ---------------------------------------------------------------------------------------------
// SlabEngine
// ---------------------------------------------------------------------------------------------
SlabEngine::SlabEngine(int device, const shape_t& dims, int nranks, int rank, int num_views)
    : device_(device), P_(nranks), rank_(rank), dims_(dims) {
  if (num_views < 1) throw std::invalid_argument("mvn: num_views must be >= 1");
  if (nranks < 1 || rank < 0 || rank >= nranks) throw std::invalid_argument("mvn: bad rank / nranks");
  if (dims[0] % nranks || dims[1] % nranks || dims[0] / nranks < 2 || dims[1] / nranks < 2)
    throw std::invalid_argument("mvn: slab mode needs d0 and d1 divisible by the rank count, "
                                "with at least two planes / rows per rank");
  be::set_device(device_);
  shape_t a = {{dims[0] / nranks, dims[1], dims[2]}};
  shape_t b = {{dims[0], dims[1] / nranks, dims[2]}};
  planA_ = PlanStore::get().add(device_, a);
  planB_ = PlanStore::get().add(device_, b);
  stream_ = be::stream_create();
  const size_t mb = planA_->main_bytes(), nb = planA_->nyq_bytes();
  psi_ = (float*)be::dmalloc(mb);
  work_ = (float*)be::dmalloc(mb);
  own_a_ = (float*)be::dmalloc(mb);
  own_b_ = (float*)be::dmalloc(mb);
  be::dzero(psi_, mb, stream_);
  be::dzero(work_, mb, stream_);
  if (nb) {
    work_nyq_ = (cfloat*)be::dmalloc(nb);
    own_an_ = (cfloat*)be::dmalloc(nb);
    own_bn_ = (cfloat*)be::dmalloc(nb);
  }
  a_main_ = own_a_;
  b_main_ = own_b_;
  a_nyq_ = own_an_;
  b_nyq_ = own_bn_;
  views_.resize((size_t)num_views);
  be::stream_sync(stream_);
}

SlabEngine::~SlabEngine() {
  try {
    be::set_device(device_);
    if (stream_) be::stream_sync(stream_);
  } catch (...) {
  }
  for (size_t v = 0; v < views_.size(); ++v) {
    be::dfree(views_[v].image);
    be::dfree(views_[v].weights);
    for (int i = 0; i < 2; ++i) {
      be::dfree(views_[v].spec[i]);
      be::dfree(views_[v].nyq[i]);
    }
  }
  be::dfree(psi_);
  be::dfree(work_);
  be::dfree(work_nyq_);
  be::dfree(own_a_);
  be::dfree(own_b_);
  be::dfree(own_an_);
  be::dfree(own_bn_);
  if (stream_) be::stream_destroy(stream_);
}

void SlabEngine::sync() {
  be::set_device(device_);
  be::stream_sync(stream_);
}

void SlabEngine::bind_buffers(float* a_main, float* b_main, float* a_nyq, float* b_nyq) {
  sync();
  const bool need_nyq = planA_->nyq_bytes() != 0;
  const bool any = a_main || b_main || a_nyq || b_nyq;
  const bool all = a_main && b_main && (!need_nyq || (a_nyq && b_nyq));
  if (any && !all)
    throw std::invalid_argument("mvn: bind all exchange buffers (main and, for even d2, Nyquist) or none");
  if (all) {
    a_main_ = a_main;
    b_main_ = b_main;
    a_nyq_ = (cfloat*)a_nyq;
    b_nyq_ = (cfloat*)b_nyq;
    external_ = true;
  } else {
    a_main_ = own_a_;
    b_main_ = own_b_;
    a_nyq_ = own_an_;
    b_nyq_ = own_bn_;
    external_ = false;
  }
}

void SlabEngine::upload_slab(float* dst, const float* host) {
  const Layout& L = planA_->L;
  if (L.RP == L.d2)
    be::h2d(dst, host, L.logical() * sizeof(float), stream_);
  else
    be::h2d_2d(dst, (size_t)L.RP * sizeof(float), host, (size_t)L.d2 * sizeof(float),
               (size_t)L.d2 * sizeof(float), L.rows, stream_);
}

void SlabEngine::set_psi(const float* host) {
  be::set_device(device_);
  if (planA_->L.RP != planA_->L.d2) be::dzero(psi_, planA_->main_bytes(), stream_);
  upload_slab(psi_, host);
  be::stream_sync(stream_);
  work_has_psi_spectrum_ = false;
}

void SlabEngine::get_psi(float* host) {
  be::set_device(device_);
  const Layout& L = planA_->L;
  be::d2h_2d(host, (size_t)L.d2 * 4, psi_, (size_t)L.RP * 4, (size_t)L.d2 * 4, L.rows, stream_);
  be::stream_sync(stream_);
}

void SlabEngine::set_view(int v, const float* image, const float* weights, const float* kernel1,
                          const int* k1dims, const float* kernel2, const int* k2dims) {
  if (v < 0 || v >= (int)views_.size()) throw std::out_of_range("mvn: view index");
  be::set_device(device_);
  SlabView& s = views_[(size_t)v];
  const size_t mb = planA_->main_bytes();
  const size_t nbT = planB_->nyq_bytes();
  if (!s.image) {
    s.image = (float*)be::dmalloc(mb);
    s.weights = (float*)be::dmalloc(mb);
    for (int i = 0; i < 2; ++i) {
      s.spec[i] = (float*)be::dmalloc(planB_->main_bytes());
      if (nbT) s.nyq[i] = (cfloat*)be::dmalloc(nbT);
    }
    if (planA_->L.RP != planA_->L.d2) {
      be::dzero(s.image, mb, stream_);
      be::dzero(s.weights, mb, stream_);
    }
  }
  upload_slab(s.image, image);
  upload_slab(s.weights, weights);
  // PSF spectra: the whole 3-D spectrum is formed locally (three passes on a temporary full
  // volume) and this rank's dim1 block cut out of it -- position order along dim1 is the same in
  // the full and in the decomposed transform, so the block is what a decomposed forward
  // transform of the PSF would leave here.
  std::shared_ptr<Plan3D> full = PlanStore::get().add(device_, dims_);
  const Layout& F = full->L;
  const float scale = (float)(1.0 / (double)F.logical());
  float* fspec = (float*)be::dmalloc(full->main_bytes());
  cfloat* fnyq = nullptr;
  float* dk = nullptr;
  try {
    if (full->nyq_bytes()) fnyq = (cfloat*)be::dmalloc(full->nyq_bytes());
    const float* ks[2] = {kernel1, kernel2};
    const int* kd[2] = {k1dims, k2dims};
    const size_t yb = (size_t)(F.d1 / P_);
    for (int i = 0; i < 2; ++i) {
      const size_t kb = sizeof(float) * (size_t)kd[i][0] * (size_t)kd[i][1] * (size_t)kd[i][2];
      dk = (float*)be::dmalloc(kb);
      be::h2d(dk, ks[i], kb, stream_);
      full->psf_spectrum(dk, kd[i], scale, fspec, fnyq, stream_);
      const size_t wrow = yb * (size_t)F.C * sizeof(cfloat);
      be::d2d_2d(s.spec[i], wrow, (const cfloat*)fspec + (size_t)rank_ * yb * F.C,
                 (size_t)F.d1 * F.C * sizeof(cfloat), wrow, (size_t)F.d0, stream_);
      if (fnyq)
        be::d2d_2d(s.nyq[i], yb * sizeof(cfloat), fnyq + (size_t)rank_ * yb,
                   (size_t)F.d1 * sizeof(cfloat), yb * sizeof(cfloat), (size_t)F.d0, stream_);
      be::stream_sync(stream_);
      be::dfree(dk);
      dk = nullptr;
    }
  } catch (...) {
    try {
      be::stream_sync(stream_);
    } catch (...) {
    }
    be::dfree(dk);
    be::dfree(fspec);
    be::dfree(fnyq);
    throw;
  }
  be::dfree(fspec);
  be::dfree(fnyq);
  s.set = true;
}

void SlabEngine::begin_sweeps() { work_has_psi_spectrum_ = false; }

// conv 0 starts from psi (its last-axis transform may already sit in the work buffer, left there
// by the previous view's fused update pass); conv 1 starts from the quotient's last-axis
// transform, left there by step_unpack(conv 0)
void SlabEngine::step_pack(int v, int conv) {
  be::set_device(device_);
  if (!views_.at((size_t)v).set) throw std::runtime_error("mvn: view " + std::to_string(v) + " was never set");
  const Plan3D& A = *planA_;
  const Layout& L = A.L;
  cfloat* W = (cfloat*)work_;
  if (conv == 0) {
    if (!work_has_psi_spectrum_) A.rows_r2c(psi_, W, work_nyq_, stream_);
    work_has_psi_spectrum_ = false;
  }
  A.axis1(MVN_ST_FWD, W, work_nyq_, stream_);
  const size_t yb = (size_t)(L.d1 / P_);
  const size_t wrow = yb * (size_t)L.C * sizeof(cfloat);
  const size_t blk = (size_t)L.d0 * yb * L.C;  // cfloats per destination rank
  for (int j = 0; j < P_; ++j) {
    be::d2d_2d((cfloat*)a_main_ + (size_t)j * blk, wrow, W + (size_t)j * yb * L.C,
               (size_t)L.d1 * L.C * sizeof(cfloat), wrow, (size_t)L.d0, stream_);
    if (work_nyq_)
      be::d2d_2d(a_nyq_ + (size_t)j * L.d0 * yb, yb * sizeof(cfloat), work_nyq_ + (size_t)j * yb,
                 (size_t)L.d1 * sizeof(cfloat), yb * sizeof(cfloat), (size_t)L.d0, stream_);
  }
}

void SlabEngine::step_mid(int v, int conv) {
  be::set_device(device_);
  const SlabView& s = views_.at((size_t)v);
  planB_->axis0(MVN_ST_FWD_MUL_INV, (cfloat*)b_main_, b_nyq_, (const cfloat*)s.spec[conv],
                s.nyq[conv], stream_);
}

void SlabEngine::step_unpack(int v, int conv, double lambda, float min_value, bool feed_next) {
  be::set_device(device_);
  const SlabView& s = views_.at((size_t)v);
  const Plan3D& A = *planA_;
  const Layout& L = A.L;
  cfloat* W = (cfloat*)work_;
  const size_t yb = (size_t)(L.d1 / P_);
  const size_t wrow = yb * (size_t)L.C * sizeof(cfloat);
  const size_t blk = (size_t)L.d0 * yb * L.C;
  for (int j = 0; j < P_; ++j) {
    be::d2d_2d(W + (size_t)j * yb * L.C, (size_t)L.d1 * L.C * sizeof(cfloat),
               (const cfloat*)a_main_ + (size_t)j * blk, wrow, wrow, (size_t)L.d0, stream_);
    if (work_nyq_)
      be::d2d_2d(work_nyq_ + (size_t)j * yb, (size_t)L.d1 * sizeof(cfloat),
                 a_nyq_ + (size_t)j * L.d0 * yb, yb * sizeof(cfloat), yb * sizeof(cfloat),
                 (size_t)L.d0, stream_);
  }
  A.axis1(MVN_ST_INV, W, work_nyq_, stream_);
  EpilogueParams e;
  std::memset(&e, 0, sizeof(e));
  e.scale = 1.f;  // 1/N lives in the PSF spectra
  if (conv == 0) {
    e.mode = MVN_EPI_DIVIDE;
    e.view = s.image;
    if (A.can_fuse_rows()) {
      A.rows_c2r_r2c(W, work_nyq_, e, stream_);
    } else {
      A.rows_c2r(W, work_nyq_, work_, e, stream_);
      A.rows_r2c(work_, W, work_nyq_, stream_);
    }
  } else {
    e.mode = MVN_EPI_UPDATE;
    e.psi = psi_;
    e.weights = s.weights;
    e.lambda = lambda;
    e.lambda_inv = lambda > 0 ? (float)(1.f / lambda) : 0.f;
    e.min_value = min_value;
    if (A.can_fuse_rows() && feed_next) {
      A.rows_c2r_r2c(W, work_nyq_, e, stream_);
      work_has_psi_spectrum_ = true;
    } else {
      A.rows_c2r(W, work_nyq_, psi_, e, stream_);
    }
  }
}

}  // namespace mvn
