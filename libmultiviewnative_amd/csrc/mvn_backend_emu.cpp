// mvn_backend_emu.cpp -- TEST-ONLY host emulation of the device backend (-DMVN_HOST_EMU).
//
// Built into libmvn_emu.so, never into the product library.  "Device memory" is host memory;
// a launch runs the very same workgroup bodies (mvn_pass_bodies.hpp) one block after the other
// with a single work-item enumerator per block, so plans, tables, addressing and the RL driver
// can be validated against the oracle and numpy on a box without a GPU.  It says nothing about
// barriers or races -- those are covered by the -m gpu parity tests.
#ifndef MVN_HOST_EMU
#error "mvn_backend_emu.cpp is only for the MVN_HOST_EMU test build"
#endif

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "mvn_backend.hpp"
#include "mvn_fixed_geom.hpp"
#include "mvn_wave_rows.hpp"

namespace mvn {
namespace be {

#define MVN_DISPATCH_T(T_, CALL)                                   \
  switch (T_) {                                                    \
    case 16: { constexpr int TT = 16; CALL; } break;               \
    case 8: { constexpr int TT = 8; CALL; } break;                 \
    case 4: { constexpr int TT = 4; CALL; } break;                 \
    case 2: { constexpr int TT = 2; CALL; } break;                 \
    case 1: { constexpr int TT = 1; CALL; } break;                 \
    default: throw std::invalid_argument("mvn: unsupported tile width"); \
  }


const char* backend_name() { return "host-emulation (test only)"; }

// MVN_EMU_DEVICES=n pretends to have n devices (tests of the per-device engine cache and locking);
// the current device is per host thread, as in HIP
int device_count() {
  const char* e = std::getenv("MVN_EMU_DEVICES");
  const int n = e ? std::atoi(e) : 1;
  return n > 0 ? n : 1;
}
static thread_local int t_device = 0;
void set_device(int d) {  // as strict as hipSetDevice: a lane key or a stale id must not pass here either
  if (d < 0 || d >= device_count()) throw std::runtime_error("emu: invalid device ordinal " + std::to_string(d));
  t_device = d;
}
int get_device() { return t_device; }
void device_name(int, char* name256) {
  std::memset(name256, 0, 256);
  std::strcpy(name256, "mvn host emulation");
}
// "device memory" accounting per fake device: total = MVN_EMU_TOTAL_MB (default 8 GiB), free =
// total - live dmalloc bytes, so that the memory heuristic of the ABI call can be tested
static std::mutex g_mem_mu;
static std::map<void*, std::pair<int, size_t>> g_live;  // pointer -> (device, bytes)
static std::map<int, size_t> g_used;
static size_t emu_total() {
  const char* e = std::getenv("MVN_EMU_TOTAL_MB");
  return e ? (size_t)std::atoll(e) << 20 : (size_t)8 << 30;
}
long long device_total_mem(int) { return (long long)emu_total(); }
void device_mem_info(size_t* f, size_t* t) {
  std::lock_guard<std::mutex> lk(g_mem_mu);
  const size_t total = emu_total(), used = g_used[t_device];
  *t = total;
  *f = used < total ? total - used : 0;
}
void device_arch(int, int* major, int* minor) {
  *major = 0;
  *minor = 0;
}

void* dmalloc(size_t bytes) {
  void* p = std::malloc(bytes ? bytes : 1);
  if (!p) throw std::bad_alloc();
  std::lock_guard<std::mutex> lk(g_mem_mu);
  g_live[p] = std::make_pair(t_device, bytes);
  g_used[t_device] += bytes;
  return p;
}
void dfree(void* p) {
  if (!p) return;
  {
    std::lock_guard<std::mutex> lk(g_mem_mu);
    auto it = g_live.find(p);
    if (it != g_live.end()) {
      g_used[it->second.first] -= it->second.second;
      g_live.erase(it);
    }
  }
  std::free(p);
}
void h2d(void* d, const void* h, size_t bytes, stream_t) { std::memcpy(d, h, bytes); }
void d2h(void* h, const void* d, size_t bytes, stream_t) { std::memcpy(h, d, bytes); }
void d2d(void* dst, const void* src, size_t bytes, stream_t) { std::memmove(dst, src, bytes); }
void h2d_2d(void* d, size_t dpitch, const void* h, size_t hpitch, size_t width, size_t height,
            stream_t) {
  for (size_t r = 0; r < height; ++r)
    std::memcpy((char*)d + r * dpitch, (const char*)h + r * hpitch, width);
}
void d2h_2d(void* h, size_t hpitch, const void* d, size_t dpitch, size_t width, size_t height,
            stream_t) {
  for (size_t r = 0; r < height; ++r)
    std::memcpy((char*)h + r * hpitch, (const char*)d + r * dpitch, width);
}
void d2d_2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height,
            stream_t) {
  for (size_t r = 0; r < height; ++r)
    std::memmove((char*)dst + r * dpitch, (const char*)src + r * spitch, width);
}
void dzero(void* d, size_t bytes, stream_t) { std::memset(d, 0, bytes); }
void enable_peer_access(int dev, int peer) {
  if (dev < 0 || dev >= device_count() || peer < 0 || peer >= device_count()) throw std::runtime_error("emu: invalid peer");
}
void copy_peer(void* dst, int, const void* src, int, size_t bytes, stream_t) { std::memmove(dst, src, bytes); }

stream_t stream_create() { return (stream_t)1; }
stream_t stream_create_upload() { return (stream_t)1; }
void stream_destroy(stream_t) {}
void stream_sync(stream_t) {}
void stream_wait_event(stream_t, event_t) {}

struct EmuEvent {
  std::chrono::steady_clock::time_point t;
};
event_t event_create() { return new EmuEvent(); }
event_t event_create_sync() { return new EmuEvent(); }
void event_destroy(event_t e) { delete (EmuEvent*)e; }
void event_record(event_t e, stream_t) { ((EmuEvent*)e)->t = std::chrono::steady_clock::now(); }
void event_sync(event_t) {}
float event_elapsed_ms(event_t a, event_t b) {
  return std::chrono::duration<float, std::milli>(((EmuEvent*)b)->t - ((EmuEvent*)a)->t).count();
}

bool graphs_supported() { return false; }
void capture_begin(stream_t) { throw std::runtime_error("mvn: no graphs in the host emulation"); }
graph_exec_t capture_end(stream_t) { throw std::runtime_error("mvn: no graphs in the host emulation"); }
void graph_launch(graph_exec_t, stream_t) { throw std::runtime_error("mvn: no graphs in the host emulation"); }
void graph_destroy(graph_exec_t) {}

// wave-row kernels (d2 = 512, mvn_wave_rows.hpp): a "grid" of about a third of the workgroups a
// one-sweep launch would have, so that the sweep loop and its ragged tail are exercised
// same knobs as the HIP backend, but read at every launch and with every pass enabled by default
// (the emulation exists to test them)
static bool emu_wave_rows(const RowsParams& p, int kind_bit) {
  const char* e = std::getenv("MVN_NO_WAVE_ROWS");
  const char* m = std::getenv("MVN_WAVE_ROWS_MASK");
  const int mask = m && *m ? std::atoi(m) : 31;
  return !(e && *e && std::strcmp(e, "0") != 0) && (mask & kind_bit) && p.fixed && !p.lines && p.h == WrCfg::H &&
         p.C == WrCfg::H;
}

template <int MODE, int EPI>
static void emu_wave_rows_run(const RowsParams& p) {
  typedef FxCtx<WrRegs, WrCfg::NT> Ctx;
  const long pairs = (p.rows + 1) / 2;
  const long full = (pairs + WrCfg::WAVES - 1) / WrCfg::WAVES;
  const long grid = full > 2 ? (full + 2) / 3 : full;
#pragma omp parallel
  {
    std::vector<char> lds(sizeof(cfloat) * WrCfg::lds_cfloats + 64);
    std::unique_ptr<Ctx> ctx(new Ctx());
#pragma omp for schedule(static)
    for (long b = 0; b < grid; ++b) wr_rows_body<MODE, EPI>(p, b, grid, (cfloat*)lds.data(), *ctx);
  }
}

template <int MODE>
static void emu_wave_rows_mode(const RowsParams& p) {
  switch (p.epi.mode) {
    case MVN_EPI_DIVIDE: emu_wave_rows_run<MODE, MVN_EPI_DIVIDE>(p); break;
    case MVN_EPI_UPDATE: emu_wave_rows_run<MODE, MVN_EPI_UPDATE>(p); break;
    case MVN_EPI_DELTA:
      if (MODE == MVN_WR_C2R) {
        emu_wave_rows_run<MVN_WR_C2R, MVN_EPI_DELTA>(p);
        break;
      }
      throw std::invalid_argument("mvn: DELTA epilogue only in the plain c2r pass");
    default: emu_wave_rows_run<MODE, MVN_EPI_STORE>(p); break;
  }
}

// fixed-length kernels: every phase is run for all thread ids in turn (real thread mapping)
// "grid" of a fixed last-axis launch: one workgroup per tile, or for the walking configurations
// about a third as many, so that the tile loop is exercised
template <int H>
static long emu_rows_grid(long ntiles) {
  return FxRowsCfg<H>::WALK && ntiles > 2 ? (ntiles + 2) / 3 : ntiles;
}

// line-layout forms of the fixed last-axis kernels (H = 256), KIND 0 r2c, 1 c2r, 2 fused
static void emu_rows_lines(const RowsParams& p, long ntiles, int kind) {
  constexpr int H = 256;
  typedef FxCtx<FxRowsRegs<H>, FxRowsCfg<H>::NT> Ctx;
  if (!fx_rows_lines_ok<H>() || !p.fixed || p.h != H || p.C != H || !p.nyq_packed || p.lines_d1 < 1 ||
      p.lines_d1 % FxRowsCfg<H>::T || p.row_base % FxRowsCfg<H>::T || p.rows != ntiles * FxRowsCfg<H>::T)
    throw std::invalid_argument("mvn: line-layout last-axis pass outside its range");
#pragma omp parallel
  {
    std::vector<char> lds(sizeof(cfloat) * FxRowsCfg<H>::lds_cfloats + 64);
    std::unique_ptr<Ctx> ctx(new Ctx());
#pragma omp for schedule(static)
    for (long t = 0; t < ntiles; ++t) {
      cfloat* l = (cfloat*)lds.data();
      if (kind == 0) {
        fx_rows_run<H, 0, MVN_EPI_STORE, Ctx, true>(p, t, ntiles, l, *ctx);
      } else if (kind == 1) {
        switch (p.epi.mode) {
          case MVN_EPI_DIVIDE: fx_rows_run<H, 1, MVN_EPI_DIVIDE, Ctx, true>(p, t, ntiles, l, *ctx); break;
          case MVN_EPI_UPDATE: fx_rows_run<H, 1, MVN_EPI_UPDATE, Ctx, true>(p, t, ntiles, l, *ctx); break;
          case MVN_EPI_DELTA: fx_rows_run<H, 1, MVN_EPI_DELTA, Ctx, true>(p, t, ntiles, l, *ctx); break;
          default: fx_rows_run<H, 1, MVN_EPI_STORE, Ctx, true>(p, t, ntiles, l, *ctx); break;
        }
      } else {
        switch (p.epi.mode) {
          case MVN_EPI_DIVIDE: fx_rows_run<H, 2, MVN_EPI_DIVIDE, Ctx, true>(p, t, ntiles, l, *ctx); break;
          case MVN_EPI_UPDATE: fx_rows_run<H, 2, MVN_EPI_UPDATE, Ctx, true>(p, t, ntiles, l, *ctx); break;
          default: fx_rows_run<H, 2, MVN_EPI_STORE, Ctx, true>(p, t, ntiles, l, *ctx); break;
        }
      }
    }
  }
}

static std::atomic<long> g_mid_fused_launches{0};
long mid_fused_launch_count() { return g_mid_fused_launches.load(); }

// the fused middle pass (mvn_mid_fused.hpp), one workgroup after the other
template <int K>
static void emu_mid_fused(const MidFusedParams& p) {
  typedef FxCtx<MfRegs<K>, MF_NT> Ctx;
  const long blocks = mf_blocks(p);
#pragma omp parallel
  {
    std::vector<cfloat> lds(MF_LDS_CFLOATS + 8);
    std::unique_ptr<Ctx> ctx(new Ctx());
#pragma omp for schedule(static)
    for (long b = 0; b < blocks; ++b) mf_body<K>(p, b, lds.data(), *ctx);
  }
}
#define MVN_MF_TAP_COUNTS(X) X(1) X(3) X(5) X(7) X(9) X(11) X(13) X(15) X(17) X(19) X(21) X(23) X(25) X(27) X(29) X(31)
void launch_mid_fused(const MidFusedParams& p, stream_t) {
  mf_check(p);
  if (p.mode == MF_TAPS) {
    typedef FxCtx<MfRegs<1>, MF_NT> Ctx;
    const long blocks = mf_blocks(p);
#pragma omp parallel
    {
      std::vector<cfloat> lds(MF_LDS_CFLOATS + 8);
      std::unique_ptr<Ctx> ctx(new Ctx());
#pragma omp for schedule(static)
      for (long b = 0; b < blocks; ++b) mf_taps_body(p, b, lds.data(), *ctx);
    }
    return;
  }
  ++g_mid_fused_launches;
  switch (mvn_dim0_taps_template(p.k)) {
#define X(K) case K: emu_mid_fused<K>(p); break;
    MVN_MF_TAP_COUNTS(X)
#undef X
    default: throw std::invalid_argument("mvn: no fused middle pass for this tap count");
  }
}

template <int H>
static void emu_rows_fused(const RowsParams& p, long ntiles) {
  typedef FxCtx<FxRowsRegs<H>, FxRowsCfg<H>::NT> Ctx;
  const long grid = emu_rows_grid<H>(ntiles);
#pragma omp parallel
  {
    std::vector<char> lds(sizeof(cfloat) * FxRowsCfg<H>::lds_cfloats + 64);
    std::unique_ptr<Ctx> ctx(new Ctx());
#pragma omp for schedule(static)
    for (long t = 0; t < grid; ++t) {
      cfloat* l = (cfloat*)lds.data();
      switch (p.epi.mode) {
        case MVN_EPI_DIVIDE: fx_rows_run<H, 2, MVN_EPI_DIVIDE>(p, t, grid, l, *ctx); break;
        case MVN_EPI_UPDATE: fx_rows_run<H, 2, MVN_EPI_UPDATE>(p, t, grid, l, *ctx); break;
        default: fx_rows_run<H, 2, MVN_EPI_STORE>(p, t, grid, l, *ctx); break;
      }
    }
  }
}

void launch_rows_c2r_r2c(const RowsParams& p0, long ntiles, int, size_t lds_bytes, stream_t) {
  RowsParams p = p0;
  mvn_arm_poison(p.epi);  // (the device kernels do this at their entry)
  if (p.lines) return emu_rows_lines(p, ntiles, 2);
  if (emu_wave_rows(p, p.epi.mode == MVN_EPI_DIVIDE ? 4 : 8)) return emu_wave_rows_mode<MVN_WR_C2R_R2C>(p);
  if (!p.fixed) {  // run-time-radix form of the fused pass (any even d2)
#pragma omp parallel
    {
      std::vector<char> lds(lds_bytes + 64);
#pragma omp for schedule(static)
      for (long t = 0; t < ntiles; ++t) {
        cfloat* l = (cfloat*)lds.data();
        MVN_DISPATCH_T(p.T, (rows_c2r_even_body<TT, true>(p, t, 0, 1, l)));
      }
    }
    return;
  }
  switch (p.h) {
#define X(H) case H: emu_rows_fused<H>(p, ntiles); return;
    MVN_FIXED_ROWS_LENGTHS(X)
#undef X
    default: throw std::invalid_argument("mvn: no fixed kernel");
  }
}

template <int H>
static void emu_rows_fixed(const RowsParams& p, long ntiles, bool r2c) {
  typedef FxCtx<FxRowsRegs<H>, FxRowsCfg<H>::NT> Ctx;
  const long grid = emu_rows_grid<H>(ntiles);
#pragma omp parallel
  {
    std::vector<char> lds(sizeof(cfloat) * FxRowsCfg<H>::lds_cfloats + 64);
    std::unique_ptr<Ctx> ctx(new Ctx());
#pragma omp for schedule(static)
    for (long t = 0; t < grid; ++t) {
      cfloat* l = (cfloat*)lds.data();
      if (r2c)
        fx_rows_run<H, 0, MVN_EPI_STORE>(p, t, grid, l, *ctx);
      else {
        switch (p.epi.mode) {
          case MVN_EPI_DIVIDE: fx_rows_run<H, 1, MVN_EPI_DIVIDE>(p, t, grid, l, *ctx); break;
          case MVN_EPI_UPDATE: fx_rows_run<H, 1, MVN_EPI_UPDATE>(p, t, grid, l, *ctx); break;
          case MVN_EPI_DELTA: fx_rows_run<H, 1, MVN_EPI_DELTA>(p, t, grid, l, *ctx); break;
          default: fx_rows_run<H, 1, MVN_EPI_STORE>(p, t, grid, l, *ctx); break;
        }
      }
    }
  }
}

template <int N, int MODE>
static void emu_strided_fixed_mode(const StridedParams& p0, long nblocks) {
  typedef FxStridedSel<N, MODE> Sel;
  typedef typename Sel::Ctx Ctx;
  StridedParams p = p0;
  p.nblocks = nblocks;
  // a "grid" of about half as many workgroups as tiles, so that the tile loop and its
  // prefetching are exercised (each workgroup walks over 1-2 tiles)
  const long grid = nblocks > 1 ? (nblocks + 1) / 2 : 1;
#pragma omp parallel
  {
    std::vector<char> lds(sizeof(cfloat) * FxStridedCfg<N>::lds_cfloats + 64);
    std::unique_ptr<Ctx> ctx(new Ctx());
#pragma omp for schedule(static)
    for (long b = 0; b < grid; ++b) Sel::run(p, b, nblocks, grid, (cfloat*)lds.data(), *ctx);
  }
}

static std::atomic<long> g_split_launches{0};
long split_launch_count() { return g_split_launches.load(); }

template <int N>
static bool emu_try_split(int mode, const StridedParams& p, long nblocks) {
  if constexpr (FxSplitCfg<N>::USE) {
    typedef FxSplitCfg<N> C;
    const char* e = std::getenv("MVN_NO_SPLIT");
    if ((e && *e && std::strcmp(e, "0") != 0) || mode == MVN_ST_FWD_MUL_INV || p.cstride != 1 ||
        p.ncols % C::T != 0 || p.tiles_per_outer < 1)
      return false;
    StridedParams q = p;
    const long outer = nblocks / p.tiles_per_outer;
    q.tiles_per_outer = p.ncols / C::T;
    q.T = q.TP = C::T;
    const long nb = outer * q.tiles_per_outer;
    q.nblocks = nb;
    ++g_split_launches;
    const long grid = nb > 1 ? (nb + 1) / 2 : 1;
    typedef FxCtx<FxSplitRegs<N>, C::NT> Ctx;
#pragma omp parallel
    {
      std::vector<char> lds(sizeof(cfloat) * C::lds_cfloats + 64);
      std::unique_ptr<Ctx> ctx(new Ctx());
#pragma omp for schedule(static)
      for (long b = 0; b < grid; ++b) {
        if (mode == MVN_ST_FWD)
          fx_strided_split_body<N, MVN_ST_FWD>(q, b, nb, grid, (cfloat*)lds.data(), *ctx);
        else
          fx_strided_split_body<N, MVN_ST_INV>(q, b, nb, grid, (cfloat*)lds.data(), *ctx);
      }
    }
    return true;
  } else {
    (void)mode; (void)p; (void)nblocks;
    return false;
  }
}

template <int N>
static void emu_strided_fixed(int mode, const StridedParams& p, long nblocks) {
  if (emu_try_split<N>(mode, p, nblocks)) return;
  if (mode == MVN_ST_FWD) emu_strided_fixed_mode<N, MVN_ST_FWD>(p, nblocks);
  if (mode == MVN_ST_INV) emu_strided_fixed_mode<N, MVN_ST_INV>(p, nblocks);
  if (mode == MVN_ST_FWD_MUL_INV) emu_strided_fixed_mode<N, MVN_ST_FWD_MUL_INV>(p, nblocks);
}

static bool emu_rows_fixed_dispatch(const RowsParams& p, long ntiles, bool r2c) {
  switch (p.h) {
#define X(H) case H: emu_rows_fixed<H>(p, ntiles, r2c); return true;
    MVN_FIXED_ROWS_LENGTHS(X)
#undef X
    default: return false;
  }
}

static bool emu_strided_fixed_dispatch(int mode, const StridedParams& p, long nblocks) {
  switch (p.ax.n) {
#define X(N) case N: emu_strided_fixed<N>(mode, p, nblocks); return true;
    MVN_FIXED_STRIDED_LENGTHS(X)
#undef X
    default: return false;
  }
}

void launch_rows_r2c(const RowsParams& p, bool even, long ntiles, int, size_t lds_bytes,
                     stream_t) {
  if (p.lines) return emu_rows_lines(p, ntiles, 0);
  if (emu_wave_rows(p, 1)) return emu_wave_rows_run<MVN_WR_R2C, MVN_EPI_STORE>(p);
  if (p.fixed) {
    if (!emu_rows_fixed_dispatch(p, ntiles, true)) throw std::invalid_argument("mvn: no fixed kernel");
    return;
  }
#pragma omp parallel
  {
    std::vector<char> lds(lds_bytes + 64);
#pragma omp for schedule(static)
    for (long t = 0; t < ntiles; ++t) {
      cfloat* l = (cfloat*)lds.data();
      if (even) {
        MVN_DISPATCH_T(p.T, rows_r2c_even_body<TT>(p, t, 0, 1, l));
      } else {
        MVN_DISPATCH_T(p.T, rows_r2c_odd_body<TT>(p, t, 0, 1, l));
      }
    }
  }
}

void launch_rows_c2r(const RowsParams& p0, bool even, long ntiles, int, size_t lds_bytes,
                     stream_t) {
  RowsParams p = p0;
  mvn_arm_poison(p.epi);
  if (p.lines) return emu_rows_lines(p, ntiles, 1);
  if (emu_wave_rows(p, p.epi.mode == MVN_EPI_DELTA ? 16 : 2)) return emu_wave_rows_mode<MVN_WR_C2R>(p);
  if (p.fixed) {
    if (!emu_rows_fixed_dispatch(p, ntiles, false)) throw std::invalid_argument("mvn: no fixed kernel");
    return;
  }
#pragma omp parallel
  {
    std::vector<char> lds(lds_bytes + 64);
#pragma omp for schedule(static)
    for (long t = 0; t < ntiles; ++t) {
      cfloat* l = (cfloat*)lds.data();
      if (even) {
        MVN_DISPATCH_T(p.T, rows_c2r_even_body<TT>(p, t, 0, 1, l));
      } else {
        MVN_DISPATCH_T(p.T, rows_c2r_odd_body<TT>(p, t, 0, 1, l));
      }
    }
  }
}

void launch_strided(int mode, const StridedParams& p, long nblocks, int, size_t lds_bytes,
                    stream_t, const StridedParams* rider, size_t rider_lds) {
  if (rider && (!p.fixed || mode == MVN_ST_FWD_MUL_INV || rider->T != 1 || rider->fixed || rider_lds > 160 * 1024))
    throw std::invalid_argument("mvn: Nyquist lines ride only in the plain fixed-length strided passes, one per workgroup");
  if (p.fixed) {
    if (!emu_strided_fixed_dispatch(mode, p, nblocks)) throw std::invalid_argument("mvn: no fixed kernel");
    if (rider) {  // the workgroups behind the walkers: one line each
      const long lines = rider->tiles_per_outer;
#pragma omp parallel
      {
        std::vector<char> lds(rider_lds + 64);
#pragma omp for schedule(static)
        for (long b = 0; b < lines; ++b) {
          cfloat* l = (cfloat*)lds.data();
          if (mode == MVN_ST_FWD)
            strided_body<MVN_ST_FWD, 1, true>(*rider, b, 0, 1, l);
          else
            strided_body<MVN_ST_INV, 1, true>(*rider, b, 0, 1, l);
        }
      }
    }
    return;
  }
#pragma omp parallel
  {
    std::vector<char> lds(lds_bytes + 64);
#pragma omp for schedule(static)
    for (long b = 0; b < nblocks; ++b) {
      cfloat* l = (cfloat*)lds.data();
      if (mode == MVN_ST_FWD) { MVN_DISPATCH_T(p.T, (strided_body<MVN_ST_FWD, TT, true>(p, b, 0, 1, l))); }
      if (mode == MVN_ST_INV) { MVN_DISPATCH_T(p.T, (strided_body<MVN_ST_INV, TT, true>(p, b, 0, 1, l))); }
      if (mode == MVN_ST_FWD_MUL_INV) { MVN_DISPATCH_T(p.T, (strided_body<MVN_ST_FWD_MUL_INV, TT, true>(p, b, 0, 1, l))); }
    }
  }
}

void launch_dim0_direct(const Dim0DirectParams& p, stream_t) {
  if (!mvn_dim0_direct_possible(p.k, p.d0) || p.kd < p.k + 1 || p.h != p.k / 2 || p.plane < 0 || p.plane + p.plane2 < 1 || (p.plane > 0 && p.in == p.out) ||
      p.plane2 < 0 || (p.plane2 > 0 && (!p.in2 || !p.out2 || !p.taps2 || p.in2 == p.out2)))
    throw std::invalid_argument("mvn: direct dim0 convolution called outside its range");
  if (p.packed) {
    if (!p.inv1 || !p.taps2 || (long)p.C * p.d1 != p.plane)
      throw std::invalid_argument("mvn: packed direct dim0 convolution needs the dim1 tables and the Nyquist taps");
    if (mvn_dim0_dc_lds_bytes(p.d0, p.k) > 64 * 1024)  // the device launch's limit
      throw std::invalid_argument("mvn: dim0 too long for the packed DC column");
#pragma omp parallel
    {
      std::vector<cfloat> lds(2 * (size_t)p.d0 + 2 * (size_t)p.k);
#pragma omp for schedule(static)
      for (int pair = 0; pair < mvn_dim0_pairs(p.d1); ++pair) {
        for (int t = 0; t < 256; ++t) mvn_dim0_dc_load(p, pair, lds.data(), t, 256);
        for (int t = 0; t < 256; ++t) mvn_dim0_dc_compute(p, pair, lds.data(), t, 256);
      }
    }
  }
  switch (mvn_dim0_taps_template(p.k)) {
#define X(K)                                                                        \
  case K: {                                                                         \
    const long nblocks = mvn_dim0_blocks(p).blocks;                                 \
    _Pragma("omp parallel for schedule(static)")                                    \
    for (long blk = 0; blk < nblocks; ++blk)                                        \
      for (int t = 0; t < MVN_D0_WG; ++t) {                                         \
        Dim0DirectParams q;                                                         \
        long b;                                                                     \
        int zs, nout;                                                               \
        if (mvn_dim0_job(p, blk, t, q, b, zs, nout)) mvn_dim0_direct_column<K, MVN_D0_PF>(q, b, zs, nout); \
      }                                                                             \
  } break;
    MVN_D0_TAP_COUNTS(X)
#undef X
    default: throw std::invalid_argument("mvn: no direct dim0 kernel for this tap count");
  }
}

void launch_scatter_psf(const float* kernel, int k0, int k1, int k2, float* target, int D0,
                        int D1, int D2, long pitch, float scale, stream_t) {
  const long total = (long)k0 * k1 * k2;
  for (long i = 0; i < total; ++i) mvn_scatter_psf_item(kernel, k0, k1, k2, target, D0, D1, D2, pitch, scale, i);
}

void launch_copy3d(float* dst, long drow, long dplane, const float* src, long srow, long splane,
                   int nx, int ny, int nz, stream_t) {
  for (long z = 0; z < nz; ++z)
    for (long y = 0; y < ny; ++y)
      std::memcpy(dst + z * dplane + y * drow, src + z * splane + y * srow, sizeof(float) * (size_t)nx);
}

void launch_divide(const float* view, float* inout, size_t n, stream_t) {
  for (size_t i = 0; i < n; ++i) inout[i] = mvn_quotient(view[i], inout[i]);
}

void launch_update(float* psi, const float* integral, const float* weights, size_t n,
                   double lambda, float min_value, stream_t) {
  const float linv = lambda > 0 ? (float)(1.f / lambda) : 0.f;
  for (size_t i = 0; i < n; ++i) {
    float last = psi[i];
    float next = mvn_next_value(last, integral[i], lambda, linv, min_value);
    psi[i] = weights[i] * (next - last) + last;
  }
}

void launch_update_legacy_tikhonov(float* image, const float* integral, const float* weights,
                                   size_t n, float lambda_f, float min_value, stream_t) {
  for (size_t i = 0; i < n; ++i)
    image[i] = mvn_legacy_tikhonov_value(image[i], integral[i], weights[i], lambda_f, min_value);
}

void launch_axpy1(float* psi, const float* delta, size_t n, stream_t) {
  for (size_t i = 0; i < n; ++i) psi[i] += delta[i];
}

}  // namespace be
}  // namespace mvn
