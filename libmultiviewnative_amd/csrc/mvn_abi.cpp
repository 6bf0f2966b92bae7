// mvn_abi.cpp -- extern "C" boundary: the reference ABI (include/multiviewnative.h) and the
// engine ABI (include/mvn_engine_api.h).  Nothing escapes as an exception; failures leave the
// caller's in/out buffers untouched, print one diagnostic to stderr and never terminate the
// host process (SURVEY.md 8b "Errors").
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mvn_engine_api.h"
#include "mvn_engine.hpp"
#include "mvn_fixed_geom.hpp"
#include "mvn_multi.hpp"

using namespace mvn;

struct mvn_engine {
  std::unique_ptr<Engine> impl;
};

struct mvn_slab {
  std::unique_ptr<SlabEngine> impl;
};

struct mvn_group {
  std::unique_ptr<HaloGroup> impl;
};

static thread_local std::string g_last_error;

static bool trace_on() {
  const char* t = std::getenv("MVN_TRACE");
  return t && *t && std::strcmp(t, "0") != 0;
}

template <typename F>
static int guarded(const char* where, F&& f) {
  try {
    f();
    return 0;
  } catch (const std::exception& e) {
    g_last_error = std::string(where) + ": " + e.what();
  } catch (...) {
    g_last_error = std::string(where) + ": unknown failure";
  }
  std::fprintf(stderr, "[libmultiviewnative] %s\n", g_last_error.c_str());
  return -1;
}

// one ABI call at a time per device (the reference is not re-entrant, SURVEY.md 8b "Threading")
static std::mutex& device_mutex(int device) {
  static std::mutex table_mu;
  static std::map<int, std::unique_ptr<std::mutex>> table;
  std::lock_guard<std::mutex> lk(table_mu);
  auto& m = table[device];
  if (!m) m.reset(new std::mutex());
  return *m;
}

// One resident engine per device is kept between inplace_gpu_deconvolve calls: Fiji deconvolves
// block after block of the same shape, and allocating / freeing 4V+2 volumes per call costs more
// than uploading them.  A call with another shape or view count replaces the
// cached engine.  MVN_ENGINE_CACHE=0 disables the cache; mvn_release_cached_engines() empties it.
// The map is process-wide while calls are serialised per DEVICE only (Fiji runs one host thread
// per GPU), so it has its own mutex, held just around find / erase / insert.
static std::mutex& engine_cache_mutex() {
  static std::mutex* m = new std::mutex();
  return *m;
}
static std::map<int, std::unique_ptr<Engine>>& engine_cache() {
  static std::map<int, std::unique_ptr<Engine>>* c = new std::map<int, std::unique_ptr<Engine>>();
  return *c;
}

static bool engine_cache_enabled() {
  const char* e = std::getenv("MVN_ENGINE_CACHE");
  return !(e && std::strcmp(e, "0") == 0);
}

// caller holds device_mutex(dev)
static std::unique_ptr<Engine> pop_cached_engine(int dev) {
  std::lock_guard<std::mutex> lk(engine_cache_mutex());
  auto& c = engine_cache();
  auto it = c.find(dev);
  if (it == c.end()) return nullptr;
  std::unique_ptr<Engine> e = std::move(it->second);
  c.erase(it);
  return e;
}

// caller holds device_mutex(dev).  The memory heuristic of src/multiviewnative.cu:94-119, restated
// for the resident layout (4 volumes per view -- view, weights, two spectra -- + psi + work + the
// spectrum scratch of the PSF preparation, + the host-shaped embedding scratch of the padded
// policies, + 2 % slack), is applied only when the call has to allocate: a cached engine of the same
// shape is re-used as it is (its memory is what the check would ask for), and a stale one of another
// shape is freed BEFORE the free memory is read -- so that "does not fit" is said here, before any
// work is queued, not by a failing hipMalloc on the staging thread.
static std::unique_ptr<Engine> take_engine(int key, int dev, const shape_t& ext, int V, size_t embed_floats) {
  std::unique_ptr<Engine> e = pop_cached_engine(key);  // key = device + lane * kLaneStride
  if (e && engine_cache_enabled()) {
    const Layout& L = e->layout();
    if (L.d0 == ext[0] && L.d1 == ext[1] && L.d2 == ext[2] && e->num_views() == V) return e;
  }
  e.reset();  // wrong shape: free its memory before the new engine allocates
  Layout L(ext[0], ext[1], ext[2]);
  const double need = ((4.0 * V + 3.0) * (double)L.B() + 4.0 * (double)embed_floats) * 1.02;
  size_t free_b = 0, total_b = 0;
  be::device_mem_info(&free_b, &total_b);
  if (trace_on())
    std::printf("[lmvn::inplace_gpu_deconvolve] FFT: %.1f MB (all-on-device), available on GPU: %.1f MB ... %s\n",
                need / 1048576.0, free_b / 1048576.0, need < free_b ? "all on device!" : "does not fit");
  if (need >= (double)free_b)
    throw std::runtime_error("FFT: Unable to run on GPU due to memory constraints");
  return std::unique_ptr<Engine>(new Engine(dev, ext, V));
}

static void give_back_engine(int dev, std::unique_ptr<Engine> e) {
  if (!engine_cache_enabled()) return;
  std::unique_ptr<Engine> old;  // destroyed outside the lock
  {
    std::lock_guard<std::mutex> lk(engine_cache_mutex());
    auto& slot = engine_cache()[dev];
    old = std::move(slot);
    slot = std::move(e);
  }
}

// ---------------------------------------------------------------------------------------------
// padding policy of inplace_gpu_deconvolve
//   MVN_PAD_ZERO (default)  the reference GPU entry's zero_padd (src/multiviewnative.cu:26-27,128;
//                           inc/padd_utils.h:121-138; src/gpu_deconvolve_methods.cuh:366-449,
//                           537-549): every stack is embedded at offset (kernel-1)/2 in a zero
//                           volume of extent >= image + kernel - 1 (maxima over views and both
//                           kernels), the loop runs cyclically on that volume, psi is cropped back
//                           on exit.  The padded extents grow to FFT-friendly sizes (see
//                           good_extent); the quotient is guarded (view == 0 -> 0) because the
//                           extra zeros lie beyond the PSF's reach.
//   MVN_PAD_ZERO_EXACT      the same with exactly image + kernel - 1 (the reference's extents; a
//                           prime factor such as 542 = 2 * 271 then goes the chirp-z route)
//   MVN_PAD_NONE            the reference CPU path's no_padd: cyclic on exactly image_dims_
//                           (inc/cpu_convolve.h:22-26) -- the parity target of the oracle tests
// Selected by mvn_set_pad_mode() (a JVM host cannot easily change its environment per call),
// else by the environment variable MVN_PAD_MODE = zero | zero_exact | none (MVN_PAD_GOOD_SIZE=0
// turns "zero" into "zero_exact"), else MVN_PAD_ZERO.
// ---------------------------------------------------------------------------------------------
enum { MVN_PAD_UNSET = -1, MVN_PAD_ZERO = 0, MVN_PAD_ZERO_EXACT = 1, MVN_PAD_NONE = 2 };
static std::atomic<int> g_pad_mode{MVN_PAD_UNSET};

static int parse_pad_mode(const char* m) {
  if (!m || !*m) return MVN_PAD_UNSET;
  if (std::strcmp(m, "zero") == 0) return MVN_PAD_ZERO;
  if (std::strcmp(m, "zero_exact") == 0) return MVN_PAD_ZERO_EXACT;
  if (std::strcmp(m, "none") == 0) return MVN_PAD_NONE;
  throw std::invalid_argument(std::string("unknown padding mode '") + m + "' (zero | zero_exact | none)");
}

static int current_pad_mode() {
  int m = g_pad_mode.load();
  if (m == MVN_PAD_UNSET) m = parse_pad_mode(std::getenv("MVN_PAD_MODE"));
  if (m == MVN_PAD_UNSET) m = MVN_PAD_ZERO;
  if (m == MVN_PAD_ZERO) {
    const char* gs = std::getenv("MVN_PAD_GOOD_SIZE");
    if (gs && std::strcmp(gs, "0") == 0) m = MVN_PAD_ZERO_EXACT;
  }
  return m;
}

// Padded extent >= n for one axis.  Any 2^a 3^b 5^c 7^d length avoids the chirp-z route; the
// lengths served by the fixed-length kernels (mvn_fixed_geom.hpp) run ~1.45x faster per element
// than the run-time-radix ones, so a slightly longer fixed length can be the cheaper transform
// (542 -> 576 rather than 560; 270 -> 320).  Candidates up to 1.3 n are priced by
// length x (fixed ? 1 : 1.45).  The last axis (d2 = 2H) must be even to have fixed kernels.
static int good_extent(int n, bool last_axis) {
  int best = 0;
  double best_cost = 0;
  for (int c = n; c <= n + n * 3 / 10 + 8; ++c) {
    int m = c;
    for (int p : {2, 3, 5, 7})
      while (m % p == 0) m /= p;
    if (m != 1) continue;
    int T = 0, threads = 0;
    size_t lds = 0;
    const bool fixed = last_axis ? (c % 2 == 0 && fixed_rows_geom(c / 2, &T, &threads, &lds))
                                 : fixed_strided_geom(c, &T, &threads, &lds);
    const double cost = (double)c * (fixed ? 1.0 : 1.45);
    if (!best || cost < best_cost) {
      best = c;
      best_cost = cost;
    }
  }
  return best ? best : next_smooth(n);
}

static int pick_device(int device) {
  if (device < 0) device = selectDeviceWithHighestComputeCapability();
  const int n = be::device_count();
  if (device < 0 || device >= n)
    throw std::runtime_error("no usable GPU (requested device " + std::to_string(device) + ", " +
                             std::to_string(n) + " present)");
  return device;
}

static shape_t to_shape(const int* d) {
  shape_t s = {{d[0], d[1], d[2]}};
  return s;
}

extern "C" {

const char* mvn_last_error(void) { return g_last_error.c_str(); }
const char* mvn_backend_name(void) { return be::backend_name(); }

// ---------------------------------------------------------------------------------------------
// device queries (inc/cuda_helpers.cuh:70-136)
// ---------------------------------------------------------------------------------------------
int getNumDevicesCUDA(void) {
  int n = 0;
  guarded("getNumDevicesCUDA", [&] { n = be::device_count(); });
  return n;
}

int getCUDAcomputeCapabilityMajorVersion(int devCUDA) {
  int major = 0, minor = 0;
  guarded("getCUDAcomputeCapabilityMajorVersion", [&] { be::device_arch(devCUDA, &major, &minor); });
  return major;
}

int getCUDAcomputeCapabilityMinorVersion(int devCUDA) {
  int major = 0, minor = 0;
  guarded("getCUDAcomputeCapabilityMinorVersion", [&] { be::device_arch(devCUDA, &major, &minor); });
  return minor;
}

void getNameDeviceCUDA(int devCUDA, char* name) {
  if (!name) return;
  std::memset(name, 0, 256);
  guarded("getNameDeviceCUDA", [&] { be::device_name(devCUDA, name); });
}

long long int getMemDeviceCUDA(int devCUDA) {
  long long v = 0;
  guarded("getMemDeviceCUDA", [&] { v = be::device_total_mem(devCUDA); });
  return v;
}

int selectDeviceWithHighestComputeCapability(void) {
  int value = -1;
  guarded("selectDeviceWithHighestComputeCapability", [&] {
    const int n = be::device_count();
    int best = 0;
    for (int d = 0; d < n; ++d) {  // first device with the highest 10*major+minor wins
      int major = 0, minor = 0;
      be::device_arch(d, &major, &minor);
      const int meta = 10 * major + minor;
      if (meta > best || value < 0) {
        best = meta;
        value = d;
      }
    }
  });
  return value;
}

// ---------------------------------------------------------------------------------------------
// reference hot path
// ---------------------------------------------------------------------------------------------
// Two LANES per device: a lane is what used to be "the device" for the blocking call -- its own
// mutex and its own cached resident engine (keys dev and dev + kLaneStride of device_mutex() /
// engine_cache()).  inplace_gpu_deconvolve and every other blocking entry point run on lane 0;
// mvn_deconvolve_submit alternates between the two, so that block k+1's stacks cross PCIe into the
// second engine while block k iterates in the first (the intent of
// inplace_gpu_deconvolve_iteration_interleaved, src/gpu_deconvolve_methods.cuh:82-326).  Uploads of
// the two lanes take turns (upload_mutex): one block at full PCIe rate starts iterating sooner than
// two at half rate each.
static const int kLaneStride = 1024;
static std::mutex& upload_mutex(int dev) { return device_mutex(2 * kLaneStride + dev); }

static void check_workspace(const imageType* psi, const workspace& input) {
  if (input.num_views_ < 0) throw std::invalid_argument("num_views_ must be >= 0");
  if (!psi || !input.data_) throw std::invalid_argument("null psi or workspace");
  for (int v = 0; v < input.num_views_; ++v) {  // before anything is dereferenced
    const view_data& d = input.data_[v];
    if (!d.image_ || !d.kernel1_ || !d.kernel2_ || !d.weights_ || !d.image_dims_ ||
        !d.kernel1_dims_ || !d.kernel2_dims_)
      throw std::invalid_argument("view " + std::to_string(v) + " has null members");
  }
}

// ---- MVN_DEVICES: one call, several devices ---------------------------------------------------
// MVN_DEVICES=0,1,2,3 makes inplace_gpu_deconvolve cut the (padded) volume into slabs of
// dim0 planes, one per listed device, and sweep them in the reference's view order with a halo exchange per
// convolution (mvn_multi.hpp) - the `device` argument is then ignored.  The call falls back to ONE device (its
// usual path) when the volume cannot be cut that way: a PSF deeper than 33 planes (no direct dim0 leg), fewer
// planes per slab than halo planes, an odd last extent, a slab that would hold padding only.  One group of slab
// engines is kept between calls like the single-device engine (mvn_release_cached_engines frees it).
static std::unique_ptr<HaloGroup>& multi_cache() {
  static std::unique_ptr<HaloGroup>* g = new std::unique_ptr<HaloGroup>();
  return *g;
}

static std::atomic<long> g_multi_calls{0};

static bool multi_device_call(imageType* psi, const workspace& input, const shape_t& dims, const shape_t& ext,
                              const int off[3], int pad_mode, const std::vector<int>& devs) {
  const int V = input.num_views_;
  int h = 1;
  for (int v = 0; v < V; ++v)
    h = std::max(h, std::max(input.data_[v].kernel1_dims_[0], input.data_[v].kernel2_dims_[0]) / 2);
  const int P = (int)devs.size();
  auto refuse = [&](const char* why) {
    if (trace_on()) std::printf("[lmvn::trace] MVN_DEVICES: %s - one device\n", why);
    return false;
  };
  if (!HaloGroup::feasible(P, ext, h)) return refuse("the volume cannot be cut into that many slabs");
  for (int r = 0; r < P; ++r) {  // every slab holds planes of the stacks, not padding only
    const int z0 = (int)((long)r * ext[0] / P), z1 = (int)((long)(r + 1) * ext[0] / P);
    if (std::min(z1, off[0] + dims[0]) <= std::max(z0, off[0])) return refuse("a slab would hold padding only");
  }
  std::vector<int> distinct(devs);
  std::sort(distinct.begin(), distinct.end());
  distinct.erase(std::unique(distinct.begin(), distinct.end()), distinct.end());
  std::vector<std::unique_lock<std::mutex>> locks;  // ascending order: no two calls can wait for each other
  for (int d : distinct) locks.emplace_back(device_mutex(d));
  std::unique_ptr<HaloGroup> group;
  {
    std::lock_guard<std::mutex> lk(engine_cache_mutex());
    group = std::move(multi_cache());
  }
  if (group && !(engine_cache_enabled() && group->matches(devs, ext, h, V))) group.reset();
  if (!group) {
    for (int d : distinct) {  // the single-device engines cached on these devices would compete for the memory
      be::set_device(d);
      pop_cached_engine(d).reset();
    }
    try {
      group.reset(new HaloGroup(devs, ext, h, V));
    } catch (const std::exception& e) {
      std::fprintf(stderr, "[libmultiviewnative] MVN_DEVICES ignored for this call: %s\n", e.what());
      return false;
    }
  }
  if (!group->all_direct(input)) {
    group.reset();
    return refuse("a PSF is not held in the direct dim0 form");
  }
  if (trace_on())
    std::printf("[lmvn::trace] MVN_DEVICES: %d slabs of %d x %d x %d, %d halo planes either side\n", P, ext[0], ext[1],
                ext[2], h);
  group->run(psi, input, dims, off, pad_mode == MVN_PAD_ZERO);  // (on failure the group is dropped)
  ++g_multi_calls;
  if (engine_cache_enabled()) {
    std::lock_guard<std::mutex> lk(engine_cache_mutex());
    multi_cache() = std::move(group);
  }
  return true;
}

static void deconvolve_call(imageType* psi, const workspace& input, int device, int lane, int pad_mode) {
  {
    check_workspace(psi, input);
    const int V = input.num_views_;
    if (V == 0 || input.num_iterations_ <= 0) return;  // 0 iterations returns psi unchanged
    const shape_t dims = to_shape(input.data_[0].image_dims_);
    for (int v = 0; v < V; ++v) {
      const view_data& d = input.data_[v];
      if (to_shape(d.image_dims_) != dims)
        throw std::invalid_argument("all views must share image_dims_ (view " + std::to_string(v) + ")");
      if (d.weights_dims_ && to_shape(d.weights_dims_) != dims)
        throw std::invalid_argument("weights_dims_ must equal image_dims_ (view " + std::to_string(v) + ")");
    }
    for (int d = 0; d < 3; ++d)
      if (dims[d] < 1) throw std::invalid_argument("image extents must be >= 1");
    shape_t ext = dims;  // padding policy: see the block comment above good_extent()
    int off[3] = {0, 0, 0};
    bool dim0_kept_exact = false;
    if (pad_mode != MVN_PAD_NONE) {
      for (int d = 2; d >= 0; --d) {
        int kmax = 1;
        for (int v = 0; v < V; ++v) {
          kmax = std::max(kmax, input.data_[v].kernel1_dims_[d]);
          kmax = std::max(kmax, input.data_[v].kernel2_dims_[d]);
        }
        ext[d] = dims[d] + kmax - 1;
        off[d] = (kmax - 1) / 2;
        if (pad_mode != MVN_PAD_ZERO) continue;
        // dim0 is not transformed when every PSF is thin enough for the direct dim0 leg (mvn_dim0_direct.hpp):
        // it then keeps the reference's exact image + kernel - 1 (542 planes for a 512-block with 31^3 PSFs, not
        // 576: 6 % less volume in every pass) - provided the rows of a plane keep whole tiles of the fixed
        // last-axis kernels whatever the plane count (d1 a multiple of 16)
        if (d == 0 && ext[1] % 16 == 0 && Engine::direct_ok_for(kmax, ext[0], ext[1], ext[2])) {
          dim0_kept_exact = true;
          continue;
        }
        ext[d] = good_extent(ext[d], d == 2);
      }
    }
    if (lane == 0) {  // (the second lane belongs to the block pipeline of mvn_deconvolve_submit)
      const std::vector<int> devs = multi_devices_from_env();
      if (!devs.empty() && multi_device_call(psi, input, dims, ext, off, pad_mode, devs)) return;
    }
    const int dev = pick_device(device);
    const int key = dev + lane * kLaneStride;
    std::lock_guard<std::mutex> lk(device_mutex(key));
    be::set_device(dev);
    const bool embedded = ext[0] != dims[0] || ext[1] != dims[1] || ext[2] != dims[2];
    // on failure the engine is simply dropped
    std::unique_ptr<Engine> eng_owner =
        take_engine(key, dev, ext, V, embedded ? (size_t)dims[0] * (size_t)dims[1] * (size_t)dims[2] : 0);
    if (dim0_kept_exact) {
      // The static rule above and the engine's own per-kernel decision (Engine::direct_form: also asks that the
      // tap arrays' plan is of the volume plan's kernel family) must agree, or an exact dim0 such as 542 = 2 * 271
      // would go through the chirp-z FFT leg: the engine has the last word, dim0 is then padded like the others.
      bool all = true;
      for (int v = 0; v < V && all; ++v)
        all = eng_owner->would_be_direct(input.data_[v].kernel1_dims_) &&
              eng_owner->would_be_direct(input.data_[v].kernel2_dims_);
      if (!all) {
        eng_owner.reset();
        ext[0] = good_extent(ext[0], false);
        if (trace_on()) std::printf("[lmvn::trace] direct dim0 leg refused by the engine: dim0 padded to %d\n", ext[0]);
        eng_owner = take_engine(key, dev, ext, V, (size_t)dims[0] * (size_t)dims[1] * (size_t)dims[2]);
      }
    }
    Engine& eng = *eng_owner;
    eng.begin_call();
    // stacks are embedded into / cropped out of the padded volume by the transfers themselves
    // (strided device copies), so the padded modes keep the pipelined upload
    const int dims_i[3] = {dims[0], dims[1], dims[2]};
    eng.set_embedding(dims_i, off);
    // extra zeros beyond the PSF's reach: the blurred estimate is exactly or nearly 0 there and
    // the reference's pointwise math would give 0 * 1/0 = NaN; the one deliberate deviation
    eng.set_quotient_guard(pad_mode == MVN_PAD_ZERO);
    static const bool no_pipeline = [] {
      const char* e = std::getenv("MVN_NO_PIPELINE");
      return e && *e && std::strcmp(e, "0") != 0;
    }();
    if (no_pipeline) {
      for (int v = 0; v < V; ++v) {
        const view_data& d = input.data_[v];
        eng.set_view(v, d.image_, d.weights_, d.kernel1_, d.kernel1_dims_, d.kernel2_, d.kernel2_dims_);
      }
      eng.set_psi(psi);
      eng.iterate(input.num_iterations_, input.lambda_, input.minValue_);
      eng.sync();
      eng.get_psi(psi);
      give_back_engine(key, std::move(eng_owner));
      return;
    }
    // stacks arrive view by view from a second host thread while the first iteration already
    // runs on the views that are in (the reference's interleaved driver)
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
      if (!trace_on()) return;
      auto t1 = std::chrono::steady_clock::now();
      std::printf("[lmvn::trace] %-28s %8.1f ms\n", what,
                  std::chrono::duration<double, std::milli>(t1 - t0).count());
      t0 = t1;
    };
    eng.reserve_views();
    {
      // the loop starts before the last view has been staged: tell it now whether every kernel of the call
      // will be held in the direct dim0 form (then the Nyquist bins ride in the DC column, mvn_dim0_direct.hpp)
      bool all = true;
      for (int v = 0; v < V && all; ++v)
        all = eng.would_be_direct(input.data_[v].kernel1_dims_) && eng.would_be_direct(input.data_[v].kernel2_dims_);
      eng.set_all_direct_hint(all);
      bool lines = all;  // ... and through the fused middle pass (mvn_mid_fused.hpp)
      for (int v = 0; v < V && lines; ++v)
        lines = eng.would_be_lines(input.data_[v].kernel1_dims_) && eng.would_be_lines(input.data_[v].kernel2_dims_);
      eng.set_all_lines_hint(lines);
    }
    lap("allocate view buffers");
    std::unique_lock<std::mutex> pcie(upload_mutex(dev));  // handed to the uploader thread's scope below
    eng.set_psi(psi);
    lap("upload psi");
    std::exception_ptr up_err;
    std::thread uploader([&] {
      try {
        be::set_device(dev);
        auto u0 = std::chrono::steady_clock::now();
        for (int v = 0; v < V; ++v) {
          const view_data& d = input.data_[v];
          eng.stage_view(v, d.image_, d.weights_, d.kernel1_, d.kernel1_dims_, d.kernel2_, d.kernel2_dims_);
        }
        eng.finish_staging();
        if (trace_on())
          std::printf("[lmvn::trace] %-28s %8.1f ms (uploader thread)\n", "stage all views",
                      std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - u0).count());
      } catch (...) {
        up_err = std::current_exception();
        eng.staging_failed();
      }
    });
    std::exception_ptr main_err;
    try {
      eng.iterate(input.num_iterations_, input.lambda_, input.minValue_);
    } catch (...) {
      main_err = std::current_exception();
    }
    lap("enqueue iterations");
    uploader.join();
    pcie.unlock();  // the other lane's block may start crossing PCIe while this one iterates
    lap("join uploader");
    if (up_err) std::rethrow_exception(up_err);
    if (main_err) std::rethrow_exception(main_err);
    eng.sync();
    lap("wait for the device");
    eng.get_psi(psi);
    lap("download psi");
    give_back_engine(key, std::move(eng_owner));
  }
}

void inplace_gpu_deconvolve(imageType* psi, struct workspace input, int device) {
  guarded("inplace_gpu_deconvolve", [&] { deconvolve_call(psi, input, device, 0, current_pad_mode()); });
}

// ---- asynchronous pair: mvn_deconvolve_submit / mvn_deconvolve_wait ---------------------------
extern "C++" {
namespace {
struct DeconvJob {
  std::thread worker;
  int rc = 0;
  std::string error;
  std::vector<view_data> views;  // the caller's view_data array and dims, copied at submit
  std::vector<int> dims;
  workspace ws;
};
std::mutex& jobs_mutex() {
  static std::mutex* m = new std::mutex();
  return *m;
}
std::map<long long, std::unique_ptr<DeconvJob>>& jobs() {  // leaked on purpose: no joinable thread is destroyed at exit
  static auto* j = new std::map<long long, std::unique_ptr<DeconvJob>>();
  return *j;
}
}  // namespace
}  // extern "C++"

int mvn_deconvolve_submit(imageType* psi, struct workspace input, int device, long long* ticket) {
  return guarded("mvn_deconvolve_submit", [&] {
    if (!ticket) throw std::invalid_argument("null ticket");
    *ticket = 0;
    check_workspace(psi, input);
    const int V = input.num_views_;
    std::unique_ptr<DeconvJob> job(new DeconvJob());
    job->views.assign(input.data_, input.data_ + V);
    job->dims.resize((size_t)V * 12);
    for (int v = 0; v < V; ++v) {  // the four int[3] of every view move into the job
      view_data& d = job->views[v];
      int* base = job->dims.data() + (size_t)v * 12;
      int** members[4] = {&d.image_dims_, &d.kernel1_dims_, &d.kernel2_dims_, &d.weights_dims_};
      for (int m = 0; m < 4; ++m) {
        if (!*members[m]) continue;  // weights_dims_ may be null
        std::memcpy(base + 3 * m, *members[m], 3 * sizeof(int));
        *members[m] = base + 3 * m;
      }
    }
    job->ws = input;
    job->ws.data_ = job->views.data();
    const int dev = pick_device(device);
    const int pad_mode = current_pad_mode();  // the policy in force at submit time
    static std::atomic<long long> next_ticket{1};
    const long long id = next_ticket.fetch_add(1);
    static std::mutex lane_mu;
    static std::map<int, int> lane_of;  // submits per device so far -> lanes alternate
    int lane;
    {
      std::lock_guard<std::mutex> lk(lane_mu);
      lane = lane_of[dev]++ & 1;
    }
    DeconvJob* j = job.get();
    // (the worker first, the map entry second: a thread that cannot be started leaves no job behind)
    j->worker = std::thread([j, psi, dev, lane, pad_mode] {
      j->rc = guarded("mvn_deconvolve_submit (worker)", [&] { deconvolve_call(psi, j->ws, dev, lane, pad_mode); });
      if (j->rc < 0) j->error = g_last_error;  // the worker's thread-local message travels with the job
    });
    try {
      std::lock_guard<std::mutex> lk(jobs_mutex());
      jobs()[id] = std::move(job);
    } catch (...) {
      j->worker.join();
      throw;
    }
    *ticket = id;
  });
}

int mvn_deconvolve_wait(long long ticket) {
  std::unique_ptr<DeconvJob> job;
  const int rc = guarded("mvn_deconvolve_wait", [&] {
    {
      std::lock_guard<std::mutex> lk(jobs_mutex());
      auto it = jobs().find(ticket);
      if (it == jobs().end()) throw std::invalid_argument("unknown or already awaited ticket");
      job = std::move(it->second);
      jobs().erase(it);
    }
    if (job->worker.joinable()) job->worker.join();
  });
  if (rc < 0) return rc;
  if (job->rc < 0) g_last_error = job->error;
  return job->rc;
}

int mvn_set_pad_mode(const char* mode) {
  return guarded("mvn_set_pad_mode", [&] { g_pad_mode.store(parse_pad_mode(mode)); });
}

const char* mvn_get_pad_mode(void) {
  switch (g_pad_mode.load()) {
    case MVN_PAD_ZERO: return "zero";
    case MVN_PAD_ZERO_EXACT: return "zero_exact";
    case MVN_PAD_NONE: return "none";
    default: return "";
  }
}

// single convolution on the engine's kernels; shared by the three convolution entry points
static void convolve_host(float* im, const int* imDim, const float* kernel, const int* kernelDim,
                          int device) {
  if (!im || !imDim || !kernel || !kernelDim) throw std::invalid_argument("null argument");
  const int dev = pick_device(device);
  std::lock_guard<std::mutex> lk(device_mutex(dev));
  be::set_device(dev);
  std::shared_ptr<Plan3D> plan = PlanStore::get().add(dev, to_shape(imDim));
  const Layout& L = plan->L;
  be::stream_t s = be::stream_create();
  float* vol = nullptr;
  float* spec = nullptr;
  cfloat *nyq = nullptr, *snyq = nullptr;
  float* dk = nullptr;
  auto cleanup = [&] {
    be::dfree(vol);
    be::dfree(spec);
    be::dfree(nyq);
    be::dfree(snyq);
    be::dfree(dk);
    be::stream_destroy(s);
  };
  try {
    vol = (float*)be::dmalloc(plan->main_bytes());
    spec = (float*)be::dmalloc(plan->main_bytes());
    if (plan->nyq_bytes()) {
      nyq = (cfloat*)be::dmalloc(plan->nyq_bytes());
      snyq = (cfloat*)be::dmalloc(plan->nyq_bytes());
    }
    const size_t kb = sizeof(float) * (size_t)kernelDim[0] * kernelDim[1] * kernelDim[2];
    dk = (float*)be::dmalloc(kb);
    be::h2d(dk, kernel, kb, s);
    be::dzero(vol, plan->main_bytes(), s);
    be::h2d_2d(vol, (size_t)L.RP * 4, im, (size_t)L.d2 * 4, (size_t)L.d2 * 4, L.rows, s);
    const float scale = (float)(1.0 / (double)L.logical());
    plan->psf_spectrum(dk, kernelDim, scale, spec, snyq, s);
    EpilogueParams e;
    std::memset(&e, 0, sizeof(e));
    e.mode = MVN_EPI_STORE;
    e.scale = 1.f;
    plan->convolve(vol, (cfloat*)vol, nyq, (const cfloat*)spec, snyq, vol, e, s);
    be::d2h_2d(im, (size_t)L.d2 * 4, vol, (size_t)L.RP * 4, (size_t)L.d2 * 4, L.rows, s);
    be::stream_sync(s);
  } catch (...) {
    try {
      be::stream_sync(s);
    } catch (...) {
    }
    cleanup();
    throw;
  }
  cleanup();
}

void inplace_gpu_convolution(imageType* im, int* imDim, imageType* kernel, int* kernelDim,
                             int device) {
  guarded("inplace_gpu_convolution", [&] { convolve_host(im, imDim, kernel, kernelDim, device); });
}

void convolution3DfftCUDAInPlace(imageType* im, int* imDim, imageType* kernel, int* kernelDim,
                                 int devCUDA) {
  guarded("convolution3DfftCUDAInPlace", [&] { convolve_host(im, imDim, kernel, kernelDim, devCUDA); });
}

// device-pointer convolution on an already selected device; the caller holds the device mutex
static void core_convolve_on_device(float* d_im, const int* imDim, const float* d_kernel,
                                    const int* kernelDim, int dev) {
  std::shared_ptr<Plan3D> plan = PlanStore::get().add(dev, to_shape(imDim));
  const Layout& L = plan->L;
  be::stream_t s = be::stream_create();
  float *vol = nullptr, *spec = nullptr;
  cfloat *nyq = nullptr, *snyq = nullptr;
  auto cleanup = [&] {
    if (vol != d_im) be::dfree(vol);
    be::dfree(spec);
    be::dfree(nyq);
    be::dfree(snyq);
    be::stream_destroy(s);
  };
  try {
    spec = (float*)be::dmalloc(plan->main_bytes());
    if (plan->nyq_bytes()) {
      nyq = (cfloat*)be::dmalloc(plan->nyq_bytes());
      snyq = (cfloat*)be::dmalloc(plan->nyq_bytes());
    }
    if (L.RP == L.d2) {
      vol = d_im;  // even d2: the caller's dense volume already is the engine layout
    } else {
      vol = (float*)be::dmalloc(plan->main_bytes());
      be::dzero(vol, plan->main_bytes(), s);
      be::d2d_2d(vol, (size_t)L.RP * 4, d_im, (size_t)L.d2 * 4, (size_t)L.d2 * 4, L.rows, s);
    }
    const float scale = (float)(1.0 / (double)L.logical());
    plan->psf_spectrum(d_kernel, kernelDim, scale, spec, snyq, s);
    EpilogueParams e;
    std::memset(&e, 0, sizeof(e));
    e.mode = MVN_EPI_STORE;
    e.scale = 1.f;
    plan->convolve(vol, (cfloat*)vol, nyq, (const cfloat*)spec, snyq, vol, e, s);
    if (vol != d_im)
      be::d2d_2d(d_im, (size_t)L.d2 * 4, vol, (size_t)L.RP * 4, (size_t)L.d2 * 4, L.rows, s);
    be::stream_sync(s);
  } catch (...) {
    try {
      be::stream_sync(s);
    } catch (...) {
    }
    cleanup();
    throw;
  }
  cleanup();
}

void convolution3DfftCUDAInPlace_core(imageType* d_im, int* imDim, imageType* d_kernel,
                                      int* kernelDim, int devCUDA) {
  guarded("convolution3DfftCUDAInPlace_core", [&] {
    if (!d_im || !imDim || !d_kernel || !kernelDim) throw std::invalid_argument("null argument");
    const int dev = pick_device(devCUDA);
    std::lock_guard<std::mutex> lk(device_mutex(dev));
    be::set_device(dev);
    core_convolve_on_device(d_im, imDim, d_kernel, kernelDim, dev);
  });
}

// One Richardson-Lucy step on a single stack, the legacy demo entry points of
// src/multiviewnative.cu:395-506 (plain) and :508-600 (tikhonov): psi_0 = view = _input,
// kernel1 = _kernel, kernel2 = 0.1 everywhere (same extents), weights = 1; both convolutions are
// cyclic on _input_dims (they go through convolution3DfftCUDAInPlace_core there as here).
static void iterate_fft_legacy(const char* what, const float* input, const float* kernel,
                               float* output, const int* input_dims, const int* kernel_dims,
                               bool tikhonov, float min_value, double lambda, int device) {
  guarded(what, [&] {
    if (!input || !kernel || !output || !input_dims || !kernel_dims)
      throw std::invalid_argument("null argument");
    const size_t n = (size_t)input_dims[0] * input_dims[1] * input_dims[2];
    const size_t nk = (size_t)kernel_dims[0] * kernel_dims[1] * kernel_dims[2];
    if (n == 0 || nk == 0) throw std::invalid_argument("empty stack or kernel");
    const int dev = pick_device(device);
    std::lock_guard<std::mutex> lk(device_mutex(dev));
    be::set_device(dev);
    float *d_image = nullptr, *d_initial = nullptr, *d_weights = nullptr, *d_kernel = nullptr;
    auto cleanup = [&] {
      be::dfree(d_image);
      be::dfree(d_initial);
      be::dfree(d_weights);
      be::dfree(d_kernel);
    };
    try {
      d_image = (float*)be::dmalloc(n * 4);
      d_initial = (float*)be::dmalloc(n * 4);
      d_weights = (float*)be::dmalloc(n * 4);
      d_kernel = (float*)be::dmalloc(nk * 4);
      std::vector<float> ones(n, 1.f), tenth(nk, .1f);
      be::h2d(d_weights, ones.data(), n * 4, nullptr);
      be::h2d(d_initial, input, n * 4, nullptr);
      be::h2d(d_image, input, n * 4, nullptr);
      be::h2d(d_kernel, kernel, nk * 4, nullptr);
      be::stream_sync(nullptr);
      int idims[3] = {input_dims[0], input_dims[1], input_dims[2]};
      int kdims[3] = {kernel_dims[0], kernel_dims[1], kernel_dims[2]};
      core_convolve_on_device(d_image, idims, d_kernel, kdims, dev);  // psi (*) kernel1
      be::launch_divide(d_initial, d_image, n, nullptr);              // view / blurred
      be::h2d(d_kernel, tenth.data(), nk * 4, nullptr);
      be::stream_sync(nullptr);
      core_convolve_on_device(d_image, idims, d_kernel, kdims, dev);  // (*) kernel2 -> integral
      if (tikhonov)
        be::launch_update_legacy_tikhonov(d_initial, d_image, d_weights, n, (float)lambda,
                                          min_value, nullptr);
      else
        be::launch_update(d_initial, d_image, d_weights, n, 0., min_value, nullptr);
      std::vector<float> tmp(n);
      be::d2h(tmp.data(), d_initial, n * 4, nullptr);
      be::stream_sync(nullptr);
      std::memcpy(output, tmp.data(), n * 4);
    } catch (...) {
      cleanup();
      throw;
    }
    cleanup();
  });
}

void iterate_fft_plain(imageType* _input, imageType* _kernel, imageType* _output, int* _input_dims,
                       int* _kernel_dims, int _device) {
  // the reference hard-codes minValue = .0001f here (src/multiviewnative.cu:485-486)
  iterate_fft_legacy("iterate_fft_plain", _input, _kernel, _output, _input_dims, _kernel_dims,
                     false, .0001f, 0., _device);
}

void iterate_fft_tikhonov(imageType* _input, imageType* _kernel, imageType* _output,
                          int* _input_dims, int* _kernel_dims, size_t _size, float _minValue,
                          double _lambda, int _device) {
  (void)_size;  // unused by the reference too (the extent comes from _input_dims)
  iterate_fft_legacy("iterate_fft_tikhonov", _input, _kernel, _output, _input_dims, _kernel_dims,
                     true, _minValue, _lambda, _device);
}

void compute_quotient(imageType* _input, imageType* _output, size_t _size, int _device) {
  guarded("compute_quotient", [&] {
    if (!_input || !_output) throw std::invalid_argument("null argument");
    if (_size == 0) return;
    const int dev = pick_device(_device);
    std::lock_guard<std::mutex> lk(device_mutex(dev));
    be::set_device(dev);
    const size_t bytes = _size * sizeof(float);
    float* d_in = (float*)be::dmalloc(bytes);
    float* d_out = nullptr;
    try {
      d_out = (float*)be::dmalloc(bytes);
      be::h2d(d_in, _input, bytes, nullptr);
      be::h2d(d_out, _output, bytes, nullptr);
      be::launch_divide(d_in, d_out, _size, nullptr);
      std::vector<float> tmp(_size);
      be::d2h(tmp.data(), d_out, bytes, nullptr);
      be::stream_sync(nullptr);
      std::memcpy(_output, tmp.data(), bytes);
    } catch (...) {
      be::dfree(d_in);
      be::dfree(d_out);
      throw;
    }
    be::dfree(d_in);
    be::dfree(d_out);
  });
}

void compute_final_values(imageType* _image, imageType* _integral, imageType* _weight,
                          size_t _size, float _minValue, double _lambda, int _device) {
  guarded("compute_final_values", [&] {
    if (!_image || !_integral || !_weight) throw std::invalid_argument("null argument");
    if (_size == 0) return;
    const int dev = pick_device(_device);
    std::lock_guard<std::mutex> lk(device_mutex(dev));
    be::set_device(dev);
    const size_t bytes = _size * sizeof(float);
    float *d_psi = nullptr, *d_int = nullptr, *d_w = nullptr;
    auto cleanup = [&] {
      be::dfree(d_psi);
      be::dfree(d_int);
      be::dfree(d_w);
    };
    try {
      d_psi = (float*)be::dmalloc(bytes);
      d_int = (float*)be::dmalloc(bytes);
      d_w = (float*)be::dmalloc(bytes);
      be::h2d(d_psi, _image, bytes, nullptr);
      be::h2d(d_int, _integral, bytes, nullptr);
      be::h2d(d_w, _weight, bytes, nullptr);
      be::launch_update(d_psi, d_int, d_w, _size, _lambda, _minValue, nullptr);
      std::vector<float> tmp(_size);
      be::d2h(tmp.data(), d_psi, bytes, nullptr);
      be::stream_sync(nullptr);
      std::memcpy(_image, tmp.data(), bytes);
    } catch (...) {
      cleanup();
      throw;
    }
    cleanup();
  });
}

// ---------------------------------------------------------------------------------------------
// plan_store
// ---------------------------------------------------------------------------------------------
int mvn_release_cached_engines(void) {
  return guarded("mvn_release_cached_engines", [&] {
    std::vector<int> devs;
    {
      std::lock_guard<std::mutex> lk(engine_cache_mutex());
      for (auto& kv : engine_cache()) devs.push_back(kv.first);
    }
    for (int d : devs) {
      std::lock_guard<std::mutex> lk(device_mutex(d));  // not while a call on that device runs
      pop_cached_engine(d).reset();
    }
    std::unique_ptr<HaloGroup> group;
    {
      std::lock_guard<std::mutex> lk(engine_cache_mutex());
      group = std::move(multi_cache());
    }
    if (group) {
      std::vector<int> distinct(group->devices());
      std::sort(distinct.begin(), distinct.end());
      distinct.erase(std::unique(distinct.begin(), distinct.end()), distinct.end());
      std::vector<std::unique_lock<std::mutex>> locks;
      for (int d : distinct) locks.emplace_back(device_mutex(d));
      group.reset();
    }
  });
}

long mvn_split_launch_count(void) { return be::split_launch_count(); }
long mvn_mid_fused_launch_count(void) { return be::mid_fused_launch_count(); }
long mvn_multi_device_calls(void) { return g_multi_calls.load(); }

// ---- resident group of slab engines (what MVN_DEVICES runs inside inplace_gpu_deconvolve) -----
int mvn_group_create(const int* devices, int ndevices, const int dims[3], int halo_planes, int num_views,
                     mvn_group** out) {
  return guarded("mvn_group_create", [&] {
    if (!devices || !dims || !out) throw std::invalid_argument("null argument");
    *out = nullptr;
    std::vector<int> devs(devices, devices + (ndevices > 0 ? ndevices : 0));
    for (int d : devs)
      if (d < 0 || d >= be::device_count()) throw std::invalid_argument("no such device");
    std::unique_ptr<mvn_group> g(new mvn_group());
    g->impl.reset(new HaloGroup(devs, to_shape(dims), halo_planes < 1 ? 1 : halo_planes, num_views));
    *out = g.release();
  });
}

int mvn_group_destroy(mvn_group* g) {
  return guarded("mvn_group_destroy", [&] { delete g; });
}

#define MVN_GROUP_CALL(where, ...)                                               \
  return guarded(where, [&] {                                                    \
    if (!g || !g->impl) throw std::invalid_argument("null group");               \
    HaloGroup& G = *g->impl;                                                     \
    __VA_ARGS__;                                                                 \
  })

int mvn_group_load(mvn_group* g, const float* psi, struct workspace input) {
  MVN_GROUP_CALL("mvn_group_load", {
    check_workspace(psi, input);
    if (input.num_views_ != G.num_views()) throw std::invalid_argument("view count of the group");
    for (int v = 0; v < input.num_views_; ++v)
      if (to_shape(input.data_[v].image_dims_) != G.extents())
        throw std::invalid_argument("stacks must have the extents the group was created for");
    if (!G.all_direct(input)) throw std::invalid_argument("every PSF must be held in the direct dim0 form (<= 33 planes)");
    const int off[3] = {0, 0, 0};
    G.load(psi, input, G.extents(), off, false);
  });
}

int mvn_group_iterate(mvn_group* g, int iterations, double lambda, float min_value, float* ms) {
  MVN_GROUP_CALL("mvn_group_iterate", {
    const double t = G.iterate(iterations, lambda, min_value);
    if (ms) *ms = (float)t;
  });
}

int mvn_group_get_psi(mvn_group* g, float* psi) {
  MVN_GROUP_CALL("mvn_group_get_psi", {
    if (!psi) throw std::invalid_argument("null psi");
    G.fetch(psi);
  });
}

int mvn_psf_cache_counters(long out[2]) {
  return guarded("mvn_psf_cache_counters", [&] {
    if (!out) throw std::invalid_argument("null out");
    out[0] = Engine::psf_cache_hits();
    out[1] = Engine::psf_cache_misses();
  });
}

int mvn_plan_store_add(int device, const int dims[3]) {
  return guarded("mvn_plan_store_add", [&] { PlanStore::get().add(pick_device(device), to_shape(dims)); });
}

int mvn_plan_store_has_key(int device, const int dims[3]) {
  int r = 0;
  int rc = guarded("mvn_plan_store_has_key",
                   [&] { r = PlanStore::get().has_key(pick_device(device), to_shape(dims)) ? 1 : 0; });
  return rc < 0 ? rc : r;
}

int mvn_plan_store_size(void) { return (int)PlanStore::get().size(); }
int mvn_plan_store_empty(void) { return PlanStore::get().empty() ? 1 : 0; }
int mvn_plan_store_clear(void) {
  return guarded("mvn_plan_store_clear", [&] {
    mvn_release_cached_engines();  // a cached engine would keep using its old plan
    PlanStore::get().clear();
  });
}

int mvn_plan_describe(int device, const int dims[3], int out[12]) {
  return guarded("mvn_plan_describe", [&] {
    std::shared_ptr<Plan3D> p = PlanStore::get().add(pick_device(device), to_shape(dims));
    out[0] = p->L.h;
    out[1] = p->L.C;
    out[2] = p->L.RP;
    out[3] = p->L.even ? 1 : 0;
    out[4] = p->g_rows.T;
    out[5] = p->g_ax1.T;
    out[6] = p->g_ax0.T;
    out[7] = p->ax2.view.nstages;
    out[8] = p->fx_rows ? 1 : 0;
    out[9] = p->fx_ax1 ? 1 : 0;
    out[10] = p->fx_ax0 ? 1 : 0;
    out[11] = 0;
  });
}

// ---------------------------------------------------------------------------------------------
// whole transforms on host buffers
// ---------------------------------------------------------------------------------------------
struct FftScratch {
  std::shared_ptr<Plan3D> plan;
  be::stream_t s = nullptr;
  float* vol = nullptr;
  cfloat* nyq = nullptr;
  FftScratch(int dev, const int* dims) {
    be::set_device(dev);
    plan = PlanStore::get().add(dev, to_shape(dims));
    s = be::stream_create();
    vol = (float*)be::dmalloc(plan->main_bytes());
    be::dzero(vol, plan->main_bytes(), s);
    if (plan->nyq_bytes()) {
      nyq = (cfloat*)be::dmalloc(plan->nyq_bytes());
      be::dzero(nyq, plan->nyq_bytes(), s);
    }
  }
  ~FftScratch() {
    try {
      be::stream_sync(s);
    } catch (...) {
    }
    be::dfree(vol);
    be::dfree(nyq);
    be::stream_destroy(s);
  }
};

// On the device a spectrum is kept in POSITION order along every axis (digit-reversed, exactly as
// the decimation-in-frequency passes leave it; the PSF spectra are stored the same way, so the
// RL loop never permutes anything).  These two host-buffer utilities translate to / from the
// natural FFTW/cuFFT bin order: bin (k0,k1,k2) lives at (inv0[k0], inv1[k1], inv2[k2]).
int mvn_fft3_r2c(int device, const int dims[3], const float* real, float* spec) {
  return guarded("mvn_fft3_r2c", [&] {
    const int dev = pick_device(device);
    std::lock_guard<std::mutex> lk(device_mutex(dev));
    FftScratch f(dev, dims);
    const Layout& L = f.plan->L;
    be::h2d_2d(f.vol, (size_t)L.RP * 4, real, (size_t)L.d2 * 4, (size_t)L.d2 * 4, L.rows, f.s);
    f.plan->forward(f.vol, f.nyq, f.s);
    const int nc = L.d2 / 2 + 1;
    std::vector<cfloat> main_h(L.rows * (size_t)L.C), nyq_h(L.nyq_cplx());
    be::d2h(main_h.data(), f.vol, main_h.size() * sizeof(cfloat), f.s);
    if (L.even) be::d2h(nyq_h.data(), f.nyq, nyq_h.size() * sizeof(cfloat), f.s);
    be::stream_sync(f.s);
    const std::vector<int>& i0 = f.plan->ax0.host.inv;
    const std::vector<int>& i1 = f.plan->ax1.host.inv;
    const std::vector<int>& i2 = f.plan->ax2.host.inv;  // length L.h; used for even d2 only
    cfloat* out = reinterpret_cast<cfloat*>(spec);
    for (int k0 = 0; k0 < L.d0; ++k0)
      for (int k1 = 0; k1 < L.d1; ++k1) {
        const size_t srow = (size_t)i0[k0] * L.d1 + (size_t)i1[k1];
        cfloat* o = out + ((size_t)k0 * L.d1 + k1) * (size_t)nc;
        for (int k2 = 0; k2 < L.C; ++k2) o[k2] = main_h[srow * L.C + (L.even ? i2[k2] : k2)];
        if (L.even) o[L.C] = nyq_h[srow];
      }
  });
}

int mvn_fft3_c2r(int device, const int dims[3], const float* spec, float* real) {
  return guarded("mvn_fft3_c2r", [&] {
    const int dev = pick_device(device);
    std::lock_guard<std::mutex> lk(device_mutex(dev));
    FftScratch f(dev, dims);
    const Layout& L = f.plan->L;
    const int nc = L.d2 / 2 + 1;
    std::vector<cfloat> main_h(L.rows * (size_t)L.C), nyq_h(L.nyq_cplx());
    const std::vector<int>& i0 = f.plan->ax0.host.inv;
    const std::vector<int>& i1 = f.plan->ax1.host.inv;
    const std::vector<int>& i2 = f.plan->ax2.host.inv;
    const cfloat* in = reinterpret_cast<const cfloat*>(spec);
    for (int k0 = 0; k0 < L.d0; ++k0)
      for (int k1 = 0; k1 < L.d1; ++k1) {
        const size_t drow = (size_t)i0[k0] * L.d1 + (size_t)i1[k1];
        const cfloat* o = in + ((size_t)k0 * L.d1 + k1) * (size_t)nc;
        for (int k2 = 0; k2 < L.C; ++k2) main_h[drow * L.C + (L.even ? i2[k2] : k2)] = o[k2];
        if (L.even) nyq_h[drow] = o[L.C];
      }
    be::h2d(f.vol, main_h.data(), main_h.size() * sizeof(cfloat), f.s);
    if (L.even) be::h2d(f.nyq, nyq_h.data(), nyq_h.size() * sizeof(cfloat), f.s);
    f.plan->backward(f.vol, f.nyq, 1.f, f.s);
    be::d2h_2d(real, (size_t)L.d2 * 4, f.vol, (size_t)L.RP * 4, (size_t)L.d2 * 4, L.rows, f.s);
    be::stream_sync(f.s);
  });
}

int mvn_fft3_time(int device, const int dims[3], int direction, int reps, float* ms) {
  return mvn_fft3_profile(device, dims, direction, reps, ms, nullptr);
}

int mvn_fft3_profile(int device, const int dims[3], int direction, int reps, float* ms,
                     double* per_kind_ms) {
  return guarded("mvn_fft3_profile", [&] {
    if (reps < 1) throw std::invalid_argument("reps must be >= 1");
    const int dev = pick_device(device);
    std::lock_guard<std::mutex> lk(device_mutex(dev));
    FftScratch f(dev, dims);
    const Layout& L = f.plan->L;
    {  // non-trivial contents (zeros flatter the clock: guide 5.4 rule 25)
      std::vector<float> host(L.real_floats());
      unsigned x = 12345u;
      for (size_t i = 0; i < host.size(); ++i) {
        x = x * 1664525u + 1013904223u;
        host[i] = (float)(x >> 8) * (1.0f / 16777216.0f) - 0.5f;
      }
      be::h2d(f.vol, host.data(), host.size() * 4, f.s);
      be::stream_sync(f.s);
    }
    const float keep = 1.0f / (float)L.logical();
    auto run = [&] {
      if (direction == 0)
        f.plan->forward(f.vol, f.nyq, f.s);
      else
        f.plan->backward(f.vol, f.nyq, keep, f.s);
    };
    run();  // warm-up
    be::event_t a = be::event_create(), b = be::event_create();
    be::event_record(a, f.s);
    for (int i = 0; i < reps; ++i) run();
    be::event_record(b, f.s);
    be::event_sync(b);
    *ms = be::event_elapsed_ms(a, b) / (float)reps;
    be::event_destroy(a);
    be::event_destroy(b);
    if (per_kind_ms) {  // second, event-instrumented round: average launch time per kernel kind
      Profiler prof;
      prof.enabled = true;
      for (int i = 0; i < reps; ++i) {
        if (direction == 0)
          f.plan->forward(f.vol, f.nyq, f.s, &prof);
        else
          f.plan->backward(f.vol, f.nyq, keep, f.s, &prof);
      }
      be::stream_sync(f.s);
      prof.collect();
      for (int k = 0; k < KK_COUNT; ++k)
        per_kind_ms[k] = prof.count[k] ? prof.total_ms[k] / (double)prof.count[k] : 0.0;
    }
  });
}

// Batched transforms: `batch` stacks of one shape through ONE cached plan, back to back on one
// stream -- the counterpart of the cufftPlanMany path of bench/bench_gpu_many_nd_fft.cu:403-463.
namespace {
struct FftBatch {
  std::shared_ptr<Plan3D> plan;
  be::stream_t s = nullptr;
  std::vector<float*> vol;
  std::vector<cfloat*> nyq;
  FftBatch(int dev, const int* dims, int batch) {
    be::set_device(dev);
    plan = PlanStore::get().add(dev, to_shape(dims));
    s = be::stream_create();
    try {
      for (int b = 0; b < batch; ++b) {
        vol.push_back((float*)be::dmalloc(plan->main_bytes()));
        be::dzero(vol.back(), plan->main_bytes(), s);
        nyq.push_back(nullptr);
        if (plan->nyq_bytes()) {
          nyq.back() = (cfloat*)be::dmalloc(plan->nyq_bytes());
          be::dzero(nyq.back(), plan->nyq_bytes(), s);
        }
      }
    } catch (...) {
      release();
      throw;
    }
  }
  void release() {
    try {
      be::stream_sync(s);
    } catch (...) {
    }
    for (float* v : vol) be::dfree(v);
    for (cfloat* n : nyq) be::dfree(n);
    vol.clear();
    nyq.clear();
    if (s) be::stream_destroy(s);
    s = nullptr;
  }
  ~FftBatch() { release(); }
};
}  // namespace

int mvn_fft3_many_r2c(int device, const int dims[3], int batch, const float* real, float* spec) {
  return guarded("mvn_fft3_many_r2c", [&] {
    if (batch < 1 || !real || !spec) throw std::invalid_argument("bad batch or null buffers");
    const int dev = pick_device(device);
    std::lock_guard<std::mutex> lk(device_mutex(dev));
    FftBatch f(dev, dims, batch);
    const Layout& L = f.plan->L;
    const size_t nreal = (size_t)L.d0 * L.d1 * L.d2;
    const int nc = L.d2 / 2 + 1;
    const size_t nspec = (size_t)L.d0 * L.d1 * nc;
    for (int b = 0; b < batch; ++b)
      be::h2d_2d(f.vol[b], (size_t)L.RP * 4, real + b * nreal, (size_t)L.d2 * 4, (size_t)L.d2 * 4,
                 L.rows, f.s);
    for (int b = 0; b < batch; ++b) f.plan->forward(f.vol[b], f.nyq[b], f.s);
    std::vector<cfloat> main_h(L.rows * (size_t)L.C), nyq_h(L.nyq_cplx());
    const std::vector<int>& i0 = f.plan->ax0.host.inv;
    const std::vector<int>& i1 = f.plan->ax1.host.inv;
    const std::vector<int>& i2 = f.plan->ax2.host.inv;
    for (int b = 0; b < batch; ++b) {
      be::d2h(main_h.data(), f.vol[b], main_h.size() * sizeof(cfloat), f.s);
      if (L.even) be::d2h(nyq_h.data(), f.nyq[b], nyq_h.size() * sizeof(cfloat), f.s);
      be::stream_sync(f.s);
      cfloat* out = reinterpret_cast<cfloat*>(spec) + b * nspec;
      for (int k0 = 0; k0 < L.d0; ++k0)
        for (int k1 = 0; k1 < L.d1; ++k1) {
          const size_t srow = (size_t)i0[k0] * L.d1 + (size_t)i1[k1];
          cfloat* o = out + ((size_t)k0 * L.d1 + k1) * (size_t)nc;
          for (int k2 = 0; k2 < L.C; ++k2) o[k2] = main_h[srow * L.C + (L.even ? i2[k2] : k2)];
          if (L.even) o[L.C] = nyq_h[srow];
        }
    }
  });
}

int mvn_fft3_many_time(int device, const int dims[3], int batch, int direction, int reps,
                       float* ms) {
  return guarded("mvn_fft3_many_time", [&] {
    if (batch < 1 || reps < 1 || !ms) throw std::invalid_argument("bad batch, reps or null ms");
    const int dev = pick_device(device);
    std::lock_guard<std::mutex> lk(device_mutex(dev));
    FftBatch f(dev, dims, batch);
    const Layout& L = f.plan->L;
    {
      std::vector<float> host(L.real_floats());
      unsigned x = 12345u;
      for (size_t i = 0; i < host.size(); ++i) {
        x = x * 1664525u + 1013904223u;
        host[i] = (float)(x >> 8) * (1.0f / 16777216.0f) - 0.5f;
      }
      for (int b = 0; b < batch; ++b) be::h2d(f.vol[b], host.data(), host.size() * 4, f.s);
      be::stream_sync(f.s);
    }
    const float keep = 1.0f / (float)L.logical();
    auto sweep = [&] {
      for (int b = 0; b < batch; ++b) {
        if (direction == 0)
          f.plan->forward(f.vol[b], f.nyq[b], f.s);
        else
          f.plan->backward(f.vol[b], f.nyq[b], keep, f.s);
      }
    };
    sweep();  // warm-up
    be::event_t a = be::event_create(), b2 = be::event_create();
    be::event_record(a, f.s);
    for (int i = 0; i < reps; ++i) sweep();
    be::event_record(b2, f.s);
    be::event_sync(b2);
    *ms = be::event_elapsed_ms(a, b2) / (float)reps;
    be::event_destroy(a);
    be::event_destroy(b2);
  });
}

// ---------------------------------------------------------------------------------------------
// resident engine
// ---------------------------------------------------------------------------------------------
int mvn_engine_create(int device, const int dims[3], int num_views, mvn_engine** out) {
  return guarded("mvn_engine_create", [&] {
    if (!out) throw std::invalid_argument("null out");
    *out = nullptr;
    std::unique_ptr<mvn_engine> h(new mvn_engine());
    h->impl.reset(new Engine(pick_device(device), to_shape(dims), num_views));
    *out = h.release();
  });
}

int mvn_engine_destroy(mvn_engine* e) {
  return guarded("mvn_engine_destroy", [&] { delete e; });
}

#define MVN_ENGINE_CALL(name, ...)                                 \
  return guarded(name, [&] {                                       \
    if (!e || !e->impl) throw std::invalid_argument("null engine"); \
    Engine& E = *e->impl;                                          \
    (void)E;                                                       \
    __VA_ARGS__;                                                   \
  })

int mvn_engine_set_view(mvn_engine* e, int v, const float* image, const float* weights,
                        const float* kernel1, const int k1dims[3], const float* kernel2,
                        const int k2dims[3]) {
  MVN_ENGINE_CALL("mvn_engine_set_view", {
    if (!image || !weights || !kernel1 || !kernel2 || !k1dims || !k2dims)
      throw std::invalid_argument("null argument");
    E.set_view(v, image, weights, kernel1, k1dims, kernel2, k2dims);
  });
}

int mvn_engine_set_psi(mvn_engine* e, const float* psi) {
  MVN_ENGINE_CALL("mvn_engine_set_psi", E.set_psi(psi));
}

int mvn_engine_get_psi(mvn_engine* e, float* psi) {
  MVN_ENGINE_CALL("mvn_engine_get_psi", E.get_psi(psi));
}

int mvn_engine_iterate(mvn_engine* e, int iterations, double lambda, float min_value) {
  MVN_ENGINE_CALL("mvn_engine_iterate", E.iterate(iterations, lambda, min_value));
}

int mvn_engine_compute_delta(mvn_engine* e, double lambda, float min_value) {
  MVN_ENGINE_CALL("mvn_engine_compute_delta", E.compute_delta(lambda, min_value));
}

int mvn_engine_apply_delta(mvn_engine* e) {
  MVN_ENGINE_CALL("mvn_engine_apply_delta", E.apply_delta());
}

int mvn_engine_delta_chunks(mvn_engine* e, int wanted) {
  int n = 1;
  int rc = guarded("mvn_engine_delta_chunks", [&] {
    if (!e || !e->impl) throw std::invalid_argument("null engine");
    n = e->impl->delta_chunks(wanted);
  });
  return rc < 0 ? rc : n;
}

int mvn_engine_delta_chunk_range(mvn_engine* e, int c, int n, size_t* first_float, size_t* n_floats) {
  MVN_ENGINE_CALL("mvn_engine_delta_chunk_range", {
    if (!first_float || !n_floats) throw std::invalid_argument("null argument");
    E.delta_chunk_range(c, n, first_float, n_floats);
  });
}

int mvn_engine_compute_delta_head(mvn_engine* e, double lambda, float min_value) {
  MVN_ENGINE_CALL("mvn_engine_compute_delta_head", E.compute_delta_head(lambda, min_value));
}

int mvn_engine_compute_delta_chunk(mvn_engine* e, int c, int n) {
  MVN_ENGINE_CALL("mvn_engine_compute_delta_chunk", E.compute_delta_chunk(c, n));
}

int mvn_engine_apply_delta_chunk(mvn_engine* e, int c, int n, int feed_next) {
  MVN_ENGINE_CALL("mvn_engine_apply_delta_chunk", E.apply_delta_chunk(c, n, feed_next != 0));
}

int mvn_engine_delta_ptr(mvn_engine* e, void** dev_ptr, size_t* n_floats) {
  MVN_ENGINE_CALL("mvn_engine_delta_ptr", {
    *dev_ptr = E.delta_ptr();
    *n_floats = E.volume_floats();
  });
}

int mvn_engine_bind_delta(mvn_engine* e, void* dev_ptr) {
  MVN_ENGINE_CALL("mvn_engine_bind_delta", E.bind_delta((float*)dev_ptr));
}

int mvn_engine_set_halo_hook(mvn_engine* e, void (*fn)(void*, void*, int, int), void* user, int drain) {
  MVN_ENGINE_CALL("mvn_engine_set_halo_hook", E.set_halo_hook(fn, user, (drain & 1) != 0, (drain & 2) != 0));
}

int mvn_engine_would_be_direct(mvn_engine* e, const int kdims[3]) {
  int yes = 0;
  const int rc = guarded("mvn_engine_would_be_direct", [&] {
    if (!e || !e->impl || !kdims) throw std::invalid_argument("null argument");
    be::set_device(e->impl->device());
    yes = e->impl->would_be_direct(kdims) ? 1 : 0;
  });
  return rc < 0 ? rc : yes;
}

int mvn_engine_set_halo_planes(mvn_engine* e, int planes, int split) {
  MVN_ENGINE_CALL("mvn_engine_set_halo_planes", E.set_halo_planes(planes, split != 0));
}

int mvn_engine_poison_ptr(mvn_engine* e, void** dev_ptr) {
  MVN_ENGINE_CALL("mvn_engine_poison_ptr", {
    if (!dev_ptr) throw std::invalid_argument("null argument");
    *dev_ptr = E.poison_ptr();
  });
}

int mvn_engine_bind_poison(mvn_engine* e, void* dev_ptr) {
  MVN_ENGINE_CALL("mvn_engine_bind_poison", E.bind_poison((unsigned*)dev_ptr));
}

int mvn_engine_poison_get(mvn_engine* e, unsigned* value) {
  MVN_ENGINE_CALL("mvn_engine_poison_get", {
    if (!value) throw std::invalid_argument("null argument");
    *value = E.poison_get();
  });
}

int mvn_engine_poison_merge(mvn_engine* e, unsigned value) {
  MVN_ENGINE_CALL("mvn_engine_poison_merge", E.poison_merge(value));
}

int mvn_engine_copy_planes(mvn_engine* e, void* spectrum, int plane0, int nplanes, void* buffer, int to_buffer) {
  MVN_ENGINE_CALL("mvn_engine_copy_planes", E.copy_planes(spectrum, plane0, nplanes, buffer, (to_buffer & 1) != 0, (to_buffer & 2) != 0, (to_buffer & 4) == 0));
}

int mvn_engine_psi_ptr(mvn_engine* e, void** dev_ptr, size_t* n_floats) {
  MVN_ENGINE_CALL("mvn_engine_psi_ptr", {
    *dev_ptr = E.psi_ptr();
    *n_floats = E.volume_floats();
  });
}

int mvn_engine_stream(mvn_engine* e, void** hip_stream) {
  MVN_ENGINE_CALL("mvn_engine_stream", *hip_stream = E.stream());
}

int mvn_engine_sync(mvn_engine* e) { MVN_ENGINE_CALL("mvn_engine_sync", E.sync()); }

int mvn_engine_time_iterate(mvn_engine* e, int iterations, double lambda, float min_value,
                            float* ms) {
  MVN_ENGINE_CALL("mvn_engine_time_iterate", {
    be::set_device(E.device());
    be::event_t a = be::event_create(), b = be::event_create();
    be::event_record(a, E.stream());
    E.iterate(iterations, lambda, min_value);
    be::event_record(b, E.stream());
    be::event_sync(b);
    *ms = be::event_elapsed_ms(a, b);
    be::event_destroy(a);
    be::event_destroy(b);
    E.sync();
  });
}

int mvn_engine_profile(mvn_engine* e, int enable) {
  MVN_ENGINE_CALL("mvn_engine_profile", {
    E.sync();
    E.profiler().reset();
    E.profiler().enabled = enable != 0;
    E.profiler().sample_every = enable > 1 ? enable : 1;
  });
}

int mvn_engine_profile_read(mvn_engine* e, int kind, double* total_ms, long* launches) {
  MVN_ENGINE_CALL("mvn_engine_profile_read", {
    if (kind < 0 || kind >= KK_COUNT) throw std::out_of_range("kernel kind");
    E.sync();
    *total_ms = E.profiler().total_ms[kind];
    *launches = E.profiler().count[kind];
  });
}

int mvn_kernel_kind_count(void) { return KK_COUNT; }
const char* mvn_kernel_kind_name(int kind) { return kernel_kind_name(kind); }

size_t mvn_engine_B(mvn_engine* e) { return (e && e->impl) ? e->impl->layout().B() : 0; }

// ---------------------------------------------------------------------------------------------
// slab-decomposed engine (sequential sweep across several GPUs)
// ---------------------------------------------------------------------------------------------
int mvn_slab_create(int device, const int dims[3], int nranks, int rank, int num_views,
                    mvn_slab** out) {
  return guarded("mvn_slab_create", [&] {
    if (!out || !dims) throw std::invalid_argument("null argument");
    *out = nullptr;
    std::unique_ptr<mvn_slab> h(new mvn_slab());
    h->impl.reset(new SlabEngine(pick_device(device), to_shape(dims), nranks, rank, num_views));
    *out = h.release();
  });
}

int mvn_slab_destroy(mvn_slab* e) {
  return guarded("mvn_slab_destroy", [&] { delete e; });
}

#define MVN_SLAB_CALL(name, ...)                                        \
  return guarded(name, [&] {                                            \
    if (!e || !e->impl) throw std::invalid_argument("null slab engine"); \
    SlabEngine& E = *e->impl;                                           \
    (void)E;                                                            \
    __VA_ARGS__;                                                        \
  })

int mvn_slab_set_view(mvn_slab* e, int v, const float* image_slab, const float* weights_slab,
                      const float* kernel1, const int k1dims[3], const float* kernel2,
                      const int k2dims[3]) {
  MVN_SLAB_CALL("mvn_slab_set_view", {
    if (!image_slab || !weights_slab || !kernel1 || !kernel2 || !k1dims || !k2dims)
      throw std::invalid_argument("null argument");
    E.set_view(v, image_slab, weights_slab, kernel1, k1dims, kernel2, k2dims);
  });
}

int mvn_slab_set_psi(mvn_slab* e, const float* psi_slab) {
  MVN_SLAB_CALL("mvn_slab_set_psi", {
    if (!psi_slab) throw std::invalid_argument("null argument");
    E.set_psi(psi_slab);
  });
}

int mvn_slab_get_psi(mvn_slab* e, float* psi_slab) {
  MVN_SLAB_CALL("mvn_slab_get_psi", {
    if (!psi_slab) throw std::invalid_argument("null argument");
    E.get_psi(psi_slab);
  });
}

int mvn_slab_buffer_sizes(mvn_slab* e, size_t* main_floats, size_t* nyq_floats) {
  MVN_SLAB_CALL("mvn_slab_buffer_sizes", {
    if (main_floats) *main_floats = E.main_floats();
    if (nyq_floats) *nyq_floats = E.nyq_floats();
  });
}

int mvn_slab_bind_buffers(mvn_slab* e, void* a_main, void* b_main, void* a_nyq, void* b_nyq) {
  MVN_SLAB_CALL("mvn_slab_bind_buffers",
                E.bind_buffers((float*)a_main, (float*)b_main, (float*)a_nyq, (float*)b_nyq));
}

int mvn_slab_buffers(mvn_slab* e, void** a_main, void** b_main, void** a_nyq, void** b_nyq) {
  MVN_SLAB_CALL("mvn_slab_buffers", {
    if (a_main) *a_main = E.a_main();
    if (b_main) *b_main = E.b_main();
    if (a_nyq) *a_nyq = E.a_nyq();
    if (b_nyq) *b_nyq = E.b_nyq();
  });
}

int mvn_slab_begin(mvn_slab* e) { MVN_SLAB_CALL("mvn_slab_begin", E.begin_sweeps()); }
int mvn_slab_pack(mvn_slab* e, int v, int conv) {
  MVN_SLAB_CALL("mvn_slab_pack", {
    if (conv != 0 && conv != 1) throw std::invalid_argument("conv must be 0 or 1");
    E.step_pack(v, conv);
  });
}
int mvn_slab_mid(mvn_slab* e, int v, int conv) {
  MVN_SLAB_CALL("mvn_slab_mid", {
    if (conv != 0 && conv != 1) throw std::invalid_argument("conv must be 0 or 1");
    E.step_mid(v, conv);
  });
}
int mvn_slab_unpack(mvn_slab* e, int v, int conv, double lambda, float min_value, int feed_next) {
  MVN_SLAB_CALL("mvn_slab_unpack", {
    if (conv != 0 && conv != 1) throw std::invalid_argument("conv must be 0 or 1");
    E.step_unpack(v, conv, lambda, min_value, feed_next != 0);
  });
}
int mvn_slab_sync(mvn_slab* e) { MVN_SLAB_CALL("mvn_slab_sync", E.sync()); }
int mvn_slab_stream(mvn_slab* e, void** hip_stream) {
  MVN_SLAB_CALL("mvn_slab_stream", {
    if (!hip_stream) throw std::invalid_argument("null argument");
    *hip_stream = (void*)E.stream();
  });
}

}  // extern "C"
