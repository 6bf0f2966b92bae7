// mvn_kernels.hip -- the product's device backend: gfx950 kernels + HIP runtime plumbing.
//
// Kernels (all HBM-bandwidth bound; no MFMA -- this path has no dense contraction):
//   k_rows_r2c / k_rows_c2r   last-axis real<->half-complex passes, LDS-staged, with the RL
//                             pointwise math fused into the c2r epilogue
//                             (replaces cufftExecR2C/C2R inc/cufft_utils.cuh:54-55,72-73 and
//                             device_divide / device_(regularized_)final_values,
//                             inc/cuda_kernels.cuh:14-112)
//   k_strided<MODE>           c2c passes along dim1 / dim0 on tiles of neighbouring columns;
//                             MODE FWD_MUL_INV fuses forward dim0, the PSF-spectrum multiply
//                             (multiply_scaled, inc/cuda_kernels.cuh:213-242) and inverse dim0
//   kx_rows_* / kx_strided    the same passes with every length, radix and tile shape a
//                             template constant (mvn_fixed.hpp); kx_rows_c2r_r2c additionally
//                             fuses c2r + pointwise step + r2c of the next convolution; the
//                             strided ones walk over several tiles per workgroup
//   k_scatter_psf             device-side wrapped insert (fftShiftKernel,
//                             src/multiviewnative.cu:154-192)
//   k_divide / k_update / k_update_legacy_tikhonov / k_axpy1   stand-alone pointwise ops for the
//                                   legacy entry points
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>
#include <stdexcept>
#include <string>

// Every switch that changes what a kernel computes or how much LDS it asks for - timing experiments - is compiled
// only with -DMVN_EXPERIMENTS, which `make all` never sets (`make probe` / `make variant EXTRA=..` do);
// tests/test_abi_symbols.py checks the product library for their traces.
#if defined(MVN_PROBE) && !defined(MVN_EXPERIMENTS)
#error "MVN_PROBE is a timing experiment: build it with -DMVN_EXPERIMENTS (make probe)"
#endif
#ifdef MVN_PROBE
// Timing probe (never in the product build): every workgroup works on tile (index mod wrap), so the
// working set of a pass is a few MB that stay in the L2 / Infinity Cache -- what a pass costs when
// HBM is out of the picture.  Results are garbage.  MVN_PROBE_WRAP_ST / MVN_PROBE_WRAP_ROWS = tiles.
__device__ int g_probe_wrap_st = 0;
__device__ int g_probe_wrap_rows = 0;
#define MVN_PROBE_BLOCK(b) (g_probe_wrap_st > 0 ? (b) % g_probe_wrap_st : (b))
#define MVN_PROBE_TILE(t) (g_probe_wrap_rows > 0 ? (t) % g_probe_wrap_rows : (t))
#else
#define MVN_PROBE_TILE(t) (t)
#endif
#include "mvn_backend.hpp"
#include "mvn_fixed_geom.hpp"
#include "mvn_wave_rows.hpp"

#define HIP_CHECK(expr)                                                                      \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess)                                                                    \
      throw std::runtime_error(std::string("HIP error '") + hipGetErrorString(_e) + "' at " + \
                               __FILE__ + ":" + std::to_string(__LINE__) + " (" #expr ")");  \
  } while (0)

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
template <bool EVEN, int T>
__global__ void __launch_bounds__(512) k_rows_r2c(const RowsParams p) {
  extern __shared__ __attribute__((aligned(16))) char mvn_smem[];
  if (EVEN)
    rows_r2c_even_body<T>(p, (long)blockIdx.x, (int)threadIdx.x, (int)blockDim.x, (cfloat*)mvn_smem);
  else
    rows_r2c_odd_body<T>(p, (long)blockIdx.x, (int)threadIdx.x, (int)blockDim.x, (cfloat*)mvn_smem);
}

template <bool EVEN, int T>
__global__ void __launch_bounds__(512) k_rows_c2r(const RowsParams p) {
  extern __shared__ __attribute__((aligned(16))) char mvn_smem[];
  if (EVEN)
    rows_c2r_even_body<T>(p, (long)blockIdx.x, (int)threadIdx.x, (int)blockDim.x, (cfloat*)mvn_smem);
  else
    rows_c2r_odd_body<T>(p, (long)blockIdx.x, (int)threadIdx.x, (int)blockDim.x, (cfloat*)mvn_smem);
}

// NYQ only tags the launches that work on the Nyquist plane, so that profilers list them apart
// run-time-radix form of the fused c2r + pointwise + r2c pass (any even d2)
template <int T>
__global__ void __launch_bounds__(512) k_rows_c2r_r2c(const RowsParams p) {
  extern __shared__ __attribute__((aligned(16))) char mvn_smem[];
  rows_c2r_even_body<T, true>(p, (long)blockIdx.x, (int)threadIdx.x, (int)blockDim.x, (cfloat*)mvn_smem);
}

template <int MODE, int T, bool NYQ>
__global__ void __launch_bounds__(512) k_strided(const StridedParams p) {
  extern __shared__ __attribute__((aligned(16))) char mvn_smem[];
  strided_body<MODE, T, NYQ>(p, (long)blockIdx.x, (int)threadIdx.x, (int)blockDim.x, (cfloat*)mvn_smem);
}

// ---- compile-time specialised kernels for power-of-two lengths (mvn_fixed.hpp) ----------------
// (LINES: the half-spectrum side in the line layout of the fused middle pass, mvn_fixed.hpp fx_lines_base)
template <int H, bool LINES = false>
__global__ void __launch_bounds__(FxRowsCfg<H>::NT) kx_rows_r2c(const RowsParams p) {
  extern __shared__ __attribute__((aligned(16))) char mvn_smem[];
  typedef FxCtx<FxRowsRegs<H>, FxRowsCfg<H>::NT> Ctx;
  Ctx ctx;
  ctx.tid = (int)threadIdx.x;
  fx_rows_run<H, 0, MVN_EPI_STORE, Ctx, LINES>(p, (long)blockIdx.x, (long)gridDim.x, (cfloat*)mvn_smem, ctx);
}

template <int H, int EPI, bool LINES = false>
__global__ void __launch_bounds__(FxRowsCfg<H>::NT) kx_rows_c2r(const RowsParams p0) {
  extern __shared__ __attribute__((aligned(16))) char mvn_smem[];
  RowsParams p = p0;
  mvn_arm_poison(p.epi);
  typedef FxCtx<FxRowsRegs<H>, FxRowsCfg<H>::NT> Ctx;
  Ctx ctx;
  ctx.tid = (int)threadIdx.x;
  fx_rows_run<H, 1, EPI, Ctx, LINES>(p, (long)blockIdx.x, (long)gridDim.x, (cfloat*)mvn_smem, ctx);
}

// (The divide form needs 93 VGPRs, just above the 84 that would let three 512-thread workgroups
// share a CU; forcing it there with a waves-per-SIMD bound spilled 56 bytes and gained nothing.)
template <int H, int EPI, bool LINES = false>
__global__ void __launch_bounds__(FxRowsCfg<H>::NT) kx_rows_c2r_r2c(const RowsParams p0) {
  extern __shared__ __attribute__((aligned(16))) char mvn_smem[];
  RowsParams p = p0;
  mvn_arm_poison(p.epi);
  typedef FxCtx<FxRowsRegs<H>, FxRowsCfg<H>::NT> Ctx;
  Ctx ctx;
  ctx.tid = (int)threadIdx.x;
  fx_rows_run<H, 2, EPI, Ctx, LINES>(p, (long)MVN_PROBE_TILE(blockIdx.x), (long)gridDim.x, (cfloat*)mvn_smem, ctx);
}

// the fused middle pass (mvn_mid_fused.hpp): one column (piece) per workgroup
template <int K>
__global__ void __launch_bounds__(MF_NT) kf_mid(const MidFusedParams p) {
  extern __shared__ __attribute__((aligned(16))) char mvn_smem[];
  FxCtx<MfRegs<K>, MF_NT> ctx;
  ctx.tid = (int)threadIdx.x;
  mf_body<K>(p, (long)blockIdx.x, (cfloat*)mvn_smem, ctx);
}
__global__ void __launch_bounds__(MF_NT) kf_mid_taps(const MidFusedParams p) {
  extern __shared__ __attribute__((aligned(16))) char mvn_smem[];
  FxCtx<MfRegs<1>, MF_NT> ctx;
  ctx.tid = (int)threadIdx.x;
  mf_taps_body(p, (long)blockIdx.x, (cfloat*)mvn_smem, ctx);
}

// last-axis passes for d2 = 512 in which a row never leaves its half-wave (mvn_wave_rows.hpp): no
// workgroup barrier inside the loop over rows, 4 LDS exchanges per fused pass
template <int MODE, int EPI>
__global__ void __launch_bounds__(WrCfg::NT, 4) kw_rows(const RowsParams p0) {
  extern __shared__ __attribute__((aligned(16))) char mvn_smem[];
  RowsParams p = p0;
  if (MODE != MVN_WR_R2C) mvn_arm_poison(p.epi);
  FxCtx<WrRegs, WrCfg::NT> ctx;
  ctx.tid = (int)threadIdx.x;
  wr_rows_body<MODE, EPI>(p, (long)blockIdx.x, (long)gridDim.x, (cfloat*)mvn_smem, ctx);
}

// `main_grid` walking workgroups work on the main array; the workgroups behind them (plain dim1 passes only) take
// ONE line of the Nyquist plane each - 4 KB at 512 points, through the run-time-radix body - so that the plane
// needs no launch, stream and fork / join pair of its own: 512 two-microsecond workgroups that fill the slots
// the walkers leave at the end of the launch, instead of 32 workgroups of 16 lines that queued for a slot beside
// the next full-volume pass on a second hardware queue (round 3: 25 % of all GPU-busy time for 0.4 % of the bytes).
template <int N, int MODE>
__global__ void __launch_bounds__((FxStridedSel<N, MODE>::NT), (FxStridedSel<N, MODE>::WAVES))
    kx_strided(const StridedParams p, const StridedParams rider, unsigned main_grid) {
  extern __shared__ __attribute__((aligned(16))) char mvn_smem[];
  if constexpr (MODE != MVN_ST_FWD_MUL_INV) {
    if (blockIdx.x >= main_grid) {
      strided_body<MODE, 1, true>(rider, (long)(blockIdx.x - main_grid), (int)threadIdx.x, (int)blockDim.x, (cfloat*)mvn_smem);
      return;
    }
  }
  typename FxStridedSel<N, MODE>::Ctx ctx;
  ctx.tid = (int)threadIdx.x;
  FxStridedSel<N, MODE>::run(p, (long)blockIdx.x, p.nblocks, (long)main_grid, (cfloat*)mvn_smem, ctx);
}

template <int N, int MODE>
__global__ void __launch_bounds__((FxSplitCfg<N>::NT)) kx_strided_split(const StridedParams p, const StridedParams rider,
                                                                       unsigned main_grid) {
  extern __shared__ __attribute__((aligned(16))) char mvn_smem[];
  if (blockIdx.x >= main_grid) {
    strided_body<MODE, 1, true>(rider, (long)(blockIdx.x - main_grid), (int)threadIdx.x, (int)blockDim.x, (cfloat*)mvn_smem);
    return;
  }
  FxCtx<FxSplitRegs<N>, FxSplitCfg<N>::NT> ctx;
  ctx.tid = (int)threadIdx.x;
  fx_strided_split_body<N, MODE>(p, (long)blockIdx.x, p.nblocks, (long)main_grid, (cfloat*)mvn_smem, ctx);
}

// direct dim0 convolution (mvn_dim0_direct.hpp): one bin per work item, all of dim0
// (K = 25 sits a register or two above the 128 that let four waves share a SIMD: bounded there)
template <int K, int PF = MVN_D0_PF>
__global__ void __launch_bounds__(MVN_D0_WG, (K == 25 ? 4 : 1)) kd_dim0(const Dim0DirectParams p, unsigned main_blocks) {
  extern __shared__ __attribute__((aligned(16))) char mvn_smem[];
  if (blockIdx.x >= main_blocks) {  // packed Nyquist: one (k1, -k1) pair of the DC column per workgroup
    const int pair = (int)(blockIdx.x - main_blocks);
    mvn_dim0_dc_load(p, pair, (cfloat*)mvn_smem, (int)threadIdx.x, MVN_D0_WG);
    __syncthreads();
    mvn_dim0_dc_compute(p, pair, (const cfloat*)mvn_smem, (int)threadIdx.x, MVN_D0_WG);
    return;
  }
  Dim0DirectParams q;
  long b;
  int zs, nout;
  // (zs, nout and the arrays in q depend on the workgroup only: scalar registers)
  if (mvn_dim0_job(p, (long)blockIdx.x, (int)threadIdx.x, q, b, zs, nout))
    mvn_dim0_direct_column<K, PF>(q, b, zs, nout);
}

__global__ void k_scatter_psf(const float* kernel, int k0, int k1, int k2, float* target, int D0,
                              int D1, int D2, long pitch, float scale) {
  const long total = (long)k0 * k1 * k2;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride)
    mvn_scatter_psf_item(kernel, k0, k1, k2, target, D0, D1, D2, pitch, scale, i);
}

// one workgroup row per (y, z): x runs over lanes, so both sides are read / written in contiguous
// runs whatever the offsets are
__global__ void k_copy3d(float* __restrict__ dst, long drow, long dplane, const float* __restrict__ src,
                         long srow, long splane, int nx, int ny, int nz) {
  const long rows = (long)ny * nz;
  for (long r = blockIdx.x; r < rows; r += gridDim.x) {
    const long z = r / ny, y = r - z * ny;
    const float* s = src + z * splane + y * srow;
    float* d = dst + z * dplane + y * drow;
    for (int x = threadIdx.x; x < nx; x += blockDim.x) d[x] = s[x];
  }
}

__global__ void k_divide(const float* __restrict__ view, float* __restrict__ inout, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    inout[i] = mvn_quotient(view[i], inout[i]);
}

__global__ void k_update(float* __restrict__ psi, const float* __restrict__ integral,
                         const float* __restrict__ weights, size_t n, double lambda,
                         float lambda_inv, float min_value) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    MVN_FP_EXACT
    const float last = psi[i];
    const float next = mvn_next_value(last, integral[i], lambda, lambda_inv, min_value);
    psi[i] = weights[i] * (next - last) + last;
  }
}

__global__ void k_update_legacy_tikhonov(float* __restrict__ image,
                                        const float* __restrict__ integral,
                                        const float* __restrict__ weights, size_t n,
                                        float lambda_f, float min_value) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    image[i] = mvn_legacy_tikhonov_value(image[i], integral[i], weights[i], lambda_f, min_value);
}

__global__ void k_axpy1(float* __restrict__ psi, const float* __restrict__ delta, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    psi[i] += delta[i];
}

// ---------------------------------------------------------------------------------------------
// runtime plumbing
// ---------------------------------------------------------------------------------------------
namespace mvn {
namespace be {

const char* backend_name() { return "hip-gfx950"; }

int device_count() {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

void set_device(int dev) { HIP_CHECK(hipSetDevice(dev)); }

int get_device() {
  int d = 0;
  HIP_CHECK(hipGetDevice(&d));
  return d;
}

void device_name(int dev, char* name256) {
  hipDeviceProp_t prop;
  HIP_CHECK(hipGetDeviceProperties(&prop, dev));
  std::memset(name256, 0, 256);
  std::strncpy(name256, prop.name, 255);
  // some ROCm builds leave the marketing name empty: fall back to the gfx target
  if (!name256[0]) std::snprintf(name256, 256, "AMD GPU %s", prop.gcnArchName);
}

long long device_total_mem(int dev) {
  hipDeviceProp_t prop;
  HIP_CHECK(hipGetDeviceProperties(&prop, dev));
  return (long long)prop.totalGlobalMem;
}

void device_mem_info(size_t* free_b, size_t* total_b) { HIP_CHECK(hipMemGetInfo(free_b, total_b)); }

// "compute capability" of a gfx target: gfx950 -> (95, 0); gfx90a -> (90, 10)
void device_arch(int dev, int* major, int* minor) {
  hipDeviceProp_t prop;
  HIP_CHECK(hipGetDeviceProperties(&prop, dev));
  const char* a = prop.gcnArchName;
  *major = 0;
  *minor = 0;
  if (std::strncmp(a, "gfx", 3) != 0) return;
  a += 3;
  size_t len = 0;
  while (a[len] && a[len] != ':') ++len;
  if (len == 0) return;
  for (size_t i = 0; i + 1 < len; ++i)
    if (a[i] >= '0' && a[i] <= '9') *major = *major * 10 + (a[i] - '0');
  const char c = a[len - 1];
  *minor = (c >= '0' && c <= '9') ? c - '0' : ((c >= 'a' && c <= 'f') ? 10 + c - 'a' : 0);
}

void* dmalloc(size_t bytes) {
  void* p = nullptr;
  HIP_CHECK(hipMalloc(&p, bytes ? bytes : 1));
  return p;
}

void dfree(void* p) {
  if (p) (void)hipFree(p);
}

static hipStream_t hs(stream_t s) { return (hipStream_t)s; }

void h2d(void* d, const void* h, size_t bytes, stream_t s) {
  HIP_CHECK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, hs(s)));
}
void d2h(void* h, const void* d, size_t bytes, stream_t s) {
  HIP_CHECK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, hs(s)));
}
void d2d(void* dst, const void* src, size_t bytes, stream_t s) {
  HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, hs(s)));
}
void h2d_2d(void* d, size_t dpitch, const void* h, size_t hpitch, size_t width, size_t height,
            stream_t s) {
  if (dpitch == width && hpitch == width) return h2d(d, h, width * height, s);
  HIP_CHECK(hipMemcpy2DAsync(d, dpitch, h, hpitch, width, height, hipMemcpyHostToDevice, hs(s)));
}
void d2h_2d(void* h, size_t hpitch, const void* d, size_t dpitch, size_t width, size_t height,
            stream_t s) {
  if (dpitch == width && hpitch == width) return d2h(h, d, width * height, s);
  HIP_CHECK(hipMemcpy2DAsync(h, hpitch, d, dpitch, width, height, hipMemcpyDeviceToHost, hs(s)));
}
void d2d_2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height,
            stream_t s) {
  if (dpitch == width && spitch == width) return d2d(dst, src, width * height, s);
  HIP_CHECK(hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyDeviceToDevice, hs(s)));
}
void dzero(void* d, size_t bytes, stream_t s) { HIP_CHECK(hipMemsetAsync(d, 0, bytes, hs(s))); }

void enable_peer_access(int dev, int peer) {
  if (dev == peer) return;
  int can = 0;
  HIP_CHECK(hipDeviceCanAccessPeer(&can, dev, peer));
  if (!can) throw std::runtime_error("mvn: device " + std::to_string(dev) + " cannot access device " + std::to_string(peer));
  int cur = 0;
  HIP_CHECK(hipGetDevice(&cur));
  HIP_CHECK(hipSetDevice(dev));
  const hipError_t e = hipDeviceEnablePeerAccess(peer, 0);
  (void)hipSetDevice(cur);
  if (e == hipErrorPeerAccessAlreadyEnabled) {
    (void)hipGetLastError();
    return;
  }
  HIP_CHECK(e);
}
void copy_peer(void* dst, int dst_dev, const void* src, int src_dev, size_t bytes, stream_t s) {
  if (dst_dev == src_dev)
    HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, hs(s)));
  else
    HIP_CHECK(hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, bytes, hs(s)));
}

stream_t stream_create() {
  hipStream_t s;
  HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  return (stream_t)s;
}
stream_t stream_create_upload() {
  int least = 0, greatest = 0;
  HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
  hipStream_t s;
  HIP_CHECK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, greatest));
  return (stream_t)s;
}
void stream_destroy(stream_t s) {
  if (s) (void)hipStreamDestroy(hs(s));
}
void stream_sync(stream_t s) { HIP_CHECK(hipStreamSynchronize(hs(s))); }

void stream_wait_event(stream_t s, event_t e) {
  HIP_CHECK(hipStreamWaitEvent(hs(s), (hipEvent_t)e, 0));
}

event_t event_create() {
  hipEvent_t e;
  HIP_CHECK(hipEventCreate(&e));
  return (event_t)e;
}
event_t event_create_sync() {
  hipEvent_t e;
  HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return (event_t)e;
}
void event_destroy(event_t e) {
  if (e) (void)hipEventDestroy((hipEvent_t)e);
}
void event_record(event_t e, stream_t s) { HIP_CHECK(hipEventRecord((hipEvent_t)e, hs(s))); }
void event_sync(event_t e) { HIP_CHECK(hipEventSynchronize((hipEvent_t)e)); }
float event_elapsed_ms(event_t a, event_t b) {
  float ms = 0.f;
  HIP_CHECK(hipEventElapsedTime(&ms, (hipEvent_t)a, (hipEvent_t)b));
  return ms;
}

bool graphs_supported() { return true; }
void capture_begin(stream_t s) {
  // thread-local mode: the staging thread of the ABI call may allocate / free meanwhile
  HIP_CHECK(hipStreamBeginCapture(hs(s), hipStreamCaptureModeThreadLocal));
}
graph_exec_t capture_end(stream_t s) {
  hipGraph_t g = nullptr;
  HIP_CHECK(hipStreamEndCapture(hs(s), &g));
  hipGraphExec_t e = nullptr;
  hipError_t err = hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  HIP_CHECK(err);
  return (graph_exec_t)e;
}
void graph_launch(graph_exec_t g, stream_t s) { HIP_CHECK(hipGraphLaunch((hipGraphExec_t)g, hs(s))); }
void graph_destroy(graph_exec_t g) {
  if (g) (void)hipGraphExecDestroy((hipGraphExec_t)g);
}

// a workgroup may ask for up to the CU's whole 160 KiB of LDS; above the 64 KiB default the
// kernel needs the attribute raised once
template <typename K>
static void ensure_lds(K kernel, size_t lds_bytes) {
  if (lds_bytes <= 64 * 1024) return;
  // once per (kernel, size): the attribute call costs a few microseconds per launch otherwise
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, size_t> done;
  int dev = 0;
  HIP_CHECK(hipGetDevice(&dev));
  const void* fn = reinterpret_cast<const void*>(kernel);
  const auto key = std::make_pair(fn, dev);
  std::lock_guard<std::mutex> lk(mu);
  auto it = done.find(key);
  if (it != done.end() && it->second >= lds_bytes) return;
  HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  done[key] = lds_bytes;
}

#ifdef MVN_PROBE
static void probe_setup() {
  static const bool once = [] {
    const char* a = std::getenv("MVN_PROBE_WRAP_ST");
    const char* b = std::getenv("MVN_PROBE_WRAP_ROWS");
    int va = a ? std::atoi(a) : 0, vb = b ? std::atoi(b) : 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_probe_wrap_st), &va, sizeof(int));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_probe_wrap_rows), &vb, sizeof(int));
    return true;
  }();
  (void)once;
}
#else
static void probe_setup() {}
#endif

static void check_launch(long nblocks, int nthreads, size_t lds_bytes) {
  probe_setup();
  if (nblocks < 1 || nblocks > 0x7fffffffL) throw std::invalid_argument("mvn: grid size out of range");
  if (nthreads < 64 || nthreads > 1024 || nthreads % 64) throw std::invalid_argument("mvn: bad block size");
  if (lds_bytes > 160 * 1024) throw std::invalid_argument("mvn: LDS request exceeds 160 KiB");
}

template <typename K, typename P>
static void launch_pass(K kernel, const P& p, long nblocks, int nthreads, size_t lds_bytes, stream_t s) {
  ensure_lds(kernel, lds_bytes);
  hipLaunchKernelGGL(kernel, dim3((unsigned)nblocks), dim3(nthreads), lds_bytes, hs(s), p);
  HIP_CHECK(hipGetLastError());
}

// fixed last-axis kernels: one workgroup per tile, or (long rows, FxRowsCfg<H>::WALK) as many
// workgroups as the device holds at once, each walking over the tiles of the launch
template <bool WALK, typename K>
static void launch_rows_fixed(K kernel, const RowsParams& p, long nblocks, int nthreads, size_t lds_bytes,
                              stream_t s);

// Fixed strided kernels walk over several tiles per workgroup (the next tile's loads overlap the
// current tile's LDS stages): the grid is what the device holds at once, not one block per tile.
// MVN_PERSIST=0 launches one workgroup per tile (same kernel, no overlap across tiles);
// MVN_PERSIST=k (k > 1) launches k times the resident number.  The LDS-staged fused pass is
// launched one workgroup per tile: measured at 512^3, 0.333 ms against 0.380 ms walking (the
// walking workgroups of a CU stay in step, all loading or all computing at once).
static int current_device() {
  int dev = 0;
  HIP_CHECK(hipGetDevice(&dev));
  return dev;
}

// CUs of the current device (asked once per device)
static int device_cu_count() {
  static std::mutex mu;
  static std::map<int, int> cache;
  const int dev = current_device();
  std::lock_guard<std::mutex> lk(mu);
  auto it = cache.find(dev);
  if (it != cache.end()) return it->second;
  hipDeviceProp_t prop;
  HIP_CHECK(hipGetDeviceProperties(&prop, dev));
  const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 1;
  cache[dev] = cus;
  return cus;
}

// resident workgroups per CU of a (device, kernel, block size, LDS) tuple: asked once, not per launch
static int resident_per_cu(const void* kernel, int nthreads, size_t lds_bytes) {
  static std::mutex mu;
  static std::map<std::tuple<int, const void*, int, size_t>, int> cache;
  const int dev = current_device();
  std::lock_guard<std::mutex> lk(mu);
  const auto key = std::make_tuple(dev, kernel, nthreads, lds_bytes);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  int per_cu = 0;
  HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, nthreads, lds_bytes));
  if (per_cu < 1) per_cu = 1;
  cache[key] = per_cu;
  return per_cu;
}

template <typename K>
static void launch_walking(K kernel, StridedParams p, long nblocks, int nthreads, size_t lds_bytes,
                           stream_t s, bool walk, const StridedParams* rider = nullptr, size_t rider_lds = 0) {
  if (rider && rider_lds > lds_bytes) lds_bytes = rider_lds;
  ensure_lds(kernel, lds_bytes);
  static const int mode = [] {
    const char* e = std::getenv("MVN_PERSIST");
    return e ? std::atoi(e) : 1;
  }();
  long grid = nblocks;
  if (mode > 0 && walk) {
    const long resident = (long)resident_per_cu(reinterpret_cast<const void*>(kernel), nthreads, lds_bytes) *
                          device_cu_count() * mode;
    if (grid > resident) grid = resident;
  }
  p.nblocks = nblocks;
  const long lines = rider ? rider->tiles_per_outer : 0;  // one line of the Nyquist plane per workgroup
  if (grid + lines > 0x7fffffffL) throw std::invalid_argument("mvn: grid size out of range");
  hipLaunchKernelGGL(kernel, dim3((unsigned)(grid + lines)), dim3(nthreads), lds_bytes, hs(s), p, rider ? *rider : p,
                     (unsigned)grid);
  HIP_CHECK(hipGetLastError());
}

// wave-row kernels (d2 = 512): workgroups sweep over the row pairs, the grid is what the device
// holds at once.  MVN_WAVE_ROWS_MASK selects the passes that use them (1 plain r2c, 2 plain c2r,
// 4 fused divide, 8 fused update / store, 16 c2r with the DELTA epilogue of the sharded step);
// MVN_NO_WAVE_ROWS=1 = mask 0.  Default 28 (round 3: the DELTA form 0.498 vs 0.512 ms tiled, 16 workgroups
// per slot; 8 / 32 per slot 0.507 / 0.504 -- a pass with three read streams is bound by the read path,
// profiles/r03_mem_counters.md).  Bits 4 and 8: measured at
// 512^3 on MI355X (tools/ab_env.sh, same box) the fused divide gains 12 % over the tiled kernel
// (0.376 -> 0.330 ms) and, with the twiddle tables transposed in the LDS, the fused update 2 %
// (0.485 -> 0.475 ms); the plain r2c / c2r passes, which the tiled kernels already run at the
// streaming ceiling, lose 3-8 % and stay tiled.
static bool wave_rows_enabled(const RowsParams& p, int kind_bit) {
  static const bool off = [] {
    const char* e = std::getenv("MVN_NO_WAVE_ROWS");
    return e && *e && std::strcmp(e, "0") != 0;
  }();
  static const int mask = [] {
    const char* e = std::getenv("MVN_WAVE_ROWS_MASK");
    return e && *e ? std::atoi(e) : 28;
  }();
  return !off && (mask & kind_bit) && p.fixed && !p.lines && p.h == WrCfg::H && p.C == WrCfg::H;
}

// `mult_default` workgroups per resident slot: short-lived workgroups that the dispatcher keeps
// feeding in address order stream better than resident ones that walk (tools/skeleton_probe.hip: a
// bare 2-reads-1-write skeleton reaches 5.27 TB/s with resident walkers, 5.50 with 64 workgroups per
// slot, 5.84 one-shot); against that the tables are built once per workgroup.  Measured at 512^3
// (tools/ab_bench.sh): fused divide 0.336 -> 0.312 ms at 16 per slot, fused update 0.535 -> 0.498 ms
// at 32; MVN_WR_GRID_MULT overrides both.
template <typename K>
static void launch_wave_rows(K kernel, const RowsParams& p, stream_t s, long mult_default = 16) {
#if defined(MVN_EXPERIMENTS) && defined(MVN_WR_LDS_PAD_KB)  // variant builds only: extra LDS = fewer resident workgroups
  const size_t pad = (size_t)MVN_WR_LDS_PAD_KB * 1024;
#else
  const size_t pad = 0;
#endif
  const size_t lds = sizeof(cfloat) * (size_t)WrCfg::lds_cfloats + pad;
  const long pairs = (p.rows + 1) / 2;
  long grid = (pairs + WrCfg::WAVES - 1) / WrCfg::WAVES;
  static const long mult_env = [] {
    const char* e = std::getenv("MVN_WR_GRID_MULT");
    return (long)(e && *e ? std::atoi(e) : 0);
  }();
  const long mult = mult_env > 0 ? mult_env : mult_default;
  const long resident = (long)resident_per_cu(reinterpret_cast<const void*>(kernel), WrCfg::NT, lds) *
                        device_cu_count() * mult;
  if (grid > resident) grid = resident;
  hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(WrCfg::NT), lds, hs(s), p);
  HIP_CHECK(hipGetLastError());
}

template <bool WALK, typename K>
static void launch_rows_fixed(K kernel, const RowsParams& p, long nblocks, int nthreads, size_t lds_bytes,
                              stream_t s) {
  if (WALK) {
    ensure_lds(kernel, lds_bytes);
    // eight workgroups per resident slot (see launch_wave_rows): measured fused divide / update
    // -5 / -3 % at 576^3, -7 / -7 % at 320 x 1920 x 1920 against resident walkers
    static const long mult = [] {
      const char* e = std::getenv("MVN_ROWS_GRID_MULT");
      return (long)(e && *e && std::atoi(e) > 0 ? std::atoi(e) : 8);
    }();
    const long resident = (long)resident_per_cu(reinterpret_cast<const void*>(kernel), nthreads, lds_bytes) *
                          device_cu_count() * mult;
    if (nblocks > resident) nblocks = resident;
  }
  launch_pass(kernel, p, nblocks, nthreads, lds_bytes, s);
}

#define MVN_DISPATCH_T(T_, KERNEL_EXPR)                                        \
  switch (T_) {                                                                \
    case 16: { constexpr int TT = 16; launch_pass(KERNEL_EXPR, p, nblocks, nthreads, lds_bytes, s); } break; \
    case 8: { constexpr int TT = 8; launch_pass(KERNEL_EXPR, p, nblocks, nthreads, lds_bytes, s); } break;   \
    case 4: { constexpr int TT = 4; launch_pass(KERNEL_EXPR, p, nblocks, nthreads, lds_bytes, s); } break;   \
    case 2: { constexpr int TT = 2; launch_pass(KERNEL_EXPR, p, nblocks, nthreads, lds_bytes, s); } break;   \
    case 1: { constexpr int TT = 1; launch_pass(KERNEL_EXPR, p, nblocks, nthreads, lds_bytes, s); } break;   \
    default: throw std::invalid_argument("mvn: unsupported tile width");       \
  }

static void check_aligned16(const void* p, const char* what) {
  if (((size_t)p) & 15) throw std::invalid_argument(std::string("mvn: fixed kernels need 16-byte aligned ") + what);
}

// line-layout forms of the fixed last-axis kernels (H = 256: d2 = 512)
static void check_lines(const RowsParams& p, long nblocks) {
  constexpr int H = 256;
  if (!fx_rows_lines_ok<H>() || !p.fixed || p.h != H || p.C != H || !p.nyq_packed || p.lines_d1 < 1 ||
      p.lines_d1 % FxRowsCfg<H>::T || p.row_base % FxRowsCfg<H>::T || p.rows != nblocks * FxRowsCfg<H>::T)
    throw std::invalid_argument("mvn: line-layout last-axis pass outside its range");
}

void launch_rows_r2c(const RowsParams& p, bool even, long nblocks, int nthreads, size_t lds_bytes,
                     stream_t s) {
  check_launch(nblocks, nthreads, lds_bytes);
  if (p.lines) {
    check_lines(p, nblocks);
    return launch_pass(kx_rows_r2c<256, true>, p, nblocks, nthreads, lds_bytes, s);
  }
  if (p.fixed) {
    check_aligned16(p.in_real, "input");
    check_aligned16(p.out_cplx, "output");
    if (wave_rows_enabled(p, 1)) return launch_wave_rows(kw_rows<MVN_WR_R2C, MVN_EPI_STORE>, p, s);
    switch (p.h) {
#define X(H) case H: launch_rows_fixed<FxRowsCfg<H>::WALK>(kx_rows_r2c<H>, p, nblocks, nthreads, lds_bytes, s); return;
      MVN_FIXED_ROWS_LENGTHS(X)
#undef X
      default: throw std::invalid_argument("mvn: no fixed rows kernel for this length");
    }
  }
  if (even) {
    MVN_DISPATCH_T(p.T, (k_rows_r2c<true, TT>));
  } else {
    MVN_DISPATCH_T(p.T, (k_rows_r2c<false, TT>));
  }
}

void launch_rows_c2r(const RowsParams& p, bool even, long nblocks, int nthreads, size_t lds_bytes,
                     stream_t s) {
  check_launch(nblocks, nthreads, lds_bytes);
  if (p.lines) {
    check_lines(p, nblocks);
    switch (p.epi.mode) {
      case MVN_EPI_DIVIDE: return launch_pass(kx_rows_c2r<256, MVN_EPI_DIVIDE, true>, p, nblocks, nthreads, lds_bytes, s);
      case MVN_EPI_UPDATE: return launch_pass(kx_rows_c2r<256, MVN_EPI_UPDATE, true>, p, nblocks, nthreads, lds_bytes, s);
      case MVN_EPI_DELTA: return launch_pass(kx_rows_c2r<256, MVN_EPI_DELTA, true>, p, nblocks, nthreads, lds_bytes, s);
      default: return launch_pass(kx_rows_c2r<256, MVN_EPI_STORE, true>, p, nblocks, nthreads, lds_bytes, s);
    }
  }
  if (p.fixed) {
    check_aligned16(p.in_cplx, "input");
    check_aligned16(p.out_real, "output");
    if (wave_rows_enabled(p, p.epi.mode == MVN_EPI_DELTA ? 16 : 2)) {
      switch (p.epi.mode) {
        case MVN_EPI_DIVIDE: return launch_wave_rows(kw_rows<MVN_WR_C2R, MVN_EPI_DIVIDE>, p, s);
        case MVN_EPI_UPDATE: return launch_wave_rows(kw_rows<MVN_WR_C2R, MVN_EPI_UPDATE>, p, s);
        case MVN_EPI_DELTA: return launch_wave_rows(kw_rows<MVN_WR_C2R, MVN_EPI_DELTA>, p, s);
        default: return launch_wave_rows(kw_rows<MVN_WR_C2R, MVN_EPI_STORE>, p, s);
      }
    }
    switch (p.h) {
#define X(H)                                                                                  \
  case H:                                                                                     \
    switch (p.epi.mode) {                                                                     \
      case MVN_EPI_DIVIDE: launch_rows_fixed<FxRowsCfg<H>::WALK>(kx_rows_c2r<H, MVN_EPI_DIVIDE>, p, nblocks, nthreads, lds_bytes, s); break; \
      case MVN_EPI_UPDATE: launch_rows_fixed<FxRowsCfg<H>::WALK>(kx_rows_c2r<H, MVN_EPI_UPDATE>, p, nblocks, nthreads, lds_bytes, s); break; \
      case MVN_EPI_DELTA: launch_rows_fixed<FxRowsCfg<H>::WALK>(kx_rows_c2r<H, MVN_EPI_DELTA>, p, nblocks, nthreads, lds_bytes, s); break;   \
      default: launch_rows_fixed<FxRowsCfg<H>::WALK>(kx_rows_c2r<H, MVN_EPI_STORE>, p, nblocks, nthreads, lds_bytes, s); break;             \
    }                                                                                         \
    return;
      MVN_FIXED_ROWS_LENGTHS(X)
#undef X
      default: throw std::invalid_argument("mvn: no fixed rows kernel for this length");
    }
  }
  if (even) {
    MVN_DISPATCH_T(p.T, (k_rows_c2r<true, TT>));
  } else {
    MVN_DISPATCH_T(p.T, (k_rows_c2r<false, TT>));
  }
}

void launch_rows_c2r_r2c(const RowsParams& p, long nblocks, int nthreads, size_t lds_bytes,
                         stream_t s) {
  check_launch(nblocks, nthreads, lds_bytes);
  if (p.lines) {
    check_lines(p, nblocks);
    switch (p.epi.mode) {
      case MVN_EPI_DIVIDE: return launch_pass(kx_rows_c2r_r2c<256, MVN_EPI_DIVIDE, true>, p, nblocks, nthreads, lds_bytes, s);
      case MVN_EPI_UPDATE: return launch_pass(kx_rows_c2r_r2c<256, MVN_EPI_UPDATE, true>, p, nblocks, nthreads, lds_bytes, s);
      default: return launch_pass(kx_rows_c2r_r2c<256, MVN_EPI_STORE, true>, p, nblocks, nthreads, lds_bytes, s);
    }
  }
  if (!p.fixed) {
    MVN_DISPATCH_T(p.T, (k_rows_c2r_r2c<TT>));
    return;
  }
  check_aligned16(p.in_cplx, "input");
  check_aligned16(p.out_cplx, "output");
  if (wave_rows_enabled(p, p.epi.mode == MVN_EPI_DIVIDE ? 4 : 8)) {
    switch (p.epi.mode) {
      case MVN_EPI_DIVIDE: return launch_wave_rows(kw_rows<MVN_WR_C2R_R2C, MVN_EPI_DIVIDE>, p, s);
      case MVN_EPI_UPDATE: return launch_wave_rows(kw_rows<MVN_WR_C2R_R2C, MVN_EPI_UPDATE>, p, s, 32);
      default: return launch_wave_rows(kw_rows<MVN_WR_C2R_R2C, MVN_EPI_STORE>, p, s);
    }
  }
  switch (p.h) {
#define X(H)                                                                                  \
  case H:                                                                                     \
    switch (p.epi.mode) {                                                                     \
      case MVN_EPI_DIVIDE: launch_rows_fixed<FxRowsCfg<H>::WALK>(kx_rows_c2r_r2c<H, MVN_EPI_DIVIDE>, p, nblocks, nthreads, lds_bytes, s); break; \
      case MVN_EPI_UPDATE: launch_rows_fixed<FxRowsCfg<H>::WALK>(kx_rows_c2r_r2c<H, MVN_EPI_UPDATE>, p, nblocks, nthreads, lds_bytes, s); break; \
      default: launch_rows_fixed<FxRowsCfg<H>::WALK>(kx_rows_c2r_r2c<H, MVN_EPI_STORE>, p, nblocks, nthreads, lds_bytes, s); break;             \
    }                                                                                         \
    return;
    MVN_FIXED_ROWS_LENGTHS(X)
#undef X
    default: throw std::invalid_argument("mvn: no fixed rows kernel for this length");
  }
}

static std::atomic<long> g_split_launches{0};
long split_launch_count() { return g_split_launches.load(); }
static std::atomic<long> g_mid_fused_launches{0};
long mid_fused_launch_count() { return g_mid_fused_launches.load(); }

// long lines: 16-column tiles through the split-window body when the columns divide (the plan's
// geometry is for the 8-column kernel); MVN_NO_SPLIT=1 keeps the 8-column kernel
template <int N>
static bool try_launch_split(int mode, const StridedParams& p, long nblocks, stream_t s, const StridedParams* rider,
                             size_t rider_lds) {
  if constexpr (FxSplitCfg<N>::USE) {
    typedef FxSplitCfg<N> C;
    static const bool off = [] {
      const char* e = std::getenv("MVN_NO_SPLIT");
      return e && *e && std::strcmp(e, "0") != 0;
    }();
    static const int fwd_env = [] {  // MVN_SPLIT_FWD=0/1 overrides the per-length default
      const char* e = std::getenv("MVN_SPLIT_FWD");
      return e && *e ? (std::strcmp(e, "0") != 0 ? 1 : 0) : -1;
    }();
    const bool fwd_ok = fwd_env < 0 ? C::FWD_DEFAULT : fwd_env == 1;
    if (off || mode == MVN_ST_FWD_MUL_INV || (mode == MVN_ST_FWD && !fwd_ok) || p.cstride != 1 ||
        p.ncols % C::T != 0 || p.tiles_per_outer < 1)
      return false;
    StridedParams q = p;
    const long outer = nblocks / p.tiles_per_outer;
    q.tiles_per_outer = p.ncols / C::T;
    q.T = q.TP = C::T;
    const long nb = outer * q.tiles_per_outer;
    const size_t lds = sizeof(cfloat) * (size_t)C::lds_cfloats;
    ++g_split_launches;
    if (mode == MVN_ST_FWD)
      launch_walking(kx_strided_split<N, MVN_ST_FWD>, q, nb, C::NT, lds, s, true, rider, rider_lds);
    else
      launch_walking(kx_strided_split<N, MVN_ST_INV>, q, nb, C::NT, lds, s, true, rider, rider_lds);
    return true;
  } else {
    (void)mode; (void)p; (void)nblocks; (void)s; (void)rider; (void)rider_lds;
    return false;
  }
}

void launch_strided(int mode, const StridedParams& p, long nblocks, int nthreads,
                    size_t lds_bytes, stream_t s, const StridedParams* rider, size_t rider_lds) {
  check_launch(nblocks, nthreads, lds_bytes);
  if (rider && (!p.fixed || mode == MVN_ST_FWD_MUL_INV || rider->T != 1 || rider->fixed || rider_lds > 160 * 1024))
    throw std::invalid_argument("mvn: Nyquist lines ride only in the plain fixed-length strided passes, one per workgroup");
  if (p.fixed) {
    check_aligned16(p.data, "data");
    if (mode == MVN_ST_FWD_MUL_INV) check_aligned16(p.spec, "spectrum");
    switch (p.ax.n) {
#define X(N)                                                                                       \
  case N:                                                                                          \
    if (try_launch_split<N>(mode, p, nblocks, s, rider, rider_lds)) return;                        \
    if (mode == MVN_ST_FWD) launch_walking(kx_strided<N, MVN_ST_FWD>, p, nblocks, FxStridedSel<N, MVN_ST_FWD>::NT, lds_bytes, s, true, rider, rider_lds); \
    else if (mode == MVN_ST_INV) launch_walking(kx_strided<N, MVN_ST_INV>, p, nblocks, FxStridedSel<N, MVN_ST_INV>::NT, lds_bytes, s, true, rider, rider_lds); \
    else launch_walking(kx_strided<N, MVN_ST_FWD_MUL_INV>, p, nblocks, FxStridedSel<N, MVN_ST_FWD_MUL_INV>::NT, lds_bytes, s, !FxStridedSel<N, MVN_ST_FWD_MUL_INV>::ONE_TILE); \
    return;
      MVN_FIXED_STRIDED_LENGTHS(X)
#undef X
      default: throw std::invalid_argument("mvn: no fixed strided kernel for this length");
    }
  }
  if (p.is_nyq) {
    switch (mode) {
      case MVN_ST_FWD: MVN_DISPATCH_T(p.T, (k_strided<MVN_ST_FWD, TT, true>)); break;
      case MVN_ST_INV: MVN_DISPATCH_T(p.T, (k_strided<MVN_ST_INV, TT, true>)); break;
      case MVN_ST_FWD_MUL_INV: MVN_DISPATCH_T(p.T, (k_strided<MVN_ST_FWD_MUL_INV, TT, true>)); break;
      default: throw std::invalid_argument("mvn: unknown strided mode");
    }
    return;
  }
  switch (mode) {
    case MVN_ST_FWD: MVN_DISPATCH_T(p.T, (k_strided<MVN_ST_FWD, TT, false>)); break;
    case MVN_ST_INV: MVN_DISPATCH_T(p.T, (k_strided<MVN_ST_INV, TT, false>)); break;
    case MVN_ST_FWD_MUL_INV: MVN_DISPATCH_T(p.T, (k_strided<MVN_ST_FWD_MUL_INV, TT, false>)); break;
    default: throw std::invalid_argument("mvn: unknown strided mode");
  }
}

void launch_dim0_direct(const Dim0DirectParams& p, stream_t s) {
  if (!mvn_dim0_direct_possible(p.k, p.d0) || p.kd < p.k + 1 || p.h != p.k / 2 || p.plane < 0 || p.plane + p.plane2 < 1 || (p.plane > 0 && p.in == p.out) ||
      p.plane2 < 0 || (p.plane2 > 0 && (!p.in2 || !p.out2 || !p.taps2 || p.in2 == p.out2)))
    throw std::invalid_argument("mvn: direct dim0 convolution called outside its range");
  if (p.plane > 0) check_aligned16(p.in, "input");  // (8-byte accesses; the volumes are 16-byte aligned anyway)
  if (p.plane + p.plane2 >= (1L << 28)) throw std::invalid_argument("mvn: plane too large for the direct dim0 leg");
  const long main_blocks = mvn_dim0_blocks(p).blocks;
  long nblocks = main_blocks;
  size_t lds = 0;
  if (p.packed) {
    if (!p.inv1 || !p.taps2 || p.C < 1 || p.d1 < 1 || (long)p.C * p.d1 != p.plane)
      throw std::invalid_argument("mvn: packed direct dim0 convolution needs the dim1 tables and the Nyquist taps");
    nblocks += mvn_dim0_pairs(p.d1);
    lds = mvn_dim0_dc_lds_bytes(p.d0, p.k);
    if (lds > 64 * 1024) throw std::invalid_argument("mvn: dim0 too long for the packed DC column");
  }
  if (nblocks > 0x7fffffffL) throw std::invalid_argument("mvn: grid size out of range");
#if defined(MVN_EXPERIMENTS) && defined(MVN_D0_LDS_PAD_KB)  // variant builds only: bounds the workgroups per CU through their LDS
  if (lds < (size_t)MVN_D0_LDS_PAD_KB * 1024) lds = (size_t)MVN_D0_LDS_PAD_KB * 1024;
#endif
  switch (mvn_dim0_taps_template(p.k)) {
#define X(K) case K: hipLaunchKernelGGL(kd_dim0<K>, dim3((unsigned)nblocks), dim3(MVN_D0_WG), lds, hs(s), p, (unsigned)main_blocks); break;
    MVN_D0_TAP_COUNTS(X)
#undef X
    default: throw std::invalid_argument("mvn: no direct dim0 kernel for this tap count");
  }
  HIP_CHECK(hipGetLastError());
}

// the fused middle pass: instantiated for odd tap counts up to 31 (33 taps would need 40 window slots: beyond the
// registers of two waves per SIMD)
#define MVN_MF_TAP_COUNTS(X) X(1) X(3) X(5) X(7) X(9) X(11) X(13) X(15) X(17) X(19) X(21) X(23) X(25) X(27) X(29) X(31)
void launch_mid_fused(const MidFusedParams& p, stream_t s) {
  mf_check(p);
  const long nblocks = mf_blocks(p);
  if (nblocks < 1 || nblocks > 0x7fffffffL) throw std::invalid_argument("mvn: grid size out of range");
  const size_t lds = sizeof(cfloat) * (size_t)MF_LDS_CFLOATS;
  if (p.mode == MF_TAPS) {
    launch_pass(kf_mid_taps, p, nblocks, MF_NT, lds, s);
    return;
  }
  ++g_mid_fused_launches;
  switch (mvn_dim0_taps_template(p.k)) {
#define X(K) case K: launch_pass(kf_mid<K>, p, nblocks, MF_NT, lds, s); break;
    MVN_MF_TAP_COUNTS(X)
#undef X
    default: throw std::invalid_argument("mvn: no fused middle pass for this tap count");
  }
}

static unsigned flat_grid(size_t n, int block) {
  size_t g = (n + (size_t)block - 1) / (size_t)block;
  const size_t cap = (size_t)device_cu_count() * 8;  // ~8 blocks per CU of THIS device, grid-stride the rest
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (unsigned)g;
}

void launch_scatter_psf(const float* kernel, int k0, int k1, int k2, float* target, int D0,
                        int D1, int D2, long pitch, float scale, stream_t s) {
  const size_t total = (size_t)k0 * k1 * k2;
  hipLaunchKernelGGL(k_scatter_psf, dim3(flat_grid(total, 256)), dim3(256), 0, hs(s), kernel, k0,
                     k1, k2, target, D0, D1, D2, pitch, scale);
  HIP_CHECK(hipGetLastError());
}

void launch_copy3d(float* dst, long drow, long dplane, const float* src, long srow, long splane,
                   int nx, int ny, int nz, stream_t s) {
  if (nx < 1 || ny < 1 || nz < 1) return;
  const long rows = (long)ny * nz;
  const int block = nx >= 256 ? 256 : (nx >= 128 ? 128 : 64);
  const unsigned grid = (unsigned)(rows < 256L * 32 ? rows : 256L * 32);
  hipLaunchKernelGGL(k_copy3d, dim3(grid), dim3(block), 0, hs(s), dst, drow, dplane, src, srow, splane,
                     nx, ny, nz);
  HIP_CHECK(hipGetLastError());
}

void launch_divide(const float* view, float* inout, size_t n, stream_t s) {
  hipLaunchKernelGGL(k_divide, dim3(flat_grid(n, 256)), dim3(256), 0, hs(s), view, inout, n);
  HIP_CHECK(hipGetLastError());
}

void launch_update(float* psi, const float* integral, const float* weights, size_t n,
                   double lambda, float min_value, stream_t s) {
  const float linv = lambda > 0 ? (float)(1.f / lambda) : 0.f;
  hipLaunchKernelGGL(k_update, dim3(flat_grid(n, 256)), dim3(256), 0, hs(s), psi, integral,
                     weights, n, lambda, linv, min_value);
  HIP_CHECK(hipGetLastError());
}

void launch_update_legacy_tikhonov(float* image, const float* integral, const float* weights,
                                   size_t n, float lambda_f, float min_value, stream_t s) {
  hipLaunchKernelGGL(k_update_legacy_tikhonov, dim3(flat_grid(n, 256)), dim3(256), 0, hs(s), image,
                     integral, weights, n, lambda_f, min_value);
  HIP_CHECK(hipGetLastError());
}

void launch_axpy1(float* psi, const float* delta, size_t n, stream_t s) {
  hipLaunchKernelGGL(k_axpy1, dim3(flat_grid(n, 256)), dim3(256), 0, hs(s), psi, delta, n);
  HIP_CHECK(hipGetLastError());
}

}  // namespace be
}  // namespace mvn
