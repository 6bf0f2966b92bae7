// skeleton_probe.hip -- what a "2 reads + 1 write in place" streaming pass can reach on this GPU as a
// function of how it is organised: one-shot workgroups vs resident workgroups that walk over rows,
// waves per CU, loads in flight per wave.  No arithmetic worth mentioning: this is the memory
// skeleton of the fused last-axis passes (spectrum row in, operand row in, spectrum row out).
//
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/skeleton_probe tools/skeleton_probe.hip && /tmp/skeleton_probe
//
// Prints GB/s (3 volumes of 512^3 floats per launch) for every variant.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                      \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      std::fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));             \
      std::exit(1);                                                                   \
    }                                                                                 \
  } while (0)

typedef float v4 __attribute__((ext_vector_type(4)));

// one-shot: thread i handles U consecutive-by-grid 16-byte chunks (the torch elementwise shape)
template <int U>
__global__ void __launch_bounds__(256) k_oneshot(v4* a, const v4* b, size_t n4) {
  size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x;
  v4 x[U], y[U];
#pragma unroll
  for (int u = 0; u < U; ++u)
    if (i + (size_t)u * 256 < n4) {
      x[u] = a[i + (size_t)u * 256];
      y[u] = b[i + (size_t)u * 256];
    }
#pragma unroll
  for (int u = 0; u < U; ++u)
    if (i + (size_t)u * 256 < n4) a[i + (size_t)u * 256] = x[u] * y[u];
}

// resident: a wave owns rows of ROWB bytes (64 lanes x 16 B x Q), walks over them with stride
// (waves in the launch), keeps DEPTH rows of both streams in flight ahead of the one it finishes
// CHUNK: a wave owns a contiguous run of rows instead of every nwaves-th row
template <int Q, int DEPTH, bool CHUNK = false>
__global__ void __launch_bounds__(256) k_walk(v4* a, const v4* b, long nrows_all) {
  const int lane = threadIdx.x & 63;
  const long wave_id = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nwaves_all = (long)gridDim.x * 4;
  long wave = wave_id, nwaves = nwaves_all, nrows = nrows_all;
  if (CHUNK) {  // rows [first, first + per) of this wave, walked with stride 1
    const long per = (nrows_all + nwaves_all - 1) / nwaves_all;
    const long first = wave_id * per;
    a += first * Q * 64;
    b += first * Q * 64;
    nrows = nrows_all - first < per ? (nrows_all - first > 0 ? nrows_all - first : 0) : per;
    wave = 0;
    nwaves = 1;
  }
  v4 x[DEPTH + 1][Q], y[DEPTH + 1][Q];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    const long r = wave + d * nwaves;
    if (r < nrows) {
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        x[d][q] = a[(r * Q + q) * 64 + lane];
        y[d][q] = b[(r * Q + q) * 64 + lane];
      }
    }
  }
  for (long r = wave; r < nrows; r += nwaves * (DEPTH + 1)) {
#pragma unroll
    for (int s = 0; s <= DEPTH; ++s) {  // rotating register sets: slot s finishes row r + s nwaves
      const long cur = r + s * nwaves;
      const long nxt = cur + DEPTH * nwaves;
      const int ld = (s + DEPTH) % (DEPTH + 1);
      if (nxt < nrows) {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          x[ld][q] = a[(nxt * Q + q) * 64 + lane];
          y[ld][q] = b[(nxt * Q + q) * 64 + lane];
        }
      }
      if (cur < nrows) {
#pragma unroll
        for (int q = 0; q < Q; ++q) a[(cur * Q + q) * 64 + lane] = x[s][q] * y[s][q];
      }
    }
  }
}

template <typename F>
static double time_ms(F launch, int reps) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main() {
  const size_t n = (size_t)512 * 512 * 512;  // floats per volume
  const size_t n4 = n / 4;
  v4 *a, *b;
  CHECK(hipMalloc(&a, n * 4));
  CHECK(hipMalloc(&b, n * 4));
  CHECK(hipMemset(a, 0, n * 4));
  CHECK(hipMemset(b, 0, n * 4));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double bytes = 3.0 * n * 4;
  auto report = [&](const char* what, double ms) {
    std::printf("%-64s %.4f ms  %6.0f GB/s\n", what, ms, bytes / ms / 1e6);
    std::fflush(stdout);
  };
  report("one-shot, 256 threads, 1 x 16 B per stream and thread", time_ms([&] {
           hipLaunchKernelGGL(k_oneshot<1>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, a, b, n4);
         }, 20));
  report("one-shot, 256 threads, 4 x 16 B per stream and thread", time_ms([&] {
           hipLaunchKernelGGL(k_oneshot<4>, dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), 0, 0, a, b, n4);
         }, 20));
  // resident walkers: rows of 2 KB x 2 (Q = 4 -> 4 KB per wave-row, the wave-row kernels' unit)
  const long nrows4 = (long)(n4 / (64 * 4));
  const long nrows2 = (long)(n4 / (64 * 2));
  for (int wg_per_cu : {2, 4, 6, 8}) {
    // extra dynamic LDS limits the resident workgroups per CU (160 KB / wg_per_cu each)
    const size_t lds = (size_t)(160 * 1024 / wg_per_cu) - 1024;
    const unsigned grid = (unsigned)(cus * wg_per_cu);
    char what[160];
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_walk<4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    std::snprintf(what, sizeof(what), "walking, %2d waves/CU, 4 KB rows, 1 row ahead  (%3d KB in flight per CU)", wg_per_cu * 4,
                  wg_per_cu * 4 * 8);
    report(what, time_ms([&] { hipLaunchKernelGGL((k_walk<4, 1>), dim3(grid), dim3(256), lds, 0, a, b, nrows4); }, 20));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_walk<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    std::snprintf(what, sizeof(what), "walking, %2d waves/CU, 4 KB rows, 2 rows ahead (%3d KB in flight per CU)", wg_per_cu * 4,
                  wg_per_cu * 4 * 16);
    report(what, time_ms([&] { hipLaunchKernelGGL((k_walk<4, 2>), dim3(grid), dim3(256), lds, 0, a, b, nrows4); }, 20));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_walk<2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    std::snprintf(what, sizeof(what), "walking, %2d waves/CU, 2 KB rows, 1 row ahead  (%3d KB in flight per CU)", wg_per_cu * 4,
                  wg_per_cu * 4 * 4);
    report(what, time_ms([&] { hipLaunchKernelGGL((k_walk<2, 1>), dim3(grid), dim3(256), lds, 0, a, b, nrows2); }, 20));
  }
  {
    const size_t lds = (size_t)(160 * 1024 / 4) - 1024;  // 16 waves per CU resident
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_walk<4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int mult : {1, 2, 4, 8, 16, 64}) {
      char what[160];
      std::snprintf(what, sizeof(what), "walking, 16 waves/CU resident, grid = %2d x resident, 4 KB rows, 1 ahead", mult);
      const unsigned grid = (unsigned)(cus * 4 * mult);
      report(what, time_ms([&] { hipLaunchKernelGGL((k_walk<4, 1>), dim3(grid), dim3(256), lds, 0, a, b, nrows4); }, 20));
    }
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_walk<4, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int mult : {1, 4, 16}) {
      char what[160];
      std::snprintf(what, sizeof(what), "chunked (contiguous rows per wave), grid = %2d x resident, 4 KB rows, 1 ahead", mult);
      const unsigned grid = (unsigned)(cus * 4 * mult);
      report(what, time_ms([&] { hipLaunchKernelGGL((k_walk<4, 1, true>), dim3(grid), dim3(256), lds, 0, a, b, nrows4); }, 20));
    }
    // one workgroup (one wave) per 4 KB row, no loop at all
    report("one-shot, one 64-thread workgroup per 4 KB row", time_ms([&] {
             hipLaunchKernelGGL((k_walk<4, 0>), dim3((unsigned)((nrows4 + 3) / 4)), dim3(256), 0, 0, a, b, nrows4);
           }, 20));
  }
  CHECK(hipFree(a));
  CHECK(hipFree(b));
  return 0;
}
