// pk_fma_probe.hip -- issue cost of v_pk_fma_f32 / v_fma_f32 / v_pk_mul_f32 on gfx950, one and two waves per SIMD.
// Never shipped.   hipcc -O3 --offload-arch=gfx950 tools/pk_fma_probe.hip -o build/probe/pk_fma_probe
// Each wave runs REP x 64 instructions of one kind on 16 independent accumulators (4 chains interleaved), timed with
// s_memtime inside the kernel; reported: shader cycles per instruction and wave, and per SIMD (waves share a SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float cfloat __attribute__((ext_vector_type(2)));
#define REP 8192

template <int KIND>
__global__ void __launch_bounds__(1024) k(cfloat* out, unsigned long long* cyc, float seed) {
  cfloat a[16], b, c;
  for (int i = 0; i < 16; ++i) a[i] = cfloat{seed + i, seed - i};
  b = cfloat{1.0001f, 0.9999f};
  c = cfloat{seed * 1e-7f, -seed * 1e-7f};
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < REP; ++r) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (KIND == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
        if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a[i]) : "v"(b), "v"(c));
        if (KIND == 2) {
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
        }
        if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (KIND == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      }
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  cfloat s = a[0];
  for (int i = 1; i < 16; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
static void run(const char* name) {
  cfloat* out;
  unsigned long long* cyc;
  hipMalloc(&out, 256 * 1024 * sizeof(cfloat));
  hipMalloc(&cyc, 256 * 16 * sizeof(unsigned long long));
  for (int threads : {256, 512, 1024}) {  // 1, 2, 4 waves per SIMD (one workgroup per CU)
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.5f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[8];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const double n = (double)REP * 64;
    const int wps = threads / 256;
    printf("%-28s %d wave(s)/SIMD: %6.2f counter ticks per instruction and wave (%5.2f per SIMD), kernel %.3f ms = %.2f ns per instruction and SIMD\n",
           name, wps, h[0] / n, h[0] / n / wps, ms, ms * 1e6 / (n * wps));
  }
}

int main() {
  run<0>("v_pk_fma_f32");
  run<1>("v_pk_fma_f32 op_sel_hi");
  run<2>("v_fma_f32");
  run<3>("v_pk_mul_f32");
  run<4>("v_pk_add_f32");
  return 0;
}
