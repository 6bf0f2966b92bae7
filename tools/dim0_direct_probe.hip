// dim0_direct_probe.hip -- feasibility probe (round 3): the dim0 leg of a convolution as a DIRECT
// cyclic convolution with the K nonzero planes of the PSF, in the spectral domain of dims 1 and 2,
// instead of forward dim0 FFT x PSF spectrum x inverse dim0 FFT.
//
//   out[z][b] = sum_{j=0}^{K-1} tap[j][b] * in[(z + K/2 - j) mod d0][b]      (b = a bin of the d1 x C plane)
//
// Reads the volume once and K / d0 of a volume of taps (31 / 512 = 6 %) where the fused FFT pass reads
// the volume and a full PSF spectrum: 1.06 read volumes instead of 2, for 31 complex multiply-adds
// per bin (62 v_pk_fma_f32).  A thread owns one bin and walks along dim0 with the K most recent
// input values and its K taps in registers.
//
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/dim0_direct_probe tools/dim0_direct_probe.hip && /tmp/dim0_direct_probe
#include <hip/hip_runtime.h>

#include "../libmultiviewnative_amd/csrc/mvn_dim0_direct.hpp"  // the product's form of the kernel, timed beside the probe's

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                          \
  do {                                                                    \
    hipError_t e_ = (x);                                                  \
    if (e_ != hipSuccess) {                                               \
      std::fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_)); \
      std::exit(1);                                                       \
    }                                                                     \
  } while (0)


// acc + a * w (complex), two packed instructions
__device__ __forceinline__ cfloat cmac(cfloat acc, cfloat a, cfloat w) {
  cfloat t, r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(t) : "v"(a), "v"(w), "v"(acc));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
  return r;
}

// K taps, PF planes requested ahead of their first use.  Window of KW = K + PF physical slots, the loop
// over z unrolled KW times so that every slot index is a compile-time constant:
//   x_j (j = -PF .. K-1) = in[z + h - j] sits in slot (j + PF - u) mod KW at unrolled step u.
// one unrolled step (u a template constant: every window slot index is static)
template <int K, int PF, int NACC, int U>
__device__ __forceinline__ void dim0_step(cfloat (&w)[K + PF], const cfloat (&tap)[K], const cfloat* __restrict__ in,
                                          cfloat* __restrict__ out, long plane, long b, int d0, int zz, int z1, int& znew) {
  constexpr int KW = K + PF;
  const int z = zz + U;
  if (z >= z1) return;
  // NACC independent partial sums: one accumulator is a chain of 2 K dependent packed instructions
  cfloat part[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a) part[a] = cfloat{0.f, 0.f};
#pragma unroll
  for (int j = 0; j < K; ++j) part[j % NACC] = cmac(part[j % NACC], w[(j + PF - U + KW) % KW], tap[j]);
  cfloat acc = part[0];
#pragma unroll
  for (int a = 1; a < NACC; ++a) acc += part[a];
  out[(long)z * plane + b] = acc;
  // the oldest value x_{K-1} leaves, in[z + h + PF + 1] takes its slot (it is x_{-PF} of step u + 1)
  w[(K - 1 + PF - U + KW) % KW] = in[(long)znew * plane + b];
  znew = znew + 1 == d0 ? 0 : znew + 1;
  if constexpr (U + 1 < KW) dim0_step<K, PF, NACC, U + 1>(w, tap, in, out, plane, b, d0, zz, z1, znew);
}

template <int K, int PF, int NACC>
__global__ void __launch_bounds__(256) k_dim0_direct(const cfloat* __restrict__ in, cfloat* __restrict__ out,
                                                     const cfloat* __restrict__ taps, int d0, long plane, int zseg) {
  constexpr int KW = K + PF, h = K / 2;
  const long b = (long)blockIdx.x * 256 + threadIdx.x;
  if (b >= plane) return;
  const int z0 = blockIdx.y * zseg;
  const int z1 = z0 + zseg < d0 ? z0 + zseg : d0;
  cfloat tap[K];
#pragma unroll
  for (int j = 0; j < K; ++j) tap[j] = taps[(long)j * plane + b];
  cfloat w[KW];
  // at u = 0: slot s holds x_{s - PF} = in[z0 + h - (s - PF)]
#pragma unroll
  for (int s = 0; s < KW; ++s) {
    int z = z0 + h + PF - s;
    z = z < 0 ? z + d0 : (z >= d0 ? z - d0 : z);
    w[s] = in[(long)z * plane + b];
  }
  int znew = z0 + h + PF + 1;  // plane requested next
  if (znew >= d0) znew -= d0;
  for (int zz = z0; zz < z1; zz += KW) dim0_step<K, PF, NACC, 0>(w, tap, in, out, plane, b, d0, zz, z1, znew);
}

template <int K>
__global__ void __launch_bounds__(256) kd_dim0(const Dim0DirectParams p) {
  const long b = (long)blockIdx.x * 256 + threadIdx.x;
  if (b < p.plane) mvn_dim0_direct_column<K, MVN_D0_PF>(p, b);
}

// plain 1 read + 1 write stream (what a dim1 pass is to the memory system), to alternate with the direct leg
__global__ void __launch_bounds__(256) k_stream(const float4* __restrict__ a, float4* __restrict__ b, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 v = a[i];
    v.x += 1.f;
    b[i] = v;
  }
}

template <int K, int PF, int NACC>
static void run(const cfloat* in, cfloat* out, const cfloat* taps, int d0, long plane, int segs, const std::vector<float>& hin,
                const std::vector<float>& htaps, bool check) {
  const int zseg = (d0 + segs - 1) / segs;
  dim3 grid((unsigned)((plane + 255) / 256), (unsigned)segs);
  auto launch = [&] { hipLaunchKernelGGL((k_dim0_direct<K, PF, NACC>), grid, dim3(256), 0, 0, in, out, taps, d0, plane, zseg); };
  launch();
  CHECK(hipDeviceSynchronize());
  double maxrel = 0;
  if (check) {
    std::vector<float> col((size_t)d0 * 2);
    for (long b : {0L, 1L, 255L, plane / 2 + 77, plane - 1}) {
      double ref_max = 0, err = 0;
      for (int z = 0; z < d0; ++z)
        CHECK(hipMemcpy(&col[2 * z], out + (long)z * plane + b, 8, hipMemcpyDeviceToHost));
      for (int z = 0; z < d0; ++z) {
        double re = 0, im = 0;
        for (int j = 0; j < K; ++j) {
          const int zi = ((z + K / 2 - j) % d0 + d0) % d0;
          const double ar = hin[2 * ((size_t)zi * plane + b)], ai = hin[2 * ((size_t)zi * plane + b) + 1];
          const double wr = htaps[2 * ((size_t)j * plane + b)], wi = htaps[2 * ((size_t)j * plane + b) + 1];
          re += ar * wr - ai * wi;
          im += ar * wi + ai * wr;
        }
        ref_max = std::fmax(ref_max, std::fmax(std::fabs(re), std::fabs(im)));
        err = std::fmax(err, std::fmax(std::fabs(re - col[2 * z]), std::fabs(im - col[2 * z + 1])));
      }
      maxrel = std::fmax(maxrel, err / ref_max);
    }
  }
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < 10; ++i) launch();
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 10;
  const double vol = (double)d0 * plane * 8;
  std::printf("K = %2d, %d planes ahead, %d partial sums, %d segment(s) along dim0: %.4f ms  (%.0f GB/s over 2 volumes)", K, PF, NACC, segs, ms,
              2 * vol / ms / 1e6);
  if (check) std::printf("   max rel err vs f64 on 5 columns %.2e", maxrel);
  std::printf("\n");
  std::fflush(stdout);
}

int main(int argc, char** argv) {
  const float scale = argc > 1 ? (float)std::atof(argv[1]) : 1.f;  // e.g. 1e-30: products become denormal
  const int d0 = 512;
  const long plane = 512L * 256;
  const size_t n = (size_t)d0 * plane;
  std::vector<float> hin(2 * n), htaps((size_t)2 * 35 * plane);
  unsigned s = 12345;
  auto rnd = [&] {
    s = s * 1664525u + 1013904223u;
    return (float)((s >> 8) & 0xffff) / 65536.f - 0.5f;
  };
  for (auto& v : hin) v = rnd() * scale;
  for (auto& v : htaps) v = rnd() * (scale < 1.f ? 1e-12f : 1.f);
  std::printf("input scale %g\n", scale);
  cfloat *in, *out, *taps;
  CHECK(hipMalloc(&in, n * 8));
  CHECK(hipMalloc(&out, n * 8));
  CHECK(hipMalloc(&taps, htaps.size() * 4));
  CHECK(hipMemcpy(in, hin.data(), n * 8, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(taps, htaps.data(), htaps.size() * 4, hipMemcpyHostToDevice));
  {
    Dim0DirectParams p;
    p.in = in; p.out = out; p.taps = taps; p.d0 = d0; p.k = 31; p.kd = 32; p.h = 15; p.plane = plane;
    for (int rep = 0; rep < 6; ++rep) {
      // alternate the product form and the probe form, every launch timed on its own
      float tp[8], tq[8];
      hipEvent_t e0, e1;
      CHECK(hipEventCreate(&e0));
      CHECK(hipEventCreate(&e1));
      for (int i = 0; i < 8; ++i) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kd_dim0<31>, dim3((unsigned)(plane / 256)), dim3(256), 0, 0, p);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&tp[i], e0, e1));
      }
      for (int i = 0; i < 8; ++i) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_dim0_direct<31, 4, 1>), dim3((unsigned)(plane / 256), 1), dim3(256), 0, 0, in, out, taps, d0, plane, d0);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&tq[i], e0, e1));
      }
      std::printf("product form:");
      for (int i = 0; i < 8; ++i) std::printf(" %.3f", tp[i]);
      std::printf("   probe form:");
      for (int i = 0; i < 8; ++i) std::printf(" %.3f", tq[i]);
      std::printf("\n");
    }
  }
  {
    // the direct leg between memory-bound kernels, as in the RL loop: [stream, stream, direct] x 60, every
    // direct launch timed on its own
    Dim0DirectParams p;
    p.in = in; p.out = out; p.taps = taps; p.d0 = d0; p.k = 31; p.kd = 32; p.h = 15; p.plane = plane; p.stagger = 64;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int nstream = 0; nstream <= 3; ++nstream) {
      float sum = 0, mx = 0, last = 0;
      for (int i = 0; i < 60; ++i) {
        for (int k = 0; k < nstream; ++k)
          hipLaunchKernelGGL(k_stream, dim3(256 * 16), dim3(256), 0, 0, (const float4*)in, (float4*)out, n / 2);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kd_dim0<31>, dim3((unsigned)(plane / 256)), dim3(256), 0, 0, p);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (i >= 20) { sum += ms; mx = ms > mx ? ms : mx; }
        last = ms;
      }
      std::printf("direct leg (K = 31) behind %d streaming kernel(s) of 0.2 ms: mean %.3f ms, max %.3f, last %.3f\n", nstream,
                  sum / 40, mx, last);
    }
  }
  run<31, 4, 1>(in, out, taps, d0, plane, 1, hin, htaps, true);
  run<31, 4, 2>(in, out, taps, d0, plane, 1, hin, htaps, true);
  run<31, 4, 4>(in, out, taps, d0, plane, 1, hin, htaps, true);
  run<31, 4, 4>(in, out, taps, d0, plane, 2, hin, htaps, false);
  run<31, 2, 4>(in, out, taps, d0, plane, 1, hin, htaps, false);
  run<31, 8, 4>(in, out, taps, d0, plane, 1, hin, htaps, false);
  run<31, 6, 3>(in, out, taps, d0, plane, 1, hin, htaps, false);
  run<15, 4, 4>(in, out, taps, d0, plane, 1, hin, htaps, true);
  run<15, 4, 4>(in, out, taps, d0, plane, 2, hin, htaps, false);
  run<15, 8, 2>(in, out, taps, d0, plane, 2, hin, htaps, false);
  run<3, 4, 1>(in, out, taps, d0, plane, 4, hin, htaps, true);
  run<3, 8, 3>(in, out, taps, d0, plane, 2, hin, htaps, false);
  run<3, 12, 3>(in, out, taps, d0, plane, 1, hin, htaps, false);
  return 0;
}
