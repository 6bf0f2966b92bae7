// mid_fused_probe.hip -- stand-alone check and timing of the fused middle pass (csrc/mvn_mid_fused.hpp) on
// synthetic data, before / beside its use in the engine.  Never shipped.
//
//   device:     hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -I libmultiviewnative_amd/csrc \
//                     tools/mid_fused_probe.hip -o /tmp/mid_fused_probe
//   emulation:  g++ -O2 -std=c++17 -DMVN_HOST_EMU -x c++ -I libmultiviewnative_amd/csrc tools/mid_fused_probe.hip \
//                     -o /tmp/mid_fused_probe_emu          (the same bodies lane after lane on the host: index math only)
//   run:        mid_fused_probe d0 H k [seg] [launches]
//
// Check: two columns against a double-precision restatement (DFT along the line, K-tap cyclic convolution along
// dim0 with the taps in NATURAL frequency order, inverse DFT).
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "mvn_mid_fused.hpp"

#if !defined(MVN_HOST_EMU)
#include <hip/hip_runtime.h>
#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(2);                                                                    \
    }                                                                             \
  } while (0)

template <int K>
__global__ void __launch_bounds__(MF_NT) kf_mid(const MidFusedParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  FxCtx<MfRegs<K>, MF_NT> ctx;
  ctx.tid = (int)threadIdx.x;
  mf_body<K>(p, (long)blockIdx.x, (cfloat*)smem, ctx);
}
#endif

typedef std::complex<double> cd;

// frequency held by bin q of a line
static int freq_of_bin(int q) {
  const int j = q >> 6, k = (q >> 3) & 7, a = q & 7;
  return k + 8 * a + 64 * j;
}

template <int K>
static void run(MidFusedParams P, int launches, const std::vector<cfloat>& h_in, const std::vector<cfloat>& h_taps,
                std::vector<cfloat>& h_out) {
  const size_t vol = (size_t)P.d0 * P.H * MF_N1;
  const long blocks = mf_blocks(P);
#if !defined(MVN_HOST_EMU)
  cfloat *d_in, *d_out, *d_taps, *d_tw;
  unsigned* d_poison;
  CK(hipMalloc(&d_in, vol * sizeof(cfloat)));
  CK(hipMalloc(&d_out, vol * sizeof(cfloat)));
  CK(hipMalloc(&d_taps, h_taps.size() * sizeof(cfloat)));
  CK(hipMalloc(&d_tw, MF_N1 * sizeof(cfloat)));
  CK(hipMalloc(&d_poison, 64));
  CK(hipMemset(d_poison, 0, 64));
  CK(hipMemset(d_out, 0, vol * sizeof(cfloat)));
  CK(hipMemcpy(d_in, h_in.data(), vol * sizeof(cfloat), hipMemcpyHostToDevice));
  CK(hipMemcpy(d_taps, h_taps.data(), h_taps.size() * sizeof(cfloat), hipMemcpyHostToDevice));
  CK(hipMemcpy(d_tw, P.tw, MF_N1 * sizeof(cfloat), hipMemcpyHostToDevice));
  P.in = d_in;
  P.out = d_out;
  P.taps = d_taps;
  P.tw = d_tw;
  P.poison = d_poison;
  P.poison_epoch = 7;
#ifdef MF_STAMPS
  unsigned long long* d_stamps;
  CK(hipMalloc(&d_stamps, 8 * 32 * sizeof(unsigned long long)));
  CK(hipMemset(d_stamps, 0, 8 * 32 * sizeof(unsigned long long)));
  P.poison_peers = reinterpret_cast<unsigned* const*>(d_stamps);  // (n_peers stays 0: never read as peers)
#endif
  const size_t lds = MF_LDS_CFLOATS * sizeof(cfloat);
  CK(hipFuncSetAttribute((const void*)kf_mid<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipFuncAttributes fa;
  CK(hipFuncGetAttributes(&fa, (const void*)kf_mid<K>));
  printf("kf_mid<%d>: %d VGPRs (+%d AGPRs?), %zu B scratch, %zu B dynamic LDS, %ld workgroups\n", K, fa.numRegs, 0,
         (size_t)fa.localSizeBytes, lds, blocks);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kf_mid<K>, dim3((unsigned)blocks), dim3(MF_NT), lds, 0, P);
  CK(hipDeviceSynchronize());
  float best = 1e9f, sum = 0.f;
  for (int it = 0; it < launches; ++it) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kf_mid<K>, dim3((unsigned)blocks), dim3(MF_NT), lds, 0, P);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
    sum += ms;
  }
  const double bytes = 2.0 * vol * sizeof(cfloat) + (double)h_taps.size() * sizeof(cfloat);
  printf("d0 %d H %d k %d seg %d: best %.4f ms, mean %.4f ms, %.2f TB/s on 2 volumes + taps (at best)\n", P.d0, P.H, P.k,
         P.seg, best, sum / launches, bytes / best * 1e-9);
  CK(hipMemcpy(h_out.data(), d_out, vol * sizeof(cfloat), hipMemcpyDeviceToHost));
  unsigned pw = 0;
  CK(hipMemcpy(&pw, d_poison, 4, hipMemcpyDeviceToHost));
  printf("poison word %u\n", pw);
#ifdef MF_STAMPS
  {
    unsigned long long st[8][32];
    CK(hipMemcpy(st, d_stamps, sizeof(st), hipMemcpyDeviceToHost));
    static const char* names[12] = {"batch start", "A requests (inv2 reads, fwd0 + stores, loads)", "filter line 0", "A inv2 arithmetic + writes",
                                    "B requests", "filter lines 1 2", "B arithmetic + writes", "C requests", "filter lines 3 4",
                                    "C arithmetic + stores / writes", "filter lines 5 6 7", "barrier"};
    printf("shader-clock ticks per step of batch 40 of workgroup 100 (per wave; the step ends at the stamp):\n");
    printf("  %-48s", "wait for the last batch's loads and stores");
    for (int w = 0; w < 8; ++w) printf(" %6lld", (long long)(st[w][12] - st[w][0]));
    printf("\n");
    for (int n = 1; n < 12; ++n) {
      printf("  %-48s", names[n]);
      for (int w = 0; w < 8; ++w) printf(" %6lld", (long long)(st[w][n] - st[w][n == 1 ? 12 : n - 1]));
      printf("\n");
    }
    printf("  %-48s", "whole batch");
    for (int w = 0; w < 8; ++w) printf(" %6lld", (long long)(st[w][11] - st[w][0]));
    printf("\n  start of the batch relative to wave 0:              ");
    for (int w = 0; w < 8; ++w) printf(" %6lld", (long long)(st[w][0] - st[0][0]));
    printf("\n");
  }
#endif
#else
  P.in = h_in.data();
  P.out = h_out.data();
  P.taps = h_taps.data();
  std::vector<cfloat> lds(MF_LDS_CFLOATS);
  static FxCtx<MfRegs<K>, MF_NT> ctx;
  for (long b = 0; b < blocks; ++b) mf_body<K>(P, b, lds.data(), ctx);
  (void)launches;
#endif
}

int main(int argc, char** argv) {
  const int d0 = argc > 1 ? atoi(argv[1]) : 512, H = argc > 2 ? atoi(argv[2]) : 256, k = argc > 3 ? atoi(argv[3]) : 31;
  const int seg = argc > 4 ? atoi(argv[4]) : 0, launches = argc > 5 ? atoi(argv[5]) : 20;
  const int packed = argc > 6 ? atoi(argv[6]) : 1;
  const size_t vol = (size_t)d0 * H * MF_N1;
  std::vector<cfloat> in(vol), out(vol), taps((size_t)k * H * MF_N1), tw(MF_N1);
  std::vector<cd> taps_nat((size_t)k * H * MF_N1);
  unsigned s = 12345u;
  auto rnd = [&]() {
    s = s * 1664525u + 1013904223u;
    return (float)((s >> 8) & 0xffff) / 65536.f - 0.5f;
  };
  for (size_t i = 0; i < vol; ++i) in[i] = cmake(rnd(), rnd());
  for (size_t i = 0; i < taps_nat.size(); ++i) taps_nat[i] = cd(rnd() * 0.01, rnd() * 0.01);
  // packed: column 0 of the taps is T0 + i TH, the line transforms of two REAL tap lines per plane
  std::vector<cd> T0((size_t)k * MF_N1), TH((size_t)k * MF_N1);
  if (packed) {
    std::vector<double> t0(MF_N1), th(MF_N1);
    for (int j = 0; j < k; ++j) {
      for (int y = 0; y < MF_N1; ++y) {
        t0[y] = rnd() * 0.001;
        th[y] = rnd() * 0.001;
      }
      for (int f = 0; f < MF_N1; ++f) {
        cd a0 = 0, ah = 0;
        for (int y = 0; y < MF_N1; ++y) {
          const cd w = std::polar(1.0, -2.0 * M_PI * ((f * y) % MF_N1) / MF_N1);
          a0 += t0[y] * w;
          ah += th[y] * w;
        }
        T0[(size_t)j * MF_N1 + f] = a0;
        TH[(size_t)j * MF_N1 + f] = ah;
        taps_nat[((size_t)j * H + 0) * MF_N1 + f] = a0 + cd(0, 1) * ah;
      }
    }
  }
  for (int j = 0; j < k; ++j)
    for (int c = 0; c < H; ++c)
      for (int q = 0; q < MF_N1; ++q) {
        const cd t = taps_nat[((size_t)j * H + c) * MF_N1 + freq_of_bin(q)];
        taps[((size_t)((j - k / 2 + k) % k) * H + c) * MF_N1 + q] = cmake((float)t.real(), (float)t.imag());
      }
  for (int j = 0; j < MF_N1; ++j) tw[j] = cmake((float)cos(-2.0 * M_PI * j / MF_N1), (float)sin(-2.0 * M_PI * j / MF_N1));
  MidFusedParams P;
  memset(&P, 0, sizeof(P));
  P.tw = tw.data();
  P.d0 = d0;
  P.H = H;
  P.k = k;
  P.h = k / 2;
  P.kd = k;
  P.seg = seg;
  P.mode = MF_CONV;
  P.packed = packed;
  switch (k | 1) {
    case 31: run<31>(P, launches, in, taps, out); break;
    case 15: run<15>(P, launches, in, taps, out); break;
    case 5: run<5>(P, launches, in, taps, out); break;
    default: fprintf(stderr, "k | 1 must be 5, 15 or 31 here\n"); return 2;
  }
  if (getenv("MF_NOCHECK")) return 0;  // timing runs
  // check two columns
  double worst = 0.;
  const int cols[2] = {0, H - 1};
  std::vector<cd> W(MF_N1);
  for (int j = 0; j < MF_N1; ++j) W[j] = std::polar(1.0, -2.0 * M_PI * j / MF_N1);
  for (int ci = 0; ci < (H > 1 ? 2 : 1); ++ci) {
    const int c = cols[ci];
    // ref(part, taps): inverse DFT of the K-tap convolution along dim0 of the DFT of the column's lines
    auto ref = [&](int part, const cd* tp, size_t tstride) {
      std::vector<cd> X((size_t)d0 * MF_N1), Y((size_t)d0 * MF_N1), R((size_t)d0 * MF_N1);
      for (int z = 0; z < d0; ++z)
        for (int f = 0; f < MF_N1; ++f) {
          cd acc = 0;
          for (int y = 0; y < MF_N1; ++y) {
            const cfloat v = in[((size_t)z * H + c) * MF_N1 + y];
            const cd x = part == 0 ? cd(v.x, v.y) : (part == 1 ? cd(v.x, 0) : cd(v.y, 0));
            acc += x * W[(f * y) % MF_N1];
          }
          X[(size_t)z * MF_N1 + f] = acc;
        }
      for (int z = 0; z < d0; ++z)
        for (int f = 0; f < MF_N1; ++f) {
          cd acc = 0;
          for (int j = 0; j < k; ++j) {
            int zi = ((z + k / 2 - j) % d0 + d0) % d0;
            acc += tp[(size_t)j * tstride + f] * X[(size_t)zi * MF_N1 + f];
          }
          Y[(size_t)z * MF_N1 + f] = acc;
        }
      for (int z = 0; z < d0; ++z)
        for (int y = 0; y < MF_N1; ++y) {
          cd acc = 0;
          for (int f = 0; f < MF_N1; ++f) acc += Y[(size_t)z * MF_N1 + f] * std::conj(W[(f * y) % MF_N1]);
          R[(size_t)z * MF_N1 + y] = acc;
        }
      return R;
    };
    std::vector<cd> R;
    if (packed && c == 0) {
      const std::vector<cd> r0 = ref(1, T0.data(), MF_N1), rh = ref(2, TH.data(), MF_N1);
      R.resize(r0.size());
      for (size_t i = 0; i < r0.size(); ++i) R[i] = r0[i] + cd(0, 1) * rh[i];
    } else {
      R = ref(0, &taps_nat[(size_t)c * MF_N1], (size_t)H * MF_N1);
    }
    double maxref = 0., maxerr = 0.;
    for (int z = 0; z < d0; ++z)
      for (int y = 0; y < MF_N1; ++y) {
        const cd acc = R[(size_t)z * MF_N1 + y];
        const cfloat g = out[((size_t)z * H + c) * MF_N1 + y];
        maxref = std::max(maxref, std::abs(acc));
        maxerr = std::max(maxerr, std::abs(acc - cd(g.x, g.y)));
      }
    printf("column %d: max |ref| %.4g, max |err| %.4g, relative %.3g\n", c, maxref, maxerr, maxerr / maxref);
    worst = std::max(worst, maxerr / maxref);
  }
  printf(worst < 2e-5 ? "OK\n" : "MISMATCH\n");
  return worst < 2e-5 ? 0 : 1;
}
