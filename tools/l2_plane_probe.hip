// l2_plane_probe.hip -- Step A2 of the L2-residency probe (profiles/r03_l2_probe.md).
//
// Memory skeleton of the plane-local part of one convolution at 512^3,
//     axis1_inv  ->  rows (c2r . pointwise . r2c)  ->  axis1_fwd ,
// once as three separate launches (what the engine does today: 7 volumes of HBM traffic with the
// divide epilogue, 9 with the update epilogue) and once as ONE persistent launch in which a team of
// workgroups of the same XCD carries a d0-plane (512 rows x 256 bins = 1 MB) through all three
// stages, handing it from stage to stage through the XCD's L2 (3 / 5 volumes of HBM traffic if the
// plane stays there).  No transforms: every stage is "load, scale, store" with the access pattern of
// the real pass (column tiles of 128-byte row segments for the dim1 passes, whole 2 KB rows for the
// last-axis pass), so the numbers are what the MEMORY SYSTEM does with the two organisations.
//
// Teams are formed from the hardware XCC id (s_getreg HW_REG_XCC_ID), not from blockIdx % 8, so
// "same L2" holds by construction wherever the dispatcher puts a workgroup.  Hand-off inside a team:
// stores, s_waitcnt vmcnt(0), workgroup barrier, one agent-scope atomic add per workgroup on the
// team's counter; consumers poll the counter, then read the plane with L1-bypassing (nt) loads.
// Every spin is bounded: a workgroup that waits too long raises ctl->error and leaves.
//
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/l2_plane_probe tools/l2_plane_probe.hip && /tmp/l2_plane_probe
//   (rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE over `/tmp/l2_plane_probe pmc` for the traffic)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x)                                                          \
  do {                                                                    \
    hipError_t e_ = (x);                                                  \
    if (e_ != hipSuccess) {                                               \
      std::fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_)); \
      std::exit(1);                                                       \
    }                                                                     \
  } while (0)

typedef float v4 __attribute__((ext_vector_type(4)));

constexpr int D0 = 512, D1 = 512, C16 = 128;           // rows of 128 x 16 B = 2 KB (256 complex bins)
constexpr long PLANE16 = (long)D1 * C16;               // 16-byte units per plane (1 MB)
constexpr int NT = 512;                                // threads per workgroup, two workgroups per CU
constexpr size_t LDS_PAD = 72 * 1024;                  // limits residency to 2 workgroups per CU

struct Ctl {
  int arrived;
  int error;
  int pad0[30];
  int xcd_count[8][32];  // [xcc][0] used; one 128-byte line per XCD
  int bar[64][32];       // one counter line per team
  int done[8][3][64][32];  // pipeline form: [xcc][stage A/B/C][plane of the XCD] arrival counters, a line each
};

enum { LD_PLAIN = 0, LD_NT = 1 };

template <int F>
__device__ __forceinline__ v4 ld(const v4* p) {
  if (F == LD_NT) return __builtin_nontemporal_load(p);
  return *p;
}
template <int F>
__device__ __forceinline__ void st(v4* p, v4 v) {
  if (F == LD_NT)
    __builtin_nontemporal_store(v, p);
  else
    *p = v;
}

// ---- the three stages on one plane, for one workgroup of a team of `ts` (rank `r`) ---------------
// column tiles: TC16 16-byte chunks per row segment (8 = 128 B = 16 bins, 4 = 64 B)
template <int TC16, int LDF, int STF>
__device__ __forceinline__ void stage_cols(const v4* src, v4* dst, int r, int ts, float f, int tid) {
  constexpr int ROWS_PER_IT = NT / TC16;        // rows covered by one sweep of the workgroup
  constexpr int IT = D1 / ROWS_PER_IT;          // 16-byte chunks per thread
  const int c = tid % TC16, r0 = tid / TC16;
  for (int t = r; t < C16 / TC16; t += ts) {
    v4 x[IT];
#pragma unroll
    for (int k = 0; k < IT; ++k) x[k] = ld<LDF>(src + (long)(r0 + k * ROWS_PER_IT) * C16 + t * TC16 + c);
#pragma unroll
    for (int k = 0; k < IT; ++k) st<STF>(dst + (long)(r0 + k * ROWS_PER_IT) * C16 + t * TC16 + c, x[k] * f);
  }
}

// rows: NIN extra real operand streams (1 = view: divide form; 2 = psi, weights: update form, which
// also writes psi back), whole 2 KB rows, 4 rows per sweep of the workgroup, U sweeps in flight
template <int NIN, int LDF_PLANE, int LDF_STREAM, int STF_STREAM>
__device__ __forceinline__ void stage_rows(const v4* src, v4* dst, const v4* o0, v4* o1, int r, int ts, int tid) {
  constexpr int U = 4;
  const int rows_per_wg = D1 / ts;              // ts divides 512
  const int first = r * rows_per_wg;
  for (int it = 0; it < rows_per_wg; it += 4 * U) {
    v4 x[U], a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long off = (long)(first + it + 4 * u) * C16 + tid;
      if (it + 4 * u < rows_per_wg) {
        x[u] = ld<LDF_PLANE>(src + off);
        a[u] = ld<LDF_STREAM>(o0 + off);
        if (NIN > 1) b[u] = ld<LDF_STREAM>(o1 + off);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long off = (long)(first + it + 4 * u) * C16 + tid;
      if (it + 4 * u < rows_per_wg) {
        if (NIN > 1) {
          st<LD_PLAIN>(dst + off, x[u] + a[u] * b[u]);
          st<STF_STREAM>(o1 + off, b[u] + 1.0f);
        } else {
          st<LD_PLAIN>(dst + off, x[u] + a[u]);
        }
      }
    }
  }
}

// ---- separate launches (today's organisation) ----------------------------------------------------
template <int TC16>
__global__ void __launch_bounds__(NT) k_cols(v4* S, float f) {
  const long tiles_per_plane = C16 / TC16;
  const long p = blockIdx.x / tiles_per_plane;
  const int t = (int)(blockIdx.x % tiles_per_plane);
  stage_cols<TC16, LD_PLAIN, LD_PLAIN>(S + p * PLANE16, S + p * PLANE16, t, 1 << 30, f, threadIdx.x);
}

template <int NIN>
__global__ void __launch_bounds__(NT) k_rows(v4* S, const v4* O0, v4* O1) {
  // 16 rows per workgroup
  const long p = blockIdx.x / 32;
  const int r = (int)(blockIdx.x % 32);
  stage_rows<NIN, LD_PLAIN, LD_PLAIN, LD_PLAIN>(S + p * PLANE16, S + p * PLANE16, O0 + p * PLANE16, O1 + p * PLANE16, r, 32,
                                                threadIdx.x);
}

// ---- fused: teams of TS workgroups of one XCD ------------------------------------------------------
__device__ __forceinline__ int ld_relaxed(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// returns false on time-out
__device__ __forceinline__ bool spin_until(const int* p, int target, Ctl* ctl) {
  for (int i = 0; i < (1 << 20); ++i) {
    if (ld_relaxed(p) >= target) return true;
    if ((i & 1023) == 1023 && ld_relaxed(&ctl->error)) return false;
    __builtin_amdgcn_s_sleep(2);
  }
  __hip_atomic_store(&ctl->error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return false;
}

// every wave's stores have left for the L2 -> workgroup barrier -> one arrival per workgroup ->
// wait for the whole team -> workgroup barrier.  `seq` counts the team's barriers.
__device__ __forceinline__ bool team_barrier(Ctl* ctl, int team, int ts, int& seq, int* s_ok, int tid) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  ++seq;
  if (tid == 0) {
    __hip_atomic_fetch_add(&ctl->bar[team][0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *s_ok = spin_until(&ctl->bar[team][0], seq * ts, ctl) ? 1 : 0;
  }
  __syncthreads();
  return *s_ok != 0;
}

// SCRATCH: the plane travels S -> scratch -> scratch -> S (scratch = 1 MB per team, re-used for every
// plane: its lines are re-dirtied before the L2 has a reason to write them back); otherwise in place.
// ACQ: 0 = nt (L1-bypassing) loads of team-written data; 1 = plain loads behind an agent acquire fence
template <int TS, int TC16, int NIN, bool SCRATCH, int ACQ, int STREAM_F>
__global__ void __launch_bounds__(NT, 4) k_fused(v4* S, const v4* O0, v4* O1, v4* scratch, Ctl* ctl) {
  __shared__ int s_info[8];
  const int tid = threadIdx.x;
  if (tid == 0) {
    const int xcc = (int)(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7);  // HW_REG_XCC_ID[3:0]
    const int li = __hip_atomic_fetch_add(&ctl->xcd_count[xcc][0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&ctl->arrived, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int team = -1, rank = 0, nteams = 0;
    if (spin_until(&ctl->arrived, (int)gridDim.x, ctl)) {
      int before = 0;
      for (int x = 0; x < 8; ++x) {
        const int n = ld_relaxed(&ctl->xcd_count[x][0]) / TS;
        if (x < xcc) before += n;
        nteams += n;
      }
      const int mine = ld_relaxed(&ctl->xcd_count[xcc][0]) / TS;
      if (li < mine * TS) {
        team = before + li / TS;
        rank = li % TS;
      }
    }
    s_info[0] = team;
    s_info[1] = rank;
    s_info[2] = nteams;
  }
  __syncthreads();
  const int team = s_info[0], rank = s_info[1], nteams = s_info[2];
  if (team < 0) return;
  constexpr int PL = ACQ ? LD_PLAIN : LD_NT;
  int seq = 0;
  v4* sc = scratch + (long)team * PLANE16;
  for (int p = team; p < D0; p += nteams) {
    v4* plane = S + (long)p * PLANE16;
    v4* mid = SCRATCH ? sc : plane;
    // A: dim1 inverse skeleton, plane in from HBM
    stage_cols<TC16, STREAM_F, LD_PLAIN>(plane, mid, rank, TS, 2.0f, tid);
    if (!team_barrier(ctl, team, TS, seq, &s_info[4], tid)) return;
    if (ACQ) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    // B: last-axis skeleton: plane rows from the L2, operand rows from HBM
    stage_rows<NIN, PL, STREAM_F, LD_PLAIN>(mid, mid, O0 + (long)p * PLANE16, O1 + (long)p * PLANE16, rank, TS, tid);
    if (!team_barrier(ctl, team, TS, seq, &s_info[4], tid)) return;
    if (ACQ) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    // C: dim1 forward skeleton, plane out to HBM
    stage_cols<TC16, PL, LD_PLAIN>(mid, plane, rank, TS, 0.5f, tid);
    if (SCRATCH) {  // the scratch plane is re-written by the next plane's stage A: everyone must have read it
      if (!team_barrier(ctl, team, TS, seq, &s_info[4], tid)) return;
    }
  }
}

// ---- pipelined: the workgroups of an XCD are SPECIALISED by stage (a quarter run stage A, half stage
// B, a quarter stage C) and the XCD's planes flow through them: stage A of plane k+2 runs beside
// stage B of plane k+1 and stage C of plane k.  Hand-off by per-plane arrival counters; the plane
// lives in a ring of R scratch planes per XCD between A and C, so stage A may run at most R planes
// ahead of stage C (that bounds the L2 footprint: R MB of 4).
__device__ __forceinline__ void signal_done(int* counter, int tid) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool wait_for(const int* counter, int target, Ctl* ctl, int* s_ok, int tid) {
  if (tid == 0) *s_ok = spin_until(counter, target, ctl) ? 1 : 0;
  __syncthreads();
  const bool ok = *s_ok != 0;
  __syncthreads();
  return ok;
}

template <int R, int NIN, int STREAM_F>
__global__ void __launch_bounds__(NT, 4) k_pipe(v4* S, const v4* O0, v4* O1, v4* scratch, Ctl* ctl) {
  __shared__ int s_info[8];
  const int tid = threadIdx.x;
  if (tid == 0) {
    const int xcc = (int)(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7);
    const int li = __hip_atomic_fetch_add(&ctl->xcd_count[xcc][0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&ctl->arrived, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int nx = 0, xr = 0, nloc = 0;
    if (spin_until(&ctl->arrived, (int)gridDim.x, ctl)) {
      for (int x = 0; x < 8; ++x) {
        const int n = ld_relaxed(&ctl->xcd_count[x][0]);
        if (n >= 4) {
          if (x < xcc) ++xr;
          ++nx;
        }
      }
      nloc = ld_relaxed(&ctl->xcd_count[xcc][0]);
    }
    s_info[0] = nloc >= 4 ? xcc : -1;
    s_info[1] = li;
    s_info[2] = nloc;
    s_info[3] = xr;
    s_info[5] = nx;
  }
  __syncthreads();
  const int xcc = s_info[0], li = s_info[1], nloc = s_info[2], xr = s_info[3], nx = s_info[5];
  if (xcc < 0) return;
  const int nA = nloc / 4, nB = nloc / 2, nC = nloc - nA - nB;
  const int role = li < nA ? 0 : (li < nA + nB ? 1 : 2);
  const int j = role == 0 ? li : (role == 1 ? li - nA : li - nA - nB);
  v4* ring = scratch + (long)xcc * R * PLANE16;
  int k = 0;
  for (int p = xr; p < D0; p += nx, ++k) {
    v4* plane = S + (long)p * PLANE16;
    v4* mid = ring + (long)(k % R) * PLANE16;
    if (role == 0) {
      if (k >= R && !wait_for(&ctl->done[xcc][2][k - R][0], nC, ctl, &s_info[4], tid)) return;
      stage_cols<8, STREAM_F, LD_PLAIN>(plane, mid, j, nA, 2.0f, tid);
      signal_done(&ctl->done[xcc][0][k][0], tid);
    } else if (role == 1) {
      if (!wait_for(&ctl->done[xcc][0][k][0], nA, ctl, &s_info[4], tid)) return;
      // rows j * 512 / nB ...: stage_rows takes (rank, team size) with 512 / ts rows per workgroup
      stage_rows<NIN, LD_NT, STREAM_F, LD_PLAIN>(mid, mid, O0 + (long)p * PLANE16, O1 + (long)p * PLANE16, j, nB, tid);
      signal_done(&ctl->done[xcc][1][k][0], tid);
    } else {
      if (!wait_for(&ctl->done[xcc][1][k][0], nB, ctl, &s_info[4], tid)) return;
      stage_cols<8, LD_NT, LD_PLAIN>(mid, plane, j, nC, 0.5f, tid);
      signal_done(&ctl->done[xcc][2][k][0], tid);
    }
  }
}

template <int R, int NIN, int STREAM_F>
static void run_pipe(const char* what, double sep_ms, bool pmc);

// ---- host ------------------------------------------------------------------------------------------
static v4 *S, *O0, *O1, *scratch;
static Ctl* ctls;
static int n_ctl = 0, next_ctl = 0;
static int cus = 256;

static Ctl* fresh_ctl() {
  if (next_ctl >= n_ctl) {
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemset(ctls, 0, sizeof(Ctl) * n_ctl));
    next_ctl = 0;
  }
  return ctls + next_ctl++;
}

template <typename F>
static double time_ms(F launch, int reps) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) launch();
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

static void fill(std::vector<float>& h, float a, float b) {
  for (size_t i = 0; i < h.size(); ++i) h[i] = a + b * (float)(i % 977);
}

// one launch from known contents, every word checked: S' = (2 S + V) / 2 (divide form),
// S' = (2 S + psi w) / 2 and w' = w + 1 (update form)
template <typename F>
static bool verify(const char* what, F launch, int nin) {
  const size_t n = (size_t)D0 * PLANE16 * 4;
  std::vector<float> hs(n), h0(n), h1(n), out(n), out1(n);
  fill(hs, 1.f, 0.25f);
  fill(h0, 3.f, 0.5f);
  fill(h1, 2.f, 0.125f);
  CHECK(hipMemcpy(S, hs.data(), n * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(O0, h0.data(), n * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(O1, h1.data(), n * 4, hipMemcpyHostToDevice));
  launch();
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(out.data(), S, n * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(out1.data(), O1, n * 4, hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (size_t i = 0; i < n; ++i) {
    const float want = nin > 1 ? (2.f * hs[i] + h0[i] * h1[i]) * 0.5f : (2.f * hs[i] + h0[i]) * 0.5f;
    if (out[i] != want) ++bad;
    if (nin > 1 && out1[i] != h1[i] + 1.f) ++bad;
  }
  std::printf("verify %-58s %s (%zu bad words)\n", what, bad ? "FAILED" : "ok", bad);
  std::fflush(stdout);
  return bad == 0;
}

static int read_error(Ctl* c) {
  Ctl h;
  CHECK(hipMemcpy(&h, c, sizeof(Ctl), hipMemcpyDeviceToHost));
  return h.error;
}

template <int TS, int TC16, int NIN, bool SCRATCH, int ACQ, int STREAM_F>
static void run_fused(const char* what, double sep_ms, bool pmc) {
  auto kern = k_fused<TS, TC16, NIN, SCRATCH, ACQ, STREAM_F>;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_PAD));
  int per_cu = 0;
  CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, NT, LDS_PAD));
  if (per_cu < 2) {
    std::printf("%-66s skipped: only %d workgroup(s) per CU resident\n", what, per_cu);
    return;
  }
  const unsigned grid = (unsigned)(2 * cus);
  Ctl* last = nullptr;
  auto launch = [&] {
    last = fresh_ctl();
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), LDS_PAD, 0, S, O0, O1, scratch, last);
  };
  if (!pmc) {
    if (!verify(what, launch, NIN) || read_error(last)) {
      std::printf("%-66s hand-off FAILED (error flag %d)\n", what, read_error(last));
      return;
    }
  }
  const double ms = time_ms(launch, pmc ? 3 : 10);
  CHECK(hipDeviceSynchronize());
  std::printf("%-66s %.4f ms   (separate launches %.4f ms: x%.2f)%s\n", what, ms, sep_ms, sep_ms / ms,
              read_error(last) ? "  TIME-OUT FLAG SET" : "");
  std::fflush(stdout);
}

template <int R, int NIN, int STREAM_F>
static void run_pipe(const char* what, double sep_ms, bool pmc) {
  auto kern = k_pipe<R, NIN, STREAM_F>;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_PAD));
  int per_cu = 0;
  CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, NT, LDS_PAD));
  if (per_cu < 2) {
    std::printf("%-66s skipped: only %d workgroup(s) per CU resident\n", what, per_cu);
    return;
  }
  const unsigned grid = (unsigned)(2 * cus);
  Ctl* last = nullptr;
  auto launch = [&] {
    last = fresh_ctl();
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), LDS_PAD, 0, S, O0, O1, scratch, last);
  };
  if (!pmc) {
    if (!verify(what, launch, NIN) || read_error(last)) {
      std::printf("%-66s hand-off FAILED (error flag %d)\n", what, read_error(last));
      return;
    }
  }
  const double ms = time_ms(launch, pmc ? 3 : 10);
  CHECK(hipDeviceSynchronize());
  std::printf("%-66s %.4f ms   (separate launches %.4f ms: x%.2f)%s\n", what, ms, sep_ms, sep_ms / ms,
              read_error(last) ? "  TIME-OUT FLAG SET" : "");
  std::fflush(stdout);
}

int main(int argc, char** argv) {
  const bool pmc = argc > 1 && !std::strcmp(argv[1], "pmc");
  const size_t bytes = (size_t)D0 * PLANE16 * 16;
  CHECK(hipMalloc(&S, bytes));
  CHECK(hipMalloc(&O0, bytes));
  CHECK(hipMalloc(&O1, bytes));
  CHECK(hipMalloc(&scratch, (size_t)64 * PLANE16 * 16));
  n_ctl = 64;
  CHECK(hipMalloc(&ctls, sizeof(Ctl) * n_ctl));
  next_ctl = n_ctl;
  CHECK(hipMemset(S, 0, bytes));
  CHECK(hipMemset(O0, 0, bytes));
  CHECK(hipMemset(O1, 0, bytes));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  cus = prop.multiProcessorCount;
  std::printf("device %s, %d CUs; volume %.0f MB, plane %.0f KB\n", prop.name, cus, bytes / 1e6, PLANE16 * 16 / 1e3);

  const int reps = pmc ? 3 : 10;
  const double vol = (double)bytes;
  double sep[3];
  for (int nin = 1; nin <= 2; ++nin) {
    auto sep_launch = [&] {
      hipLaunchKernelGGL(k_cols<8>, dim3(D0 * 16), dim3(NT), 0, 0, S, 2.0f);
      if (nin == 1)
        hipLaunchKernelGGL(k_rows<1>, dim3(D0 * 32), dim3(NT), 0, 0, S, O0, O1);
      else
        hipLaunchKernelGGL(k_rows<2>, dim3(D0 * 32), dim3(NT), 0, 0, S, O0, O1);
      hipLaunchKernelGGL(k_cols<8>, dim3(D0 * 16), dim3(NT), 0, 0, S, 0.5f);
    };
    if (!pmc) verify(nin == 1 ? "separate launches, divide form" : "separate launches, update form", sep_launch, nin);
    sep[nin] = time_ms(sep_launch, reps);
    const double vols = nin == 1 ? 7 : 9;
    std::printf("separate launches, %s form: %.4f ms = %.0f GB/s over %g volumes\n", nin == 1 ? "divide" : "update", sep[nin],
                vols * vol / sep[nin] / 1e6, vols);
  }
  {
    const double a = time_ms([&] { hipLaunchKernelGGL(k_cols<8>, dim3(D0 * 16), dim3(NT), 0, 0, S, 1.0f); }, reps);
    const double b = time_ms([&] { hipLaunchKernelGGL(k_rows<1>, dim3(D0 * 32), dim3(NT), 0, 0, S, O0, O1); }, reps);
    const double c = time_ms([&] { hipLaunchKernelGGL(k_rows<2>, dim3(D0 * 32), dim3(NT), 0, 0, S, O0, O1); }, reps);
    std::printf("  alone: column-tile pass %.4f ms (%.0f GB/s), rows divide form %.4f ms (%.0f GB/s), rows update form %.4f ms (%.0f GB/s)\n",
                a, 2 * vol / a / 1e6, b, 3 * vol / b / 1e6, c, 5 * vol / c / 1e6);
  }
  std::fflush(stdout);

  //        TS  TC16 NIN scratch acq stream-flavour
  run_fused<16, 8, 1, false, 0, LD_PLAIN>("fused div: teams of 16, 128-B tiles, in place, nt plane loads", sep[1], pmc);
  run_fused<16, 8, 1, true, 0, LD_PLAIN>("fused div: teams of 16, 128-B tiles, 1 MB scratch per team", sep[1], pmc);
  run_fused<16, 8, 1, true, 0, LD_NT>("fused div: teams of 16, scratch, nt on the HBM streams too", sep[1], pmc);
  run_fused<16, 8, 1, true, 1, LD_PLAIN>("fused div: teams of 16, scratch, acquire fence + plain loads", sep[1], pmc);
  run_fused<32, 4, 1, false, 0, LD_PLAIN>("fused div: teams of 32, 64-B tiles, in place", sep[1], pmc);
  run_fused<32, 4, 1, true, 0, LD_PLAIN>("fused div: teams of 32, 64-B tiles, scratch", sep[1], pmc);
  run_fused<32, 4, 1, true, 0, LD_NT>("fused div: teams of 32, 64-B tiles, scratch, nt streams", sep[1], pmc);
  run_fused<32, 8, 1, true, 0, LD_PLAIN>("fused div: teams of 32, 128-B tiles (half idle in A/C), scratch", sep[1], pmc);
  run_fused<64, 4, 1, true, 0, LD_PLAIN>("fused div: teams of 64 (one per XCD), 64-B tiles, scratch", sep[1], pmc);
  run_fused<8, 8, 1, true, 0, LD_PLAIN>("fused div: teams of 8, 128-B tiles, scratch", sep[1], pmc);
  run_fused<16, 8, 2, false, 0, LD_PLAIN>("fused upd: teams of 16, 128-B tiles, in place", sep[2], pmc);
  run_fused<16, 8, 2, true, 0, LD_PLAIN>("fused upd: teams of 16, 128-B tiles, scratch", sep[2], pmc);
  run_fused<16, 8, 2, true, 0, LD_NT>("fused upd: teams of 16, scratch, nt streams", sep[2], pmc);
  run_fused<32, 4, 2, true, 0, LD_PLAIN>("fused upd: teams of 32, 64-B tiles, scratch", sep[2], pmc);
  run_fused<32, 4, 2, true, 0, LD_NT>("fused upd: teams of 32, 64-B tiles, scratch, nt streams", sep[2], pmc);
  run_pipe<2, 1, LD_PLAIN>("pipelined div: stage-specialised workgroups, ring of 2 planes per XCD", sep[1], pmc);
  run_pipe<3, 1, LD_PLAIN>("pipelined div: ring of 3 planes", sep[1], pmc);
  run_pipe<4, 1, LD_PLAIN>("pipelined div: ring of 4 planes", sep[1], pmc);
  run_pipe<3, 1, LD_NT>("pipelined div: ring of 3 planes, nt streams", sep[1], pmc);
  run_pipe<6, 1, LD_PLAIN>("pipelined div: ring of 6 planes", sep[1], pmc);
  run_pipe<3, 2, LD_PLAIN>("pipelined upd: ring of 3 planes", sep[2], pmc);
  run_pipe<3, 2, LD_NT>("pipelined upd: ring of 3 planes, nt streams", sep[2], pmc);
  run_pipe<4, 2, LD_PLAIN>("pipelined upd: ring of 4 planes", sep[2], pmc);
  return 0;
}
