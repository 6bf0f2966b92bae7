/*
 * oracle/mvn_oracle.c  --  TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's FFTW/CPU multi-view Richardson-Lucy path
 * (psteinb/libmultiviewnative).  It is the parity checker for the HIP product
 * library and the timed "port" CPU baseline of bench.py.  Nothing in the product
 * (libmultiviewnative_amd/, the shipped libmultiviewnative.so) links, imports or
 * calls this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may.
 *
 * Pinning status: the reference itself cannot be built in this image (it needs
 * Boost.MultiArray and FFTW3, both absent).  The FFT is therefore
 * our own float32 mixed-radix Stockham transform standing in for FFTW's
 * fftwf_plan_dft_r2c_3d / c2r_3d (un-normalised, last axis halved; third-party
 * dependency "fftw 3.1 or later", unpinned in the reference: README.md:20).  The
 * oracle is pinned against the reference's own synthetic test fixtures
 * (tests/test_fixtures.hpp, tests/test_plan_store.cpp:83-142,
 * tests/test_fftw_numerical_stability.cpp, tests/test_gpu_kernels_impl.cu
 * constants, bench/synthetic_data.hpp closed form) in tests/test_oracle_*.py and
 * cross-checked against numpy's pocketfft.  End-to-end RL results of the reference
 * are pinned only by TIFF fixtures that are not in the repository.
 *
 * Every function cites the reference file:line whose arithmetic it follows.
 * Plain C99 + OpenMP; build: make -C oracle.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef float imageType; /* inc/multiviewnative.h:4 */

/* inc/multiviewnative.h:15-26 */
typedef struct view_data {
  imageType* image_;
  imageType* kernel1_;
  imageType* kernel2_;
  imageType* weights_;
  int* image_dims_;
  int* kernel1_dims_;
  int* kernel2_dims_;
  int* weights_dims_;
} view_data;

/* inc/multiviewnative.h:28-35 */
typedef struct workspace {
  view_data* data_;
  unsigned short num_views_;
  double lambda_;
  float minValue_;
  int num_iterations_;
} workspace;

typedef struct { float re, im; } cpx;

/* ------------------------------------------------------------------------------------------
 * 1-D complex FFT plans (stand-in for fftwf plans; cached per length like
 * inc/plan_store.h:99-124 caches per shape).
 * ---------------------------------------------------------------------------------------- */
typedef struct plan1d {
  int n;
  int nfac;
  int fac[40];
  cpx* w; /* w[j] = exp(-2 pi i j / n), computed in double, rounded to float */
  struct plan1d* next;
} plan1d;

static plan1d* g_plans = NULL;

static void factorize(int n, int* fac, int* nfac) {
  int k = 0;
  while (n % 4 == 0) { fac[k++] = 4; n /= 4; }
  while (n % 2 == 0) { fac[k++] = 2; n /= 2; }
  for (int p = 3; (long)p * p <= n; p += 2)
    while (n % p == 0) { fac[k++] = p; n /= p; }
  if (n > 1) fac[k++] = n;
  *nfac = k;
}

static plan1d* get_plan(int n) {
  plan1d* p;
#ifdef _OPENMP
  if (omp_in_parallel()) {
    for (p = g_plans; p; p = p->next)
      if (p->n == n) return p;
    fprintf(stderr, "[mvn_oracle] plan for n=%d requested inside a parallel region\n", n);
    abort();
  }
#endif
  for (p = g_plans; p; p = p->next)
    if (p->n == n) return p;
  p = (plan1d*)calloc(1, sizeof(plan1d));
  p->n = n;
  factorize(n, p->fac, &p->nfac);
  p->w = (cpx*)malloc(sizeof(cpx) * (size_t)(n > 0 ? n : 1));
  for (int j = 0; j < n; ++j) {
    double a = -2.0 * M_PI * (double)j / (double)n;
    p->w[j].re = (float)cos(a);
    p->w[j].im = (float)sin(a);
  }
  p->next = g_plans;
  g_plans = p;
  return p;
}

/*
 * Batched Stockham autosort FFT.  x holds `n` rows of `s0` interleaved sequences
 * (element j of sequence q at x[q + s0*j]); y is scratch of the same size.  The
 * result ends up in x.  sign = -1 forward, +1 backward; un-normalised either way
 * (FFTW convention, inc/fft_utils.h:84,103).
 */
static void cfft_batch(const plan1d* pl, cpx* x, cpx* y, int s0, int sign) {
  const int N = pl->n;
  int n = N;      /* current sub-transform length */
  long s = s0;    /* current stride (number of interleaved sequences) */
  cpx* in = x;
  cpx* out = y;
  const float fs = (float)sign;
  for (int f = 0; f < pl->nfac; ++f) {
    const int r = pl->fac[f];
    const int m = n / r;
    const int wstep = N / n; /* w_n^k = w_N^(k*wstep) */
    if (r == 2) {
      for (int p = 0; p < m; ++p) {
        cpx w = pl->w[(size_t)p * wstep];
        w.im *= -fs; /* table holds exp(-i..): forward keeps, backward conjugates */
        const cpx* a = in + s * p;
        const cpx* b = in + s * (p + m);
        cpx* o0 = out + s * (2 * p);
        cpx* o1 = out + s * (2 * p + 1);
        for (long q = 0; q < s; ++q) {
          float ar = a[q].re, ai = a[q].im, br = b[q].re, bi = b[q].im;
          float dr = ar - br, di = ai - bi;
          o0[q].re = ar + br;
          o0[q].im = ai + bi;
          o1[q].re = dr * w.re - di * w.im;
          o1[q].im = dr * w.im + di * w.re;
        }
      }
    } else if (r == 4) {
      for (int p = 0; p < m; ++p) {
        cpx w1 = pl->w[(size_t)p * wstep];
        cpx w2 = pl->w[(size_t)2 * p * wstep];
        cpx w3 = pl->w[(size_t)3 * p * wstep];
        w1.im *= -fs; w2.im *= -fs; w3.im *= -fs;
        const cpx* a0 = in + s * p;
        const cpx* a1 = in + s * (p + m);
        const cpx* a2 = in + s * (p + 2 * m);
        const cpx* a3 = in + s * (p + 3 * m);
        cpx* o0 = out + s * (4 * p);
        cpx* o1 = out + s * (4 * p + 1);
        cpx* o2 = out + s * (4 * p + 2);
        cpx* o3 = out + s * (4 * p + 3);
        for (long q = 0; q < s; ++q) {
          float t0r = a0[q].re + a2[q].re, t0i = a0[q].im + a2[q].im;
          float t1r = a0[q].re - a2[q].re, t1i = a0[q].im - a2[q].im;
          float t2r = a1[q].re + a3[q].re, t2i = a1[q].im + a3[q].im;
          /* (a1 - a3) * (sign*i): forward -i, backward +i */
          float dr = a1[q].re - a3[q].re, di = a1[q].im - a3[q].im;
          float t3r = -fs * di, t3i = fs * dr;
          float b1r = t1r + t3r, b1i = t1i + t3i;
          float b2r = t0r - t2r, b2i = t0i - t2i;
          float b3r = t1r - t3r, b3i = t1i - t3i;
          o0[q].re = t0r + t2r;
          o0[q].im = t0i + t2i;
          o1[q].re = b1r * w1.re - b1i * w1.im;
          o1[q].im = b1r * w1.im + b1i * w1.re;
          o2[q].re = b2r * w2.re - b2i * w2.im;
          o2[q].im = b2r * w2.im + b2i * w2.re;
          o3[q].re = b3r * w3.re - b3i * w3.im;
          o3[q].im = b3r * w3.im + b3i * w3.re;
        }
      }
    } else {
      /* generic radix: O(r^2) butterfly, roots of unity taken from the length-N table */
      const int rstep = N / r; /* w_r^k = w_N^(k*rstep) */
      for (int p = 0; p < m; ++p) {
        for (int k = 0; k < r; ++k) {
          cpx tw = pl->w[(size_t)((long)p * k % n) * wstep];
          tw.im *= -fs;
          cpx* o = out + s * ((long)r * p + k);
          for (long q = 0; q < s; ++q) {
            float accr = 0.f, acci = 0.f;
            for (int j = 0; j < r; ++j) {
              cpx wr = pl->w[(size_t)((long)j * k % r) * rstep];
              float wi = -fs * wr.im;
              const cpx a = in[q + s * (p + (long)j * m)];
              accr += a.re * wr.re - a.im * wi;
              acci += a.re * wi + a.im * wr.re;
            }
            o[q].re = accr * tw.re - acci * tw.im;
            o[q].im = accr * tw.im + acci * tw.re;
          }
        }
      }
    }
    n = m;
    s *= r;
    cpx* t = in; in = out; out = t;
  }
  if (in != x) memcpy(x, in, sizeof(cpx) * (size_t)N * (size_t)s0);
}

/* CPUs this process may really use: the affinity mask, capped by the cgroup CPU quota (a container on
 * a 256-thread host with a 16-CPU share runs 256 OpenMP threads 25x SLOWER than 1 on a 64^3 stack) */
static int usable_cpus(void) {
#ifdef _OPENMP
  int n = omp_get_num_procs();
#else
  int n = 1;
#endif
  const char* files[2] = {"/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"};
  FILE* f = fopen(files[0], "r");
  if (f) { /* cgroup v2: "<quota|max> <period>" */
    char q[64];
    long period = 0;
    if (fscanf(f, "%63s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
      const long quota = atol(q);
      const int c = (int)((quota + period - 1) / period);
      if (c >= 1 && c < n) n = c;
    }
    fclose(f);
  } else if ((f = fopen(files[1], "r")) != NULL) { /* cgroup v1 */
    long quota = -1, period = 0;
    if (fscanf(f, "%ld", &quota) != 1) quota = -1;
    fclose(f);
    f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
    if (f) {
      if (fscanf(f, "%ld", &period) != 1) period = 0;
      fclose(f);
    }
    if (quota > 0 && period > 0) {
      const int c = (int)((quota + period - 1) / period);
      if (c >= 1 && c < n) n = c;
    }
  }
  return n < 1 ? 1 : n;
}

static int resolve_threads(int nthreads) {
#ifdef _OPENMP
  if (nthreads <= 0) return usable_cpus();
  return nthreads;
#else
  (void)nthreads;
  return 1;
#endif
}

/* transform along a strided axis of a [outer][n][inner] complex array, in column blocks */
static void axis_pass(cpx* data, long outer, int n, long inner, int sign, int nthreads) {
  const plan1d* pl = get_plan(n);
  const long BLK = 16;
  const long nblk = (inner + BLK - 1) / BLK;
  const long njobs = outer * nblk;
#pragma omp parallel num_threads(nthreads)
  {
    cpx* a = (cpx*)malloc(sizeof(cpx) * (size_t)n * BLK);
    cpx* b = (cpx*)malloc(sizeof(cpx) * (size_t)n * BLK);
#pragma omp for schedule(static)
    for (long job = 0; job < njobs; ++job) {
      const long o = job / nblk;
      const long c0 = (job % nblk) * BLK;
      const int w = (int)((inner - c0) < BLK ? (inner - c0) : BLK);
      cpx* base = data + o * (long)n * inner + c0;
      for (int j = 0; j < n; ++j) memcpy(a + (long)j * w, base + (long)j * inner, sizeof(cpx) * w);
      cfft_batch(pl, a, b, w, sign);
      for (int j = 0; j < n; ++j) memcpy(base + (long)j * inner, a + (long)j * w, sizeof(cpx) * w);
    }
    free(a);
    free(b);
  }
}

/*
 * In-place un-normalised 3-D real->half-complex transform on the FFTW in-place layout
 * [d0][d1][2*(d2/2+1)] floats (inc/fft_utils.h:55-85 -> fftwf_execute_dft_r2c with the plan
 * of inc/plan_store.h:116-118; layout inc/image_stack_utils.h:24-42).
 */
static void builtin_forward_nt(float* buf, int d0, int d1, int d2, int nthreads) {
  const int nc = d2 / 2 + 1;
  const long rows = (long)d0 * d1;
  const plan1d* pl = get_plan(d2);
  get_plan(d1);
  get_plan(d0);
#pragma omp parallel num_threads(nthreads)
  {
    cpx* a = (cpx*)malloc(sizeof(cpx) * (size_t)d2);
    cpx* b = (cpx*)malloc(sizeof(cpx) * (size_t)d2);
#pragma omp for schedule(static)
    for (long r = 0; r < rows; ++r) {
      float* row = buf + r * 2L * nc;
      for (int j = 0; j < d2; ++j) { a[j].re = row[j]; a[j].im = 0.f; }
      cfft_batch(pl, a, b, 1, -1);
      memcpy(row, a, sizeof(cpx) * (size_t)nc);
    }
    free(a);
    free(b);
  }
  axis_pass((cpx*)buf, d0, d1, nc, -1, nthreads);
  axis_pass((cpx*)buf, 1, d0, (long)d1 * nc, -1, nthreads);
}

/* inverse of the above, un-normalised (inc/fft_utils.h:87-104 -> fftwf_execute_dft_c2r) */
static void builtin_backward_nt(float* buf, int d0, int d1, int d2, int nthreads) {
  const int nc = d2 / 2 + 1;
  const long rows = (long)d0 * d1;
  const plan1d* pl = get_plan(d2);
  get_plan(d1);
  get_plan(d0);
  axis_pass((cpx*)buf, 1, d0, (long)d1 * nc, +1, nthreads);
  axis_pass((cpx*)buf, d0, d1, nc, +1, nthreads);
#pragma omp parallel num_threads(nthreads)
  {
    cpx* a = (cpx*)malloc(sizeof(cpx) * (size_t)d2);
    cpx* b = (cpx*)malloc(sizeof(cpx) * (size_t)d2);
#pragma omp for schedule(static)
    for (long r = 0; r < rows; ++r) {
      float* row = buf + r * 2L * nc;
      const cpx* h = (const cpx*)row;
      /* Hermitian extension; imaginary parts of DC / Nyquist drop out of the real part */
      for (int k = 0; k < nc; ++k) a[k] = h[k];
      for (int k = nc; k < d2; ++k) { a[k].re = h[d2 - k].re; a[k].im = -h[d2 - k].im; }
      cfft_batch(pl, a, b, 1, +1);
      for (int j = 0; j < d2; ++j) row[j] = a[j].re;
    }
    free(a);
    free(b);
  }
}

/* ------------------------------------------------------------------------------------------
 * Optional FFTW backend.  The reference's CPU path calls FFTW3 single precision
 * (fftwf_plan_dft_r2c_3d / c2r_3d with FFTW_MEASURE, inc/plan_store.h:116-122;
 * fftwf_execute_dft_r2c / c2r, inc/fft_utils.h:84,103; threads via fftwf_init_threads +
 * fftwf_plan_with_nthreads, inc/fft_utils.h:180-197).  FFTW is not in this image, so nothing is
 * linked: when libfftw3f.so.3 can be dlopen'ed at run time (MVN_ORACLE_FFTW_LIB overrides the
 * name, MVN_ORACLE_FFTW=0 disables the probe) the 3-D transforms go through it -- bench.py's CPU
 * baseline is then literally the reference's FFT library ("kind": "fftw") -- otherwise through
 * the built-in transform above ("port").  Plans are cached per (shape, threads), like plan_store.
 * ---------------------------------------------------------------------------------------- */
#include <dlfcn.h>

typedef void* fftwf_plan_t;
typedef struct fftw_api {
  int state; /* 0 not probed, 1 loaded, -1 unavailable */
  int threads_ok;
  fftwf_plan_t (*plan_r2c)(int, int, int, float*, float*, unsigned);
  fftwf_plan_t (*plan_c2r)(int, int, int, float*, float*, unsigned);
  void (*exec_r2c)(fftwf_plan_t, float*, float*);
  void (*exec_c2r)(fftwf_plan_t, float*, float*);
  void (*destroy)(fftwf_plan_t);
  int (*init_threads)(void);
  void (*plan_with_nthreads)(int);
} fftw_api;
static fftw_api g_fftw;

typedef struct fftw_plans {
  int d0, d1, d2, nthreads;
  fftwf_plan_t fwd, bwd;
  struct fftw_plans* next;
} fftw_plans;
static fftw_plans* g_fftw_plans = NULL;

static int fftw_ready(void) {
  int st;
#pragma omp critical(mvn_oracle_fftw)
  {
    if (g_fftw.state == 0) {
      g_fftw.state = -1;
      const char* off = getenv("MVN_ORACLE_FFTW");
      if (!(off && strcmp(off, "0") == 0)) {
        const char* name = getenv("MVN_ORACLE_FFTW_LIB");
        void* h = dlopen(name && *name ? name : "libfftw3f.so.3", RTLD_NOW | RTLD_GLOBAL);
        if (h) {
          *(void**)&g_fftw.plan_r2c = dlsym(h, "fftwf_plan_dft_r2c_3d");
          *(void**)&g_fftw.plan_c2r = dlsym(h, "fftwf_plan_dft_c2r_3d");
          *(void**)&g_fftw.exec_r2c = dlsym(h, "fftwf_execute_dft_r2c");
          *(void**)&g_fftw.exec_c2r = dlsym(h, "fftwf_execute_dft_c2r");
          *(void**)&g_fftw.destroy = dlsym(h, "fftwf_destroy_plan");
          if (g_fftw.plan_r2c && g_fftw.plan_c2r && g_fftw.exec_r2c && g_fftw.exec_c2r) {
            g_fftw.state = 1;
            /* the threads API lives in the same library or in libfftw3f_omp / _threads
             * (cmake/FindFFTW.cmake:47-115) */
            const char* tl[3] = {NULL, "libfftw3f_omp.so.3", "libfftw3f_threads.so.3"};
            for (int i = 0; i < 3 && !g_fftw.threads_ok; ++i) {
              void* th = i == 0 ? h : dlopen(tl[i], RTLD_NOW | RTLD_GLOBAL);
              if (!th) continue;
              *(void**)&g_fftw.init_threads = dlsym(th, "fftwf_init_threads");
              *(void**)&g_fftw.plan_with_nthreads = dlsym(th, "fftwf_plan_with_nthreads");
              if (g_fftw.init_threads && g_fftw.plan_with_nthreads && g_fftw.init_threads())
                g_fftw.threads_ok = 1;
            }
          }
        }
      }
    }
    st = g_fftw.state;
  }
  return st == 1;
}

/* "fftw" or "port": which transform the 3-D FFTs of this process go through */
const char* oracle_fft_backend(void) { return fftw_ready() ? "fftw" : "port"; }

static fftw_plans* fftw_get_plans(int d0, int d1, int d2, int nthreads) {
  fftw_plans* p = NULL;
#pragma omp critical(mvn_oracle_fftw)
  {
    for (p = g_fftw_plans; p; p = p->next)
      if (p->d0 == d0 && p->d1 == d1 && p->d2 == d2 && p->nthreads == nthreads) break;
    if (!p) {
      /* FFTW_MEASURE (= 0, inc/plan_store.h:117,122) overwrites the arrays while planning:
       * plan on a scratch volume, execute with the new-array interface.  The arrays executed on
       * are the callers' (numpy buffers of any alignment), so the plans carry FFTW_UNALIGNED
       * (1 << 1): the new-array interface is only defined for arrays as aligned as the planned one
       * otherwise. */
      const unsigned flags = 0u /* FFTW_MEASURE */ | (1u << 1) /* FFTW_UNALIGNED */;
      const size_t nfl = (size_t)d0 * d1 * 2u * (size_t)(d2 / 2 + 1);
      float* scratch = (float*)calloc(nfl, sizeof(float));
      if (g_fftw.threads_ok) g_fftw.plan_with_nthreads(nthreads);
      p = (fftw_plans*)calloc(1, sizeof(fftw_plans));
      p->d0 = d0; p->d1 = d1; p->d2 = d2; p->nthreads = nthreads;
      p->fwd = g_fftw.plan_r2c(d0, d1, d2, scratch, scratch, flags);
      p->bwd = g_fftw.plan_c2r(d0, d1, d2, scratch, scratch, flags);
      free(scratch);
      p->next = g_fftw_plans;
      g_fftw_plans = p;
    }
  }
  return p;
}

static void rfft3_forward_nt(float* buf, int d0, int d1, int d2, int nthreads) {
  if (fftw_ready()) {
    fftw_plans* p = fftw_get_plans(d0, d1, d2, nthreads);
    if (p->fwd) { g_fftw.exec_r2c(p->fwd, buf, buf); return; }
  }
  builtin_forward_nt(buf, d0, d1, d2, nthreads);
}

static void rfft3_backward_nt(float* buf, int d0, int d1, int d2, int nthreads) {
  if (fftw_ready()) {
    fftw_plans* p = fftw_get_plans(d0, d1, d2, nthreads);
    if (p->bwd) { g_fftw.exec_c2r(p->bwd, buf, buf); return; }
  }
  builtin_backward_nt(buf, d0, d1, d2, nthreads);
}

/* the built-in transform whatever the backend (lets a test double of libfftw3f call back in) */
void oracle_builtin_rfft3_forward(float* buf, int d0, int d1, int d2, int nthreads) {
  builtin_forward_nt(buf, d0, d1, d2, resolve_threads(nthreads));
}
void oracle_builtin_rfft3_backward(float* buf, int d0, int d1, int d2, int nthreads) {
  builtin_backward_nt(buf, d0, d1, d2, resolve_threads(nthreads));
}

void oracle_rfft3_forward(float* buf, int d0, int d1, int d2, int nthreads) {
  rfft3_forward_nt(buf, d0, d1, d2, resolve_threads(nthreads));
}
void oracle_rfft3_backward(float* buf, int d0, int d1, int d2, int nthreads) {
  rfft3_backward_nt(buf, d0, d1, d2, resolve_threads(nthreads));
}

/* number of floats in the in-place r2c layout (inc/image_stack_utils.h:24-42) */
size_t oracle_padded_floats(int d0, int d1, int d2) {
  return (size_t)d0 * (size_t)d1 * 2u * (size_t)(d2 / 2 + 1);
}

/* ------------------------------------------------------------------------------------------
 * Padding helpers
 * ---------------------------------------------------------------------------------------- */

/* inc/padd_utils.h:11-40 wrapped_insert_at_point with _point = target extents
 * (no_padd::wrapped_insert_at_offsets, inc/padd_utils.h:91-95).  target must be
 * pre-zeroed by the caller when that is wanted (src/multiviewnative.cpp:161). */
void oracle_wrapped_insert(const float* kernel, const int* kdims, float* target,
                           const int* tdims) {
  for (long z = 0; z < kdims[0]; ++z)
    for (long y = 0; y < kdims[1]; ++y)
      for (long x = 0; x < kdims[2]; ++x) {
        long ix = x - kdims[2] / 2;
        long iy = y - kdims[1] / 2;
        long iz = z - kdims[0] / 2;
        if (ix < 0) ix += tdims[2];
        if (iy < 0) iy += tdims[1];
        if (iz < 0) iz += tdims[0];
        target[(iz * tdims[1] + iy) * (long)tdims[2] + ix] =
            kernel[(z * kdims[1] + y) * (long)kdims[2] + x];
      }
}

/* fft.padd_for_fft: Boost.MultiArray resize [d0][d1][d2] -> [d0][d1][2(d2/2+1)], old cells kept,
 * new cells value-initialised (inc/fft_utils.h:108-122) */
static void pad_rows(const float* src, float* dst, int d0, int d1, int d2) {
  const int rp = 2 * (d2 / 2 + 1);
  const long rows = (long)d0 * d1;
  for (long r = 0; r < rows; ++r) {
    memcpy(dst + r * rp, src + r * d2, sizeof(float) * (size_t)d2);
    for (int j = d2; j < rp; ++j) dst[r * rp + j] = 0.f;
  }
}

/* fft.resize_after_fft: crop back to [d0][d1][d2] (inc/fft_utils.h:124-128) */
static void crop_rows(const float* src, float* dst, int d0, int d1, int d2) {
  const int rp = 2 * (d2 / 2 + 1);
  const long rows = (long)d0 * d1;
  for (long r = 0; r < rows; ++r) memcpy(dst + r * d2, src + r * rp, sizeof(float) * (size_t)d2);
}

/* forwarded kernel: zero volume, wrapped insert, pad, forward r2c
 * (src/multiviewnative.cpp:160-173) */
static float* forwarded_kernel(const float* kernel, const int* kdims, const int* idims,
                               int nthreads) {
  const size_t n = (size_t)idims[0] * idims[1] * idims[2];
  float* vol = (float*)calloc(n, sizeof(float));
  oracle_wrapped_insert(kernel, kdims, vol, idims);
  float* padded = (float*)malloc(sizeof(float) * oracle_padded_floats(idims[0], idims[1], idims[2]));
  pad_rows(vol, padded, idims[0], idims[1], idims[2]);
  free(vol);
  rfft3_forward_nt(padded, idims[0], idims[1], idims[2], nthreads);
  return padded;
}

/* cpu_convolve<..., no_padd>::half_inplace (inc/cpu_convolve.h:217-291): image is
 * replaced by its cyclic convolution with the kernel whose forward transform is given. */
static void half_inplace(float* image, const int* dims, const float* fwd_kernel, float* work,
                         int nthreads) {
  const int d0 = dims[0], d1 = dims[1], d2 = dims[2];
  const size_t np = oracle_padded_floats(d0, d1, d2);
  pad_rows(image, work, d0, d1, d2);            /* :63-90 ctor copy + padd_for_fft :222 */
  rfft3_forward_nt(work, d0, d1, d2, nthreads); /* :223 */
  cpx* a = (cpx*)work;
  const cpx* b = (const cpx*)fwd_kernel;
  const size_t ncpx = np / 2;
  for (size_t i = 0; i < ncpx; ++i) { /* :256-266, serial scalar loop in the reference too */
    float re = a[i].re * b[i].re - a[i].im * b[i].im;
    float im = a[i].re * b[i].im + a[i].im * b[i].re;
    a[i].re = re;
    a[i].im = im;
  }
  rfft3_backward_nt(work, d0, d1, d2, nthreads); /* :268 */
  crop_rows(work, image, d0, d1, d2);            /* :269 + :280-290 */
  const size_t n = (size_t)d0 * d1 * d2;
  const float scale = (float)(1.0 / (double)n); /* :271-274 value_type scale = 1.0/size */
  for (size_t i = 0; i < n; ++i) image[i] *= scale; /* :275-278 */
}

/* ------------------------------------------------------------------------------------------
 * Pointwise kernels (serial forms are canonical: inc/cpu_kernels.h:19-90)
 * ---------------------------------------------------------------------------------------- */

/* Not in the reference: mirrors the product's MVN_PAD_GOOD_SIZE mode, where a view voxel that is
 * exactly 0 gives quotient 0 (instead of 0 * 1/0 = NaN) -- see include/multiviewnative.h. */
static int g_quotient_guard = 0;
void oracle_set_quotient_guard(int on) { g_quotient_guard = on; }

/* inc/cpu_kernels.h:19-26 */
void oracle_compute_quotient(const float* input, float* output, size_t size) {
  for (size_t i = 0; i < size; ++i) {
    if (g_quotient_guard && input[i] == 0.f) {
      output[i] = 0.f;
      continue;
    }
    float temp = (float)(1. / (double)output[i]);
    output[i] = input[i] * temp;
  }
}

/* inc/cpu_kernels.h:28-54 */
void oracle_final_values(float* psi, const float* integral, const float* weight, size_t size,
                         float minValue) {
  for (size_t i = 0; i < size; ++i) {
    float last_value = psi[i];
    float value = last_value * integral[i];
    float next_value;
    if (!(value > 0.f)) value = minValue;
    if (isnan(value) || isinf(value))
      next_value = minValue;
    else
      next_value = value > minValue ? value : minValue; /* std::max(value,_minValue) */
    next_value = weight[i] * (next_value - last_value) + last_value;
    psi[i] = next_value;
  }
}

/* inc/cpu_kernels.h:59-90 */
void oracle_regularized_final_values(float* psi, const float* integral, const float* weight,
                                     size_t size, double lambda, float minValue) {
  const float lambda_inv = (float)(1.f / lambda); /* :71 */
  for (size_t i = 0; i < size; ++i) {
    float last_value = psi[i];
    float value = last_value * integral[i];
    float next_value;
    if (value > 0.f)
      value = (float)((double)lambda_inv * (sqrt(1. + 2. * lambda * (double)value) - 1.)); /* :77 */
    else
      value = minValue;
    if (isnan(value) || isinf(value))
      next_value = minValue;
    else
      next_value = value > minValue ? value : minValue;
    next_value = weight[i] * (next_value - last_value) + last_value;
    psi[i] = next_value;
  }
}

/* inc/cuda_kernels.cuh:162-193 (device_finalValues_tikhonov, the update of the legacy
 * iterate_fft_tikhonov entry, src/multiviewnative.cu:588-590): lambda is a float there, the
 * regularised value is divided by lambda in double, and the weight blends against the new value. */
void oracle_legacy_tikhonov_final_values(float* image, const float* integral, const float* weight,
                                         size_t size, float minValue, float lambda) {
  for (size_t i = 0; i < size; ++i) {
    float temp_image = image[i];
    temp_image *= integral[i];                                                   /* :180-181 */
    float temp_weight = weight[i];
    if (temp_image > 0.f)
      temp_image = (float)((sqrt(1.0 + 2.0 * lambda * temp_image) - 1.) / lambda); /* :185 */
    else
      temp_image = minValue;
    float new_value = (minValue > temp_image) ? minValue : temp_image;           /* cmax, :189 */
    new_value = temp_weight * (new_value - temp_image) + temp_image;             /* :190 */
    image[i] = new_value;
  }
}

/* the per-view additive correction w*(next-last) of the formulas above, WITHOUT applying it:
 * building block of the simultaneous (Jacobi) multi-GPU mode. */
void oracle_update_delta(const float* psi, const float* integral, const float* weight,
                         float* delta, int accumulate, size_t size, double lambda,
                         float minValue) {
  const float lambda_inv = (float)(1.f / lambda);
  for (size_t i = 0; i < size; ++i) {
    float last_value = psi[i];
    float value = last_value * integral[i];
    float next_value;
    if (value > 0.f) {
      if (lambda > 0)
        value = (float)((double)lambda_inv * (sqrt(1. + 2. * lambda * (double)value) - 1.));
    } else
      value = minValue;
    if (isnan(value) || isinf(value))
      next_value = minValue;
    else
      next_value = value > minValue ? value : minValue;
    float d = weight[i] * (next_value - last_value);
    delta[i] = accumulate ? delta[i] + d : d;
  }
}

/* ------------------------------------------------------------------------------------------
 * ABI twins of the reference's CPU entry points (same names, same signatures)
 * ---------------------------------------------------------------------------------------- */

/* src/multiviewnative.cpp:273-293 -> cpu_convolve<>::inplace (inc/cpu_convolve.h:147-202) */
void inplace_cpu_convolution(imageType* im, int* imDim, imageType* kernel, int* kernelDim,
                             int nthreads) {
  const int nt = resolve_threads(nthreads);
  float* fk = forwarded_kernel(kernel, kernelDim, imDim, nt);
  float* work = (float*)malloc(sizeof(float) * oracle_padded_floats(imDim[0], imDim[1], imDim[2]));
  half_inplace(im, imDim, fk, work, nt);
  free(work);
  free(fk);
}

static double now_s(void) {
#ifdef _OPENMP
  return omp_get_wtime();
#else
  return 0.0;
#endif
}

/* wall time of the last inplace_cpu_deconvolve call, split the way the benches need it:
 * [0] = PSF preparation (src/multiviewnative.cpp:146-174), [1] = iteration loop (:191-229) */
static double g_last_timing[2] = {0.0, 0.0};
void oracle_last_timing(double* out2) {
  out2[0] = g_last_timing[0];
  out2[1] = g_last_timing[1];
}
int oracle_threads(int nthreads) { return resolve_threads(nthreads); }

/* src/multiviewnative.cpp:101-240 driver, :244-256 dispatch.  Sequential (Gauss-Seidel)
 * sweep over views: psi is updated in place after each view. */
void inplace_cpu_deconvolve(imageType* psi, workspace input, int nthreads) {
  const int nt = resolve_threads(nthreads);
  const int V = input.num_views_;
  if (V == 0) return;
  const double t0 = now_s();
  float** fk1 = (float**)calloc((size_t)V, sizeof(float*));
  float** fk2 = (float**)calloc((size_t)V, sizeof(float*));
  for (int v = 0; v < V; ++v) { /* :146-174 */
    const view_data* d = &input.data_[v];
    fk1[v] = forwarded_kernel(d->kernel1_, d->kernel1_dims_, d->image_dims_, nt);
    fk2[v] = forwarded_kernel(d->kernel2_, d->kernel2_dims_, d->image_dims_, nt);
  }
  const int* dims0 = input.data_[0].image_dims_; /* :181 psi has the shape of view 0 */
  const size_t n = (size_t)dims0[0] * dims0[1] * dims0[2];
  float* integral = (float*)malloc(sizeof(float) * n);
  float* work = (float*)malloc(sizeof(float) * oracle_padded_floats(dims0[0], dims0[1], dims0[2]));
  const double t1 = now_s();
  for (int it = 0; it < input.num_iterations_; ++it) { /* :191 */
    for (int v = 0; v < V; ++v) {                       /* :192 */
      const view_data* d = &input.data_[v];
      memcpy(integral, psi, sizeof(float) * n);                   /* :195 */
      half_inplace(integral, d->image_dims_, fk1[v], work, nt);   /* :198-200 */
      oracle_compute_quotient(d->image_, integral, n);            /* :203-204 */
      half_inplace(integral, d->image_dims_, fk2[v], work, nt);   /* :208-210 */
      if (input.lambda_ > 0)                                      /* :216-227 */
        oracle_regularized_final_values(psi, integral, d->weights_, n, input.lambda_,
                                        input.minValue_);
      else
        oracle_final_values(psi, integral, d->weights_, n, input.minValue_);
    }
  }
  g_last_timing[0] = t1 - t0;
  g_last_timing[1] = now_s() - t1;
  free(work);
  free(integral);
  for (int v = 0; v < V; ++v) { free(fk1[v]); free(fk2[v]); }
  free(fk1);
  free(fk2);
}

/*
 * Simultaneous (Jacobi) counterpart used as the parity oracle of the one-view-per-GPU mode
 * (SURVEY.md 8e): every view's correction is computed from the SAME psi_k and
 * psi_{k+1} = psi_k + sum_v w_v (next_v - psi_k).  For one view it equals the sequential sweep.
 * Restricting to views [v_begin, v_end) and returning the un-applied partial sum in `delta`
 * (apply==0) is what one rank computes before the all-reduce.
 */
void oracle_deconvolve_simultaneous_step(const float* psi, workspace input, int v_begin,
                                         int v_end, float* delta, int nthreads) {
  const int nt = resolve_threads(nthreads);
  const int* dims0 = input.data_[0].image_dims_;
  const size_t n = (size_t)dims0[0] * dims0[1] * dims0[2];
  float* integral = (float*)malloc(sizeof(float) * n);
  float* work = (float*)malloc(sizeof(float) * oracle_padded_floats(dims0[0], dims0[1], dims0[2]));
  memset(delta, 0, sizeof(float) * n);
  for (int v = v_begin; v < v_end; ++v) {
    const view_data* d = &input.data_[v];
    float* fk1 = forwarded_kernel(d->kernel1_, d->kernel1_dims_, d->image_dims_, nt);
    float* fk2 = forwarded_kernel(d->kernel2_, d->kernel2_dims_, d->image_dims_, nt);
    memcpy(integral, psi, sizeof(float) * n);
    half_inplace(integral, d->image_dims_, fk1, work, nt);
    oracle_compute_quotient(d->image_, integral, n);
    half_inplace(integral, d->image_dims_, fk2, work, nt);
    oracle_update_delta(psi, integral, d->weights_, delta, 1, n, input.lambda_, input.minValue_);
    free(fk1);
    free(fk2);
  }
  free(work);
  free(integral);
}

void oracle_deconvolve_simultaneous(float* psi, workspace input, int nthreads) {
  const int* dims0 = input.data_[0].image_dims_;
  const size_t n = (size_t)dims0[0] * dims0[1] * dims0[2];
  float* delta = (float*)malloc(sizeof(float) * n);
  for (int it = 0; it < input.num_iterations_; ++it) {
    oracle_deconvolve_simultaneous_step(psi, input, 0, input.num_views_, delta, nthreads);
    for (size_t i = 0; i < n; ++i) psi[i] += delta[i];
  }
  free(delta);
}

/* direct O(N*K) spatial convolution with zero outside, kernel flipped (true convolution):
 * the reference's oracle-of-oracle, tests/test_algorithms.hpp:10-58, over the whole volume. */
void oracle_spatial_convolve(const float* image, const int* idims, const float* kernel,
                             const int* kdims, float* result) {
  const int hz = kdims[0] / 2, hy = kdims[1] / 2, hx = kdims[2] / 2;
  for (int z = 0; z < idims[0]; ++z)
    for (int y = 0; y < idims[1]; ++y)
      for (int x = 0; x < idims[2]; ++x) {
        float value = 0.f;
        for (int kz = 0; kz < kdims[0]; ++kz)
          for (int ky = 0; ky < kdims[1]; ++ky)
            for (int kx = 0; kx < kdims[2]; ++kx) {
              int iz = z - hz + kz, iy = y - hy + ky, ix = x - hx + kx;
              if (iz < 0 || iy < 0 || ix < 0 || iz >= idims[0] || iy >= idims[1] || ix >= idims[2])
                continue;
              float kv = kernel[((long)(kdims[0] - 1 - kz) * kdims[1] + (kdims[1] - 1 - ky)) * kdims[2] +
                                (kdims[2] - 1 - kx)];
              value += kv * image[((long)iz * idims[1] + iy) * idims[2] + ix];
            }
        result[((long)z * idims[1] + y) * idims[2] + x] = value;
      }
}
