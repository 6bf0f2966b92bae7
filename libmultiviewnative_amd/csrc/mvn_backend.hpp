// mvn_backend.hpp -- the thin device interface the engine is written against.
//
// Product build (libmultiviewnative.so): implemented in mvn_kernels.hip on HIP streams and
// gfx950 kernels.  There is NO CPU implementation in the product.
// Test-only build (libmvn_emu.so, -DMVN_HOST_EMU): implemented in mvn_backend_emu.cpp, where
// "device memory" is host memory and a launch runs the same workgroup bodies one block at a
// time on the CPU.  It exists to validate plans, index math and the RL driver on a box without
// a GPU; it is never linked into, or loaded by, the product library.
#pragma once

#include <cstddef>

#include "mvn_pass_bodies.hpp"
#include "mvn_dim0_direct.hpp"
#include "mvn_mid_fused.hpp"

namespace mvn {
namespace be {

typedef void* stream_t;
typedef void* event_t;

const char* backend_name();

int device_count();
void set_device(int dev);
int get_device();
void device_name(int dev, char* name256);
long long device_total_mem(int dev);
void device_mem_info(size_t* free_b, size_t* total_b);
void device_arch(int dev, int* major, int* minor);

void* dmalloc(size_t bytes);
void dfree(void* p);
void h2d(void* d, const void* h, size_t bytes, stream_t s);
void d2h(void* h, const void* d, size_t bytes, stream_t s);
void d2d(void* dst, const void* src, size_t bytes, stream_t s);
void h2d_2d(void* d, size_t dpitch, const void* h, size_t hpitch, size_t width, size_t height,
            stream_t s);
void d2h_2d(void* h, size_t hpitch, const void* d, size_t dpitch, size_t width, size_t height,
            stream_t s);
void d2d_2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height,
            stream_t s);
void dzero(void* d, size_t bytes, stream_t s);
// several devices in one process (slabs of one volume, mvn_multi.cpp): let kernels and copies running on `dev`
// reach memory of `peer` (once per pair; a no-op for dev == peer), and a device-to-device copy between two
// devices' memories enqueued on a stream of either
void enable_peer_access(int dev, int peer);
void copy_peer(void* dst, int dst_dev, const void* src, int src_dev, size_t bytes, stream_t s);

stream_t stream_create();
// a stream of the device's highest priority: it gets hardware queues of its own, so that host->device
// copies never sit behind another engine's kernels in a SHARED hardware queue (ROCm multiplexes a
// process's normal-priority streams onto 4 queues: with two resident engines x (compute, Nyquist,
// upload) streams the upload stream of one aliased the compute stream of the other and block k+1's
// upload ran at 31 instead of 56 GB/s under block k's iterations -- profiles/r03_pipeline.md)
stream_t stream_create_upload();
void stream_destroy(stream_t s);
void stream_sync(stream_t s);

// make every later operation on `s` wait for event `e` (recorded on another stream)
void stream_wait_event(stream_t s, event_t e);
event_t event_create();
// an event that only orders streams (no timestamps: cheaper to record and to wait on)
event_t event_create_sync();
void event_destroy(event_t e);
void event_record(event_t e, stream_t s);
void event_sync(event_t e);
float event_elapsed_ms(event_t a, event_t b);

// stream capture into an executable graph (launch-bound small volumes replay a whole sweep with
// one submission); the host emulation has none
typedef void* graph_exec_t;
bool graphs_supported();
void capture_begin(stream_t s);
graph_exec_t capture_end(stream_t s);  // ends the capture and instantiates the graph
void graph_launch(graph_exec_t g, stream_t s);
void graph_destroy(graph_exec_t g);

// ---- kernel launches --------------------------------------------------------------------
void launch_rows_r2c(const RowsParams& p, bool even, long ntiles, int nthreads, size_t lds_bytes,
                     stream_t s);
void launch_rows_c2r(const RowsParams& p, bool even, long ntiles, int nthreads, size_t lds_bytes,
                     stream_t s);
// even d2 only: c2r + pointwise epilogue + r2c of the result in one pass (reads p.in_cplx /
// p.in_nyq, writes p.out_cplx / p.out_nyq, UPDATE also writes epi.psi); p.fixed selects the
// compile-time specialised kernel
void launch_rows_c2r_r2c(const RowsParams& p, long ntiles, int nthreads, size_t lds_bytes,
                         stream_t s);
// launches that went through the long-line (split-window, 16-column) kernels since process start
long split_launch_count();
// launches of the fused middle pass (mvn_mid_fused.hpp) since process start
long mid_fused_launch_count();
// `rider` (plain fixed-length passes only): a second pass of the same mode with tiles of ONE line (T = 1, run-time
// radix body) - the lines of the Nyquist plane, taken by rider->tiles_per_outer further workgroups of the same launch
void launch_strided(int mode, const StridedParams& p, long nblocks, int nthreads,
                    size_t lds_bytes, stream_t s, const StridedParams* rider = nullptr, size_t rider_lds = 0);

// the dim0 leg of a convolution as a direct cyclic convolution with the PSF's few planes
// (mvn_dim0_direct.hpp); p.k must satisfy mvn_dim0_direct_possible(p.k, p.d0)
void launch_dim0_direct(const Dim0DirectParams& p, stream_t s);

// dim1 forward + direct dim0 leg + dim1 inverse in ONE pass over a half-spectrum in the line layout, or (p.mode ==
// MF_TAPS) the forward transform of a PSF's planes into the bin order that pass filters in (mvn_mid_fused.hpp)
void launch_mid_fused(const MidFusedParams& p, stream_t s);

// target[(z-kz/2 mod D0, y-ky/2 mod D1, x-kx/2 mod D2)] = kernel[z][y][x] * scale
// (device-side wrapped_insert_at_point, inc/padd_utils.h:11-40; the reference's GPU twin is
// fftShiftKernel, src/multiviewnative.cu:154-192).  target has row pitch `pitch` floats.
void launch_scatter_psf(const float* kernel, int k0, int k1, int k2, float* target, int D0,
                        int D1, int D2, long pitch, float scale, stream_t s);

// strided 3-D copy of floats, dst[z][y][x] = src[z][y][x] for x < nx, y < ny, z < nz with row /
// plane pitches in floats: embeds a dense stack into a zero-padded volume and crops it back
// (insert_at_offsets / the crop on exit, inc/padd_utils.h:160-190, src/gpu_deconvolve_methods.cuh:
// 537-549, done by the reference on the host)
void launch_copy3d(float* dst, long drow, long dplane, const float* src, long srow, long splane,
                   int nx, int ny, int nz, stream_t s);

// stand-alone pointwise ops on flat arrays (legacy ABI: compute_quotient / compute_final_values)
void launch_divide(const float* view, float* inout, size_t n, stream_t s);
void launch_update(float* psi, const float* integral, const float* weights, size_t n,
                   double lambda, float min_value, stream_t s);
// psi += delta  (applies the all-reduced correction in simultaneous mode)
// legacy iterate_fft_tikhonov final step (inc/cuda_kernels.cuh:162-193)
void launch_update_legacy_tikhonov(float* image, const float* integral, const float* weights,
                                   size_t n, float lambda_f, float min_value, stream_t s);
void launch_axpy1(float* psi, const float* delta, size_t n, stream_t s);

}  // namespace be
}  // namespace mvn
