/*
 * mvn_engine_api.h -- C ABI of the device-resident engine underneath multiviewnative.h.
 *
 * Not part of the reference's interface: these entry points expose what the reference keeps
 * internal (gpu::plan_store, inc/plan_store.cuh:20-216; the all-on-device RL driver,
 * src/gpu_deconvolve_methods.cuh:345-562; inplace_3d_transform_on_device,
 * inc/cufft_utils.cuh:41-75) so that benches, the multi-GPU launcher and the parity tests can
 * keep data resident in HBM between steps.  Plain pointers and sizes only.  Every function
 * returns 0 on success and a negative value on failure (message via mvn_last_error()).
 */
#ifndef MVN_ENGINE_API_H
#define MVN_ENGINE_API_H

#include <stddef.h>

#include "multiviewnative.h"

typedef struct mvn_engine mvn_engine; /* opaque */

MVN_API const char* mvn_last_error(void);
MVN_API const char* mvn_backend_name(void);

/* inplace_gpu_deconvolve keeps one resident engine per device between calls (same shape and view
 * count are re-used without re-allocating; MVN_ENGINE_CACHE=0 disables it).  This frees them. */
MVN_API int mvn_release_cached_engines(void);

/* Block-after-block callers (Fiji deconvolves a large volume as a sequence of blocks): the
 * asynchronous form of inplace_gpu_deconvolve.  Submit starts the call on a worker thread and
 * returns a ticket; wait blocks until that call is complete (psi written) and returns its status.
 * Submits to one device alternate between two resident engines, so the stacks of block k+1 cross
 * PCIe while block k iterates -- what the reference's interleaved driver
 * (inplace_gpu_deconvolve_iteration_interleaved, src/gpu_deconvolve_methods.cuh:82-326, "not
 * supported yet" there) was meant to do inside one call.  The workspace struct, its view_data array
 * and the int[3] dims are copied at submit; psi and every stack / kernel buffer they point to must
 * stay valid and untouched until the matching wait returns.  The padding policy is the one in
 * force at submit.  Results are those of inplace_gpu_deconvolve, bit for bit.  Every ticket must
 * be waited for exactly once; inplace_gpu_deconvolve == submit + wait on the first engine. */
MVN_API int mvn_deconvolve_submit(imageType* psi, struct workspace input, int device, long long* ticket);
MVN_API int mvn_deconvolve_wait(long long ticket);
/* Padding policy of inplace_gpu_deconvolve for the calls that follow (process-wide): "zero"
 * (default: the reference GPU entry's zero_padd with FFT-friendly padded extents), "zero_exact"
 * (exactly image + kernel - 1), "none" (the reference CPU path's cyclic no_padd); NULL or ""
 * returns to the environment variable MVN_PAD_MODE / the default.  See multiviewnative.h. */
MVN_API int mvn_set_pad_mode(const char* mode);
/* The mode last selected with mvn_set_pad_mode ("zero" | "zero_exact" | "none"), or "" when the
 * environment / default decides -- what a caller that switches the policy for one call restores. */
MVN_API const char* mvn_get_pad_mode(void);
/* A resident engine keeps, per view slot, the PSF spectra of the last call together with host
 * copies of the kernels they were made from; a call (or mvn_engine_set_view) that brings
 * bytewise identical kernels for a slot re-uses the spectra (the reference's
 * GPU path recomputes them for every view and iteration, inc/gpu_convolve.cuh:121-124).
 * MVN_PSF_CACHE=0 disables it.  out[0] = spectra re-used, out[1] = spectra prepared, both since
 * process start. */
MVN_API int mvn_psf_cache_counters(long out[2]);
/* passes launched through the long-line (16-column, split-window) kernels since process start
 * (test / diagnostics; MVN_NO_SPLIT=1 keeps the 8-column kernels) */
MVN_API long mvn_split_launch_count(void);
/* launches of the fused middle pass (dim1 forward + direct dim0 leg + dim1 inverse in one pass over the line layout;
   csrc/mvn_mid_fused.hpp) since process start: tests check that the shapes that have it take it */
MVN_API long mvn_mid_fused_launch_count(void);
/* inplace_gpu_deconvolve calls that ran as dim0 slabs on the devices of MVN_DEVICES (multiviewnative.h) since
 * process start - a call that could not be cut that way ran on one device and is not counted */
MVN_API long mvn_multi_device_calls(void);

/* ---- plan_store (inc/plan_store.cuh: get()/add/has_key/empty/size/clear) ---------------- */
MVN_API int mvn_plan_store_add(int device, const int dims[3]);
MVN_API int mvn_plan_store_has_key(int device, const int dims[3]); /* 1 / 0 */
MVN_API int mvn_plan_store_size(void);
MVN_API int mvn_plan_store_empty(void);
MVN_API int mvn_plan_store_clear(void);
/* layout facts of a shape: {h, C, RP, even, rows_T, ax1_T, ax0_T, n_stages(d2 axis),
 * fixed-length kernel used for the last-axis / dim1 / dim0 passes (1/0 each), reserved} */
MVN_API int mvn_plan_describe(int device, const int dims[3], int out[12]);

/* ---- whole 3-D transforms on host buffers (test/bench utility) --------------------------
 * real:  dense [d0][d1][d2] floats.  spec: [d0][d1][d2/2+1] complex64 in the FFTW/cuFFT
 * in-place order (inc/image_stack_utils.h:24-42).  Un-normalised both ways. */
MVN_API int mvn_fft3_r2c(int device, const int dims[3], const float* real, float* spec);
MVN_API int mvn_fft3_c2r(int device, const int dims[3], const float* spec, float* real);
/* times `reps` forward (direction 0) or backward (1) transforms of a resident volume with
 * stream events; returns average milliseconds per transform in *ms */
MVN_API int mvn_fft3_time(int device, const int dims[3], int direction, int reps, float* ms);
/* same, plus the average launch time of every kernel kind (array of mvn_kernel_kind_count()) */
MVN_API int mvn_fft3_profile(int device, const int dims[3], int direction, int reps, float* ms,
                             double* per_kind_ms);

/* Batched transforms, the counterpart of the cufftPlanMany path of the reference's
 * bench/bench_gpu_many_nd_fft.cu:403-463: `batch` stacks of one shape, contiguous in `real`
 * ([batch][d0][d1][d2]), go through ONE cached plan back to back on one stream; `spec` receives
 * [batch][d0][d1][d2/2+1] complex64 in natural bin order. */
MVN_API int mvn_fft3_many_r2c(int device, const int dims[3], int batch, const float* real,
                              float* spec);
/* resident timing of the same sweep over `batch` stacks (direction 0 forward, 1 backward):
 * *ms = average milliseconds per sweep over the batch, stacks already in HBM */
MVN_API int mvn_fft3_many_time(int device, const int dims[3], int batch, int direction, int reps,
                               float* ms);

/* ---- resident RL engine ----------------------------------------------------------------- */
MVN_API int mvn_engine_create(int device, const int dims[3], int num_views, mvn_engine** out);
MVN_API int mvn_engine_destroy(mvn_engine* e);
MVN_API int mvn_engine_set_view(mvn_engine* e, int v, const float* image, const float* weights,
                                const float* kernel1, const int k1dims[3], const float* kernel2,
                                const int k2dims[3]);
MVN_API int mvn_engine_set_psi(mvn_engine* e, const float* psi);
MVN_API int mvn_engine_get_psi(mvn_engine* e, float* psi);
/* enqueue `iterations` sequential (Gauss-Seidel) sweeps over the views */
MVN_API int mvn_engine_iterate(mvn_engine* e, int iterations, double lambda, float min_value);
/* simultaneous (Jacobi) mode, one step: delta = sum_v w_v (next_v - psi) over this engine's
 * views; the caller all-reduces the delta buffer across ranks, then applies it */
MVN_API int mvn_engine_compute_delta(mvn_engine* e, double lambda, float min_value);
MVN_API int mvn_engine_apply_delta(mvn_engine* e);
/* The same step in pieces, so that the all-reduce runs under the compute.  The
 * correction leaves the LAST pass of the last local view and is consumed by plane-local passes
 * (psi += delta, then the forward last-axis and dim1 passes of the next step), so both ends go
 * chunk by chunk over ranges of dim0 planes:
 *   n = mvn_engine_delta_chunks(e, wanted)        (<= wanted; tile alignment of the shape)
 *   mvn_engine_compute_delta_head(e, lambda, minValue)
 *   for c in 0..n-1: mvn_engine_compute_delta_chunk(e, c, n); start all-reduce of floats
 *                    [first, first + count) of the delta buffer (mvn_engine_delta_chunk_range)
 *   for c in 0..n-1: wait for chunk c's all-reduce; mvn_engine_apply_delta_chunk(e, c, n, feed_next)
 * feed_next != 0 also leaves psi's transformed spectrum for the next _head call (another step
 * follows).  Everything is asynchronous on mvn_engine_stream(); an engine created with 0 views
 * contributes zeros (a rank without views still holds a replica of psi). */
MVN_API int mvn_engine_delta_chunks(mvn_engine* e, int wanted);
MVN_API int mvn_engine_delta_chunk_range(mvn_engine* e, int c, int n, size_t* first_float,
                                         size_t* n_floats);
MVN_API int mvn_engine_compute_delta_head(mvn_engine* e, double lambda, float min_value);
MVN_API int mvn_engine_compute_delta_chunk(mvn_engine* e, int c, int n);
MVN_API int mvn_engine_apply_delta_chunk(mvn_engine* e, int c, int n, int feed_next);
MVN_API int mvn_engine_delta_ptr(mvn_engine* e, void** dev_ptr, size_t* n_floats);
/* make the engine write its delta into caller-owned DEVICE memory (same size as the engine's
 * own buffer) so a collective library can reduce it in place; NULL restores the internal one */
MVN_API int mvn_engine_bind_delta(mvn_engine* e, void* dev_ptr);
/* Halo mode: one volume cut into dim0 slabs over several ranks, swept in the REFERENCE's view order
 * (src/multiviewnative.cpp:194-227) - the engine's volume is this rank's planes plus h = (deepest PSF) / 2 halo
 * planes either side (image 1, weights 0, psi anything there).  `fn(user, spectrum, view, conv)` is called on the
 * calling thread right before every convolution's dim0 leg, with the engine's stream drained: it must fill planes
 * [0, h) and [d0 - h, d0) of `spectrum` ([d0][d1][d2/2] complex, DEVICE memory) with the lower neighbour's last
 * and the upper neighbour's first h own planes; mvn_engine_copy_planes moves whole planes between `spectrum` and
 * an exchange buffer (to_buffer bit 0: spectrum -> buffer; bit 1: the buffer is HOST memory; bit 2: only enqueue the
 * copy on the engine's stream, do not wait for it).  Every PSF must have
 * at most 33 planes (direct dim0 leg); NULL switches the mode off.  drain == 0: `fn` is called WITHOUT waiting for the
 * stream and must order everything it does on mvn_engine_stream() itself (copies with bit 2, collectives issued with
 * that stream current): the host then never waits inside a sweep.  libmultiviewnative_amd/sharded.py drives it. */
MVN_API int mvn_engine_set_halo_hook(mvn_engine* e, void (*fn)(void* user, void* spectrum, int view, int conv),
                                     void* user, int drain);
MVN_API int mvn_engine_copy_planes(mvn_engine* e, void* spectrum, int plane0, int nplanes, void* buffer,
                                   int to_buffer);
/* After mvn_engine_set_halo_hook: the first and last `planes` planes of the engine's volume are halo planes that no
 * pass needs to compute - the last-axis and dim1 passes and the dim0 leg then run on the own planes only (a slab of
 * 64 + 2 x 15 planes does the work of 64, not of 94).  split != 0: the leg runs in two parts - first the own planes
 * that do not depend on the halos - and fn is called once more in between, with conv + 4: the halo planes have to be
 * in place (in stream order) only when THAT call returns, so an exchange started at the first call on another
 * stream runs beside the first part (mvn_multi.cpp does that; sharded.py passes 0). */
MVN_API int mvn_engine_set_halo_planes(mvn_engine* e, int planes, int split);
/* 1 / 0: would this engine hold a kernel of these extents in the direct dim0 form (which halo mode needs)?
 * Lets a driver refuse a PSF when it is handed over instead of failing inside the first sweep. */
MVN_API int mvn_engine_would_be_direct(mvn_engine* e, const int kdims[3]);
/* Non-finite values in halo mode.  An FFT-based convolution turns ONE Inf / NaN voxel of its input into a volume of
 * NaN (inc/cpu_convolve.h:256-268), which the update then clamps to minValue everywhere (inc/cpu_kernels.h:40-47,
 * 76-83).  The direct dim0 leg reproduces that through a 4-byte device word per engine, the "poison word": a leg
 * that met a non-finite input stores its epoch (the engine's count of direct legs, so the word never needs clearing)
 * there and the last-axis pass that ends the convolution turns every voxel into NaN when it finds the epoch of its
 * own convolution.  Slabs of one volume on several engines must agree: with bit 1 of `drain` set in
 * mvn_engine_set_halo_hook the hook is called a second time per convolution, right after the dim0 leg and the dim1
 * pass behind it have been enqueued, with conv + 2, and must leave the MAXIMUM of all slabs' words in every slab's
 * word before it returns / in stream order (epochs only grow and the slabs count in step: the maximum is the latest
 * report).  _ptr: the word's device address; _bind: make the engine use caller-owned device memory (4 bytes, zeroed;
 * NULL: back to its own) so that a collective can reduce it in place; _get drains the stream and reads the word;
 * _merge: word = max(word, value). */
MVN_API int mvn_engine_poison_ptr(mvn_engine* e, void** dev_ptr);
MVN_API int mvn_engine_bind_poison(mvn_engine* e, void* dev_ptr);
MVN_API int mvn_engine_poison_get(mvn_engine* e, unsigned* value);
MVN_API int mvn_engine_poison_merge(mvn_engine* e, unsigned value);
MVN_API int mvn_engine_psi_ptr(mvn_engine* e, void** dev_ptr, size_t* n_floats);
MVN_API int mvn_engine_stream(mvn_engine* e, void** hip_stream);
MVN_API int mvn_engine_sync(mvn_engine* e);
/* wall time of `iterations` sweeps measured with events on the engine stream */
MVN_API int mvn_engine_time_iterate(mvn_engine* e, int iterations, double lambda,
                                    float min_value, float* ms);
/* per-kernel event timing: enable (1 = every launch, n > 1 = the launches of every n-th
 * (view, iteration) only, which keeps the events' own cost below 1 %), run, then read totals.
 * kind indexes mvn_kernel_kind_name */
MVN_API int mvn_engine_profile(mvn_engine* e, int enable);
MVN_API int mvn_engine_profile_read(mvn_engine* e, int kind, double* total_ms, long* launches);
MVN_API int mvn_kernel_kind_count(void);
MVN_API const char* mvn_kernel_kind_name(int kind);
/* algorithmic bytes B = 4*d0*d1*2(d2/2+1) of the engine's shape */
MVN_API size_t mvn_engine_B(mvn_engine* e);

/* ---- one volume on several devices of ONE process (what MVN_DEVICES runs inside inplace_gpu_deconvolve) ------
 * The volume is cut into slabs of dim0 planes, one per entry of `devices` (an entry may repeat: two slabs on one
 * device), each an ordinary resident engine on its planes plus `halo_planes` = (deepest PSF) / 2 halo planes either
 * side, each driven by its own host thread; before every dim0 leg a slab pulls its neighbours' boundary planes with
 * peer copies (cyclically: the reference's convolution is cyclic), under the part of the leg that does not need them.
 * The sweep is the reference's view-after-view order (src/multiviewnative.cpp:194-227): results equal the
 * one-device engine's bit for bit.  Needs PSFs of at most 33 planes (direct dim0 leg), an even last extent, at least
 * halo_planes planes per slab, at most 8 slabs.  No communication library: events and hipMemcpyPeerAsync.
 *   mvn_group_create -> mvn_group_load (psi and the workspace's stacks, extents == dims, uploaded slab by slab)
 *   -> mvn_group_iterate (blocking; *ms = wall time of the sweeps, stacks resident) -> mvn_group_get_psi */
typedef struct mvn_group mvn_group; /* opaque */
MVN_API int mvn_group_create(const int* devices, int ndevices, const int dims[3], int halo_planes, int num_views,
                             mvn_group** out);
MVN_API int mvn_group_destroy(mvn_group* g);
MVN_API int mvn_group_load(mvn_group* g, const float* psi, struct workspace input);
MVN_API int mvn_group_iterate(mvn_group* g, int iterations, double lambda, float min_value, float* ms);
MVN_API int mvn_group_get_psi(mvn_group* g, float* psi);

/* ---- slab-decomposed engine: the SEQUENTIAL sweep on several GPUs ------
 * Rank `rank` of `nranks` keeps planes [rank*d0/nranks, (rank+1)*d0/nranks) of psi, of every
 * view and of every weight stack; all host arrays below are those slabs, dense
 * [d0/nranks][d1][d2] (kernels are passed whole).  The reference has no multi-GPU code
 * (src/gpu_deconvolve_methods.cuh runs one device); the arithmetic is the view-after-view sweep
 * of src/multiviewnative.cpp:191-227, pass for pass.
 *
 * Last-axis and dim1 passes are plane-local; the dim0 pass needs whole lines, so every
 * convolution exchanges the half-transformed slab twice.  The exchange is the CALLER's (one
 * all-to-all with equal splits per buffer, e.g. torch.distributed.all_to_all_single over RCCL):
 *
 *   for conv in (0, 1):                       conv 0: psi (*) kernel1, conv 1: quotient (*) kernel2
 *     mvn_slab_pack(h, v, conv)               ... fills A_main / A_nyq
 *     mvn_slab_sync(h); all-to-all A_main -> B_main and A_nyq -> B_nyq
 *     mvn_slab_mid(h, v, conv)                dim0 forward * PSF * inverse, in place on B
 *     mvn_slab_sync(h); all-to-all B_main -> A_main and B_nyq -> A_nyq
 *     mvn_slab_unpack(h, v, conv, lambda, minValue, feed_next)
 *                                             conv 0: view / blurred; conv 1: psi update
 *
 * Buffers hold main_floats / nyq_floats floats (nyq_floats == 0 for odd d2) and are split in
 * `nranks` equal contiguous parts by the all-to-all.  feed_next != 0 says another view update
 * follows (the update pass then leaves psi's last-axis transform for it).  Needs d0 and d1
 * divisible by nranks with quotients >= 2. */
typedef struct mvn_slab mvn_slab; /* opaque */
MVN_API int mvn_slab_create(int device, const int dims[3], int nranks, int rank, int num_views,
                            mvn_slab** out);
MVN_API int mvn_slab_destroy(mvn_slab* h);
MVN_API int mvn_slab_set_view(mvn_slab* h, int v, const float* image_slab,
                              const float* weights_slab, const float* kernel1,
                              const int k1dims[3], const float* kernel2, const int k2dims[3]);
MVN_API int mvn_slab_set_psi(mvn_slab* h, const float* psi_slab);
MVN_API int mvn_slab_get_psi(mvn_slab* h, float* psi_slab);
MVN_API int mvn_slab_buffer_sizes(mvn_slab* h, size_t* main_floats, size_t* nyq_floats);
/* device pointers of the exchange buffers; bind caller-owned device memory (e.g. torch tensors)
 * so that a collective library can work on them in place, all-null returns to engine-owned */
MVN_API int mvn_slab_buffers(mvn_slab* h, void** a_main, void** b_main, void** a_nyq,
                             void** b_nyq);
MVN_API int mvn_slab_bind_buffers(mvn_slab* h, void* a_main, void* b_main, void* a_nyq,
                                  void* b_nyq);
MVN_API int mvn_slab_begin(mvn_slab* h); /* psi was replaced: forget its cached transform */
MVN_API int mvn_slab_pack(mvn_slab* h, int v, int conv);
MVN_API int mvn_slab_mid(mvn_slab* h, int v, int conv);
MVN_API int mvn_slab_unpack(mvn_slab* h, int v, int conv, double lambda, float min_value,
                            int feed_next);
MVN_API int mvn_slab_sync(mvn_slab* h);
MVN_API int mvn_slab_stream(mvn_slab* h, void** hip_stream);

#endif
