// mvn_engine.hpp -- plans, plan_store and the device-resident Richardson-Lucy engine.
//
// Replaces, MI355X-first, the reference's
//   gpu::plan_store<float>                         inc/plan_store.cuh:20-216
//   inplace_convolve_on_device                     inc/gpu_convolve.cuh:113-142
//   inplace_gpu_deconvolve_iteration_all_on_device src/gpu_deconvolve_methods.cuh:345-562
// Everything a call needs (views, weights, both PSF spectra per view, psi, one work volume)
// stays resident in HBM for the whole call; PSF spectra are computed once per call, not once
// per (view, iteration) as the reference GPU path does (inc/gpu_convolve.cuh:121-124).
#pragma once

#include <array>
#include <condition_variable>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "mvn_backend.hpp"
#include "mvn_plan.hpp"

namespace mvn {

enum KernelKind {
  KK_ROWS_R2C = 0,
  KK_ROWS_C2R,
  KK_ROWS_FUSED,      // c2r + divide + r2c
  KK_ROWS_FUSED_UPD,  // c2r + psi update + r2c
  KK_AXIS1_FWD,
  KK_AXIS1_INV,
  KK_AXIS0_FUSED,
  KK_AXIS0_FWD,
  KK_AXIS0_INV,
  KK_NYQ,
  KK_OTHER,
  KK_AXIS0_DIRECT,  // dim0 leg as a direct convolution with the PSF's planes (mvn_dim0_direct.hpp)
  KK_MID_FUSED,     // dim1 forward + direct dim0 leg + dim1 inverse in one pass (mvn_mid_fused.hpp)
  KK_COUNT
};
const char* kernel_kind_name(int k);

// Optional per-launch timing with events on the launch stream (bench.py's roofline leg).
class Profiler {
 public:
  bool enabled = false;
  int sample_every = 1;  // time every n-th (view, iteration) only: keeps the events' own cost low
  void begin(int kind, be::stream_t s);
  void end(be::stream_t s);
  // waits for the recorded events and folds them into the totals
  void collect();
  void reset();
  double total_ms[KK_COUNT] = {0};
  long count[KK_COUNT] = {0};
  ~Profiler();

 private:
  struct Rec {
    int kind;
    be::event_t a, b;
  };
  std::vector<Rec> recs_;
  std::vector<be::event_t> pool_;
  be::event_t get_event();
};

struct DevAxis {
  AxisPlanHost host;
  cfloat* tw = nullptr;
  cfloat* tws = nullptr;
  cfloat* chirp = nullptr;  // bluestein axes only
  cfloat* bhat = nullptr;
  int* rev = nullptr;
  int* inv = nullptr;
  AxisPlan view;
  explicit DevAxis(int n, bool composite = false);
  ~DevAxis();
  DevAxis(const DevAxis&) = delete;
  DevAxis& operator=(const DevAxis&) = delete;
};

struct PassGeom {
  int T = 1, TP = 1, threads = 256;
  size_t lds_bytes = 0;
  long lds_alt = 0;
  long lds_tw = 0;
};

typedef std::array<int, 3> shape_t;

// a second stream plus the two events used to fork it from / join it to the main stream
struct SideStream {
  be::stream_t s = nullptr;
  be::event_t fork = nullptr, join = nullptr;
  void create();
  void destroy();
  // side stream waits for everything enqueued on `main` so far / `main` waits for the side stream
  void fork_from(be::stream_t main);
  void join_into(be::stream_t main);
};

class Plan3D {
 public:
  const int device;
  const Layout L;
  DevAxis ax2, ax1, ax0;
  cfloat* twr = nullptr;  // d2-th roots of unity (even d2)
  unsigned* no_poison = nullptr;  // a zero word: what EpilogueParams::poison points to after an FFT dim0 leg
  PassGeom g_rows, g_ax1, g_ax0, g_ax0f, g_nyq1, g_nyq0, g_nyq1_line;
  bool nyq_rides() const;
  // compile-time specialised kernels (mvn_fixed.hpp) are used where the shape allows
  bool fx_rows = false, fx_ax1 = false, fx_ax0 = false;
  PassGeom gx_rows, gx_ax1, gx_ax0;

  Plan3D(int device, int d0, int d1, int d2);
  ~Plan3D();

  size_t main_bytes() const { return L.real_floats() * sizeof(float); }
  size_t nyq_bytes() const { return L.nyq_cplx() * sizeof(cfloat); }

  // last-axis passes
  // `row0` / `nrows` (default: all d0*d1 rows) restrict a pass to a range of rows: every array a
  // last-axis pass touches is indexed by row, so a range is the same launch on shifted pointers.
  // The view-sharded driver uses it to produce and consume the correction chunk by chunk under
  // the all-reduce (Engine::compute_delta_chunk / apply_delta_chunk).
  // `lines`: the half-spectrum is in the LINE layout of the fused middle pass ([plane][position][row], Nyquist bins
  // packed: the Nyquist pointer must be null); out-of-place for rows_r2c, in place tile by tile for rows_c2r_r2c
  void rows_r2c(const float* in_real, cfloat* out, cfloat* out_nyq, be::stream_t s,
                Profiler* prof = nullptr, long row0 = 0, long nrows = -1, bool lines = false) const;
  void rows_c2r(const cfloat* in, const cfloat* in_nyq, float* out_real,
                const EpilogueParams& epi, be::stream_t s, Profiler* prof = nullptr,
                long row0 = 0, long nrows = -1, bool lines = false) const;
  // the shape has the line-layout forms of the last-axis kernels and the fused middle pass (d1 = d2 = 512)
  bool lines_capable() const;
  // dim1 forward + K-tap direct convolution along dim0 + dim1 inverse, `in` -> `out` (both in the line layout);
  // taps: [kd][C][d1] as prepared by taps_to_lines()
  // zcount > 0: the output planes [zbeg, zbeg + zcount) only (slabs with halo planes); peers: further poison words
  void mid_fused(const cfloat* in, cfloat* out, const cfloat* taps, int k, int kd, unsigned* poison,
                 unsigned poison_epoch, be::stream_t s, Profiler* prof = nullptr, int zbeg = 0, int zcount = 0,
                 int n_peers = 0, unsigned* const* peers = nullptr) const;
  // PSF planes after the last-axis pass (line layout, this plan = the small plan of the tap arrays) -> transformed
  // along dim1 into the bin order mid_fused() filters in; in place
  void taps_to_lines(cfloat* taps, be::stream_t s) const;
  // rows per tile of the last-axis passes and whether a launch needs whole tiles
  int rows_tile() const { return fx_rows ? gx_rows.T : g_rows.T; }
  bool rows_need_full_tiles() const { return fx_rows; }
  // fused c2r + pointwise + r2c (even d2 only, see can_fuse_rows()); in place on
  // (data, nyq); epi.mode is DIVIDE, UPDATE or STORE
  bool can_fuse_rows() const { return L.even; }
  void rows_c2r_r2c(cfloat* data, cfloat* nyq, const EpilogueParams& epi, be::stream_t s,
                    Profiler* prof = nullptr, long row0 = 0, long nrows = -1, bool lines = false) const;
  // strided passes on the main array and its Nyquist plane; mode = MvnStridedMode
  // `s_nyq` (default: s) is the stream of the small Nyquist-plane launches; giving them their own
  // stream lets the 2 MB plane ride along with the full-volume passes (see SideStream)
  // `z0` / `nz` (default: all d0 planes) restrict the pass to a range of planes (lines along d1
  // never leave their plane)
  void axis1(int mode, cfloat* data, cfloat* nyq, be::stream_t s, Profiler* prof = nullptr,
             be::stream_t s_nyq = nullptr, int z0 = 0, int nz = -1) const;
  // `src` / `src_nyq` (optional) make the pass out of place: read there, write to data / nyq
  // `spec_tiled`: the main-array spectrum is in the tile-contiguous layout of retile_spectrum()
  void axis0(int mode, cfloat* data, cfloat* nyq, const cfloat* spec, const cfloat* spec_nyq,
             be::stream_t s, Profiler* prof = nullptr, be::stream_t s_nyq = nullptr,
             const cfloat* src = nullptr, const cfloat* src_nyq = nullptr, bool spec_tiled = false) const;
  // PSF spectra of a resident engine are kept TILE-CONTIGUOUS for the fused dim0 pass where that
  // pass runs the fixed-length kernels: [tile][row][T columns] instead of [row][d1 * C columns], so
  // that the operand fetch of a tile is one stream of d0 * T * 8 bytes (64 KB at 512^3) instead of d0
  // row segments of 128 bytes, each on another 1 MB-strided page.
  bool tiles_spectra() const;
  void retile_spectrum(const cfloat* natural, cfloat* tiled, be::stream_t s) const;

  // whole transforms, un-normalised, in place on (vol, nyq)
  void forward(float* vol, cfloat* nyq, be::stream_t s, Profiler* prof = nullptr) const;
  void backward(float* vol, cfloat* nyq, float scale, be::stream_t s,
                Profiler* prof = nullptr) const;
  // cyclic convolution: out <- epilogue( IFFT( FFT(in) * spec ) ); `work` may alias `in`
  void convolve(const float* in_real, cfloat* work, cfloat* work_nyq, const cfloat* spec,
                const cfloat* spec_nyq, float* out_real, const EpilogueParams& epi,
                be::stream_t s, Profiler* prof = nullptr) const;
  // the three strided passes of a convolution (dim1 forward, dim0 forward*PSF*inverse, dim1
  // inverse); with `side` the Nyquist plane's three launches run on a second stream, forked
  // after and joined before the last-axis passes that produce / consume the plane
  void middle_passes(cfloat* work, cfloat* work_nyq, const cfloat* spec, const cfloat* spec_nyq,
                     be::stream_t s, Profiler* prof, struct SideStream* side, bool spec_tiled = false) const;
  // spectrum of a PSF: zero volume, centre->origin wrapped insert scaled by `scale`, forward FFT
  void psf_spectrum(const float* d_kernel, const int* kdims, float scale, float* spec_vol,
                    cfloat* spec_nyq, be::stream_t s) const;

 private:
  static PassGeom pick_geom(int n, bool generic, bool rows, int max_t);
};

// Process-wide plan cache keyed by (device, logical shape); the reference keys by logical shape
// only (inc/plan_store.cuh:31-33) because it knows a single device.  Thread-safe, unlike the
// reference's (SURVEY.md 8b "Threading").
class PlanStore {
 public:
  static PlanStore& get();
  std::shared_ptr<Plan3D> add(int device, const shape_t& shape);
  bool has_key(int device, const shape_t& shape);
  // throws std::runtime_error on a miss, like the reference (inc/plan_store.cuh:140-152)
  std::shared_ptr<Plan3D> lookup(int device, const shape_t& shape);
  bool empty();
  size_t size();
  void clear();

 private:
  std::mutex mu_;
  std::map<std::array<int, 4>, std::shared_ptr<Plan3D>> plans_;
};

struct ViewSlot {
  float* image = nullptr;
  float* weights = nullptr;
  float* spec1 = nullptr;  // main array of the kernel1 spectrum (pre-scaled by 1/N)
  cfloat* nyq1 = nullptr;
  float* spec2 = nullptr;
  cfloat* nyq2 = nullptr;
  bool set = false;
  // host copies of the kernels whose spectra spec1 / spec2 currently hold: a later call that
  // brings the same PSF for this slot (Fiji's block-after-block calls do) skips the preparation
  std::vector<float> kcopy[2];
  int kdims[2][3] = {{0, 0, 0}, {0, 0, 0}};
  // direct dim0 form of kernel i (mvn_dim0_direct.hpp): its planes after the last-axis and dim1
  // transforms, [tap_kd][d1][C] (+ Nyquist [tap_kd][d1]), pre-scaled by 1 / (d1 d2).  tap_k = PSF planes
  // when this form is what the slot holds for kernel i, 0 when the 3-D spectrum (spec / nyq) is.
  float* taps[2] = {nullptr, nullptr};
  cfloat* taps_nyq[2] = {nullptr, nullptr};
  int tap_k[2] = {0, 0};
  int tap_kd[2] = {0, 0};
  // the same PSF planes for the fused middle pass (mvn_mid_fused.hpp): [tap_kd][C][d1], Nyquist bins packed, bins
  // along dim1 in that pass's own order; taps_scr: the scattered PSF the last-axis pass reads (it cannot run in
  // place into the line layout).  taps_l_ok: valid for the kernel tap_k / tap_kd describe.
  float* taps_l[2] = {nullptr, nullptr};
  float* taps_scr[2] = {nullptr, nullptr};
  bool taps_l_ok[2] = {false, false};
};

class Engine {
 public:
  Engine(int device, const shape_t& dims, int num_views);
  ~Engine();
  Engine(const Engine&) = delete;
  Engine& operator=(const Engine&) = delete;

  int device() const { return device_; }
  const Layout& layout() const { return plan_->L; }
  int num_views() const { return (int)views_.size(); }
  be::stream_t stream() const { return stream_; }
  Profiler& profiler() { return prof_; }

  // host -> device staging (blocking); arrays are dense [d0][d1][d2] floats
  void set_view(int v, const float* image, const float* weights, const float* kernel1,
                const int* k1dims, const float* kernel2, const int* k2dims);
  void set_psi(const float* host);
  void get_psi(float* host);

  // Pipelined staging for the blocking ABI call (what the reference's "interleaved" driver was
  // after, src/gpu_deconvolve_methods.cuh:82-326): a second host thread uploads view after view
  // on its own stream while the first RL iteration already runs on the views that have arrived.
  //   reserve_views()            main thread: allocate every view's buffers up front
  //   stage_view(v, ...)         uploader thread: H2D + PSF spectra of view v on the upload stream
  //   staging_failed()           uploader thread: wake the main thread up after an error
  //   iterate(...)               main thread: waits for view v only before its first use
  void reserve_views();
  void stage_view(int v, const float* image, const float* weights, const float* kernel1,
                  const int* k1dims, const float* kernel2, const int* k2dims);
  void staging_failed();
  void finish_staging();  // uploader thread, after the last view: drain and free scratch

  // `iterations` Gauss-Seidel sweeps over all views (the reference order,
  // src/gpu_deconvolve_methods.cuh:487-535); asynchronous on stream()
  void iterate(int iterations, double lambda, float min_value);
  // simultaneous (Jacobi) mode for view sharding: delta <- sum over this engine's views of
  // w_v (next_v - psi), computed from the current psi without changing it
  void compute_delta(double lambda, float min_value);
  void apply_delta();
  // The same step in pieces, so that the caller's all-reduce can run under the compute
  // ("overlap ... by chunking along dim0").  The correction is produced by the LAST
  // pass of the last local view and consumed by passes that never leave a dim0 plane (psi += delta,
  // forward last-axis and dim1 passes of the next iteration), so both ends can go chunk by chunk:
  //   n = delta_chunks(wanted)
  //   compute_delta_head(lambda, min)            all passes but the last view's final one
  //   for c < n: compute_delta_chunk(c, n)       planes [c d0/n, (c+1) d0/n) of delta are final;
  //                                              the caller starts that chunk's all-reduce
  //   for c < n: [wait for chunk c's collective] apply_delta_chunk(c, n, feed_next)
  // feed_next != 0 additionally leaves the chunk's part of psi's dim1-transformed spectrum for the
  // next compute_delta_head (which then skips those two shared passes).
  int delta_chunks(int wanted) const;
  void delta_chunk_range(int c, int n, size_t* first_float, size_t* n_floats) const;
  void compute_delta_head(double lambda, float min_value);
  void compute_delta_chunk(int c, int n);
  void apply_delta_chunk(int c, int n, bool feed_next);
  float* delta_ptr();
  // use caller-owned device memory (volume_floats() floats) as the delta buffer, so that a
  // collective library can all-reduce it in place; nullptr returns to an engine-owned buffer
  void bind_delta(float* external);
  // Halo mode (dim0 slabs of one volume on several ranks, reference update order): the engine's volume is this
  // rank's planes with h halo planes either side; `fn` is called on the host right before every dim0 leg, with
  // the stream drained, and fills the halo planes of `spectrum` (the leg's input, [d0][d1][C] complex, packed
  // Nyquist layout) with the neighbours' planes (copy_planes moves planes between it and exchange buffers).
  // Needs every PSF in the direct form and the packed layout; throws at the first sweep otherwise.
  typedef void (*halo_fn_t)(void* user, void* spectrum, int view, int conv);
  // drain: wait for the stream before calling fn (fn works from the host); false: fn only enqueues work on stream()
  // post: fn is called a second time right AFTER the leg has been enqueued, with conv + 2 - where the ranks merge
  // their poison words (a non-finite input met by ONE slab's leg must turn EVERY slab's volume into NaN)
  void set_halo_hook(halo_fn_t fn, void* user, bool drain = true, bool post = false);
  // Tells the engine how many planes at either end of its volume are halo planes (after set_halo_hook): no pass
  // computes them any more - the last-axis and dim1 passes and the leg run on the own planes only (where those
  // start and end on tile boundaries of the last-axis passes; otherwise everything is computed as before).
  // split: the leg runs in two parts, the own planes that do not depend on the halos first, and fn is called
  // once more in between, with conv + 4 - the halo planes need to be in place only behind THAT call, so that an
  // exchange started at the first call runs beside the first part.
  void set_halo_planes(int planes, bool split);
  // The hook also fills the halo planes of the Nyquist plane (leg_input_nyq(), [d0][d1] complex) when the engine
  // runs the split layout: the engine then need not pack the Nyquist bins into the DC column, which costs 6 % at
  // 512^3 (profiles/r04_layouts.md).  layout: -1 the engine's own rule (packed up to 256 MB per volume), 0 split,
  // 1 packed - slabs of one volume take the WHOLE volume's layout, so that they run its arithmetic.
  void set_halo_nyq_aware(int layout) {
    halo_nyq_aware_ = true;
    layout_override_ = layout;
    if (!plan_->nyq_rides()) layout_override_ = 1;  // (run-time-radix dim1 kernels carry no riders)
  }
  // input of the dim0 leg in flight: its Nyquist plane (nullptr in the packed layout); valid inside the hook
  void* leg_input_nyq() const { return (packed_ || lines_) ? nullptr : (void*)work_nyq_; }
  // The poison word (mvn_dim0_direct.hpp, EpilogueParams::poison): a direct dim0 leg that met a non-finite input
  // stores its epoch there.  poison_ptr(): the device word; bind_poison(): use caller-owned device memory (4 bytes,
  // zeroed) instead, e.g. a torch tensor a collective can MAX-reduce in place; poison_get() drains the stream and
  // reads it; poison_merge(v): word = max(word, v), enqueued on stream(); add_poison_peer(): a further word (another
  // slab's, reachable from this device) every report is also written to.
  unsigned* poison_ptr() { return poison_; }
  void bind_poison(unsigned* external);
  unsigned poison_get();
  void poison_merge(unsigned value);
  void add_poison_peer(unsigned* word);
  void clear_poison_peers() { poison_peers_.clear(); }  // (the device table is only read up to n_peers)
  // slabs of one volume on several engines count their direct legs together: same epoch for the same leg
  unsigned poison_epoch() const { return epoch_; }
  void set_poison_epoch(unsigned e) { epoch_ = e; }
  void copy_planes(void* spectrum, int plane0, int nplanes, void* buffer, bool to_buffer, bool host_buffer = false,
                   bool wait = true);
  float* psi_ptr() { return psi_; }
  size_t volume_floats() const { return plan_->L.real_floats(); }
  // Host stacks of extents `dims` live at offset `off` inside the engine's (larger) volume, the
  // rest of which holds zeros: set_view / stage_view / set_psi embed, get_psi crops (the
  // reference's zero_padd policy, inc/padd_utils.h:121-190, without the host-side staging
  // copies).  dims == engine extents and off == 0 is the dense default.
  void set_embedding(const int dims[3], const int off[3]);
  // quotient 0 wherever the view is exactly 0 (see EpilogueParams::guard_zero_view)
  void set_quotient_guard(bool on) { quotient_guard_ = on; }
  // pipelined ABI call: will a kernel of these extents be held in the direct dim0 form? / every kernel of the
  // call will (so the loop may start on the packed Nyquist layout before the last view has been staged)
  bool would_be_direct(const int* kdims);
  // the direct dim0 leg is enabled and is the better leg for PSFs of this depth on a volume of these extents
  static bool direct_ok_for(int k0, int d0, int d1, int d2);
  // does a volume of this size keep its Nyquist bins packed in the DC column (when every PSF is in the direct form)?
  static bool packed_layout_for(size_t volume_bytes);
  void set_all_direct_hint(bool all) { packed_hint_ = all; }
  // the fused middle pass (mvn_mid_fused.hpp): will a kernel of these extents run through it? / every kernel of
  // the call will (pipelined calls: the loop starts before the last view has been staged)
  bool would_be_lines(const int* kdims);
  void set_all_lines_hint(bool all) { lines_hint_ = all; }
  // Halo mode (slabs of one volume): the slabs must all run the same middle - they exchange planes of its input -, so
  // whoever drives them decides for all of them (mvn_multi.cpp): on = the fused middle pass where this slab has it.
  // A hook set without this call keeps the three passes.
  void set_lines_in_halo_mode(bool on) { lines_override_ = on ? 1 : 0; }
  bool lines_in_use() const { return lines_; }
  // a cached engine starts every ABI call from a clean per-call state
  void begin_call() {
    pending_rows_ = nullptr;  // (a deferred pass of a call that failed half-way must never run on the next call's stacks)
    pipelined_ = false;
    quotient_guard_ = false;
    work_has_psi_spectrum_ = false;
    psi_spec_valid_ = false;
  }
  void sync();
  // PSF spectra re-used / prepared since process start (all engines)
  static long psf_cache_hits();
  static long psf_cache_misses();

 private:
  // true when slot spectrum i already belongs to this kernel; otherwise records the kernel
  bool psf_resident(ViewSlot& s, int i, const float* kernel, const int* kdims);
  void conv_pair(int v, double lambda, float min_value, int final_mode, int accumulate,
                 bool feed_next);
  void upload_volume(float* dst, const float* host, be::stream_t s);
  bool embedded_ = false;
  int host_dims_[3] = {0, 0, 0}, host_off_[3] = {0, 0, 0};
  float* embed_scratch_ = nullptr;  // one dense host-shaped stack: H2D lands here, a strided device copy embeds it
  size_t host_floats() const { return (size_t)host_dims_[0] * host_dims_[1] * host_dims_[2]; }
  void alloc_view(ViewSlot& s);
  // PSF spectrum of slot array `spec` from a device-resident kernel: forward transform, then (where
  // the plan wants it) the tile-contiguous re-ordering through `scratch` (one volume)
  void make_spectrum(const float* d_kernel, const int* kdims, float scale, float* spec, cfloat* nyq,
                     float* scratch, be::stream_t s);
  // Kernel i of slot s from its device-resident copy: the direct dim0 form (taps) where the PSF is thin
  // enough along dim0 (direct_form), the 3-D spectrum otherwise; buffers are allocated on first use.
  // `scratch` (one volume, or nullptr = allocate one with the staging scratch) serves the re-tiling of a
  // 3-D spectrum.
  void prepare_psf(ViewSlot& s, int i, const float* d_kernel, const int* kdims, float* scratch, bool staging,
                   be::stream_t st);
  bool direct_form(const int* kdims);
  Plan3D* taps_plan(int kd);
  // dim1 forward -> dim0 leg (direct or fused FFT) -> dim1 inverse on the work volume, with kernel i of s
  // `produce` (optional): the last-axis pass that produces this convolution's input, as a function of a row range
  // (row0, nrows; nrows < 0 = all): launched by middle() - in one piece, or boundary planes first (halo mode)
  typedef std::function<void(long, long)> RowsProducer;
  void middle(const ViewSlot& s, int i, Profiler* prof, SideStream* side, const RowsProducer* produce = nullptr);
  RowsProducer pending_rows_;  // a fused update + forward pass deferred to the next convolution (boundary-first order)
  void flush_pending_rows();
  bool boundary_first() const;
  // zcount > 0: only the output planes [zbeg, zbeg + zcount); first = false: a further launch of the same leg
  void dim0_conv(const ViewSlot& s, int i, const cfloat* in, const cfloat* in_nyq, cfloat* out, cfloat* out_nyq,
                 Profiler* prof, be::stream_t sn, int zbeg = 0, int zcount = 0, bool first = true);
  void ensure_work2();
  // Nyquist layout of the spectra between the last-axis passes of one call (mvn_dim0_direct.hpp, RowsParams::
  // nyq_packed): packed into the DC column when every kernel of every view is in the direct form, else the
  // separate plane.  Decided when a loop starts; a pipelined ABI call decides from the kernel extents up front.
  bool all_direct() const;
  void decide_layout();
  cfloat* wn() const { return packed_ ? nullptr : work_nyq_; }
  cfloat* pn() const { return packed_ ? nullptr : psi_spec_nyq_; }
  bool packed_ = false, packed_hint_ = false, packed_allowed_ = true;
  // the loops run their convolutions as last-axis pass -> fused middle pass -> last-axis pass on the line layout
  // (decided per iterate() call / per simultaneous step; halo and slab modes keep the three-pass middle)
  bool lines_capable_ = false, lines_ = false, lines_hint_ = false, lines_last_sweep_ = false;
  bool psi_spec_lines_ = false;  // the shared spectrum of psi (simultaneous steps) is in the line layout
  bool lines_forced_ = false;    // MVN_MID_FUSED=2: whenever the shape has the pass, however few planes
  bool lines_worth(int k0) const;
  void decide_lines();
  void mid_fused_conv(const ViewSlot& s, int i, Profiler* prof, int zbeg = 0, int zcount = 0);
  int lines_override_ = -1;  // halo mode: -1 = three-pass middle (a hook of unknown kind), 0 / 1 = the slabs' common decision
  halo_fn_t halo_fn_ = nullptr;
  void* halo_user_ = nullptr;
  bool halo_drain_ = true, halo_post_ = false, halo_split_ = false, halo_nyq_aware_ = false;
  int layout_override_ = -1;
  int halo_planes_ = 0;
  bool halo_ranged() const;
  unsigned* poison_ = nullptr;      // the word in use: poison_own_, or caller-owned memory (bind_poison)
  unsigned* poison_own_ = nullptr;  // 256 bytes: [0] the engine's own word, byte 64: table of the peers' words
  std::vector<unsigned*> poison_peers_;
  unsigned epoch_ = 0;        // direct dim0 legs enqueued so far = the epoch of the latest
  unsigned armed_epoch_ = 0;  // != 0: the convolution in flight ran a direct leg with this epoch
  // hands the report of the convolution's direct leg (if it had one) to the last-axis pass that ends it
  void arm(EpilogueParams& e);
  bool direct_enabled_ = true;
  int d0_stagger_ = 0;
  int direct_max_taps_ = MVN_D0_MAX_TAPS;
  long direct_min_plane_ = 131072, direct_min_items_ = 0;
  std::map<int, std::unique_ptr<Plan3D>> taps_plans_;  // (kd, d1, d2) plans of the tap arrays, private to the engine
  // second work volume: the direct dim0 leg is out of place, work_ and work2_ swap roles after it
  float* work2_ = nullptr;
  cfloat* work2_nyq_ = nullptr;
  float *work_alloc_ = nullptr, *work2_alloc_ = nullptr;  // what the two work volumes were allocated as
  bool spec_tiled_ = false;
  float* stage_spec_scratch_ = nullptr;  // owned by stage_scratch_
  void wait_staged(int v);
  int device_;
  std::shared_ptr<Plan3D> plan_;
  be::stream_t stream_ = nullptr;
  SideStream side_;
  // staging pipeline
  be::stream_t upload_stream_ = nullptr;
  std::vector<be::event_t> staged_ev_;
  std::vector<int> staged_;  // per view: 0 pending, 1 enqueued on the upload stream, -1 failed
  bool pipelined_ = false;
  std::mutex stage_mu_;
  std::condition_variable stage_cv_;
  std::vector<float*> stage_scratch_;
  float* psi_ = nullptr;
  float* work_ = nullptr;
  cfloat* work_nyq_ = nullptr;
  float* delta_ = nullptr;
  bool delta_external_ = false;
  // simultaneous mode: psi after the forward last-axis and dim1 passes, shared by all local views
  float* psi_spec_ = nullptr;
  cfloat* psi_spec_nyq_ = nullptr;
  bool work_has_psi_spectrum_ = false;  // work_ holds the last-axis transform of the current psi
  bool psi_spec_valid_ = false;         // psi_spec_ holds the last-axis + dim1 transform of the current psi
  // chunked simultaneous step: the last local view's final pass, parked by compute_delta_head
  EpilogueParams tail_epi_;
  bool tail_pending_ = false;
  int fed_n_ = 0, fed_ = 0;  // apply_delta_chunk(.., feed_next): chunks of the current round fed so far
  Profiler* tail_prof_ = nullptr;
  void chunk_planes(int c, int n, int* z0, int* nz) const;
  bool quotient_guard_ = false;
  long pair_counter_ = 0;
  // one steady-state sweep over all views captured as a graph (small, launch-bound volumes);
  // valid for the parameters it was captured with, buffers never move during an engine's life
  be::graph_exec_t sweep_graph_ = nullptr;
  double graph_lambda_ = 0;
  float graph_min_ = 0;
  bool graph_guard_ = false;
  // PSF preparations and layout changes so far / at capture: the graph bakes in the PSF buffers, their form
  // (taps or 3-D spectrum), depth and the Nyquist layout
  unsigned long graph_gen_ = 0, graph_captured_gen_ = 0;
  const float* graph_work_ = nullptr;  // which of the two work volumes was current when the sweep was captured
  std::vector<ViewSlot> views_;
  Profiler prof_;
};

// ---------------------------------------------------------------------------------------------
// Slab-decomposed engine: the reference's SEQUENTIAL sweep over views on several GPUs (SURVEY.md
// 8e row 3).  Rank r of P keeps planes [r d0/P, (r+1) d0/P) of psi and of every view and weight
// stack.  Last-axis and dim1 passes are local to a plane; the dim0 pass needs whole lines along
// d0, so each convolution exchanges the half-transformed slab twice:
//     W [d0/P][d1][C]  --pack-->  A [P][d0/P][d1/P][C]  ==all-to-all==>  B [d0][d1/P][C]
//     dim0 FWD * PSF * INV on B (PSF spectra kept in the same transposed layout)
//     B, seen as [P][d0/P][d1/P][C]  ==all-to-all==>  A  --unpack-->  W
// The exchanges themselves are the caller's (torch.distributed / RCCL on buffers it binds);
// everything between them is the single-GPU engine's kernels on two plans, (d0/P, d1, d2) for the
// plane-local passes and (d0, d1/P, d2) for the dim0 pass, so the arithmetic is the sequential
// sweep's, pass for pass.  Requires d0 % P == 0, d1 % P == 0, d0/P >= 2, d1/P >= 2.
// ---------------------------------------------------------------------------------------------
class SlabEngine {
 public:
  SlabEngine(int device, const shape_t& dims, int nranks, int rank, int num_views);
  ~SlabEngine();
  SlabEngine(const SlabEngine&) = delete;
  SlabEngine& operator=(const SlabEngine&) = delete;

  int device() const { return device_; }
  int nranks() const { return P_; }
  int rank() const { return rank_; }
  const Layout& slab_layout() const { return planA_->L; }  // (d0/P, d1, d2)
  size_t main_floats() const { return planA_->L.real_floats(); }    // per exchange buffer
  size_t nyq_floats() const { return 2 * planA_->L.nyq_cplx(); }

  // image / weights: this rank's planes, dense [d0/P][d1][d2]; kernels: whole
  void set_view(int v, const float* image_slab, const float* weights_slab, const float* kernel1,
                const int* k1dims, const float* kernel2, const int* k2dims);
  void set_psi(const float* psi_slab);
  void get_psi(float* psi_slab);
  // exchange buffers (device memory): A is packed / unpacked here, B is what the dim0 pass
  // works on.  nullptr arguments return to engine-owned buffers.
  void bind_buffers(float* a_main, float* b_main, float* a_nyq, float* b_nyq);
  float* a_main() { return a_main_; }
  float* b_main() { return b_main_; }
  float* a_nyq() { return (float*)a_nyq_; }
  float* b_nyq() { return (float*)b_nyq_; }

  // one view update = for conv in {0, 1}: pack, [A -> B all-to-all], mid, [B -> A all-to-all],
  // unpack.  All steps are asynchronous on stream(); sync() before handing a buffer to a
  // collective on another stream.
  void begin_sweeps();  // psi may have been replaced: forget the cached last-axis spectrum
  void step_pack(int v, int conv);
  void step_mid(int v, int conv);
  void step_unpack(int v, int conv, double lambda, float min_value, bool feed_next);
  void sync();
  be::stream_t stream() const { return stream_; }

 private:
  void upload_slab(float* dst, const float* host);
  int device_, P_, rank_;
  shape_t dims_;
  std::shared_ptr<Plan3D> planA_, planB_;
  be::stream_t stream_ = nullptr;
  float* psi_ = nullptr;
  float* work_ = nullptr;
  cfloat* work_nyq_ = nullptr;
  float *a_main_ = nullptr, *b_main_ = nullptr;
  cfloat *a_nyq_ = nullptr, *b_nyq_ = nullptr;
  bool external_ = false;
  float *own_a_ = nullptr, *own_b_ = nullptr;
  cfloat *own_an_ = nullptr, *own_bn_ = nullptr;
  bool work_has_psi_spectrum_ = false;
  struct SlabView {
    float* image = nullptr;
    float* weights = nullptr;
    float* spec[2] = {nullptr, nullptr};   // transposed slabs [d0][d1/P][C]
    cfloat* nyq[2] = {nullptr, nullptr};   // [d0][d1/P]
    bool set = false;
  };
  std::vector<SlabView> views_;
};

}  // namespace mvn
