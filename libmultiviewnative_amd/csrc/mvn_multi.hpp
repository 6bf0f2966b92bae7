// mvn_multi.hpp -- one volume on several devices inside ONE inplace_gpu_deconvolve call (MVN_DEVICES).
//
// The reference's GPU entry (inc/multiviewnative.h:66-67, src/multiviewnative.cu:89-142) drives one device.  Here
// the (padded) volume of a call is cut into slabs of dim0 planes, one per entry of MVN_DEVICES, and swept in the
// REFERENCE's view order (Gauss-Seidel, src/multiviewnative.cpp:194-227) - the exact arithmetic of the one-device
// path, not the Jacobi variant of view sharding.  What makes that possible is the direct dim0 leg
// (mvn_dim0_direct.hpp): the last-axis and dim1 passes never leave a plane and the dim0 leg of a slab needs
// h = K / 2 planes of either neighbour, so per convolution 2 h planes per neighbour pair cross the links
// (31.5 MB at 512^3, K = 31) instead of the volume.
//
// Every slab is an ORDINARY resident engine on its planes plus h halo planes either side (which it never computes:
// Engine::set_halo_planes), driven by its own host thread; the engines' halo hook (Engine::set_halo_hook) is where the
// slabs meet:
//   before a dim0 leg   record "my dim1 pass is done"; [host barrier]; my HALO stream waits for both neighbours' records
//                       and PULLS their boundary planes into my halo planes (peer copies), while my engine's stream
//                       already runs the leg on the planes that do not depend on them
//   before its 2nd part my engine's stream waits for those copies (and for the neighbours' copies of the previous
//                       convolution, whose source planes this part overwrites); then the planes next to the halos
//   behind a dim0 leg   record "my leg is done"; [host barrier]; my stream waits for every other slab's record
//                       before the last-axis pass, which reads the poison word every leg reports non-finite
//                       inputs to (through peer access: Dim0DirectParams::poison_peers)
// Nothing waits on the host for the device; the host barriers only order the RECORDING of an event before the
// enqueueing of the waits on it.  No torch, no RCCL: events and peer copies.
#pragma once

#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <vector>

#include "../../include/multiviewnative.h"
#include "mvn_engine.hpp"

namespace mvn {

// reusable barrier for the slab threads; abort() makes every present and future wait() throw, so that a slab
// that failed does not leave the others waiting for it
class HostBarrier {
 public:
  explicit HostBarrier(int n) : n_(n) {}
  void wait();
  void abort();
  void reset(int n);

 private:
  std::mutex mu_;
  std::condition_variable cv_;
  int n_, count_ = 0;
  unsigned long generation_ = 0;
  bool aborted_ = false;
};

// devices of MVN_DEVICES ("0,1,2,3"; an entry may repeat: "0,0" runs two slabs on one device - the rehearsal a
// one-GPU box allows); empty when the variable is unset, has fewer than two entries or names a device that
// does not exist
std::vector<int> multi_devices_from_env();

class HaloGroup {
 public:
  // ext = extents of the (padded) volume, h = halo planes = (deepest PSF) / 2, V = views
  HaloGroup(const std::vector<int>& devices, const shape_t& ext, int h, int V);
  ~HaloGroup();
  HaloGroup(const HaloGroup&) = delete;
  HaloGroup& operator=(const HaloGroup&) = delete;

  bool matches(const std::vector<int>& devices, const shape_t& ext, int h, int V) const {
    return devices == devices_ && ext == ext_ && h == h_ && V == V_;
  }
  // can a volume of these extents be cut over this many slabs at all (planes per slab >= h >= 1, even d2, the
  // packed layout's dim0 limit)?
  static bool feasible(int nslabs, const shape_t& ext, int h);
  // every kernel of the call will be held in the direct form by every slab's engine
  bool all_direct(const workspace& input);
  // Host stacks of extents `dims` sit at offset `off` inside the volume (pad policy of the ABI):
  //   load     every slab uploads its planes of psi and of every view (one host thread per slab)
  //   iterate  `iterations` sweeps in the reference's view order; blocking; returns the wall time in ms between the
  //            moment every slab is ready and the moment the last one has finished
  //   fetch    every slab writes its planes of psi back
  //   run      the three in a row = one inplace_gpu_deconvolve call
  void load(const imageType* psi, const workspace& input, const shape_t& dims, const int off[3], bool quotient_guard);
  double iterate(int iterations, double lambda, float min_value);
  void fetch(imageType* psi);
  void run(imageType* psi, const workspace& input, const shape_t& dims, const int off[3], bool quotient_guard);
  const shape_t& extents() const { return ext_; }
  int num_views() const { return V_; }
  int slabs() const { return (int)slabs_.size(); }
  const std::vector<int>& devices() const { return devices_; }

 private:
  struct Slab {
    HaloGroup* group = nullptr;
    int index = 0, dev = 0;
    int z0 = 0, nz = 0;  // planes [z0, z0 + nz) of the volume; the engine holds nz + 2 h planes
    std::unique_ptr<Engine> eng;
    be::event_t e_fwd[2] = {nullptr, nullptr};   // my dim1 pass before a leg is done (parity of the convolution count)
    // my lower [.][0] / upper [.][1] halo planes have arrived = that neighbour's planes have been read
    be::event_t e_copy[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    be::event_t e_leg[2] = {nullptr, nullptr};   // my leg (and the dim1 pass behind it) is done
    // the peer copies run here, beside the interior part of the leg: one stream per side (two neighbours, two links)
    be::stream_t halo_stream[2] = {nullptr, nullptr};
    void* spectrum = nullptr;                    // input of the leg in flight (the engine's work volume)
    void* spectrum_nyq = nullptr;                // ... and its Nyquist plane (split layout), nullptr when packed
    unsigned long convs = 0;
    int host_a = 0, host_b = 0, embed_z = 0;     // planes [host_a, host_b) of the host stacks, at plane embed_z of the engine
  };
  static void hook(void* user, void* spectrum, int view, int conv);
  void before_leg(Slab& s, void* spectrum);
  void before_boundary(Slab& s);
  void behind_leg(Slab& s);
  void on_every_slab(const std::function<void(Slab&)>& body);
  shape_t host_dims_ = {{0, 0, 0}};
  bool loaded_ = false;

  std::vector<int> devices_;
  shape_t ext_;
  int h_, V_;
  std::vector<Slab> slabs_;
  HostBarrier barrier_;
};

}  // namespace mvn
