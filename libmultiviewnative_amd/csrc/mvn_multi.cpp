// mvn_multi.cpp -- slabs of one volume on several devices, swept in the reference's order (see mvn_multi.hpp).
#include "mvn_multi.hpp"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <stdexcept>
#include <string>
#include <thread>

namespace mvn {

// ---- HostBarrier ----------------------------------------------------------------------------------------------
void HostBarrier::wait() {
  std::unique_lock<std::mutex> lk(mu_);
  if (aborted_) throw std::runtime_error("mvn: another slab of the call failed");
  const unsigned long gen = generation_;
  if (++count_ == n_) {
    count_ = 0;
    ++generation_;
    cv_.notify_all();
    return;
  }
  cv_.wait(lk, [&] { return generation_ != gen || aborted_; });
  if (generation_ == gen) throw std::runtime_error("mvn: another slab of the call failed");
}

void HostBarrier::abort() {
  {
    std::lock_guard<std::mutex> lk(mu_);
    aborted_ = true;
  }
  cv_.notify_all();
}

void HostBarrier::reset(int n) {
  std::lock_guard<std::mutex> lk(mu_);
  n_ = n;
  count_ = 0;
  aborted_ = false;
}

std::vector<int> multi_devices_from_env() {
  std::vector<int> devs;
  const char* e = std::getenv("MVN_DEVICES");
  if (!e || !*e) return devs;
  const int have = be::device_count();
  const char* p = e;
  while (*p) {
    char* end = nullptr;
    const long v = std::strtol(p, &end, 10);
    if (end == p || v < 0 || v >= have) return std::vector<int>();  // not a list of existing devices: ignored
    devs.push_back((int)v);
    p = end;
    while (*p == ',' || *p == ' ') ++p;
  }
  if (devs.size() < 2) devs.clear();
  return devs;
}

// ---- HaloGroup ------------------------------------------------------------------------------------------------
static void slab_range(int r, int P, int d0, int* z0, int* nz) {
  const int a = (int)((long)r * d0 / P), b = (int)((long)(r + 1) * d0 / P);
  *z0 = a;
  *nz = b - a;
}

bool HaloGroup::feasible(int P, const shape_t& ext, int h) {
  // (one slab is its own neighbour both ways: the cyclic exchange with itself - what the mode costs on one device)
  if (P < 1 || P > MVN_D0_MAX_PEERS + 1 || h < 1 || ext[2] % 2 != 0) return false;
  for (int r = 0; r < P; ++r) {
    int z0 = 0, nz = 0;
    slab_range(r, P, ext[0], &z0, &nz);
    if (nz < h || !mvn_dim0_packed_possible(nz + 2 * h)) return false;
  }
  return true;
}

HaloGroup::HaloGroup(const std::vector<int>& devices, const shape_t& ext, int h, int V)
    : devices_(devices), ext_(ext), h_(h), V_(V), barrier_((int)devices.size()) {
  const int P = (int)devices.size();
  if (!feasible(P, ext, h)) throw std::invalid_argument("mvn: this volume cannot be cut over the devices asked for");
  for (int a = 0; a < P; ++a)
    for (int b = 0; b < P; ++b) be::enable_peer_access(devices[(size_t)a], devices[(size_t)b]);
  slabs_.resize((size_t)P);
  for (int r = 0; r < P; ++r) {
    Slab& s = slabs_[(size_t)r];
    s.group = this;
    s.index = r;
    s.dev = devices[(size_t)r];
    slab_range(r, P, ext[0], &s.z0, &s.nz);
    be::set_device(s.dev);
    const shape_t shape = {{s.nz + 2 * h, ext[1], ext[2]}};
    s.eng.reset(new Engine(s.dev, shape, V));
    for (int i = 0; i < 2; ++i) {
      s.e_fwd[i] = be::event_create_sync();
      s.e_copy[i][0] = be::event_create_sync();
      s.e_copy[i][1] = be::event_create_sync();
      s.e_leg[i] = be::event_create_sync();
      s.halo_stream[i] = be::stream_create();
    }
    s.eng->set_halo_hook(&HaloGroup::hook, &s, /*drain=*/false, /*post=*/true);
    s.eng->set_halo_planes(h, /*split=*/true);
    // the slabs run the layout the WHOLE volume would have on one device (packed Nyquist bins up to 256 MB): the
    // same arithmetic as the one-device call, whatever the slab size
    s.eng->set_halo_nyq_aware(Engine::packed_layout_for(Layout(ext[0], ext[1], ext[2]).real_floats() * sizeof(float)) ? 1 : 0);
  }
  // a leg that meets a non-finite input tells every slab (Dim0DirectParams::poison_peers)
  for (int a = 0; a < P; ++a)
    for (int b = 0; b < P; ++b)
      if (a != b) slabs_[(size_t)a].eng->add_poison_peer(slabs_[(size_t)b].eng->poison_ptr());
}

HaloGroup::~HaloGroup() {
  for (size_t r = 0; r < slabs_.size(); ++r) {
    Slab& s = slabs_[r];
    try {
      be::set_device(s.dev);
      if (s.eng) s.eng->sync();
    } catch (...) {
    }
  }
  for (size_t r = 0; r < slabs_.size(); ++r) {
    Slab& s = slabs_[r];
    try {
      be::set_device(s.dev);
    } catch (...) {
    }
    s.eng.reset();
    for (int i = 0; i < 2; ++i) {
      try {
        if (s.halo_stream[i]) be::stream_sync(s.halo_stream[i]);
      } catch (...) {
      }
      be::stream_destroy(s.halo_stream[i]);
      be::event_destroy(s.e_fwd[i]);
      be::event_destroy(s.e_copy[i][0]);
      be::event_destroy(s.e_copy[i][1]);
      be::event_destroy(s.e_leg[i]);
    }
  }
}

bool HaloGroup::all_direct(const workspace& input) {
  for (size_t r = 0; r < slabs_.size(); ++r) {
    be::set_device(slabs_[r].dev);
    for (int v = 0; v < input.num_views_; ++v)
      if (!slabs_[r].eng->would_be_direct(input.data_[v].kernel1_dims_) ||
          !slabs_[r].eng->would_be_direct(input.data_[v].kernel2_dims_))
        return false;
  }
  return true;
}

void HaloGroup::hook(void* user, void* spectrum, int /*view*/, int conv) {
  Slab& s = *static_cast<Slab*>(user);
  if (conv < 2)
    s.group->before_leg(s, spectrum);
  else if (conv < 4)
    s.group->behind_leg(s);
  else
    s.group->before_boundary(s);
}

// The neighbours' boundary planes become my halo planes: pulled by peer copies on my HALO stream, beside the part
// of the leg that does not need them (the engine launches it right after this call).  The host barrier only makes
// sure that an event has been RECORDED (by its slab's thread) before a wait on it is enqueued.
void HaloGroup::before_leg(Slab& s, void* spectrum) {
  const int P = (int)slabs_.size();
  const int par = (int)(s.convs & 1);
  s.spectrum = spectrum;
  s.spectrum_nyq = s.eng->leg_input_nyq();  // (split layout: the Nyquist plane has halo planes of its own)
  be::event_record(s.e_fwd[par], s.eng->stream());
  barrier_.wait();
  Slab& lo = slabs_[(size_t)((s.index + P - 1) % P)];
  Slab& up = slabs_[(size_t)((s.index + 1) % P)];
  const Layout& L = s.eng->layout();
  const size_t pb = (size_t)L.d1 * (size_t)L.C * sizeof(cfloat);  // one plane of the spectrum
  const size_t nb = (size_t)L.d1 * sizeof(cfloat);                // one plane of the Nyquist plane: a line of d1 bins
  char* mine = static_cast<char*>(spectrum);
  char* mine_n = static_cast<char*>(s.spectrum_nyq);
  // one halo stream per side: the two neighbours sit behind two different links, their planes travel side by side
  // (two slabs: the one neighbour is both, one link, one stream - a process's streams share 4 hardware queues)
  for (int side = 0; side < 2; ++side) {
    Slab& nb_slab = side == 0 ? lo : up;
    be::stream_t hs = s.halo_stream[&lo == &up ? 0 : side];
    be::stream_wait_event(hs, s.e_fwd[par]);  // (my halo planes are free: every earlier reader is behind this event)
    be::stream_wait_event(hs, nb_slab.e_fwd[par]);
    // side 0: planes [0, h) <- the lower neighbour's last h own planes [nz, nz + h) of its extended slab
    // side 1: planes [nz + h, nz + 2 h) <- the upper neighbour's first h own planes [h, 2 h)
    const size_t dst = side == 0 ? 0 : (size_t)(s.nz + h_), src = side == 0 ? (size_t)nb_slab.nz : (size_t)h_;
    be::copy_peer(mine + dst * pb, s.dev, static_cast<const char*>(nb_slab.spectrum) + src * pb, nb_slab.dev,
                  (size_t)h_ * pb, hs);
    if (mine_n)
      be::copy_peer(mine_n + dst * nb, s.dev, static_cast<const char*>(nb_slab.spectrum_nyq) + src * nb, nb_slab.dev,
                    (size_t)h_ * nb, hs);
    be::event_record(s.e_copy[par][side], hs);
  }
}

// In front of the part of the leg that reads the halo planes - and WRITES, in the engine's other work volume, the
// planes the neighbours read during the PREVIOUS convolution (the two work volumes swap roles at every leg): my
// halos have arrived, and the neighbours' copies of the previous convolution have completed.  A whole convolution
// lies between those copies and this point, so the second wait never stalls in a balanced run.
void HaloGroup::before_boundary(Slab& s) {
  const int P = (int)slabs_.size();
  const int par = (int)(s.convs & 1);
  be::stream_t st = s.eng->stream();
  be::stream_wait_event(st, s.e_copy[par][0]);
  be::stream_wait_event(st, s.e_copy[par][1]);
  if (s.convs > 0) {
    Slab& lo = slabs_[(size_t)((s.index + P - 1) % P)];
    Slab& up = slabs_[(size_t)((s.index + 1) % P)];
    be::stream_wait_event(st, lo.e_copy[par ^ 1][1]);  // the lower neighbour's UPPER halo came from my first own planes
    be::stream_wait_event(st, up.e_copy[par ^ 1][0]);  // the upper neighbour's LOWER halo from my last
  }
}

// Every slab's leg has reported to every slab's poison word before any slab's last-axis pass reads its own.
void HaloGroup::behind_leg(Slab& s) {
  const int par = (int)(s.convs & 1);
  be::stream_t st = s.eng->stream();
  be::event_record(s.e_leg[par], st);
  barrier_.wait();
  for (size_t r = 0; r < slabs_.size(); ++r)
    if ((int)r != s.index) be::stream_wait_event(st, slabs_[r].e_leg[par]);
  ++s.convs;
}

// One host thread per slab runs `body`; a slab that throws makes the barrier throw in the others; afterwards every
// stream is drained and the error of the slab that failed FIRST is re-thrown.
void HaloGroup::on_every_slab(const std::function<void(Slab&)>& body) {
  const int P = (int)slabs_.size();
  barrier_.reset(P);
  std::vector<std::exception_ptr> errs((size_t)P);
  auto run = [&](int r) {
    try {
      be::set_device(slabs_[(size_t)r].dev);
      body(slabs_[(size_t)r]);
    } catch (...) {
      errs[(size_t)r] = std::current_exception();
      barrier_.abort();
    }
  };
  std::vector<std::thread> threads;
  for (int r = 1; r < P; ++r) threads.emplace_back(run, r);
  run(0);
  for (size_t i = 0; i < threads.size(); ++i) threads[i].join();
  bool failed = false;
  for (int r = 0; r < P; ++r) failed = failed || errs[(size_t)r];
  if (!failed) return;
  for (int r = 0; r < P; ++r) {  // (a slab that failed early may have left the others with work in flight)
    try {
      be::set_device(slabs_[(size_t)r].dev);
      slabs_[(size_t)r].eng->sync();
      be::stream_sync(slabs_[(size_t)r].halo_stream[0]);
      be::stream_sync(slabs_[(size_t)r].halo_stream[1]);
    } catch (...) {
    }
  }
  for (int r = 0; r < P; ++r) {  // the first failure is the one whose message is not "another slab failed"
    if (!errs[(size_t)r]) continue;
    try {
      std::rethrow_exception(errs[(size_t)r]);
    } catch (const std::exception& e) {
      if (std::strstr(e.what(), "another slab") == nullptr) throw;
    }
  }
  for (int r = 0; r < P; ++r)
    if (errs[(size_t)r]) std::rethrow_exception(errs[(size_t)r]);
}

void HaloGroup::load(const imageType* psi, const workspace& input, const shape_t& dims, const int off[3],
                     bool quotient_guard) {
  const int P = (int)slabs_.size();
  if (input.num_views_ != V_) throw std::invalid_argument("mvn: view count of the group");
  for (int r = 0; r < P; ++r) {
    Slab& s = slabs_[(size_t)r];
    // the host planes this slab holds: [lo, hi) of the volume = [ha, hb) of the host stacks
    const int lo = std::max(s.z0, off[0]), hi = std::min(s.z0 + s.nz, off[0] + dims[0]);
    if (hi <= lo) throw std::invalid_argument("mvn: a slab of the padded volume holds no plane of the stacks");
    s.host_a = lo - off[0];
    s.host_b = hi - off[0];
    s.embed_z = h_ + lo - s.z0;
  }
  host_dims_ = dims;
  on_every_slab([&](Slab& s) {
    Engine& e = *s.eng;
    e.begin_call();
    const int sd[3] = {s.host_b - s.host_a, dims[1], dims[2]};
    const int so[3] = {s.embed_z, off[1], off[2]};
    e.set_embedding(sd, so);
    e.set_quotient_guard(quotient_guard);
    const size_t first = (size_t)s.host_a * (size_t)dims[1] * (size_t)dims[2];
    for (int v = 0; v < V_; ++v) {
      const view_data& d = input.data_[v];
      e.set_view(v, d.image_ + first, d.weights_ + first, d.kernel1_, d.kernel1_dims_, d.kernel2_, d.kernel2_dims_);
    }
    e.set_psi(psi + first);
  });
  // the slabs exchange planes of the middle's input: all of them take the fused middle pass (mvn_mid_fused.hpp), or none
  bool lines = true;
  for (int r = 0; r < P && lines; ++r) {
    be::set_device(slabs_[(size_t)r].dev);
    for (int v = 0; v < V_ && lines; ++v)
      lines = slabs_[(size_t)r].eng->would_be_lines(input.data_[v].kernel1_dims_) &&
              slabs_[(size_t)r].eng->would_be_lines(input.data_[v].kernel2_dims_);
  }
  for (int r = 0; r < P; ++r) slabs_[(size_t)r].eng->set_lines_in_halo_mode(lines);
  loaded_ = true;
}

double HaloGroup::iterate(int iterations, double lambda, float min_value) {
  if (!loaded_) throw std::logic_error("mvn: group without stacks");
  if (iterations < 1) return 0.;
  const int P = (int)slabs_.size();
  unsigned epoch = 0;  // the slabs count their legs together: the same leg has the same epoch everywhere
  for (int r = 0; r < P; ++r) epoch = std::max(epoch, slabs_[(size_t)r].eng->poison_epoch());
  for (int r = 0; r < P; ++r) {
    slabs_[(size_t)r].eng->set_poison_epoch(epoch);
    slabs_[(size_t)r].convs = 0;
  }
  std::vector<double> ms((size_t)P, 0.);
  on_every_slab([&](Slab& s) {
    barrier_.wait();  // (every slab is here, its engine idle: the first exchange reads the neighbours' volumes)
    const auto t0 = std::chrono::steady_clock::now();
    s.eng->iterate(iterations, lambda, min_value);
    s.eng->sync();
    barrier_.wait();  // the sweeps are over when the LAST slab has finished
    ms[(size_t)s.index] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  });
  return *std::max_element(ms.begin(), ms.end());
}

void HaloGroup::fetch(imageType* psi) {
  if (!loaded_) throw std::logic_error("mvn: group without stacks");
  on_every_slab([&](Slab& s) {
    s.eng->get_psi(psi + (size_t)s.host_a * (size_t)host_dims_[1] * (size_t)host_dims_[2]);
  });
}

void HaloGroup::run(imageType* psi, const workspace& input, const shape_t& dims, const int off[3],
                    bool quotient_guard) {
  load(psi, input, dims, off, quotient_guard);
  iterate(input.num_iterations_, input.lambda_, input.minValue_);
  fetch(psi);  // psi is written only once EVERY slab has finished: a failed call leaves it untouched
}

}  // namespace mvn
