"""View-sharded ("simultaneous update") RL driver: one process per GPU, one all-reduce of the
per-view correction per iteration (the reference has no multi-GPU code).

Every rank holds a replica of psi and the stacks of its own views.  Per iteration each rank
computes, from the same psi_k, ``delta = sum_{v in my views} w_v (next_v - psi_k)`` on its GPU,
the deltas are summed over ranks with ONE all-reduce of the psi-sized buffer
(``torch.distributed``: backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests), and
``psi_{k+1} = psi_k + delta`` is applied on every rank.  For a single view this equals the
reference's sequential sweep; for several views it is its Jacobi counterpart, and the parity
oracle is the CPU restatement run in the same mode.

The all-reduce is issued in CHUNKS of dim0 planes and runs under the compute: the correction
leaves the last pass of the last local view chunk by chunk (``compute_delta_chunk``), each
chunk's collective starts as soon as its kernel has been enqueued, and ``psi += delta`` plus the
next iteration's forward last-axis and dim1 passes (all local to a dim0 plane) follow chunk by
chunk behind the collectives that have completed (``apply_delta_chunk``).  What stays exposed is
the all-reduce of the first chunks minus the compute it overlaps with.
"""


def view_partition(num_views, world_size, rank):
    """Contiguous, balanced shard of view indices for `rank` (first ranks take the remainder)."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    base, rem = divmod(num_views, world_size)
    begin = rank * base + min(rank, rem)
    return list(range(begin, begin + base + (1 if rank < rem else 0)))


class _Ordering:
    """How a collective on a buffer the engine writes / reads is ordered against the engine's own
    HIP stream.

    * ``stream`` given (a ``torch.cuda.ExternalStream`` wrapping ``engine.stream()``): collectives
      are issued with that stream current, so the communication library's stream waits for the
      engine work enqueued so far, and ``work.wait()`` makes the engine stream wait for the
      collective -- no host synchronisation anywhere, compute keeps being enqueued meanwhile.
    * no stream: host-synchronised fallback, valid for every backend and for host tensors: the
      engine is drained before the collective reads the buffer and the collective is complete
      (``work.wait()`` plus, for device tensors, a synchronise of torch's current stream) before
      the engine touches it again.
    """

    def __init__(self, engine, stream=None, after_collective=None):
        self.engine = engine
        self.stream = stream
        self.after_collective = after_collective

    def _ctx(self):
        import contextlib
        if self.stream is None:
            return contextlib.nullcontext()
        import torch
        return torch.cuda.stream(self.stream)

    def start(self, issue):
        """`issue()` starts the collective with async_op=True and returns its work handle."""
        if self.stream is None:
            self.engine.sync()  # the buffer must be complete before the collective reads it
        with self._ctx():
            return issue()

    def finish(self, work, tensor):
        with self._ctx():
            if work is not None:
                work.wait()
            if self.stream is None and getattr(tensor, "is_cuda", False):
                import torch
                torch.cuda.current_stream(tensor.device).synchronize()
        if self.after_collective is not None:
            self.after_collective()


class SimultaneousDriver:
    """Runs the sharded iteration loop on an engine (``native.EngineHandle`` or anything with
    ``compute_delta_head / compute_delta_chunk / apply_delta_chunk / delta_chunks /
    delta_chunk_range / sync``).

    `delta` is the torch tensor aliasing the engine's delta buffer (device memory for the HIP
    engine, bound with ``bind_delta``); `dist` is ``torch.distributed`` (None: single rank);
    `chunks` the wanted number of dim0 chunks (the engine may allow fewer); `stream` see
    :class:`_Ordering`.
    """

    def __init__(self, engine, delta, dist=None, after_collective=None, chunks=4, stream=None,
                 force_collective=False):
        self.engine = engine
        self.delta = delta
        self.dist = dist
        self.order = _Ordering(engine, stream, after_collective)
        # force_collective: issue the (trivial) all-reduce even on one rank -- rehearsals of the
        # multi-rank code path on a one-GPU box
        self.multi = dist is not None and (dist.get_world_size() > 1 or force_collective)
        self.n = engine.delta_chunks(chunks if self.multi else 1)
        self.parts = []
        if self.multi:
            flat = delta.view(-1)
            for c in range(self.n):
                first, count = engine.delta_chunk_range(c, self.n)
                self.parts.append(flat[first:first + count])

    def step(self, lambda_, min_value, feed_next=False):
        eng, n = self.engine, self.n
        eng.compute_delta_head(lambda_, min_value)
        works = []
        for c in range(n):
            eng.compute_delta_chunk(c, n)
            if self.multi:
                part = self.parts[c]
                works.append(self.order.start(
                    lambda: self.dist.all_reduce(part, op=self.dist.ReduceOp.SUM, async_op=True)))
        for c in range(n):
            if self.multi:
                self.order.finish(works[c], self.parts[c])
            eng.apply_delta_chunk(c, n, feed_next)

    def run(self, iterations, lambda_, min_value):
        for it in range(iterations):
            self.step(lambda_, min_value, feed_next=it + 1 < iterations)
        self.engine.sync()


class SlabDriver:
    """The reference's sequential view-after-view sweep on several GPUs, volume split in slabs of
    planes: exact single-GPU arithmetic, four all-to-all exchanges per
    (view, iteration).

    `engine` is a ``native.SlabHandle`` (or anything with pack / mid / unpack / sync);
    `a_main, b_main, a_nyq, b_nyq` are torch tensors aliasing its exchange buffers (``a_nyq`` /
    ``b_nyq`` may be None for odd d2); `dist` is ``torch.distributed``; `stream` see
    :class:`_Ordering`.
    """

    def __init__(self, engine, a_main, b_main, a_nyq, b_nyq, dist, after_collective=None, stream=None):
        self.engine = engine
        self.a_main, self.b_main, self.a_nyq, self.b_nyq = a_main, b_main, a_nyq, b_nyq
        self.dist = dist
        self.order = _Ordering(engine, stream, after_collective)

    def _exchange(self, src_main, dst_main, src_nyq, dst_nyq):
        w1 = self.order.start(lambda: self.dist.all_to_all_single(dst_main, src_main, async_op=True))
        w2 = None
        if src_nyq is not None:
            w2 = self.order.start(lambda: self.dist.all_to_all_single(dst_nyq, src_nyq, async_op=True))
        self.order.finish(w1, dst_main)
        if w2 is not None:
            self.order.finish(w2, dst_nyq)

    def view_update(self, v, lambda_, min_value, feed_next):
        for conv in (0, 1):
            self.engine.pack(v, conv)
            self._exchange(self.a_main, self.b_main, self.a_nyq, self.b_nyq)
            self.engine.mid(v, conv)
            self._exchange(self.b_main, self.a_main, self.b_nyq, self.a_nyq)
            self.engine.unpack(v, conv, lambda_, min_value, feed_next)

    def run(self, iterations, num_views, lambda_, min_value):
        for it in range(iterations):
            for v in range(num_views):
                last = it == iterations - 1 and v == num_views - 1
                self.view_update(v, lambda_, min_value, not last)
        self.engine.sync()


class HaloSlabDriver:
    """One volume cut into dim0 slabs over the ranks, swept in the REFERENCE's view order (Gauss-Seidel,
    src/multiviewnative.cpp:194-227): exact parity with the single-GPU path on any rank count.

    Possible because the dim0 leg of a convolution is a direct K-tap convolution along dim0 in the (dim1, dim2)
    spectral domain (``mvn_dim0_direct.hpp``): the last-axis and dim1 passes are local to a plane, and the dim0
    leg of this rank's planes needs only the h = K // 2 planes either side of them - a halo exchange with the
    two neighbouring ranks (cyclically: the reference's convolution is cyclic) of 2 h planes per convolution
    instead of an all-reduce of the volume (view sharding) or two transposes of it (``SlabDriver``).

    The engine is an ordinary resident engine on the EXTENDED slab, ``nz + 2 h`` planes: halo planes carry
    image 1 / weights 0 and whatever the passes leave in them; before every dim0 leg the engine calls back
    (``mvn_engine_set_halo_hook``) and the planes of the neighbours overwrite the halo planes of the leg's input.
    Every PSF must have at most 33 planes along dim0 and ``nz >= h``.  Built and parity-tested on gloo ranks
    (CPU emulation) and on one GPU; un-measured on several GPUs.  With device exchange buffers everything the
    exchange does is ordered on the engine's stream (the host never waits inside a sweep); with host buffers
    (gloo) it is synchronous.

    Non-finite values: the reference's FFT convolution turns one Inf / NaN voxel into a volume of NaN.  A slab's
    direct leg reports such an input in its engine's poison word (``mvn_engine_api.h``); behind every leg the
    hook is called a second time and the ranks MAX-reduce their words (4 bytes), so that every slab's last-axis
    pass floods its part of the volume.

    A failed exchange cannot raise through the engine's C frames: the rank records the error, keeps taking part
    in the remaining exchanges (so that its peers do not block in a receive that never gets matched), and
    ``run()`` raises on EVERY rank once the sweep is over (the ranks all-reduce an error flag).
    """

    def __init__(self, binding, full_shape, num_views, max_psf_depth, dist=None, rank=0, world=1, device=0,
                 torch_device=None, force_collective=False):
        import torch
        self.torch = torch
        # force_collective: one rank that still exchanges through isend / irecv with itself and reduces its poison
        # word - a rehearsal of the multi-rank code path (device tensors, stream ordering) on a one-GPU box
        self.dist = dist if (world > 1 or (force_collective and dist is not None)) else None
        self.rank, self.world = rank, world
        d0, d1, d2 = (int(x) for x in full_shape)
        if d0 % world:
            raise ValueError("halo mode: dim0 must divide by the rank count")
        self.nz = d0 // world
        self.h = int(max_psf_depth) // 2
        if self.h < 1:
            self.h = 1
        if self.nz < self.h:
            raise ValueError("halo mode: fewer planes per rank than halo planes")
        self.z0 = rank * self.nz
        self.ext_shape = (self.nz + 2 * self.h, d1, d2)
        self.eng = binding.engine(self.ext_shape, num_views, device=device)
        plane_floats = d1 * (d2 // 2) * 2
        dev = torch_device if torch_device is not None else torch.device("cpu")
        mk = lambda: torch.zeros(self.h * plane_floats, dtype=torch.float32, device=dev)
        self.send_lo, self.send_hi, self.recv_lo, self.recv_hi = mk(), mk(), mk(), mk()
        # exchange buffers in host memory (gloo has no device send / recv) under an engine on the GPU: the plane
        # copies are device <-> host then; on the host emulation "device" memory is host memory either way
        self.host_staging = (not self.send_lo.is_cuda) and binding.backend_name().startswith("hip")
        # exchange buffers on the device: everything the exchange does is ORDERED ON THE ENGINE'S STREAM (plane copies
        # enqueued there, torch copies / send / recv issued with it current, their completion awaited by the stream,
        # not by the host) and the engine does not drain its stream before calling back - no host wait in a sweep
        self.ext = None
        if self.send_lo.is_cuda:
            self.ext = torch.cuda.ExternalStream(self.eng.stream(), device=dev)
        self.error = None
        # the poison word: a 1-element int32 tensor the collective can reduce in place (device path), or the
        # engine's own word read / merged through the host (host buffers)
        self.flag = torch.zeros(1, dtype=torch.int32, device=dev)
        if self.ext is not None:
            self.eng.bind_poison(self.flag.data_ptr())
        self.eng.set_halo_hook(self._exchange, drain=self.ext is None, post=self.dist is not None)
        self.eng.set_halo_planes(self.h)  # the halo planes are the neighbours': no pass of this engine computes them

    # --- data in / out: the caller hands over its own planes [z0, z0 + nz) ----------------------------------
    def _extend(self, local, fill):
        import numpy as np
        out = np.full(self.ext_shape, np.float32(fill), np.float32)
        out[self.h:self.h + self.nz] = local
        return out

    def set_view(self, v, image_local, weights_local, kernel1, kernel2):
        if max(kernel1.shape[0], kernel2.shape[0]) // 2 > self.h:
            raise ValueError("halo mode: PSF deeper than the halo this driver was created for")
        for k in (kernel1, kernel2):  # refused here, not by an engine that throws in the middle of a sweep
            if not self.eng.would_be_direct(k.shape):
                raise ValueError("halo mode: a PSF of extents %s is not held in the direct dim0 form (at most 33 "
                                 "planes along dim0)" % (tuple(k.shape),))
        self.eng.set_view(v, self._extend(image_local, 1.0), self._extend(weights_local, 0.0), kernel1, kernel2)

    def set_psi(self, psi_local):
        self.eng.set_psi(self._extend(psi_local, float(psi_local.flat[0])))

    def get_psi(self):
        return self.eng.get_psi()[self.h:self.h + self.nz].copy()

    # --- the exchange, called by the engine before every dim0 leg ------------------------------------------------
    def _exchange(self, spectrum, view, conv):
        # (after an error this rank goes on exchanging - whatever its planes hold - so that its peers' receives
        # and collectives stay matched; run() raises on every rank at the end)
        try:
            body = self._merge_poison if conv >= 2 else self._exchange_body
            if self.ext is not None:
                with self.torch.cuda.stream(self.ext):
                    body(spectrum, False)
            else:
                body(spectrum, True)
        except Exception as ex:  # a Python exception cannot cross the C frames of the engine
            if self.error is None:
                self.error = ex

    def _merge_poison(self, spectrum, host_waits):
        d = self.dist
        if self.ext is not None:  # the bound word itself, in stream order
            d.all_reduce(self.flag, op=d.ReduceOp.MAX)
            return
        self.flag[0] = self.eng.poison_get()
        d.all_reduce(self.flag, op=d.ReduceOp.MAX)
        self.eng.poison_merge(int(self.flag[0]))

    def _exchange_body(self, spectrum, host_waits):
        e, h, nz, hs = self.eng, self.h, self.nz, self.host_staging
        e.copy_planes(spectrum, h, h, self.send_lo.data_ptr(), True, hs, wait=False)        # my first h planes -> lower
        e.copy_planes(spectrum, nz, h, self.send_hi.data_ptr(), True, hs, wait=host_waits)  # my last h planes -> upper
        if self.dist is None:  # one rank: its own neighbour both ways (cyclic)
            self.recv_hi.copy_(self.send_lo)
            self.recv_lo.copy_(self.send_hi)
        else:
            d = self.dist
            lower, upper = (self.rank - 1) % self.world, (self.rank + 1) % self.world
            # (tags keep the two messages of a pair apart on gloo when lower == upper; the order of the receives
            # matches the order of the peer's sends for backends that match in issue order)
            ops = [d.P2POp(d.isend, self.send_lo, lower, tag=0), d.P2POp(d.isend, self.send_hi, upper, tag=1),
                   d.P2POp(d.irecv, self.recv_hi, upper, tag=0), d.P2POp(d.irecv, self.recv_lo, lower, tag=1)]
            for r in d.batch_isend_irecv(ops):
                r.wait()  # device tensors: the CURRENT stream (the engine's) waits, not the host
        # (the halo planes are read by kernels on the same stream: no wait; host buffers are re-used by the next
        # exchange, which drains the stream first)
        e.copy_planes(spectrum, 0, h, self.recv_lo.data_ptr(), False, hs, wait=False)
        e.copy_planes(spectrum, nz + h, h, self.recv_hi.data_ptr(), False, hs, wait=False)

    def run(self, iterations, lam, min_value):
        self.eng.iterate(iterations, lam, min_value)
        self.eng.sync()
        failed = self.error is not None
        if self.dist is not None:  # a rank whose exchange failed makes every rank raise
            t = self.torch.tensor([1 if failed else 0], dtype=self.torch.int32, device=self.flag.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            failed = int(t[0]) != 0
        if self.error is not None:
            raise self.error
        if failed:
            raise RuntimeError("halo mode: the exchange failed on another rank")

    def close(self):
        self.eng.set_halo_hook(None)
        if self.ext is not None:
            self.eng.bind_poison(None)
        self.eng.close()
