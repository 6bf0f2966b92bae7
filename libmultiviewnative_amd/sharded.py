"""View-sharded ("simultaneous update") RL driver: one process per GPU, one all-reduce of the
per-view correction per iteration (the reference has no multi-GPU code).

Every rank holds a replica of psi and the stacks of its own views.  Per iteration each rank
computes, from the same psi_k, ``delta = sum_{v in my views} w_v (next_v - psi_k)`` on its GPU
(``mvn_engine_compute_delta``), the deltas are summed over ranks with ONE all-reduce
(``torch.distributed``: backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests), and
``psi_{k+1} = psi_k + delta`` is applied on every rank.  For a single view this equals the
reference's sequential sweep; for several views it is its Jacobi counterpart, and the parity
oracle is the CPU restatement run in the same mode.
"""


def view_partition(num_views, world_size, rank):
    """Contiguous, balanced shard of view indices for `rank` (first ranks take the remainder)."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    base, rem = divmod(num_views, world_size)
    begin = rank * base + min(rank, rem)
    return list(range(begin, begin + base + (1 if rank < rem else 0)))


class SimultaneousDriver:
    """Runs the sharded iteration loop on any engine-like object.

    `engine` needs ``compute_delta(lambda, min_value)``, ``apply_delta()`` and ``sync()``;
    `delta` is the torch tensor aliasing the engine's delta buffer (device memory for the HIP
    engine); `dist` is ``torch.distributed`` (or None for a single rank).
    """

    def __init__(self, engine, delta, dist=None, after_collective=None):
        self.engine = engine
        self.delta = delta
        self.dist = dist
        self.after_collective = after_collective

    def step(self, lambda_, min_value):
        self.engine.compute_delta(lambda_, min_value)
        if self.dist is not None and self.dist.get_world_size() > 1:
            self.engine.sync()  # the delta must be complete before the collective reads it
            self.dist.all_reduce(self.delta, op=self.dist.ReduceOp.SUM)
            if self.after_collective is not None:
                self.after_collective()  # e.g. torch.cuda.current_stream().synchronize()
        self.engine.apply_delta()

    def run(self, iterations, lambda_, min_value):
        for _ in range(iterations):
            self.step(lambda_, min_value)
        self.engine.sync()


class SlabDriver:
    """The reference's sequential view-after-view sweep on several GPUs, volume split in slabs of
    planes: exact single-GPU arithmetic, four all-to-all exchanges per
    (view, iteration).

    `engine` is a ``native.SlabHandle`` (or anything with pack / mid / unpack / sync);
    `a_main, b_main, a_nyq, b_nyq` are torch tensors aliasing its exchange buffers (``a_nyq`` /
    ``b_nyq`` may be None for odd d2); `dist` is ``torch.distributed``.
    """

    def __init__(self, engine, a_main, b_main, a_nyq, b_nyq, dist, after_collective=None):
        self.engine = engine
        self.a_main, self.b_main, self.a_nyq, self.b_nyq = a_main, b_main, a_nyq, b_nyq
        self.dist = dist
        self.after_collective = after_collective

    def _exchange(self, src_main, dst_main, src_nyq, dst_nyq):
        self.engine.sync()  # the buffers must be complete before the collective reads them
        self.dist.all_to_all_single(dst_main, src_main)
        if src_nyq is not None:
            self.dist.all_to_all_single(dst_nyq, src_nyq)
        if self.after_collective is not None:
            self.after_collective()

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
