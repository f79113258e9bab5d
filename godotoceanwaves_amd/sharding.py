"""Cascade sharding across the GPUs of a node.

Cascades are independent units (wave_generator.gd:65-85 touches nothing shared between cascades): rank r owns
the global cascades r*C .. r*C+C-1 with all their state (h0, foam, time) and runs the two frame kernels on them;
there is NO data-path collective.  The only exchange is the gather of the finished RGBA16F layers to the consumer
(one collective over RCCL/xGMI), and it is optional: consumers on the owning GPU read in place.
torch is used for device memory, streams and the collective only.

`MapGatherer` is that exchange:
  * only the OWNED layers travel (the reference allocates max(2, C) array layers, water.gd:91; a rank that owns one cascade
    does not ship the spare one), both maps in ONE message per rank (one collective per gather, not two);
  * the collective reads a SNAPSHOT taken in the generator's stream order (device-to-device, 16 B/texel), never the
    live maps, so the next ticks may overwrite them while the bytes are on the wire;
  * it runs on a side stream: compute continues, and only the next snapshot waits for the previous gather;
  * `mode="all"`: all_gather (every rank ends up with every layer);  `mode="root"`: gather to one consumer rank
    (1/world of the receive volume -- SURVEY 8e's "gather to the consumer GPU").
With CPU tensors (the 2-rank gloo tests) the same object degrades to the synchronous collective.
"""


def owned_cascades(rank, world, per_rank):
    """global cascade ids of `rank`: a contiguous block (weak scaling: per-rank work is fixed)"""
    if not (0 <= rank < world) or per_rank < 1:
        raise ValueError(f"bad shard request rank={rank} world={world} per_rank={per_rank}")
    return list(range(rank * per_rank, (rank + 1) * per_rank))


def owner_of(cascade, per_rank):
    """(rank, local layer) that holds global cascade `cascade`"""
    return cascade // per_rank, cascade % per_rank


class MapGatherer:
    def __init__(self, torch, dist, world, rank, disp, norm, owned, mode="all", root=0, overlap=True, compute_stream=None):
        """disp / norm: this rank's [layers >= owned][N][N][4] map tensors (FP16, or their bytes); `owned` leading layers travel.
        compute_stream: the torch stream the generator enqueues on (CUDA tensors); the snapshot is ordered on it."""
        if mode not in ("all", "root"):
            raise ValueError(f"mode {mode!r}: 'all' (all_gather) or 'root' (gather to one rank)")
        if disp.shape != norm.shape or disp.dtype != norm.dtype or not (1 <= owned <= disp.shape[0]):
            raise ValueError("disp / norm must have the same shape and dtype, and 1 <= owned <= layers")
        self.torch, self.dist, self.world, self.rank = torch, dist, world, rank
        self.disp, self.norm, self.owned, self.mode, self.root = disp, norm, owned, mode, root
        self.cuda = disp.is_cuda
        self.overlap = bool(overlap) and self.cuda
        self.compute = compute_stream
        self.comm = torch.cuda.Stream(device=disp.device) if self.cuda else None
        shape = (2, owned) + tuple(disp.shape[1:])  # [map][layer][N][N][4]: one message per rank
        self.snap = torch.empty(shape, dtype=disp.dtype, device=disp.device)
        self.out = None
        if mode == "all" or rank == root:
            self.out = torch.empty((world,) + shape, dtype=disp.dtype, device=disp.device)
        self.work = None
        self.fallback = None  # set (at construction, by all ranks together) when the backend has no gather-to-root: all-gather instead
        self.bytes_sent = self.snap.numel() * self.snap.element_size()
        self.bytes_received = (self.out.numel() * self.out.element_size()) if self.out is not None else 0
        if mode == "root" and world > 1 and dist.is_initialized():
            self._agree_on_gather_support()

    # -- one gather: snapshot now (in compute-stream order), bytes move on the side stream --
    def begin(self):
        torch, dist = self.torch, self.dist
        if not self.cuda:
            self.snap[0].copy_(self.disp[:self.owned])
            self.snap[1].copy_(self.norm[:self.owned])
            self._collective(False)
            return
        compute = self.compute if self.compute is not None else torch.cuda.current_stream(self.disp.device)
        with torch.cuda.stream(compute):
            if self.work is not None:
                self.work.wait()  # stream-level: the snapshot buffer is free once the previous gather has read it
            self.snap[0].copy_(self.disp[:self.owned], non_blocking=True)
            self.snap[1].copy_(self.norm[:self.owned], non_blocking=True)
        ready = torch.cuda.Event()
        ready.record(compute)
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(ready)
            self.work = self._collective(True)
        if not self.overlap:  # the serialised form (for the with / without-overlap comparison): compute waits for the wire
            with torch.cuda.stream(compute):
                self.work.wait()

    def _collective(self, async_op):
        dist = self.dist
        if self.world == 1 and not dist.is_initialized():
            self.out[0].copy_(self.snap)
            return None
        u8 = self.torch.uint8  # the wire carries bytes (RGBA16F bits): every backend takes uint8
        if self.mode == "all":
            # flat views: the concatenating form of all_gather_into_tensor, accepted by both RCCL and gloo
            return dist.all_gather_into_tensor(self.out.view(u8).view(-1), self.snap.view(u8).view(-1), async_op=async_op)
        parts = [self.out[r].view(u8) for r in range(self.world)] if self.rank == self.root else None
        return dist.gather(self.snap.view(u8), gather_list=parts, dst=self.root, async_op=async_op)

    _UNSUPPORTED = ("not supported", "not implemented", "does not support", "unsupported", "no backend type associated")

    def _agree_on_gather_support(self):
        """Decided ONCE, before any map travels, and by ALL ranks together: a tiny gather-to-root probes the backend (a backend without
        gather refuses at the call, on every rank alike: NotImplementedError, or a RuntimeError that says "not supported"), then a MIN
        all-reduce of the outcome makes the ranks agree -- no rank can end up issuing all_gather while another issues gather.  Any other
        exception is a real failure and is raised, not masked."""
        torch, dist = self.torch, self.dist
        probe = torch.zeros(8, dtype=torch.uint8, device=self.snap.device)
        parts = [torch.empty_like(probe) for _ in range(self.world)] if self.rank == self.root else None
        ok, why = 1, ""
        try:
            dist.gather(probe, gather_list=parts, dst=self.root)
        except NotImplementedError as e:
            ok, why = 0, f"{type(e).__name__}: {str(e)[:120]}"
        except RuntimeError as e:
            if not any(m in str(e).lower() for m in self._UNSUPPORTED):
                raise
            ok, why = 0, f"{type(e).__name__}: {str(e)[:120]}"
        flag = torch.tensor([ok], dtype=torch.int32, device=self.snap.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            return
        # the exchange degrades to the all-gather for the whole run -- N times the receive volume, said so in `fallback`
        self.fallback = f"gather-to-root refused by the backend ({why or 'on another rank'}): all_gather instead"
        self.mode = "all"
        if self.out is None:
            self.out = torch.empty((self.world,) + tuple(self.snap.shape), dtype=self.snap.dtype, device=self.snap.device)
        self.bytes_received = self.out.numel() * self.out.element_size()

    def wait(self):
        """host-level: returns when the most recent gather's bytes are in `out`"""
        if self.cuda:
            if self.work is not None:
                self.work.wait()  # the caller's current stream waits for the collective ...
            self.torch.cuda.current_stream(self.disp.device).synchronize()  # ... and the host for that stream
            self.comm.synchronize()

    def maps(self):
        """(displacement, normal) as [world * owned][N][N][4]: entry g is global cascade g (None on a non-root rank in
        mode "root")"""
        if self.out is None:
            return None
        shape = (self.world * self.owned,) + tuple(self.disp.shape[1:])
        return self.out[:, 0].reshape(shape), self.out[:, 1].reshape(shape)
