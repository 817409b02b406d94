"""Multi-GPU sharding of the separation path (one process per GPU, ``torch.distributed``).

The reference has no parallelism of its own; what shards naturally is its tile axis
(``examples/dsd100/separate_dsd.py:114-135``): tiles are independent through the network, and a
tile only sees the 30 frames it covers.  Two modes:

* many independent batches / files: every rank separates its own work and the PCM is gathered --
  what ``bench.py --gpus N`` measures (weak scaling);
* one long file: ranks own contiguous sample ranges cut on the tile grid; each rank separates its
  range plus a halo (cross-fade: the 5 neighbouring tiles; iSTFT: N/hop-1 frames) and keeps the
  interior, so the only exchange is the final gather of PCM (6x smaller than the masked tiles).

No collective other than that final gather is needed.  Backend ``nccl`` (= RCCL over
xGMI) on GPUs, ``gloo`` in the CPU tests.
"""
import numpy as np


def shard_ranges(n_units, world):
    """Contiguous, balanced ranges ``[(lo, hi)] * world`` over ``n_units``."""
    base, extra = divmod(int(n_units), int(world))
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


def plan_long_file(n_samples, world, frame, hop, time_context, overlap):
    """Per-rank plan for one long signal.

    Returns a list of dicts ``{a0, a1, s0, s1}``: the rank separates ``audio[a0:a1]`` (its own range
    plus halo, ``a0`` on the tile grid so that the segment's tiles coincide with the global ones) and
    owns output samples ``[s0, s1)``.  Halo = two tile lengths + the iSTFT overlap, in frames: every
    owned sample then only depends on frames and tiles that exist identically in the segment.
    """
    st = time_context - overlap
    grid = st * hop                                  # samples between tile starts
    frames = -(-int(n_samples) // hop) + 2           # ceil(L/hop)+2, transform.py:309
    n_slots = max(1, -(-frames // st))               # tile-grid slots over the signal
    halo_frames = 2 * time_context + (frame + hop - 1) // hop + 4
    halo_slots = -(-halo_frames // st)
    plan = []
    for r, (k0, k1) in enumerate(shard_ranges(n_slots, world)):
        s0 = 0 if r == 0 else min(n_samples, k0 * grid)
        s1 = n_samples if r == world - 1 else min(n_samples, k1 * grid)
        a0 = max(0, (k0 - halo_slots) * grid)
        a1 = min(n_samples, (k1 + halo_slots) * grid)
        if r == world - 1:
            a1 = n_samples
        plan.append(dict(a0=int(a0), a1=int(a1), s0=int(s0), s1=int(s1)))
    return plan


def separate_long_file(separate_fn, audio, frame, hop, time_context, overlap, group=None, gather_to_all=True):
    """Shard one long signal over the ranks of ``group``.

    ``separate_fn(segment) -> [S, len(segment)]`` (ndarray or torch tensor) is the single-GPU path
    (``Separator.separate`` or the device-tensor ``Network.separate``).  Returns ``[S, L]`` float32/64
    as a torch tensor on every rank (``gather_to_all``) or on rank 0 only (others get ``None``).
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    L = int(audio.shape[0])
    plan = plan_long_file(L, world, frame, hop, time_context, overlap)
    me = plan[rank]
    seg = audio[me['a0']:me['a1']]
    out = separate_fn(seg)
    out = out if torch.is_tensor(out) else torch.from_numpy(np.ascontiguousarray(out))
    own = out[:, me['s0'] - me['a0']:me['s1'] - me['a0']].contiguous()
    if world == 1:
        return own
    # equal-size buffers for the collective: pad to the largest owned range
    longest = max(p['s1'] - p['s0'] for p in plan)
    buf = torch.zeros((own.shape[0], longest), dtype=own.dtype, device=own.device)
    buf[:, :own.shape[1]] = own
    if gather_to_all:
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf, group=group)
    else:
        parts = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
        dist.gather(buf, parts, dst=0, group=group)
        if rank != 0:
            return None
    return torch.cat([parts[r][:, :plan[r]['s1'] - plan[r]['s0']] for r in range(world)], dim=1)


def gather_batches(pcm, group=None):
    """Weak-scaling mode: every rank holds ``pcm [S, L]`` of its own batch; returns ``[world*S, L]``
    on every rank (one all-gather -- the 'final gather' of BASELINE.json)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return pcm
    out = torch.empty((world * pcm.shape[0], pcm.shape[1]), dtype=pcm.dtype, device=pcm.device)
    dist.all_gather_into_tensor(out, pcm.contiguous(), group=group)
    return out


class RcclComm:
    """An RCCL communicator of this process's own -- what ``dcs_gather`` (include/dcs.h) takes.  One rank per GPU; the
    unique id is made by rank 0 and handed to the others by whatever side channel the launcher has
    (``from_process_group`` uses ``torch.distributed``).  libdcs dlopens the same ``librccl.so.1``, so the communicator and
    the collective calls live in one library instance."""

    _rccl = None

    @classmethod
    def _lib(cls):
        import ctypes
        if cls._rccl is None:
            lib = None
            for name in ("librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"):
                try:
                    lib = ctypes.CDLL(name, mode=ctypes.RTLD_GLOBAL)
                    break
                except OSError:
                    continue
            if lib is None:
                raise NotImplementedError("librccl.so not found")
            cls._rccl = lib
        return cls._rccl

    @classmethod
    def unique_id(cls):
        import ctypes
        buf = (ctypes.c_char * 128)()
        rc = cls._lib().ncclGetUniqueId(ctypes.byref(buf))
        if rc != 0:
            raise RuntimeError("ncclGetUniqueId: %d" % rc)
        return bytes(buf)

    def __init__(self, n_ranks, rank, uid, device=None):
        """``device``: the GPU ordinal this rank's libdcs context lives on (a ``Context``, a ``torch.device`` or an int;
        default: torch's current device).  ``ncclCommInitRank`` binds the communicator to the CURRENT HIP device and
        ``dcs_gather`` later selects ``ctx->device``: the two must be the same GPU, so the device is made current here."""
        import ctypes
        import torch
        if device is None:
            device = torch.cuda.current_device()
        elif hasattr(device, "device_index"):            # a deepconvsep_amd.runtime.Context
            device = device.device_index
        elif isinstance(device, torch.device):
            device = device.index if device.index is not None else torch.cuda.current_device()
        self.device = int(device)
        torch.cuda.set_device(self.device)

        class _Uid(ctypes.Structure):
            _fields_ = [("internal", ctypes.c_char * 128)]
        lib = self._lib()
        u = _Uid()
        ctypes.memmove(ctypes.byref(u), uid, 128)
        self._h = ctypes.c_void_p()
        lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _Uid, ctypes.c_int]
        rc = lib.ncclCommInitRank(ctypes.byref(self._h), int(n_ranks), u, int(rank))
        if rc != 0:
            raise RuntimeError("ncclCommInitRank: %d" % rc)
        self.n_ranks, self.rank = int(n_ranks), int(rank)

    @classmethod
    def from_process_group(cls, group=None, device=None):
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        return cls(world, rank, box[0], device=device)

    def gather(self, ctx, shard, full=None, root=-1):
        """``shard``: this rank's contiguous device tensor; ``full``: ``[n_ranks, *shard.shape]`` of the same dtype on the
        receiving rank(s) (allocated if omitted); ``root`` < 0: every rank receives.  Enqueued on the context's stream."""
        import ctypes
        import torch
        from ._lib import check
        if ctx.device_index != self.device:
            raise ValueError("the communicator was made on GPU %d, the context lives on GPU %d" % (self.device, ctx.device_index))
        if not shard.is_contiguous():
            shard = shard.contiguous()
        receives = root < 0 or root == self.rank
        if full is None and receives:
            with torch.cuda.stream(ctx.torch_stream):
                full = torch.empty((self.n_ranks,) + tuple(shard.shape), dtype=shard.dtype, device=shard.device)
        nbytes = shard.numel() * shard.element_size()
        check(ctx._lib.dcs_gather(ctx._h, self._h, ctypes.c_void_p(shard.data_ptr()), nbytes,
                                  ctypes.c_void_p(full.data_ptr()) if full is not None else None, int(root)))
        return full if receives else None

    def close(self):
        if getattr(self, "_h", None):
            self._lib().ncclCommDestroy.argtypes = [__import__("ctypes").c_void_p]
            self._lib().ncclCommDestroy(self._h)
            self._h = None
