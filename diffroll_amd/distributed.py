"""Batch sharding over the GPUs of one node (one process per GPU, torch.distributed 'nccl' = RCCL).

Each clip's reverse chain is independent (no cross-sample operation anywhere on the path), so the
batch is partitioned contiguously across ranks, weights are replicated, nothing is exchanged inside
the 200-step loop, and the ONLY collective is one all-gather of the finished rolls over xGMI
(<= 3.6 MB per rank: latency bound).  The reference itself never gathers (Lightning only shards the
DataLoader under ``Trainer(gpus=N)``, sampling.py:70); the gather exists so rank 0 can return the
whole batch.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch


def _dist():
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def world() -> Tuple[int, int]:
    d = _dist()
    return (d.get_rank(), d.get_world_size()) if d else (0, 1)


def shard_bounds(n: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of n samples for ``rank``; the first n % world ranks get one extra."""
    base, rem = divmod(n, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_rolls(local: torch.Tensor, group=None, comm: Optional["NativeComm"] = None) -> torch.Tensor:
    """All-gather equally sized local rolls (b, 1, T, 88) into (world*b, 1, T, 88), rank-major: through
    torch.distributed (backend 'nccl' = RCCL), or through the C-ABI's dr_gather when a NativeComm is given."""
    if comm is not None:
        return comm.all_gather(local)
    d = _dist()
    if d is None:
        return local
    ws = d.get_world_size(group)
    local = local.contiguous()
    if local.is_cuda and d.get_backend(group) == "gloo":
        # a gloo group (CPU-only rendezvous, or several ranks sharing one GPU in tests): gather host copies
        parts = [torch.empty(local.shape, dtype=local.dtype) for _ in range(ws)]
        d.all_gather(parts, local.cpu(), group=group)
        return torch.cat(parts, 0).to(local.device)
    out = torch.empty((ws * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    d.all_gather_into_tensor(out, local, group=group)
    return out


class NativeComm:
    """An RCCL communicator reached through the C-ABI (dr_comm_create / dr_gather, include/diffroll_amd.h): what a
    caller WITHOUT torch.distributed uses for the path's one collective.  Here it is created next to an existing
    torch process group only to hand the 128-byte unique id from rank 0 to the others (any side channel would do -
    a file, MPI, a socket); the all-gather itself runs in librccl via the engine library, on the caller's stream."""

    def __init__(self, device: torch.device, rank: Optional[int] = None, world_size: Optional[int] = None,
                 unique_id: Optional[bytes] = None, group=None):
        import ctypes as C
        from . import _cabi
        self.lib = _cabi.load_library()
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:      # 'cuda' == the current device: compare indexed
            self.device = torch.device("cuda", torch.cuda.current_device())
        d = _dist()
        if rank is None or world_size is None:
            rank, world_size = (d.get_rank(group), d.get_world_size(group)) if d else (0, 1)
        if unique_id is None:
            if rank == 0:
                buf = C.create_string_buffer(128)
                if self.lib.dr_comm_unique_id(buf) != 0:
                    raise RuntimeError("dr_comm_unique_id: " + self.lib.dr_comm_last_error().decode())
                unique_id = buf.raw
            if world_size > 1:
                if d is None:
                    raise ValueError("unique_id is required on every rank when no torch process group exists")
                box = [unique_id]
                d.broadcast_object_list(box, src=0, group=group)
                unique_id = box[0]
        assert isinstance(unique_id, (bytes, bytearray)) and len(unique_id) == 128
        self.rank, self.world_size = rank, world_size
        h = C.c_void_p()
        rc = self.lib.dr_comm_create(C.byref(h), bytes(unique_id), world_size, rank, self.device.index or 0)
        if rc != 0:
            raise RuntimeError(f"dr_comm_create failed ({rc}): " + self.lib.dr_comm_last_error().decode())
        self.h = h

    def rccl_version(self) -> int:
        import ctypes as C
        v = C.c_int(0)
        if self.lib.dr_comm_info(self.h, None, None, C.byref(v)) != 0:
            raise RuntimeError(self.lib.dr_comm_last_error().decode())
        return int(v.value)

    def all_gather(self, local: torch.Tensor, engine=None) -> torch.Tensor:
        """(b, 1, T, 88) or (b, T, 88) fp32 on the communicator's device -> (world * b, ...), rank-major.  Synchronous.
        engine (optional, the Engine that produced `local`): its pending fused-kernel time-out, if any, is reported to EVERY
        rank of the gather (dr_gather's status word) - all of them raise, none keeps a result that holds the invalid shard."""
        local = local.contiguous()
        assert local.device == self.device and local.dtype == torch.float32 and local.shape[-1] == 88
        b, T = local.shape[0], local.shape[-2]
        out = torch.empty((self.world_size * b,) + tuple(local.shape[1:]), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.lib.dr_gather(engine.h if engine is not None else None, self.h, local.data_ptr(), out.data_ptr(), b, T,
                                    torch.cuda.current_stream(self.device).cuda_stream)
        if rc != 0:
            raise RuntimeError(f"dr_gather failed ({rc}): " + self.lib.dr_comm_last_error().decode())
        return out

    def close(self):
        if getattr(self, "h", None):
            self.lib.dr_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001
            pass


def unpad_gathered(full: torch.Tensor, n_total: int, world_size: int) -> torch.Tensor:
    """Rank-major gather of shards padded to the largest shard -> the n_total real samples, in order."""
    mx = (n_total + world_size - 1) // world_size
    parts = []
    for r in range(world_size):
        lo, hi = shard_bounds(n_total, r, world_size)
        parts.append(full[r * mx: r * mx + (hi - lo)])
    return torch.cat(parts, 0)


def pad_shard(local: torch.Tensor, n_total: int, world_size: int) -> torch.Tensor:
    """Zero-pad a shard_bounds() shard to the size of the largest one (what every rank contributes to the gather)."""
    mx = (n_total + world_size - 1) // world_size
    pad = mx - local.shape[0]
    if pad:
        local = torch.cat([local, local.new_zeros((pad,) + tuple(local.shape[1:]))], 0)
    return local


def gather_rolls_uneven(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """As gather_rolls for shard_bounds() partitions whose sizes differ by one: pad to the largest
    shard, gather, and drop the padding."""
    d = _dist()
    if d is None:
        return local
    ws = d.get_world_size(group)          # a 1-rank group still goes through the collective
    return unpad_gathered(gather_rolls(pad_shard(local, n_total, ws), group), n_total, ws)


def sample_shard(model, x_T: torch.Tensor, waveform: Optional[torch.Tensor], noise: Optional[torch.Tensor],
                 seed: int, rank: int, world_size: int) -> torch.Tensor:
    """The part of sample_sharded one rank computes: the chain of its contiguous shard of the GLOBAL batch,
    returned as (b_local, 1, T', 88) on the engine's device (b_local may be 0).  Philox noise is keyed by the
    global sample index (first_sample = lo), so the rolls do not depend on the world size."""
    B = x_T.shape[0]
    lo, hi = shard_bounds(B, rank, world_size)
    wav = None if waveform is None else waveform[lo:hi]
    z = None if noise is None else noise[:, lo:hi]
    if hi > lo:
        roll, _ = model.sample(x_T[lo:hi], wav, noise=z, seed=seed, first_sample=lo)
        return roll
    # more ranks than clips: an empty shard with the frame count the other ranks will produce (the model's own rule:
    # trim_spec_roll incl. the 641-frame learned spectrogram of condition='trainable_spec')
    T = model.output_frames(x_T.shape[2], None if waveform is None else waveform.shape[-1])
    return torch.zeros((0, 1, T, x_T.shape[3]), dtype=torch.float32, device=model.engine.device)


def sample_sharded(model, x_T: torch.Tensor, waveform: Optional[torch.Tensor], noise: Optional[torch.Tensor] = None,
                   seed: int = 0, group=None) -> torch.Tensor:
    """Every rank passes the SAME global batch (x_T (B,1,T,88), waveform (B,L), optional injected noise
    (S,B,1,T,88)); each runs its contiguous shard and all ranks return the full (B,1,T',88) result.
    Philox noise is keyed by the global sample index, so the result does not depend on the world size."""
    rank, ws = world()
    roll = sample_shard(model, x_T, waveform, noise, seed, rank, ws)
    return gather_rolls_uneven(roll, x_T.shape[0], group)


def sample_sharded_sequential(model, x_T: torch.Tensor, waveform: Optional[torch.Tensor],
                              noise: Optional[torch.Tensor] = None, seed: int = 0, world_size: int = 2) -> torch.Tensor:
    """The world_size-rank job emulated on ONE device: every rank's sample_shard() is run in turn through the same
    slicing / padding / un-padding code the collective path uses (the all-gather itself is the concatenation of
    the padded shards in rank order).  Used by the GPU tests to hold the N-rank result to the unsharded one when
    only one GPU is leased."""
    B = x_T.shape[0]
    shards = [pad_shard(sample_shard(model, x_T, waveform, noise, seed, r, world_size), B, world_size)
              for r in range(world_size)]
    return unpad_gathered(torch.cat(shards, 0), B, world_size)
