"""The two collectives of the sharded path through the C ABI (tt_allgather_topk, tt_allreduce_grads: SURVEY 8e).

The C entry points take a CALLER-CREATED ncclComm_t.  In a PyTorch host the caller is torch.distributed: its NCCL
(= RCCL on ROCm) process group owns one communicator per device and `ProcessGroupNCCL._comm_ptr()` hands out the raw
handle, so the library's calls and torch's own collectives share ONE communicator (RCCL serialises what is
enqueued on it, whatever the stream).  `Collective` wraps that choice:

  * group backend nccl and a usable handle  -> the C-ABI exports, on whatever HIP stream the caller is on
  * anything else (gloo rehearsals on a shared GPU, a missing `_comm_ptr`) -> torch.distributed's own calls

`RcclComm` creates a communicator through the tt_comm_* helpers instead (hosts without torch.distributed; the
single-rank GPU test), shipping rank 0's 128-byte id with any torch.distributed backend or a user callback.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Optional

import torch

from . import _lib

__all__ = ["Collective", "RcclComm", "rccl_comm_of_group"]


def rccl_comm_of_group(group, device: torch.device) -> Optional[int]:
    """Raw ncclComm_t (as an int) of torch.distributed's NCCL process group for `device`, or None."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None
    pg = group if group is not None else dist.distributed_c10d._get_default_group()
    try:
        if dist.get_backend(pg) != "nccl":
            return None
        backend = pg._get_backend(torch.device("cuda", device.index if device.index is not None else torch.cuda.current_device()))
        with torch.cuda.device(device):
            ptr = int(backend._comm_ptr())
    except Exception:  # noqa: BLE001 -- no such accessor in this torch build, communicator not created yet, ...
        return None
    return ptr or None


class RcclComm:
    """An RCCL communicator created through libtt.so's helpers (tt_comm_unique_id / tt_comm_init_rank)."""

    def __init__(self, world: int, rank: int, device: torch.device, exchange_id: Optional[Callable[[bytes], bytes]] = None):
        """exchange_id(id_bytes_of_this_rank) -> rank 0's id bytes.  Default: torch.distributed.broadcast_object_list
        on the default group (any backend); world == 1 needs no exchange."""
        L = _lib.lib()
        self.world, self.rank, self.device = int(world), int(rank), torch.device(device)
        buf = (C.c_char * 128)()
        if rank == 0:
            _lib.check(L.tt_comm_unique_id(buf))
        raw = bytes(buf)
        if world > 1:
            if exchange_id is not None:
                raw = exchange_id(raw)
            else:
                import torch.distributed as dist
                box = [raw]
                dist.broadcast_object_list(box, src=0)
                raw = box[0]
        self._handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(L.tt_comm_init_rank(C.byref(self._handle), world, raw, rank))

    @property
    def ptr(self) -> int:
        return int(self._handle.value or 0)

    def close(self) -> None:
        if self._handle.value:
            _lib.check(_lib.lib().tt_comm_destroy(self._handle))
            self._handle = C.c_void_p()


class Collective:
    """all-gather of equal byte blocks / in-place summing all-reduce of a flat fp32 buffer on the CURRENT stream.

    CONSTRUCTION IS A COLLECTIVE: with an nccl group the constructor runs two all-reduces and a 16-byte all-gather (the
    transport self-test), so every object that owns one -- `ShardedIndex`, `FusedClipAdam(group=...)`,
    `DataParallelTrainer` -- must be constructed by ALL ranks of the group, in the SAME ORDER, or the job deadlocks; the same
    holds for their searches / steps, as for any collective.  The C-ABI calls share torch.distributed's communicator: RCCL
    executes what is enqueued on one communicator in issue order whatever the stream, so the host must issue collectives in the
    same order on every rank (ShardedIndex puts all of an index's collectives on ONE exchange stream for that reason)."""

    def __init__(self, group=None, device: Optional[torch.device] = None, comm: Optional[RcclComm] = None):
        import torch.distributed as dist
        self.group = group
        self.comm_ptr: Optional[int] = None
        self.via = "none"
        if comm is not None:
            self.comm_ptr, self.world, self.rank, self.via = comm.ptr, comm.world, comm.rank, "rccl-c-abi (own communicator)"
            return
        if dist.is_available() and dist.is_initialized():
            self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
            self.via = f"torch.distributed ({dist.get_backend(group)})"
            if device is not None and device.type == "cuda" and dist.get_backend(group) == "nccl":
                ptr = rccl_comm_of_group(group, device)
                if self._self_test(ptr, device):  # collective: every rank of an nccl group gets here
                    self.comm_ptr, self.via = ptr, "rccl-c-abi (torch.distributed's communicator)"
        else:
            self.world, self.rank = 1, 0

    def _usable(self, ptr: int) -> bool:
        """The handle answers for the world size / rank torch reports (guards against binding a different RCCL)."""
        w, r = C.c_int(-1), C.c_int(-1)
        try:
            _lib.check(_lib.lib().tt_comm_info(C.c_void_p(ptr), C.byref(w), C.byref(r)))
        except Exception:  # noqa: BLE001
            return False
        return w.value == self.world and r.value == self.rank

    def _self_test(self, ptr: Optional[int], device: torch.device) -> bool:
        """One 16-byte all-gather through the C ABI, checked on the host (every rank takes part: a collective).  All
        ranks agree on the outcome -- a second gather, by torch.distributed, of each rank's verdict -- so either every
        rank uses the C-ABI transport or none does."""
        import torch.distributed as dist
        usable = torch.tensor([1 if (ptr is not None and self._usable(ptr)) else 0], dtype=torch.int32, device=device)
        dist.all_reduce(usable, op=dist.ReduceOp.MIN, group=self.group)
        if usable.item() != 1:  # some rank has no handle: nobody takes the C-ABI transport
            return False
        ok = True
        try:
            send = torch.full((2,), self.rank, dtype=torch.int64, device=device)
            recv = torch.full((2 * self.world,), -1, dtype=torch.int64, device=device)
            with torch.cuda.device(device):
                _lib.check(_lib.lib().tt_allgather_topk(C.c_void_p(ptr), send.data_ptr(), recv.data_ptr(), 16,
                                                        torch.cuda.current_stream(device).cuda_stream))
            torch.cuda.synchronize(device)
            want = torch.arange(self.world, dtype=torch.int64).repeat_interleave(2)
            ok = bool(torch.equal(recv.cpu(), want))
        except Exception:  # noqa: BLE001
            ok = False
        verdict = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
        dist.all_reduce(verdict, op=dist.ReduceOp.MIN, group=self.group)
        return bool(verdict.item() == 1)

    def all_gather_blocks(self, send: torch.Tensor, recv: torch.Tensor) -> None:
        """recv (world * send.numel() bytes) <- every rank's `send` block, rank order."""
        if self.comm_ptr is not None:
            with torch.cuda.device(send.device):
                _lib.check(_lib.lib().tt_allgather_topk(C.c_void_p(self.comm_ptr), send.data_ptr(), recv.data_ptr(),
                                                        send.numel() * send.element_size(),
                                                        torch.cuda.current_stream(send.device).cuda_stream))
        elif self.world > 1 or self.via.startswith("torch.distributed"):
            import torch.distributed as dist
            dist.all_gather_into_tensor(recv, send, group=self.group)
        else:
            recv.copy_(send)

    def all_reduce_sum(self, flat: torch.Tensor) -> None:
        if self.comm_ptr is not None:
            with torch.cuda.device(flat.device):
                _lib.check(_lib.lib().tt_allreduce_grads(C.c_void_p(self.comm_ptr), flat.data_ptr(), flat.numel(),
                                                         torch.cuda.current_stream(flat.device).cuda_stream))
        elif self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
