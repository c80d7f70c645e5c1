"""Row-sharded multi-GPU support: torch.distributed (backend "nccl" == RCCL on ROCm) behind the C ABI's collective hooks.

One process per GPU; every rank holds a contiguous block of rows and calls ``GBRL.step`` with its shard.  The engine
calls the hooks installed here at its exchange points (include/gbrl_hip.h, gbrl_hip_collective): gradient statistics,
min/max or quantile-selection counts, the per-level integer histograms and the leaf sums.  All sums are integer (or
max/min), so every rank grows the same tree bit for bit and the result does not depend on the number of GPUs.
``predict`` needs no exchange: rows are independent.

The hooks receive raw device pointers.  They are wrapped zero-copy as torch tensors through ``__cuda_array_interface__``
and reduced in place with ``torch.distributed.all_reduce`` on the current torch stream; the hook synchronises before it
returns, which is the contract the engine expects.  With ``device=None`` the pointers are host pointers (used by the
world_size-2 gloo tests, which exercise the same hook code on CPU buffers).  With a device and a ``gloo`` process group the
reduction is staged through host memory -- slow, but it lets two ranks share ONE GPU, which RCCL refuses; the GPU tests
use it to run the real row-sharded step with world_size 2 on a single-GPU box.
"""
from __future__ import annotations

import ctypes as C

import numpy as np


class _Coll(C.Structure):
    _I64 = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)
    _fields_ = [("ctx", C.c_void_p), ("world_size", C.c_int), ("rank", C.c_int),
                ("allreduce_sum_i64", _I64), ("allreduce_sum_f64", _I64),
                ("allreduce_max_f32", _I64), ("allreduce_min_f32", _I64)]


class _DevArray:
    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class TorchCollective:
    """Owns the ctypes callbacks (must outlive the model's use of them)."""

    def __init__(self, device=None, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.device, self.group = torch, dist, device, group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.stage_through_host = device is not None and str(dist.get_backend(group)).lower() == "gloo"
        self.calls = 0
        self.bytes = 0
        mk = lambda typestr, np_dtype, op: _Coll._I64(lambda ctx, ptr, n: self._reduce(ptr, n, typestr, np_dtype, op))  # noqa: E731
        self._cbs = (mk("<i8", np.int64, dist.ReduceOp.SUM), mk("<f8", np.float64, dist.ReduceOp.SUM),
                     mk("<f4", np.float32, dist.ReduceOp.MAX), mk("<f4", np.float32, dist.ReduceOp.MIN))
        self.struct = _Coll(None, self.world_size, self.rank, *self._cbs)

    def _tensor(self, ptr, n, typestr, np_dtype):
        if self.device is None:
            buf = (C.c_char * (int(n) * np.dtype(np_dtype).itemsize)).from_address(int(ptr))
            return self.torch.from_numpy(np.frombuffer(buf, dtype=np_dtype))
        return self.torch.as_tensor(_DevArray(ptr, n, typestr), device=self.device)

    def _reduce(self, ptr, n, typestr, np_dtype, op):
        try:
            t = self._tensor(ptr, n, typestr, np_dtype)
            if self.stage_through_host:
                h = t.cpu()
                self.dist.all_reduce(h, op=op, group=self.group)
                t.copy_(h)
            else:
                self.dist.all_reduce(t, op=op, group=self.group)
            if self.device is not None:
                self.torch.cuda.synchronize(self.device)
            self.calls += 1
            self.bytes += int(n) * np.dtype(np_dtype).itemsize
            return 0
        except Exception as e:  # never let an exception unwind through the C frame
            print("gbrl_amd.dist: all_reduce failed:", repr(e), flush=True)
            return -1

    # direct (Python-side) access to the same code path, used by the tests
    def allreduce_sum_i64(self, arr): return self._cbs[0](None, arr.ctypes.data, arr.size)
    def allreduce_sum_f64(self, arr): return self._cbs[1](None, arr.ctypes.data, arr.size)
    def allreduce_max_f32(self, arr): return self._cbs[2](None, arr.ctypes.data, arr.size)
    def allreduce_min_f32(self, arr): return self._cbs[3](None, arr.ctypes.data, arr.size)


_LIB = None


def _lib():
    """The library handle with the argument types of the exchange entry points (include/gbrl_hip.h), opened once."""
    global _LIB
    if _LIB is None:
        from . import LIB_PATH
        lib = C.CDLL(LIB_PATH)
        lib.gbrl_hip_rccl_available.restype = C.c_int
        lib.gbrl_hip_rccl_unique_id.argtypes = [C.c_void_p]
        lib.gbrl_hip_rccl_unique_id.restype = C.c_int
        lib.gbrl_hip_set_rccl.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        lib.gbrl_hip_set_rccl.restype = C.c_int
        lib.gbrl_hip_set_rccl_flags.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint]
        lib.gbrl_hip_set_rccl_flags.restype = C.c_int
        lib.gbrl_hip_set_collective.argtypes = [C.c_void_p, C.c_void_p]
        lib.gbrl_hip_set_collective.restype = C.c_int
        lib.gbrl_hip_last_error.restype = C.c_char_p
        _LIB = lib
    return _LIB


def install_torch_collective(model, device, group=None) -> TorchCollective:
    """Attach torch.distributed hooks to a gbrl_amd.GBRL model.  Keep the returned object alive."""
    coll = TorchCollective(device, group)
    lib = _lib()
    rc = lib.gbrl_hip_set_collective(C.c_void_p(model._handle()), C.cast(C.byref(coll.struct), C.c_void_p))
    if rc != 0:
        raise RuntimeError(lib.gbrl_hip_last_error().decode())
    return coll   # the caller must keep this object alive for as long as the model may call the hooks


def install_rccl(model, device, group=None) -> None:
    """Native exchange: give the model its own RCCL communicator (collectives enqueued on the model's stream, no host
    synchronisation at the exchange points).  Collective call: every rank of `group` must make it.  The unique id is created
    by rank 0 and distributed with torch.distributed.broadcast.  Ranks agree (all-reduce of a flag) before and after the
    communicator is created, so a rank that cannot take part makes EVERY rank raise instead of leaving the others waiting."""
    import torch
    import torch.distributed as dist
    lib = _lib()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    on_dev = str(dist.get_backend(group)).lower() == "nccl"
    where = device if on_dev else "cpu"

    def all_ok(ok: bool) -> bool:
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=where)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        return bool(flag.item())

    buf = (C.c_char * 128)()
    ok = lib.gbrl_hip_rccl_available() == 1
    if ok and rank == 0:
        ok = lib.gbrl_hip_rccl_unique_id(buf) == 0
    if not all_ok(ok):
        raise RuntimeError("native RCCL exchange is not available on every rank: " + (lib.gbrl_hip_last_error() or b"").decode())
    t = torch.frombuffer(bytearray(bytes(buf)), dtype=torch.uint8).clone().to(where)
    dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    idbuf = C.create_string_buffer(bytes(t.cpu().numpy().tobytes()), 128)
    ok = lib.gbrl_hip_set_rccl(C.c_void_p(model._handle()), idbuf, world, rank) == 0
    if not all_ok(ok):
        lib.gbrl_hip_set_collective(C.c_void_p(model._handle()), None)   # drop a communicator that only some ranks hold
        raise RuntimeError("gbrl_hip_set_rccl failed on some rank: " + (lib.gbrl_hip_last_error() or b"").decode())


def install_rccl_single(model) -> None:
    """A world-size-1 RCCL communicator of the model's own, without torch.distributed: the row-sharded code path (statistics / selection
    count / histogram reduce-scatter / winner / leaf-sum exchanges, all enqueued on the model's stream) on ONE GPU.  `bench.py` uses it to
    report what that path costs before any byte crosses xGMI.  The engine drops a world-size-1 communicator unless the caller asks to keep
    it: `gbrl_hip_set_rccl_flags(..., GBRL_HIP_RCCL_KEEP_WORLD1)` (a call argument -- no process-global state is touched)."""
    lib = _lib()
    if lib.gbrl_hip_rccl_available() != 1:
        raise RuntimeError("RCCL is not available: " + (lib.gbrl_hip_last_error() or b"").decode())
    buf = (C.c_char * 128)()
    if lib.gbrl_hip_rccl_unique_id(buf) != 0:
        raise RuntimeError("ncclGetUniqueId failed: " + (lib.gbrl_hip_last_error() or b"").decode())
    if lib.gbrl_hip_set_rccl_flags(C.c_void_p(model._handle()), buf, 1, 0, 1) != 0:   # 1 = GBRL_HIP_RCCL_KEEP_WORLD1
        raise RuntimeError("gbrl_hip_set_rccl_flags failed: " + (lib.gbrl_hip_last_error() or b"").decode())
