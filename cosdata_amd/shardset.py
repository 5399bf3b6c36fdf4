"""Host-side mirror of the shard-set entry points of include/cosdata_hip.h.

ShardSet          every shard in this process (the reference's single-process host): batch_search() fans a query batch out to
                  the S resident shards and returns the merged global top-k — walk + rerank per shard, ONE all-gather of the
                  packed records (RCCL), merge kernel; everything below the call is C++/HIP.
ProcessShardSet   one process per GPU (bench.py under torchrun): the process owns one shard; exchange_device() is the
                  per-launch all-gather + merge on the caller's stream.  torch.distributed is used ONLY to hand rank 0's
                  ncclUniqueId to the other ranks.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np

from . import _lib
from ._lib import CosdataError, check

UNIQUE_ID_BYTES = 128


class ShardSet:
    def __init__(self, shards: Sequence["HNSWIndex"]):  # noqa: F821
        self.shards = list(shards)
        arr = (C.c_void_p * len(self.shards))(*[s._h for s in self.shards])
        self._h = C.c_void_p()
        check(_lib.lib().cos_shardset_create(arr, len(self.shards), 0, len(self.shards), None, C.byref(self._h)))
        self.dim = self.shards[0].dim

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            _lib.lib().cos_shardset_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def batch_search(self, queries, top_k: int):
        """[B][dim] raw f32 -> merged (global ids [B][k], exact cosine scores [B][k], counts [B])"""
        q = np.ascontiguousarray(np.atleast_2d(queries), dtype=np.float32)
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise CosdataError(_lib.ERR_INVALID, f"queries must be [B][{self.dim}] f32, got shape {tuple(q.shape)}")
        B = q.shape[0]
        ids = np.full((B, top_k), 0xFFFFFFFF, np.uint32)
        scores = np.zeros((B, top_k), np.float32)
        counts = np.zeros(B, np.uint32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        check(_lib.lib().cos_shardset_search_batch(self._h, p(q), B, top_k, p(ids), p(scores), p(counts)))
        return ids, scores, counts


class ProcessShardSet:
    """One local shard of a world-size shard set.  Construction is COLLECTIVE: every rank must call it, and every rank gets the
    same outcome — either all ranks hold a working communicator, or all of them raise (so a caller can fall back together
    instead of leaving some ranks blocked in a collective the others never enter)."""

    def __init__(self, index, rank: int, world: int, local_rank: int, dist=None):
        import torch
        self._h = C.c_void_p()
        if world <= 1:
            arr = (C.c_void_p * 1)(index._h)
            check(_lib.lib().cos_shardset_create(arr, 1, 0, 1, None, C.byref(self._h)))
            return
        dev = f"cuda:{local_rank}"
        # 1. rank 0 obtains the ncclUniqueId; [ok flag | 128 id bytes] goes to everybody
        msg = np.zeros(1 + UNIQUE_ID_BYTES, np.uint8)
        err0 = ""
        if rank == 0:
            rc = _lib.lib().cos_shardset_unique_id(msg[1:].ctypes.data_as(C.c_void_p))
            msg[0] = 1 if rc == _lib.OK else 0
            if rc != _lib.OK:
                err0 = _lib.lib().cos_last_error_string().decode("utf-8", "replace")
        t = torch.from_numpy(msg).to(dev)
        dist.broadcast(t, 0)
        msg = t.cpu().numpy()
        if msg[0] != 1:
            raise CosdataError(_lib.ERR_HIP, "rank 0 could not create an ncclUniqueId" + (": " + err0 if err0 else ""))
        # 2. every rank joins the communicator; the outcome is agreed with a MIN all-reduce
        self._uid = np.ascontiguousarray(msg[1:])
        arr = (C.c_void_p * 1)(index._h)
        rc = _lib.lib().cos_shardset_create(arr, 1, rank, world, self._uid.ctypes.data_as(C.c_void_p), C.byref(self._h))
        err = _lib.lib().cos_last_error_string().decode("utf-8", "replace") if rc != _lib.OK else ""
        ok = torch.tensor([1 if rc == _lib.OK else 0], device=dev, dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) != 1:
            self.close()
            raise CosdataError(rc if rc != _lib.OK else _lib.ERR_HIP, "shard set creation failed on " + ("this rank: " + err if err else "another rank"))

    def exchange_device(self, packed_ptr: int, B: int, k: int, gathered_ptr: int, out_ids_ptr: int, out_scores_ptr: int, out_counts_ptr: int,
                        stream: int = 0):
        check(_lib.lib().cos_shardset_exchange_device(self._h, C.c_void_p(packed_ptr), B, k, C.c_void_p(gathered_ptr), C.c_void_p(out_ids_ptr),
                                                      C.c_void_p(out_scores_ptr), C.c_void_p(out_counts_ptr), C.c_void_p(stream)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            _lib.lib().cos_shardset_destroy(self._h)
            self._h = C.c_void_p()
