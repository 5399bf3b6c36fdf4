"""ID-range sharding across the GPUs of one node.

One process per GPU; shard s owns the contiguous internal ids [s*n, (s+1)*n) with its own graph
(cos_params.id_base = s*n), every query goes to every shard, each shard runs walk + local exact rerank
and emits its local top-k as (global id, cosine score).  The ONLY exchange step on the path is one
all-gather of those lists (RCCL over xGMI when the backend is "nccl"; "gloo" in the CPU tests) followed
by the S-way merge kernel (cos_merge_topk_device).  The reference has no sharding; parity for S > 1 is
defined against the oracle running the same S-shard scheme.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_total: int, world: int, rank: int):
    """Contiguous id range of shard `rank` (ids are allocated sequentially, collection.rs:451-468)."""
    base = n_total // world
    lo = rank * base
    hi = n_total if rank == world - 1 else lo + base
    return lo, hi


def allgather_topk(ids: torch.Tensor, scores: torch.Tensor, counts: torch.Tensor, out_ids=None, out_scores=None, out_counts=None):
    """ids/scores [B][k], counts [B] of this shard -> gathered [S][B][k], [S][B][k], [S][B] on every rank.
    Payload per rank: B*k*8 + B*4 bytes (20 KB at B=256, k=10): latency-bound, far below xGMI bandwidth."""
    world = dist.get_world_size()
    B, k = ids.shape
    if out_ids is None:
        out_ids = torch.empty(world, B, k, dtype=ids.dtype, device=ids.device)
        out_scores = torch.empty(world, B, k, dtype=scores.dtype, device=scores.device)
        out_counts = torch.empty(world, B, dtype=counts.dtype, device=counts.device)
    # outputs viewed as the concatenation along dim 0 (the layout both the nccl and the gloo backend accept)
    dist.all_gather_into_tensor(out_ids.view(world * B, k), ids.contiguous())
    dist.all_gather_into_tensor(out_scores.view(world * B, k), scores.contiguous())
    dist.all_gather_into_tensor(out_counts.view(world * B), counts.contiguous())
    return out_ids, out_scores, out_counts


def merge_topk_device(g_ids, g_scores, g_counts, out_ids, out_scores, out_counts, device_index: int, stream: int = 0):
    """S-way merge on the GPU: total_cmp score desc, larger id first on ties (same rule as everywhere else)."""
    from . import _lib
    S, B, k = g_ids.shape
    _lib.check(_lib.lib().cos_merge_topk_device(g_ids.data_ptr(), g_scores.data_ptr(), g_counts.data_ptr(), S, B, k,
                                                out_ids.data_ptr(), out_scores.data_ptr(), out_counts.data_ptr(), device_index, stream))


# ---- packed exchange: one collective per batch -----------------------------------------------------------------
def packed_words(B: int, k: int) -> int:
    """4-byte words of one shard's packed result record [ids B*k | scores B*k | counts B]."""
    return B * (2 * k + 1)


def packed_views(buf: torch.Tensor, B: int, k: int):
    """Views (ids u32-as-int32 [B][k], scores f32 [B][k], counts int32 [B]) into one packed int32 record, so the
    search kernels write straight into the buffer that is exchanged."""
    assert buf.dtype == torch.int32 and buf.numel() == packed_words(B, k) and buf.is_contiguous()
    ids = buf[: B * k].view(B, k)
    scores = buf[B * k: 2 * B * k].view(torch.float32).view(B, k)
    counts = buf[2 * B * k:]
    return ids, scores, counts


def allgather_packed(buf: torch.Tensor, out: torch.Tensor | None = None):
    """ONE all-gather of the packed record: [words] -> [S][words].  Three small collectives per batch are
    latency-bound (tens of microseconds each on xGMI); one keeps the exchange step off the critical path."""
    world = dist.get_world_size()
    if out is None:
        out = torch.empty(world * buf.numel(), dtype=buf.dtype, device=buf.device)
    dist.all_gather_into_tensor(out.view(-1), buf)
    return out.view(world, buf.numel())


def merge_topk_packed_device(g_packed, B: int, k: int, out_ids, out_scores, out_counts, device_index: int, stream: int = 0):
    from . import _lib
    S = g_packed.shape[0]
    _lib.check(_lib.lib().cos_merge_topk_packed_device(g_packed.data_ptr(), S, B, k, out_ids.data_ptr(), out_scores.data_ptr(),
                                                       out_counts.data_ptr(), device_index, stream))


def global_topk_by_score(ids_global: torch.Tensor, scores: torch.Tensor, k: int) -> torch.Tensor:
    """Recall bookkeeping for S shards (not on the query path): every rank contributes its [Q][k'] candidate ids (global)
    with their exact scores; returns the [Q][k] ids of the best scores over all shards, identical on every rank."""
    world = dist.get_world_size()
    all_ids = [torch.zeros_like(ids_global) for _ in range(world)]
    all_s = [torch.zeros_like(scores) for _ in range(world)]
    dist.all_gather(all_ids, ids_global.contiguous())
    dist.all_gather(all_s, scores.contiguous())
    ci, cs = torch.cat(all_ids, 1), torch.cat(all_s, 1)
    return torch.gather(ci, 1, cs.topk(k, dim=1).indices)
