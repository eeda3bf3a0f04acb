"""Data-parallel plumbing: one process per GPU, documents sharded across ranks, gradients
all-reduced over RCCL/xGMI (``backend="nccl"`` is RCCL on ROCm) in large bf16-compressed buckets that
overlap with the remaining backward stages (the model's autograd stages hand their parameter
gradients to DDP layer by layer, decoder first).

Reference counterpart: the implicit ``torchrun`` + HF Trainer -> accelerate -> DistributedDataParallel
path (README.md:206-218); evaluation metrics are merged there with barrier + all_gather_object
(pipeline/evaluation.py:150-156), mirrored by ``all_gather_counts``.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None) -> tuple:
    """(rank, local_rank, world_size); initialises the default process group when WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:   # PENEO_DIST_BACKEND=gloo: test the multi-process path with several ranks on ONE GPU
            backend = os.environ.get("PENEO_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            return rank, local_rank, world
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if "PENEO_DEVICE" in os.environ:      # all ranks on one device (single-GPU test box)
        local_rank = int(os.environ["PENEO_DEVICE"])
    return rank, local_rank, world


def wrap_data_parallel(model: torch.nn.Module, device_ids=None, bucket_cap_mb: int = 128, compress: str = "bf16"):
    """DDP with settings chosen for point-to-point xGMI: few, large buckets (per-link bound ring), gradient
    buckets aliased to ``.grad`` (no extra copy), bf16 wire format, no per-step buffer broadcast."""
    from torch.nn.parallel import DistributedDataParallel as DDP
    ddp = DDP(model, device_ids=device_ids, broadcast_buffers=False, gradient_as_bucket_view=True,
              bucket_cap_mb=bucket_cap_mb, find_unused_parameters=False)
    if compress == "bf16" and dist.get_backend() == "nccl":
        from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
        ddp.register_comm_hook(None, default_hooks.bf16_compress_hook)
    return ddp


def shard_documents(n_docs: int, rank: int, world: int) -> range:
    """Contiguous, balanced shard of document indices for this rank (documents are independent)."""
    base, rem = divmod(n_docs, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def all_gather_counts(counts: List[int]) -> List[List[int]]:
    """Per-file metric counts from every rank (reference: barrier + all_gather_object)."""
    if not (dist.is_available() and dist.is_initialized()):
        return [counts]
    dist.barrier()
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, counts)
    return out


def max_over_ranks(value: float, device=None) -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
