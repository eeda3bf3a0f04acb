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


class FlatGradDataParallel(torch.nn.Module):
    """Data parallel for the staged model: ONE flat wire buffer, ONE all-reduce per step, no per-parameter hooks.

    torch's DistributedDataParallel copies every gradient into its bucket from a per-parameter autograd hook; on this
    stack that is 241 `hipMemcpyAsync` calls per step issued by the autograd thread, and the host falls behind the GPU
    (measured with RCCL at a world of one rank: 252 docs/s against 370 without the wrapper, 5 ms of idle gaps per step).
    Here the gradients are packed into a flat bf16 buffer by one multi-tensor copy when the backward pass has finished (an
    end-of-backward engine callback armed from the outputs), summed across ranks by a single RCCL all-reduce (few large
    transfers suit the per-link bound xGMI ring), and unpacked / averaged by two more multi-tensor launches.
    `module`, `no_sync()` and the initial parameter broadcast follow DistributedDataParallel."""

    def __init__(self, module: torch.nn.Module, compress: str = "bf16"):
        super().__init__()
        self.module = module
        self.world = dist.get_world_size()
        self.params = [p for p in module.parameters() if p.requires_grad]
        dev = self.params[0].device
        wire = torch.bfloat16 if (compress == "bf16" and dev.type == "cuda" and dist.get_backend() == "nccl") else torch.float32
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += (p.numel() + 7) // 8 * 8
        self.flat = torch.zeros(total, dtype=wire, device=dev)
        self.views = [self.flat[o:o + p.numel()].view(p.shape) for p, o in zip(self.params, offs)]
        self.require_sync = True
        self._armed = False
        self.sync_calls = 0                          # all-reduces issued so far (tests)
        # Optional overlap (PENEO_DP_CHUNKS=2, off by default: it cannot be measured on one GPU).  The upper half of the flat
        # buffer (by bytes; mostly the decoder and the upper encoder layers) is packed and all-reduced asynchronously while
        # the backward of the rest is still running.  Gradient arrival order is NOT parameter order (the embedding-stage
        # parameters registered after encoder.layer.* get theirs last; the order inside one stage is undefined), so the
        # early collective starts only when EVERY upper-half parameter has reported its gradient this step (counted
        # post-accumulate hooks); parameters that report late, or never, are handled by the fall-back in sync_gradients.
        self._split = None
        self._early = None
        self.early_calls = 0
        self._upper_ids = set()
        chunks = int(os.environ.get("PENEO_DP_CHUNKS", "1"))
        if chunks >= 2 and len(self.params) >= 2:
            half, acc, split = total // 2, 0, len(self.params) - 1
            for i in range(len(self.params) - 1, 0, -1):
                acc += self.params[i].numel()
                if acc >= half:
                    split = i
                    break
            self._split = max(1, split)
            self._split_off = offs[self._split]
            for p in self.params[self._split:]:
                p.register_post_accumulate_grad_hook(self._upper_grad_ready)
        with torch.no_grad():                       # every rank starts from rank 0's parameters and buffers
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, 0)
        _invalidate_working_weights()

    def no_sync(self):
        import contextlib

        @contextlib.contextmanager
        def ctx():
            old, self.require_sync = self.require_sync, False
            try:
                yield
            finally:
                self.require_sync = old
        return ctx()

    def forward(self, *args, **kwargs):
        out = self.module(*args, **kwargs)
        if torch.is_grad_enabled() and self.require_sync:
            vals = out.values() if isinstance(out, dict) else (out if isinstance(out, (tuple, list)) else [out])
            for t in vals:
                if isinstance(t, torch.Tensor) and t.requires_grad:
                    t.register_hook(self._backward_started)
        return out

    def _backward_started(self, grad):
        if not self._armed:
            self._armed = True
            torch.autograd.Variable._execution_engine.queue_callback(self.sync_gradients)
        return grad

    def _pack(self, lo: int, hi: int) -> None:
        _join_side_streams()                        # gradients may still be in flight on the model's side streams
        have = [(p, v) for p, v in zip(self.params[lo:hi], self.views[lo:hi]) if p.grad is not None]
        for p, v in zip(self.params[lo:hi], self.views[lo:hi]):
            if p.grad is None:
                v.zero_()
        if have:
            torch._foreach_copy_([v for _, v in have], [p.grad for p, _ in have])

    def _upper_grad_ready(self, param) -> None:
        if not (self._armed and self.require_sync) or self._early is not None:
            return
        if id(param) in self._upper_ids:             # a second accumulation into the same parameter: counted once
            return
        self._upper_ids.add(id(param))
        if len(self._upper_ids) == len(self.params) - self._split:
            self._pack(self._split, len(self.params))
            self._early = dist.all_reduce(self.flat[self._split_off:], async_op=True)
            self.early_calls += 1

    def sync_gradients(self) -> None:
        """Average the gradients over the ranks (runs by itself at the end of backward())."""
        self._armed = False
        self.sync_calls += 1
        self._upper_ids.clear()
        have = [(p, v) for p, v in zip(self.params, self.views) if p.grad is not None]
        if self._split is not None:
            if self._early is None:                  # the hook did not fire on this rank: same two collectives, same order
                self._pack(self._split, len(self.params))
                self._early = dist.all_reduce(self.flat[self._split_off:], async_op=True)
            self._pack(0, self._split)               # the upper half is already on the wire
            dist.all_reduce(self.flat[:self._split_off])
            self._early.wait()
            self._early = None
        else:
            self._pack(0, len(self.params))
            dist.all_reduce(self.flat)
        if self.world > 1:
            self.flat.mul_(1.0 / self.world)         # one pass over the wire buffer (half the bytes of the fp32 gradients)
        if have:
            torch._foreach_copy_([p.grad for p, _ in have], [v for _, v in have])
        for p, v in zip(self.params, self.views):
            if p.grad is None:                       # unused on this rank, used elsewhere
                p.grad = v.to(torch.float32)


def _join_side_streams() -> None:
    try:
        from .model.engine import join_pending
    except Exception:
        return
    if torch.cuda.is_available():
        join_pending()


def _invalidate_working_weights() -> None:
    """The broadcast wrote the parameters through ``.data``: neither ``_version`` nor the parameter epoch moved, so
    working-precision copies cached by an earlier forward would be stale on ranks != 0."""
    try:
        from .model.engine import bump_param_epoch
    except Exception:                                # plain torch modules (CPU tests) have no weight caches
        return
    bump_param_epoch()


def wrap_data_parallel(model: torch.nn.Module, device_ids=None, bucket_cap_mb: int = 128, compress: str = "bf16",
                       impl: Optional[str] = None):
    """Data-parallel wrapper.  impl = "flat" (default, PENEO_DP_IMPL): FlatGradDataParallel above.  impl = "ddp": torch's
    DistributedDataParallel with settings chosen for point-to-point xGMI: few, large buckets (per-link bound ring),
    gradient buckets aliased to ``.grad``, bf16 wire format, no per-step buffer broadcast."""
    impl = impl or os.environ.get("PENEO_DP_IMPL", "flat")
    if impl == "flat":
        return FlatGradDataParallel(model, compress=compress)
    from torch.nn.parallel import DistributedDataParallel as DDP
    from .model.engine import DEFER_ALLOWED
    DEFER_ALLOWED[0] = False    # DDP's hooks read .grad on the main stream as soon as autograd stores it: no deferred joins
    ddp = DDP(model, device_ids=device_ids, broadcast_buffers=False, gradient_as_bucket_view=True,
              bucket_cap_mb=bucket_cap_mb, find_unused_parameters=False)
    if compress == "bf16" and dist.get_backend() == "nccl":
        from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
        ddp.register_comm_hook(None, default_hooks.bf16_compress_hook)
    _invalidate_working_weights()                    # DDP's constructor broadcast rank 0's parameters in place
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
