"""Data-parallel plumbing: one process per GPU, documents sharded across ranks, gradients
all-reduced over RCCL/xGMI (``backend="nccl"`` is RCCL on ROCm) from one flat wire buffer laid out in
stage-execution order, in chunks of ~64 MB that go on the wire as the backward completes them
(``FlatGradDataParallel``; the first step learns the order and runs un-overlapped).

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
    """Data parallel for the staged model: ONE flat wire buffer laid out in the order the backward PRODUCES the gradients,
    all-reduced in a few large chunks while the rest of the backward is still running.

    torch's DistributedDataParallel copies every gradient into its bucket from a per-parameter autograd hook; on this
    stack that is 241 `hipMemcpyAsync` calls per step issued by the autograd thread, and the host falls behind the GPU
    (measured with RCCL at a world of one rank: 252 docs/s against 370 without the wrapper, 5 ms of idle gaps per step).
    Here a parameter's post-accumulate hook only counts.  The flat buffer is cut into chunks of about `chunk_mb` of wire
    bytes (few large transfers suit the per-link bound xGMI ring); when the last gradient of chunk c has been stored (and
    chunks 0..c-1 are on the wire: every rank issues the collectives in index order) the chunk is packed by ONE multi-tensor
    copy and all-reduced asynchronously.  The last chunk goes out when the backward has ended (an engine callback armed
    from the outputs); then the buffer is scaled and unpacked by two more multi-tensor launches.

    Layout.  Gradient arrival is not parameter registration order: the decoder's gradients come first, then encoder layers
    L-1 .. 0, then the embedding stage (whose rel-pos tables, patch embedding and LayerNorms are registered AFTER
    encoder.layer.*), and the decoder's held dW1 GEMM completes only when the backward ends (engine.mark_late).  The
    first synchronised step therefore runs with one chunk and records the arrival order; rank 0's order is broadcast and
    the buffer is re-laid-out: arrival order, parameters that never reported next, late parameters last.  From the
    second step on the chunks follow the stages (about four chunks of 64 MB for the 254 MB of LayoutLMv3-base in bf16).

    Wire precision: "bf16" (default on RCCL) halves the bytes; the ranks' gradients are then SUMMED in bf16 by the
    collective (relative error of an 8-rank sum <= 2^-8 * log2(8) per element, tested), "fp32" (PENEO_DP_WIRE=fp32) sums
    exactly like the reference's DistributedDataParallel.  `module`, `no_sync()` and the initial parameter broadcast
    follow DistributedDataParallel."""

    def __init__(self, module: torch.nn.Module, compress: str = "bf16", chunk_mb: Optional[float] = None,
                 wire_dtype: Optional[torch.dtype] = None):
        super().__init__()
        self.module = module
        self.world = dist.get_world_size()
        self.params = [p for p in module.parameters() if p.requires_grad]
        self._index = {id(p): i for i, p in enumerate(self.params)}
        dev = self.params[0].device
        compress = os.environ.get("PENEO_DP_WIRE", compress)
        wire = torch.bfloat16 if (compress == "bf16" and dev.type == "cuda" and dist.get_backend() == "nccl") else torch.float32
        if wire_dtype is not None:                   # tests: the bf16 wire format on a backend that would default to fp32
            wire = wire_dtype
        self.flat = torch.zeros(sum((p.numel() + 7) // 8 * 8 for p in self.params), dtype=wire, device=dev)
        # PENEO_DP_CHUNKS=1: one all-reduce after the backward (no overlap); otherwise chunks of PENEO_DP_CHUNK_MB wire bytes
        self.overlap = os.environ.get("PENEO_DP_CHUNKS", "") != "1"
        self.chunk_bytes = int(float(os.environ.get("PENEO_DP_CHUNK_MB", chunk_mb if chunk_mb is not None else 64)) * (1 << 20))
        self.require_sync = True
        self._armed = False
        self._learned = not self.overlap            # False until the arrival order of one step has been recorded
        self._arrival: List[int] = []
        self._works: list = []
        self._next = 0                               # next chunk to go on the wire
        self.sync_calls = 0                          # synchronised steps so far (tests)
        self.early_calls = 0                         # chunks that went on the wire before the backward had ended (tests)
        self._layout(list(range(len(self.params))), chunked=False)
        if self.overlap:
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._grad_ready)
        with torch.no_grad():                       # every rank starts from rank 0's parameters and buffers
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, 0)
        _invalidate_working_weights()

    # ---- layout -------------------------------------------------------------------------------------------------------
    def _layout(self, order: List[int], chunked: bool, n_tail: int = 0) -> None:
        """Views of the flat buffer in `order`; chunk boundaries at ~chunk_bytes.  The last `n_tail` parameters of `order`
        (late / never reported) always share the final chunk."""
        esz = self.flat.element_size()
        self.order = list(order)
        self.views: List[Optional[torch.Tensor]] = [None] * len(self.params)
        offs, off = {}, 0
        for i in order:
            p = self.params[i]
            offs[i] = off
            self.views[i] = self.flat[off:off + p.numel()].view(p.shape)
            off += (p.numel() + 7) // 8 * 8
        self.chunks: List[tuple] = []                # (parameter indices, first element, one past the last element)
        if not chunked or len(order) < 2:
            self.chunks = [(list(order), 0, off)]
        else:
            cur, lo = [], 0
            body = order[:len(order) - n_tail] if n_tail else order
            for i in body:
                cur.append(i)
                hi = offs[i] + (self.params[i].numel() + 7) // 8 * 8
                if (hi - lo) * esz >= self.chunk_bytes:
                    self.chunks.append((cur, lo, hi))
                    cur, lo = [], hi
            tail = cur + (order[len(order) - n_tail:] if n_tail else [])
            if tail:
                # a small remainder rides with the previous chunk unless it carries the late parameters' tail
                if self.chunks and not n_tail and (off - lo) * esz < self.chunk_bytes // 2:
                    pc, plo, _ = self.chunks.pop()
                    self.chunks.append((pc + tail, plo, off))
                else:
                    self.chunks.append((tail, lo, off))
        self._chunk_of = [0] * len(self.params)
        for c, (idxs, _, _) in enumerate(self.chunks):
            for i in idxs:
                self._chunk_of[i] = c
        self._reset_counts()

    def _reset_counts(self) -> None:
        self._remaining = [len(idxs) for idxs, _, _ in self.chunks]
        self._seen = bytearray(len(self.params))
        self._next = 0

    def _relayout_from_arrival(self) -> None:
        """After the first synchronised step: rank 0's arrival order becomes everybody's layout."""
        late = {i for i, p in enumerate(self.params) if _is_late(p)}
        seen = set(self._arrival)
        early = [i for i in self._arrival if i not in late]
        never = [i for i in range(len(self.params)) if i not in seen and i not in late]
        tail = never + [i for i in range(len(self.params)) if i in late]
        order = torch.tensor(early + tail + [len(tail)], dtype=torch.int64, device=self.flat.device)
        dist.broadcast(order, 0)
        order = order.tolist()
        n_tail = order.pop()
        assert sorted(order) == list(range(len(self.params)))
        self._layout(order, chunked=True, n_tail=n_tail)
        self._learned = True

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
            self._reset_counts()
            self._arrival = []
            torch.autograd.Variable._execution_engine.queue_callback(self.sync_gradients)
        return grad

    # ---- the step -----------------------------------------------------------------------------------------------------
    def _pack(self, c: int, final: bool) -> None:
        # gradients may still be in flight on the model's side streams; an early chunk must not wait for the HELD work
        # (the decoder's dW1 GEMM runs beside the whole encoder backward and normally only the last chunk contains its
        # output).  Decided per launch, not from the layout: if the step that taught the layout could not hold the GEMM
        # (pre-existing .grad tensors, a user hook) its outputs sit in an early chunk, and a later step that does hold it must
        # join it before that chunk is packed.
        idxs = self.chunks[c][0]
        _join_side_streams(held=final or any(_is_late(self.params[i]) for i in idxs))
        src, dst = [], []
        for i in idxs:
            g = self.params[i].grad
            if g is None:
                self.views[i].zero_()
            else:
                src.append(g)
                dst.append(self.views[i])
        if dst:
            torch._foreach_copy_(dst, src)

    def _launch(self, c: int, final: bool) -> None:
        self._pack(c, final)
        _, lo, hi = self.chunks[c]
        self._works.append(dist.all_reduce(self.flat[lo:hi], async_op=True))
        self._next = c + 1

    def _grad_ready(self, param) -> None:
        if not (self._armed and self.require_sync):
            return
        i = self._index[id(param)]
        if self._seen[i]:
            return
        self._seen[i] = 1
        if not self._learned:
            self._arrival.append(i)
            return
        c = self._chunk_of[i]
        self._remaining[c] -= 1
        # collectives are issued in chunk order on every rank; the last chunk always waits for the end of the backward
        while self._next < len(self.chunks) - 1 and self._remaining[self._next] == 0:
            self._launch(self._next, final=False)
            self.early_calls += 1

    # this hook only counts; when it packs a chunk, _pack joins the side streams before any gradient is read (engine.can_defer)
    _grad_ready._peneo_joins_before_read = True

    def sync_gradients(self) -> None:
        """Average the gradients over the ranks (runs by itself at the end of backward())."""
        self._armed = False
        self.sync_calls += 1
        have = [i for i, p in enumerate(self.params) if p.grad is not None]
        while self._next < len(self.chunks):
            self._launch(self._next, final=True)
        for w in self._works:
            w.wait()
        self._works = []
        if self.world > 1:
            self.flat.mul_(1.0 / self.world)         # one pass over the wire buffer (half the bytes of the fp32 gradients)
        if have:
            torch._foreach_copy_([self.params[i].grad for i in have], [self.views[i] for i in have])
        for i, p in enumerate(self.params):
            if p.grad is None:                       # unused on this rank, used elsewhere
                p.grad = self.views[i].to(torch.float32)
        if not self._learned:
            self._relayout_from_arrival()


def _is_late(p) -> bool:
    try:
        from .model.engine import is_late
    except Exception:
        return False
    return is_late(p)


def _join_side_streams(held: bool = True) -> None:
    try:
        from .model.engine import join_pending
    except Exception:
        return
    if torch.cuda.is_available():
        join_pending(held=held)


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
