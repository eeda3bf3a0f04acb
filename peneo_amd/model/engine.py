"""Shared runtime pieces of the HIP model path: working-precision weight cache and dropout seeds."""
from __future__ import annotations

import threading
import weakref
from typing import Dict, Optional, Sequence, Tuple

import torch

from .. import ops


# Bumped by optimizers that update parameters through the C ABI (peneo_amd.optim.FusedAdamW): such in-place updates do not
# touch torch's per-tensor version counters, so the caches below also key on this epoch.
PARAM_EPOCH = [0]


def bump_param_epoch() -> None:
    PARAM_EPOCH[0] += 1


class WeightCache:
    """fp32 master parameters -> working-precision copies (cast / concatenated / re-packed on the
    device by libpeneo_hip kernels), refreshed whenever a parameter's ``_version`` changes, i.e.
    after every optimizer step, and reused as-is during evaluation."""

    def __init__(self) -> None:
        self._store: Dict[Tuple, Tuple[Tuple, object]] = {}

    @staticmethod
    def _stamp(params: Sequence[torch.Tensor]) -> Tuple:
        return (PARAM_EPOCH[0],) + tuple((p.data_ptr(), p._version) for p in params)

    def get(self, key: Tuple, params: Sequence[torch.Tensor], build):
        stamp = self._stamp(params)
        hit = self._store.get(key)
        if hit is not None and hit[0] == stamp:
            return hit[1]
        val = build()
        self._store[key] = (stamp, val)
        return val

    def clear(self) -> None:
        self._store.clear()

    # ---- common builders -------------------------------------------------------------------
    def cast(self, name: str, p: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
        """Row-major working copy of one matrix (fp32 mode returns the parameter itself)."""
        if dtype == torch.float32:
            return p.detach()
        return self.get((name, dtype), [p], lambda: ops.cast(p.detach().contiguous(), dtype))

    def cat_rows(self, name: str, ps: Sequence[torch.Tensor], dtype: torch.dtype) -> torch.Tensor:
        """[sum rows, K] working copy of several [rows_i, K] matrices stacked along dim 0."""
        def build():
            rows = sum(p.shape[0] for p in ps)
            out = torch.empty((rows, ps[0].shape[1]), dtype=dtype, device=ps[0].device)
            r = 0
            for p in ps:
                ops.cast(p.detach().contiguous(), dtype, out=out[r:r + p.shape[0]])
                r += p.shape[0]
            return out
        return self.get((name, dtype), list(ps), build)

    def cat_vec(self, name: str, ps: Sequence[Optional[torch.Tensor]], sizes: Sequence[int]) -> torch.Tensor:
        """fp32 concatenation of bias vectors (None -> zeros)."""
        real = [p for p in ps if p is not None]

        def build():
            out = torch.zeros(sum(sizes), dtype=torch.float32, device=real[0].device)
            r = 0
            for p, n in zip(ps, sizes):
                if p is not None:
                    ops.cast(p.detach().contiguous(), torch.float32, out=out[r:r + n])
                r += n
            return out
        return self.get((name, "vec"), real, build)


class DropoutSeeds:
    """Per-forward dropout seeds: seed(site) = f(base, step, site).  The backward of a stage reuses
    the seeds recorded by its forward, so masks are regenerated, never stored."""

    _lock = threading.Lock()
    _step = 0

    def __init__(self, training: bool, p_hidden: float, p_attn: float) -> None:
        self.active = bool(training)
        self.p_hidden = float(p_hidden) if training else 0.0
        self.p_attn = float(p_attn) if training else 0.0
        with DropoutSeeds._lock:
            DropoutSeeds._step += 1
            step = DropoutSeeds._step
        self.base = (int(torch.initial_seed()) * 0x9E3779B1 + step * 0x85EBCA6B) & 0xFFFFFFFF
        self._attn_words = None
        self._attn_words_ready = None

    def prepare_attn_words(self, n_layers: int, B: int, nh: int, T: int, device) -> None:
        """Make the keep bits of the attention dropout of ALL encoder layers (ops.attn_drop_words, one launch, 7 MB per layer
        at 8 documents) on a side stream: called by the embedding stage, whose small kernels leave the GPU mostly idle, so
        the 0.1 ms of integer work is off the critical path; the first attention call waits for its event."""
        if self.p_attn <= 0.0 or self._attn_words is not None:
            return
        from .. import ops
        main = torch.cuda.current_stream(device)
        side = side_stream(device, "rel")
        # the buffer belongs to the MAIN stream's allocator pool (it is read and freed there); the side stream only fills it,
        # behind whatever the main stream had queued on that memory before
        words = torch.empty(ops.attn_drop_words_shape(B, nh, T, n_layers), dtype=torch.int32, device=device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            self._attn_words = ops.attn_drop_words(B, nh, T, self.p_attn, self.seed(7), device, sets=n_layers, out=words)
            self._attn_words_ready = torch.cuda.Event()
            self._attn_words_ready.record(side)

    def attn_words(self, layer: int, n_layers: int, B: int, nh: int, T: int, device):
        """Keep bits of the attention dropout of encoder layer `layer`; they stay with the forward's seeds for the backward.
        None when dropout is off."""
        if self.p_attn <= 0.0:
            return None
        if self._attn_words is None or tuple(self._attn_words.shape[:2]) != (n_layers, B * nh):
            from .. import ops
            self._attn_words = ops.attn_drop_words(B, nh, T, self.p_attn, self.seed(7), device, sets=n_layers)
            self._attn_words_ready = None
        if self._attn_words_ready is not None:
            torch.cuda.current_stream(device).wait_event(self._attn_words_ready)
            self._attn_words_ready = None
        return self._attn_words[layer]

    def seed(self, site: int) -> int:
        x = (self.base ^ (site * 0xC2B2AE35)) & 0xFFFFFFFF
        x = ((x ^ (x >> 15)) * 0x2C1B3C6D) & 0xFFFFFFFF
        x = ((x ^ (x >> 12)) * 0x297A2D39) & 0xFFFFFFFF
        return (x ^ (x >> 15)) & 0xFFFFFFFF


def zeros_like_param(p: torch.Tensor) -> torch.Tensor:
    return torch.zeros(p.shape, dtype=torch.float32, device=p.device)


def zeros_like_params(params, fill_stream=None):
    """{id(p): zeroed fp32 gradient buffer}: views of ONE zero-filled allocation (one fill launch instead of one per
    parameter); every view starts on a 16-byte boundary.  With `fill_stream` the buffer is allocated here (the current stream's
    pool) but zeroed on that stream, and (dict, event) is returned: the user waits for the event before the first write."""
    params = [p for p in params if p is not None]
    if not params:
        return {} if fill_stream is None else ({}, None)
    offs, total = [], 0
    for p in params:
        offs.append(total)
        total += (p.numel() + 3) // 4 * 4
    dev = params[0].device
    if fill_stream is None:
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        return {id(p): flat[o:o + p.numel()].view(p.shape) for p, o in zip(params, offs)}
    flat = torch.empty(total, dtype=torch.float32, device=dev)
    fill_stream.wait_stream(torch.cuda.current_stream(dev))   # behind whatever used this memory before
    with torch.cuda.stream(fill_stream):
        flat.zero_()
        ev = torch.cuda.Event()
        ev.record(fill_stream)
    return {id(p): flat[o:o + p.numel()].view(p.shape) for p, o in zip(params, offs)}, ev


# ---- side streams ------------------------------------------------------------------------------------------------
# All extra HIP streams of the step come from ONE small pool per device.  More than four streams with work in flight
# (ours + RCCL's) make the step collapse on this stack (measured: a fifth stream costs 35-45 % of the throughput even
# with GPU_MAX_HW_QUEUES=8), so the roles can share streams (PENEO_SIDE_STREAMS = 1..3): the decoder's second stage never
# overlaps in time with the encoder's weight gradients or the bias-table reduction.
_SIDE_STREAMS: dict = {}
_ROLE_SLOT = {"wgrad": 0, "rel": 1, "dec": 2}


def side_stream(device, role: str) -> "torch.cuda.Stream":
    import os
    limit = int(os.environ.get("PENEO_SIDE_STREAMS", "2"))
    slot = _ROLE_SLOT.get(role)
    if slot is None:                                   # optional experiments (third decoder stream, encoder groups ...)
        key = (str(device), role)
    else:
        key = (str(device), min(slot, max(limit, 1) - 1))
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]



# ---- step-sized buffers --------------------------------------------------------------------------------------------
# The decoder's three multi-gigabyte buffers of a train step (saved pre-activations, x rows, dz: 4.2 + 0.8 + 4.2 GB at 8 x 511
# tokens) are kept between steps instead of going back to torch's allocator.  With per-step allocation the allocator reaches its
# steady state only in the third or fourth step (the previous step's graph is still alive while the next one allocates), and on
# some boxes the driver needs 30 - 100 ms of host time for one such hipMalloc (memory handed out cleared): a training loop pays that
# once, a 10-step measurement after two warm-up steps saw it as 18 - 24 ms per step instead of 16 (profiles/r05_pair_saved.txt).
# acquire() hands out a free buffer of at least the size asked for (a view of a flat byte buffer) or allocates one; release()
# returns it once every launch that touches it has been issued or ordered behind the main stream (stream order does the rest).
# A buffer that is never released (a forward without its backward) is simply dropped with its graph.
_BIG_FREE: dict = {}


def big_acquire(tag: str, shape, dtype, device):
    """-> (tensor of `shape` / `dtype`, handle for big_release)."""
    import math
    nbytes = int(math.prod(shape)) * dtype.itemsize
    free = _BIG_FREE.setdefault((tag, str(device)), [])
    best = None
    for i, b in enumerate(free):
        if b.numel() >= nbytes and (best is None or b.numel() < free[best].numel()):
            best = i
    base = free.pop(best) if best is not None else torch.empty(-(-nbytes // (2 << 20)) * (2 << 20), dtype=torch.uint8, device=device)
    if best is None:
        free.clear()                                  # (a larger request: the smaller buffers of this tag will not be used again)
    return base[:nbytes].view(dtype).view(shape), base


def big_release(tag: str, base) -> None:
    free = _BIG_FREE.setdefault((tag, str(base.device)), [])
    if len(free) < 2 and all(b is not base for b in free):
        free.append(base)


def big_clear(device=None) -> int:
    """Drop the step-sized buffers kept between steps (all devices, or one) and return how many bytes that hands back to the caching
    allocator (torch.cuda.empty_cache() then returns them to the driver).  Called by PEneoDecoder.train(False) -- after training,
    inference on the same model needs none of them (9.2 GB at 8 x 511 tokens, ~35 GB at N = 1023) -- and by anyone who wants the
    memory; the next training step simply allocates them again.  Buffers of a step in flight are not in the pool and are untouched."""
    n = 0
    for key in list(_BIG_FREE):
        if device is None or key[1] == str(device):
            n += sum(b.numel() for b in _BIG_FREE[key])
            del _BIG_FREE[key]
    return n


# ---- deferred joins of side-stream work ------------------------------------------------------------------------------
# A backward stage that put its weight-gradient kernels on a side stream used to end with main.wait_stream(side): the main
# stream (= the activation-gradient critical path) then idles until the last weight gradient of the layer is done (~50 us
# per encoder layer, the QKV wgrad can only start once the attention backward has produced dqkv).  When the gradients are
# freshly assigned (p.grad is None, no grad mode, no hooks: autograd's AccumulateGrad only stores the tensor, no kernel reads
# it) the join can wait: the stage leaves an event here, the NEXT stage's end (one layer of slack) or the end-of-backward
# callback waits for it on the main stream.  Anything that reads gradients on the main stream before the backward has ended
# (the data-parallel wrapper's early pack, a user hook) must call join_pending() first.
#
# State is per DEVICE (one record per device index).  The product runs one process per GPU and calls the model from one
# thread; two models on ONE device share that device's record, which is correct (the joins are stream-level
# waits, whoever issues them) but serialises their side streams.
class _Pending:
    __slots__ = ("next_stage", "held", "armed")

    def __init__(self) -> None:
        self.next_stage: list = []   # (event, kept tensors): joined by the next stage that defers, or by join_pending()
        self.held: list = []         # joins that only the end of the backward (or join_pending) waits for
        self.armed = False           # an end-of-backward callback is queued on the running graph task


_STATE: Dict[int, _Pending] = {}
DEFER_ALLOWED = [True]       # False: every stage joins its side stream before it returns (torch DDP reads .grad in hooks)
# Parameters whose gradient DATA is complete only when the backward has ended (outputs of held joins): id -> weak reference
# (an id can be reused once a model is freed: an entry counts only while its reference still is the asking parameter).
_LATE: Dict[int, "weakref.ref"] = {}
# A post-accumulate hook that calls join_pending() before it reads a gradient carries this attribute (set on the function; a
# bound method forwards the lookup): only such hooks leave deferral allowed.  Anybody else's hook (an optimizer-in-backward
# hook, a logging hook, a second wrapper) reads .grad on the main stream at once, so the stage joins before it returns.
JOINS_BEFORE_READ = "_peneo_joins_before_read"


def mark_late(params) -> None:
    for p in params:
        key = id(p)
        # the weak reference's callback drops the entry when the parameter dies (the table does not grow across model rebuilds);
        # it only deletes an entry that still is this reference: the id may have been re-registered by a new parameter meanwhile
        def _drop(ref, key=key):
            if _LATE.get(key) is ref:
                del _LATE[key]
        _LATE[key] = weakref.ref(p, _drop)


def is_late(p) -> bool:
    r = _LATE.get(id(p))
    if r is None:
        return False
    if r() is p:
        return True
    del _LATE[id(p)]          # the id belonged to a parameter that is gone
    return False


def _state(device: Optional[int] = None) -> _Pending:
    idx = torch.cuda.current_device() if device is None else device
    st = _STATE.get(idx)
    if st is None:
        st = _STATE[idx] = _Pending()
    return st


def can_defer(params) -> bool:
    """True when nothing can read the stage's parameter gradients on the main stream before the backward has ended:
    deferral is allowed, the backward is not being recorded (create_graph makes AccumulateGrad clone), every .grad is unset
    (AccumulateGrad stores the tensor instead of adding into an existing one), and no parameter carries a tensor hook or a
    post-accumulate hook other than the data-parallel wrapper's (which joins before it reads)."""
    if not DEFER_ALLOWED[0] or torch.is_grad_enabled():
        return False
    for p in params:
        if p is None:
            continue
        if p.grad is not None or getattr(p, "_backward_hooks", None):
            return False
        hooks = getattr(p, "_post_accumulate_grad_hooks", None)
        if hooks and not all(getattr(h, JOINS_BEFORE_READ, False) for h in hooks.values()):
            return False
    return True


def _end_of_backward_join(device: int) -> None:
    st = _state(device)
    st.armed = False
    with torch.cuda.device(device):
        join_pending()


def defer_join(side: "torch.cuda.Stream", keep=(), hold: bool = False) -> None:
    """Record `side`'s progress; waits for the events left by EARLIER stages (they are long complete) on the current stream.
    `keep`: tensors the side stream is still READING; they were allocated on the main stream, so they must stay referenced
    until the join (the caching allocator would hand their memory to the next main-stream allocation otherwise).  Never a
    tensor the stage returns as a gradient: the extra reference makes AccumulateGrad clone it on the main stream.
    `hold`: long side work (the decoder's dW1 GEMM) that later stages must NOT wait for: joined by join_pending() only."""
    st = _state()
    ev = torch.cuda.Event()
    ev.record(side)
    if hold:
        st.held.append((ev, tuple(keep)))
    else:
        older = list(st.next_stage)
        st.next_stage.clear()
        st.next_stage.append((ev, tuple(keep)))
        main = torch.cuda.current_stream()
        for e, _ in older:
            main.wait_event(e)
    if not st.armed:
        st.armed = True
        device = torch.cuda.current_device()
        torch.autograd.Variable._execution_engine.queue_callback(lambda: _end_of_backward_join(device))


def join_pending(held: bool = True) -> None:
    """Make the current stream wait for every deferred side-stream event of the current device (`held=False`: all but the
    held ones, whose outputs the caller does not read)."""
    st = _state()
    main = torch.cuda.current_stream()
    for pending in ((st.next_stage, st.held) if held else (st.next_stage,)):
        while pending:
            main.wait_event(pending.pop()[0])


def reset_pending() -> None:
    """Called at the start of every model forward: a backward that raised after defer_join() (an out-of-memory error the
    training loop caught) never ran its end-of-backward callback, so its events would stay queued and `armed` would stay
    set, and no later backward would arm a join again.  Joining here is free when the lists are empty."""
    if not torch.cuda.is_available():
        return
    st = _state()
    if st.next_stage or st.held:
        join_pending()
    st.armed = False
