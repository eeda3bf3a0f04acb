"""Optimizer step of the training loop: fused multi-tensor AdamW on the C ABI
(``peneo_adamw_step``) and the reference's four parameter groups.

Reference: ``PEneoTrainer.create_optimizer`` (pipeline/trainer.py:275-330) — parameters whose name contains
``"peneo_decoder"`` train at ``lr * peneo_downstream_speedup_ratio``; biases and LayerNorm weights get no weight
decay (HF ``Trainer.get_decay_parameter_names``); optimizer = AdamW.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterable, List

import torch
import torch.nn as nn

from . import hip
from .hip import check, lib, ptr, stream
from .model.engine import bump_param_epoch


def decay_parameter_names(model: nn.Module) -> List[str]:
    """Names that receive weight decay: everything except LayerNorm parameters and biases (the HF Trainer rule)."""
    no_decay = set()
    for mod_name, mod in model.named_modules():
        if isinstance(mod, nn.LayerNorm):
            for pn, _ in mod.named_parameters(recurse=False):
                no_decay.add(f"{mod_name}.{pn}" if mod_name else pn)
    return [n for n, _ in model.named_parameters() if n not in no_decay and "bias" not in n]


def peneo_param_groups(model: nn.Module, lr: float, weight_decay: float, speedup_ratio: float) -> List[Dict]:
    """The four groups of pipeline/trainer.py:286-322, in the reference's order."""
    decay = set(decay_parameter_names(model))
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    sel = lambda d, s: [p for n, p in named if (n in decay) == d and ("peneo_decoder" in n) == s]
    return [
        {"params": sel(True, True), "weight_decay": weight_decay, "lr": lr * speedup_ratio},
        {"params": sel(False, True), "weight_decay": 0.0, "lr": lr * speedup_ratio},
        {"params": sel(True, False), "weight_decay": weight_decay, "lr": lr},
        {"params": sel(False, False), "weight_decay": 0.0, "lr": lr},
    ]


class FusedAdamW(torch.optim.Optimizer):
    """AdamW over all groups in one kernel launch per step (fp32 master parameters, contiguous fp32 gradients).
    Learning rates may be changed between steps through ``param_groups`` (schedulers work unchanged).

    State layout is torch.optim.AdamW's (``state[p] = {"step", "exp_avg", "exp_avg_sq"}``), so ``state_dict()`` /
    ``load_state_dict()`` round-trip and a reference ``optimizer.pt`` resumes with its own step counts: the kernel takes
    one global step plus a per-tensor offset (0 unless a checkpoint holds differing counts or a parameter joined late).
    ``state[p]["step"]`` is written back lazily (in ``state_dict()`` and before the device tables are rebuilt), not by
    241 host tensor updates per step.

    ``max_grad_norm`` (None = off): global gradient-norm clipping inside the step, i.e. what HF ``Trainer`` does between
    backward and ``optimizer.step()`` at its default ``max_grad_norm = 1.0`` (the reference trains through it,
    start/run_rfund.py:307-321): ``torch.nn.utils.clip_grad_norm_`` is ~1000 small torch launches over 241 tensors; here
    one more launch sums the squares of all gradients into a device scalar and the AdamW kernel applies
    ``min(1, max_grad_norm / (norm + 1e-6))`` while it reads the gradients (``.grad`` itself is left as it is; a Trainer
    driving this optimizer sets its own ``max_grad_norm`` to 0).  ``last_grad_norm()`` returns the un-clipped norm of the
    last step as a device tensor (no host sync unless the caller reads it)."""

    def __init__(self, params: Iterable, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 max_grad_norm: float = None):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if max_grad_norm is not None and not max_grad_norm > 0:
            raise ValueError("max_grad_norm must be positive (None switches clipping off)")
        self.max_grad_norm = max_grad_norm
        self._sqnorm = None
        self._tables = None
        self._step = 0
        self._entries = []
        self._offsets = []

    # ---- step bookkeeping -------------------------------------------------------------------------------------------
    def _flush_steps(self) -> None:
        """Write the step counts of the tracked tensors into ``self.state`` (torch's format: a CPU float tensor)."""
        for (p, st, _), off in zip(self._entries, self._offsets):
            st["step"] = torch.tensor(float(self._step + off))

    def state_dict(self):
        self._flush_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict) -> None:
        super().load_state_dict(state_dict)
        # self.state was replaced: the cached per-parameter dicts and the device table (moment pointers) are stale
        self._tables, self._entries, self._offsets, self._step = None, [], [], 0

    def __setstate__(self, state) -> None:
        super().__setstate__(state)
        self._tables, self._entries, self._offsets = None, [], []
        self._step = getattr(self, "_step", 0)

    def _wanted(self):
        return [(p, gi) for gi, group in enumerate(self.param_groups) for p in group["params"] if p.grad is not None]

    def _build(self):
        self._flush_steps()                          # a rebuild (new parameters with gradients) keeps everyone's count
        chunk = lib().peneo_adamw_chunk_elems()
        entries, chunk_t, chunk_i, steps = [], [], [], []
        for p, gi in self._wanted():
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                raise hip.PeneoHipError("FusedAdamW needs contiguous fp32 parameters on the GPU")
            st = self.state[p]
            if "exp_avg" not in st:
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            for k in ("exp_avg", "exp_avg_sq"):      # e.g. moments loaded from a checkpoint in another dtype / layout
                if st[k].dtype != torch.float32 or not st[k].is_contiguous() or st[k].device != p.device:
                    st[k] = st[k].to(device=p.device, dtype=torch.float32).contiguous()
            steps.append(int(float(st.get("step", 0))))
            t = len(entries)
            entries.append((p, st, gi))
            for c in range((p.numel() + chunk - 1) // chunk):
                chunk_t.append(t)
                chunk_i.append(c)
        if not entries:
            raise hip.PeneoHipError("FusedAdamW.step(): no parameter has a gradient")
        dev = entries[0][0].device
        # global step = the most common count (all equal in practice), the others ride on per-tensor offsets
        self._step = max(set(steps), key=steps.count)
        self._offsets = [s - self._step for s in steps]
        self._entries = entries
        self._chunk_t = torch.tensor(chunk_t, dtype=torch.int32, device=dev)
        self._chunk_i = torch.tensor(chunk_i, dtype=torch.int32, device=dev)
        self._host = (hip.AdamwTensor * len(entries))()
        self._table = torch.empty(C.sizeof(self._host), dtype=torch.uint8, device=dev)
        self._grads = [None] * len(entries)
        self._tables = True

    def _refresh(self):
        dirty = False
        for k, (p, st, gi) in enumerate(self._entries):
            g = p.grad
            if g is None or not g.is_contiguous() or g.dtype != torch.float32:
                raise hip.PeneoHipError("FusedAdamW: every tracked parameter needs a contiguous fp32 .grad each step")
            group = self.param_groups[gi]
            e = self._host[k]
            vals = (p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                    float(group["lr"]), float(group["weight_decay"]), self._offsets[k])
            if self._grads[k] != vals:
                e.param, e.grad, e.exp_avg, e.exp_avg_sq = vals[:4]
                e.numel, e.lr, e.weight_decay, e.step_offset = p.numel(), vals[4], vals[5], vals[6]
                self._grads[k] = vals
                dirty = True
        if dirty:   # pointer / lr table changed (new .grad tensors, scheduler step): one small H2D copy
            src = torch.frombuffer(memoryview(self._host).cast("B"), dtype=torch.uint8)
            self._table.copy_(src, non_blocking=False)

    def last_grad_norm(self):
        """Global L2 norm of the gradients the last step saw (before clipping); None without max_grad_norm."""
        return None if self._sqnorm is None else self._sqnorm.sum().sqrt().to(torch.float32).reshape(1)

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        if self._tables is not None:                 # the set of parameters with gradients changed: rebuild the tables
            want = self._wanted()
            if len(want) != len(self._entries) or any(a is not b[0] for (a, _), b in zip(want, self._entries)):
                self._tables = None
        if self._tables is None:
            self._build()
        self._refresh()
        self._step += 1
        g0 = self.param_groups[0]
        sq = None
        if self.max_grad_norm is not None:
            if self._sqnorm is None or self._sqnorm.device != self._table.device:
                self._sqnorm = torch.zeros(int(lib().peneo_grad_sqnorm_slots()), dtype=torch.float64, device=self._table.device)
            sq = self._sqnorm
            check(lib().peneo_grad_sqnorm(ptr(self._table), ptr(self._chunk_t), ptr(self._chunk_i), self._chunk_t.numel(),
                                          ptr(sq), stream()), "peneo_grad_sqnorm")
        check(lib().peneo_adamw_step_clip(ptr(self._table), ptr(self._chunk_t), ptr(self._chunk_i), self._chunk_t.numel(),
                                          float(g0["betas"][0]), float(g0["betas"][1]), float(g0["eps"]), self._step,
                                          ptr(sq), float(self.max_grad_norm or 0.0), stream()), "peneo_adamw_step_clip")
        bump_param_epoch()   # the parameters changed in place behind torch's version counters: invalidate the working copies
        return loss
