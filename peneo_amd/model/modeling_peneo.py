"""PEneoModel — drop-in for the reference's model/modeling_peneo.py on libpeneo_hip kernels.

Same constructor (``PEneoModel(config, backbone_name_or_path=None)``), ``from_pretrained`` /
``save_pretrained`` (inherited from transformers), state-dict keys (``backbone.*``,
``peneo_decoder.*``), forward signature (``input_ids, bbox, orig_bbox, attention_mask, image=None,
**kwargs`` with the five ``*_shaking_tag`` label maps and arbitrary extra keys in kwargs) and
outputs (``PEneoOutput`` or the 6-tuple of ``inference_mode``).
"""
from __future__ import annotations

import inspect
import logging
import math

import torch
import torch.nn as nn
from transformers import PreTrainedModel

from .. import ops
from ..hip import PeneoHipError, load_library
from .backbone_mapping import BACKBONE_MAPPING
from .configuration_peneo import PEneoConfig
from .engine import reset_pending
from .peneo_decoder import PEneoDecoder

logger = logging.getLogger(__name__)


class _CropStage(torch.autograd.Function):
    """hidden[:, lo:hi] -> contiguous [B, N, H] with the model-level dropout (reference :138-165)."""

    @staticmethod
    def forward(ctx, hidden, lo, hi, drop_p, seed):
        B, T, H = hidden.shape
        out = torch.empty((B, hi - lo, H), dtype=hidden.dtype, device=hidden.device)
        ops.copy_rows(hidden[:, lo:hi], out, drop_p=drop_p, drop_seed=seed)
        ctx.meta = (B, T, H, lo, hi, drop_p, seed)
        return out

    @staticmethod
    def backward(ctx, d_out):
        B, T, H, lo, hi, drop_p, seed = ctx.meta
        d_hidden = torch.zeros((B, T, H), dtype=d_out.dtype, device=d_out.device)
        ops.copy_rows(d_out.contiguous(), d_hidden[:, lo:hi], drop_p=drop_p, drop_seed=seed)
        return d_hidden, None, None, None, None


def build_backbone(info, backbone_config: dict):
    """Backbone from a config dict (unknown keys of a hub config.json are dropped)."""
    bcfg = dict(backbone_config)
    bcfg.pop("model_type", None)
    known = set(inspect.signature(info.config.__init__).parameters) | \
        set(inspect.signature(info.config.__mro__[1].__init__).parameters)
    return info.model(info.config(**{k: v for k, v in bcfg.items() if k in known}))


def load_pretrained_backbone(info, name_or_path: str):
    """``info.model.from_pretrained(name_or_path)`` of the reference (model/modeling_peneo.py:58-79) for LOCAL checkpoint
    directories: ``config.json`` + ``model.safetensors`` / ``pytorch_model.bin`` in the HF layout of the backbone
    (keys optionally prefixed ``layoutlmv3.`` / ``lilt.`` / ``backbone.``).  "auto" and unreadable paths resolve to the
    registry's hub name, which needs network access: OSError."""
    import json
    import os

    def read(path):
        if not os.path.isdir(path) or not os.path.isfile(os.path.join(path, "config.json")):
            raise OSError(f"{path!r} is not a local checkpoint directory (config.json + weights); the HuggingFace hub is "
                          f"not reachable from this process")
        with open(os.path.join(path, "config.json")) as f:
            cfg = json.load(f)
        st = os.path.join(path, "model.safetensors")
        if os.path.isfile(st):
            from safetensors.torch import load_file
            state = load_file(st)
        elif os.path.isfile(os.path.join(path, "pytorch_model.bin")):
            state = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu", weights_only=True)
        else:
            raise OSError(f"no model.safetensors / pytorch_model.bin in {path!r}")
        return cfg, state

    if name_or_path == "auto":
        cfg, state = read(info.hf_name)
    else:
        try:
            cfg, state = read(name_or_path)
        except OSError:
            logger.warning(f"Could not load pretrained model from {name_or_path}. "
                           f"Load from {info.hf_name} in huggingface_hub instead.")
            cfg, state = read(info.hf_name)
    backbone = build_backbone(info, cfg)
    own = backbone.state_dict()
    mapped = {}
    for k, v in state.items():
        for prefix in ("backbone.", "layoutlmv3.", "lilt.", ""):
            if k.startswith(prefix) and k[len(prefix):] in own:
                mapped[k[len(prefix):]] = v
                break
    missing = [k for k in own if k not in mapped and not k.endswith("position_ids")]
    if missing:
        logger.warning(f"backbone checkpoint lacks {len(missing)} tensors (left at their initial values): {missing[:8]}")
    if not mapped:
        raise OSError(f"{name_or_path!r} holds no tensor of a {type(backbone).__name__} state dict")
    backbone.load_state_dict(mapped, strict=False)
    backbone._loaded_from_checkpoint = set(mapped)
    return backbone


class PEneoPreTrainedModel(PreTrainedModel):
    config_class = PEneoConfig
    base_model_prefix = "backbone"
    _supports_sdpa = False

    def _init_weights(self, module) -> None:
        if isinstance(module, nn.Linear):
            module.weight.data.normal_(mean=0.0, std=self.config.initializer_range)
            if module.bias is not None:
                module.bias.data.zero_()
        elif isinstance(module, nn.Embedding):
            module.weight.data.normal_(mean=0.0, std=self.config.initializer_range)
            if module.padding_idx is not None:
                module.weight.data[module.padding_idx].zero_()


class PEneoModel(PEneoPreTrainedModel):
    """Visual information extraction model with switchable backbones and the PEneo pair decoder."""

    def __init__(self, config: PEneoConfig, backbone_name_or_path: str = None) -> None:
        super().__init__(config)
        load_library()  # fail loudly right here if libpeneo_hip.so is missing
        self.backbone_name = config.backbone_name
        self.backbone_info = BACKBONE_MAPPING[config.backbone_name]
        if config.backbone_config is None and backbone_name_or_path is None:
            raise ValueError(
                "You cannot initialize a model with a config file that has no backbone config "
                "and without specifying the path to a pretrained model.")
        if backbone_name_or_path is not None:
            # fine-tuning start (reference :58-79): backbone weights from a checkpoint directory, "auto" = the registry's hub
            # name; a path that cannot be read falls back to the hub name like the reference, and with no network that
            # raises OSError instead of silently training from random weights
            self.backbone = load_pretrained_backbone(self.backbone_info, backbone_name_or_path)
            if config.backbone_config is None:
                config.backbone_config = self.backbone.config.to_dict()
        else:
            self.backbone = build_backbone(self.backbone_info, config.backbone_config)
        self.dropout = nn.Dropout(config.backbone_config["hidden_dropout_prob"])
        self.loss_ratio = config.peneo_loss_ratio
        if self.loss_ratio is not None:
            assert len(self.loss_ratio) == 5, "loss_ratio must be a list of 5 elements"
        if "lilt" in self.backbone_name.lower():
            downstream_input_size = (config.backbone_config["hidden_size"]
                                     + config.backbone_config["hidden_size"] // config.backbone_config["channel_shrink_ratio"])
        else:
            downstream_input_size = config.backbone_config["hidden_size"]
        self.peneo_decoder = PEneoDecoder(config=config, input_size=downstream_input_size)
        self._compute_dtype = torch.float32
        self._step = 0
        loaded = None
        if backbone_name_or_path is not None:        # post_init() re-initialises every module: keep the checkpoint's tensors
            loaded = {k: v.detach().clone() for k, v in self.backbone.state_dict().items()
                      if k in self.backbone._loaded_from_checkpoint}
        try:
            self.post_init()
        except AttributeError:  # transformers < 4.x naming
            self.init_weights()
        if loaded is not None:
            self.backbone.load_state_dict(loaded, strict=False)

    # ---- precision of the HIP path ------------------------------------------------------------
    def set_compute_dtype(self, dtype: torch.dtype) -> "PEneoModel":
        """torch.bfloat16: bf16 MFMA inputs / fp32 accumulate (throughput mode);
        torch.float32: exact fp32 MFMA (parity mode, the default)."""
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("compute dtype must be torch.float32 or torch.bfloat16")
        self._compute_dtype = dtype
        self.backbone.compute_dtype = dtype
        return self

    def _init_weights(self, module) -> None:
        self.backbone._init_weights(module)

    def forward(self, input_ids, bbox, orig_bbox, attention_mask, image=None, **kwargs):
        kwargs.update({"input_ids": input_ids, "bbox": bbox, "orig_bbox": orig_bbox, "attention_mask": attention_mask,
                       "image": image})
        reset_pending()   # joins left behind by a backward that raised (engine.py)
        names = [p.name for p in inspect.signature(self.backbone.forward).parameters.values()]
        backbone_kwargs = {n: kwargs.get(n, None) for n in names if n not in ("unused", "kwargs")}
        self.backbone.compute_dtype = self._compute_dtype
        hidden = self.backbone(**backbone_kwargs)[0]

        bbox = kwargs.pop("bbox", None)
        orig_bbox = kwargs.pop("orig_bbox", None)
        attention_mask = kwargs.pop("attention_mask", None)
        seq_len = input_ids.shape[1]
        if self.backbone_info.has_visual_embeds:
            lo, hi = (1, seq_len) if self.backbone_info.add_cls_token else (0, seq_len)
        else:
            lo, hi = (1, hidden.shape[1]) if self.backbone_info.add_cls_token else (0, hidden.shape[1])
        bbox = bbox[:, lo:hi] if bbox is not None else None
        orig_bbox = orig_bbox[:, lo:hi] if orig_bbox is not None else None
        attention_mask = attention_mask[:, lo:hi] if attention_mask is not None else None

        p = self.dropout.p if self.training else 0.0
        self._step += 1
        seed = (int(torch.initial_seed()) * 0x9E3779B1 + self._step * 0x7F4A7C15 + 0x5bd1) & 0xFFFFFFFF
        seq = _CropStage.apply(hidden, lo, hi, p, seed)
        return self.peneo_decoder(sequence_output=seq, bbox=bbox, orig_bbox=orig_bbox, attention_mask=attention_mask,
                                  **kwargs)
