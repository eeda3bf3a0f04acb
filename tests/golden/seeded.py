"""Platform-independent seeded weights / configs for fixtures (test infrastructure).

``seeded_state_dict`` fills a reference-layout state dict from per-tensor CPU
generators seeded by crc32(name), so the 127 M-parameter base model used by the
full-size golden can be regenerated bit-identically on the GPU box instead of being
committed.  Scales are chosen so that logits are O(0.1..1) (random ``N(0, 0.02)``
weights give logits ~1e-3, which makes absolute tolerances meaningless).
"""
from __future__ import annotations

import zlib
from typing import Dict

import torch


def layoutlmv3_config(size: str = "tiny") -> dict:
    """backbone_config dicts; 'base' mirrors microsoft/layoutlmv3-base's public config.json
    (hyper-parameters only), 'large' is BASELINE config 4."""
    common = dict(
        model_type="layoutlmv3", pad_token_id=1, bos_token_id=0, eos_token_id=2, type_vocab_size=1,
        max_2d_position_embeddings=1024, has_relative_attention_bias=True, has_spatial_attention_bias=True,
        rel_pos_bins=32, max_rel_pos=128, rel_2d_pos_bins=64, max_rel_2d_pos=256, visual_embed=True,
        input_size=224, hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
        layer_norm_eps=1e-5, initializer_range=0.02, is_decoder=False, add_cross_attention=False,
        chunk_size_feed_forward=0,
    )
    if size == "tiny":
        common.update(vocab_size=1000, hidden_size=64, num_hidden_layers=2, num_attention_heads=4,
                      intermediate_size=128, max_position_embeddings=66, coordinate_size=11, shape_size=10)
    elif size == "base":
        common.update(vocab_size=50265, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                      intermediate_size=3072, max_position_embeddings=514, coordinate_size=128, shape_size=128)
    elif size == "large":
        common.update(vocab_size=50265, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16,
                      intermediate_size=4096, max_position_embeddings=1026, coordinate_size=171, shape_size=170)
    else:
        raise ValueError(size)
    return common


def lilt_config(size: str = "tiny") -> dict:
    common = dict(
        model_type="lilt", pad_token_id=1, type_vocab_size=1, max_2d_position_embeddings=1024,
        hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, layer_norm_eps=1e-5,
        initializer_range=0.02, channel_shrink_ratio=4, position_embedding_type="absolute",
        is_decoder=False, add_cross_attention=False, chunk_size_feed_forward=0,
    )
    if size == "tiny":
        # H/2 = 96 (decoder width, multiple of 32 for the MFMA pair heads); d = 48, d_layout = 12
        common.update(vocab_size=1000, hidden_size=192, num_hidden_layers=2, num_attention_heads=4,
                      intermediate_size=384, max_position_embeddings=66)
    elif size == "base":
        common.update(vocab_size=50265, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                      intermediate_size=3072, max_position_embeddings=514)
    else:
        raise ValueError(size)
    return common


def peneo_config(backbone_name: str, backbone_config: dict) -> dict:
    """PEneoConfig as a dict with tools/generate_peneo_weights.py:63-74 defaults."""
    return dict(
        model_type="peneo", backbone_name=backbone_name, backbone_config=backbone_config,
        initializer_range=0.02, peneo_decoder_shrink=True, peneo_classifier_num_layers=2,
        peneo_loss_ratio=[1.0, 1.0, 1.0, 1.0, 1.0], peneo_category_weights=[1.0, 10.0, 10.0],
        peneo_ohem_num_positive=-1, peneo_ohem_num_negative=-1, peneo_downstream_speedup_ratio=30.0,
        inference_mode=False,
    )


def _gen(name: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def seeded_fill_(sd: Dict[str, torch.Tensor], seed: int = 0, decoder_gain: float = 2.0) -> Dict[str, torch.Tensor]:
    """Overwrite every floating tensor of ``sd`` in place with seeded values."""
    for name, t in sd.items():
        if not t.is_floating_point() or name.endswith("_loss.weight"):
            continue
        g = _gen(name, seed)
        dec = name.startswith("peneo_decoder.")
        if "LayerNorm" in name or name.endswith("norm.weight") or name.endswith("norm.bias"):
            if name.endswith("weight"):
                v = 1.0 + 0.1 * torch.randn(t.shape, generator=g)
            else:
                v = 0.05 * torch.randn(t.shape, generator=g)
        elif name.endswith(".bias"):
            v = 0.02 * torch.randn(t.shape, generator=g)
        elif "rel_pos" in name:
            v = 0.5 * torch.randn(t.shape, generator=g)
        elif name.endswith("cls_token") or name.endswith("pos_embed"):
            v = 0.02 * torch.randn(t.shape, generator=g)
        elif "embeddings" in name:
            v = 0.05 * torch.randn(t.shape, generator=g)
        else:  # Linear / conv weights: ~1/sqrt(fan_in) keeps activations O(1)
            fan_in = t[0].numel()
            std = (decoder_gain if dec else 1.0) * 0.6 / (fan_in ** 0.5)
            v = std * torch.randn(t.shape, generator=g)
        t.copy_(v.to(t.dtype))
    return sd
