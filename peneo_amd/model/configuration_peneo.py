"""PEneoConfig / backbone configs — same fields, defaults and ``model_type`` as the reference
(model/configuration_peneo.py:6-37, model/backbone/layoutlmv3/configuration_layoutlmv3.py:36-86,
model/backbone/lilt/configuration_lilt.py:6-47), so a reference ``config.json`` loads unchanged."""
from __future__ import annotations

from typing import List, Optional

from transformers import PretrainedConfig


class PEneoConfig(PretrainedConfig):
    model_type = "peneo"

    def __init__(
        self,
        backbone_name: Optional[str] = None,
        backbone_config: Optional[dict] = None,
        initializer_range: float = 0.02,
        peneo_decoder_shrink: bool = True,
        peneo_classifier_num_layers: int = 2,
        peneo_loss_ratio: List[float] = [1.0, 1.0, 1.0, 1.0, 1.0],
        peneo_category_weights: List[float] = [1.0, 1.0, 1.0],
        peneo_ohem_num_positive: int = -1,
        peneo_ohem_num_negative: int = -1,
        peneo_downstream_speedup_ratio: float = 1.0,
        inference_mode: bool = False,
        **kwargs,
    ):
        super().__init__(**kwargs)
        if backbone_config is not None and not isinstance(backbone_config, dict):
            backbone_config = backbone_config.to_dict()
        self.backbone_name = backbone_name
        self.backbone_config = backbone_config
        self.initializer_range = initializer_range
        self.peneo_decoder_shrink = peneo_decoder_shrink
        self.peneo_classifier_num_layers = peneo_classifier_num_layers
        self.peneo_category_weights = peneo_category_weights
        self.peneo_loss_ratio = peneo_loss_ratio
        self.peneo_ohem_num_positive = peneo_ohem_num_positive
        self.peneo_ohem_num_negative = peneo_ohem_num_negative
        self.peneo_downstream_speedup_ratio = peneo_downstream_speedup_ratio
        self.inference_mode = inference_mode


class _BackboneConfig(PretrainedConfig):
    """BERT-style hyper-parameters shared by the two backbones."""

    def __init__(self, vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                 intermediate_size=3072, hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
                 max_position_embeddings=512, type_vocab_size=2, initializer_range=0.02, layer_norm_eps=1e-12,
                 pad_token_id=0, max_2d_position_embeddings=1024, is_decoder=False, add_cross_attention=False,
                 chunk_size_feed_forward=0, **kwargs):
        super().__init__(pad_token_id=pad_token_id, **kwargs)
        self.pad_token_id = pad_token_id
        self.vocab_size = vocab_size
        self.hidden_size = hidden_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.intermediate_size = intermediate_size
        self.hidden_act = hidden_act
        self.hidden_dropout_prob = hidden_dropout_prob
        self.attention_probs_dropout_prob = attention_probs_dropout_prob
        self.max_position_embeddings = max_position_embeddings
        self.type_vocab_size = type_vocab_size
        self.initializer_range = initializer_range
        self.layer_norm_eps = layer_norm_eps
        self.max_2d_position_embeddings = max_2d_position_embeddings
        self.is_decoder = is_decoder
        self.add_cross_attention = add_cross_attention
        self.chunk_size_feed_forward = chunk_size_feed_forward
        if hidden_act != "gelu":
            raise ValueError("only the exact-erf 'gelu' activation is implemented (reference configs use it)")
        if is_decoder or add_cross_attention:
            raise ValueError("decoder / cross-attention variants are not supported (reference asserts the same)")


class LayoutLMv3Config(_BackboneConfig):
    model_type = "layoutlmv3"

    def __init__(self, pad_token_id=1, bos_token_id=0, eos_token_id=2, max_2d_position_embeddings=1024,
                 coordinate_size=None, shape_size=None, has_relative_attention_bias=False, rel_pos_bins=32,
                 max_rel_pos=128, has_spatial_attention_bias=False, rel_2d_pos_bins=64, max_rel_2d_pos=256,
                 visual_embed=True, input_size=224, **kwargs):
        super().__init__(pad_token_id=pad_token_id, bos_token_id=bos_token_id, eos_token_id=eos_token_id,
                         max_2d_position_embeddings=max_2d_position_embeddings, **kwargs)
        self.coordinate_size = coordinate_size
        self.shape_size = shape_size
        self.has_relative_attention_bias = has_relative_attention_bias
        self.rel_pos_bins = rel_pos_bins
        self.max_rel_pos = max_rel_pos
        self.has_spatial_attention_bias = has_spatial_attention_bias
        self.rel_2d_pos_bins = rel_2d_pos_bins
        self.max_rel_2d_pos = max_rel_2d_pos
        self.visual_embed = visual_embed
        self.input_size = input_size


class LiltConfig(_BackboneConfig):
    model_type = "lilt"

    def __init__(self, pad_token_id=0, channel_shrink_ratio=4, position_embedding_type="absolute",
                 classifier_dropout=None, **kwargs):
        super().__init__(pad_token_id=pad_token_id, **kwargs)
        self.channel_shrink_ratio = channel_shrink_ratio
        self.position_embedding_type = position_embedding_type
        self.classifier_dropout = classifier_dropout
        if position_embedding_type != "absolute":
            raise ValueError("only absolute position embeddings are implemented for LiLT")
