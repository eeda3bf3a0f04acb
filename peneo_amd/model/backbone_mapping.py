"""Backbone registry — the model-side subset of the reference's ``BACKBONE_MAPPING``
(model/backbone_mapping.py:260-349): same keys, same *order* (tools/generate_peneo_weights.py:26-27
matches names by substring in dict order) and the same flags consumed by the model and the data
pipeline, including the tokenizer fetchers (``tokenizer_fetchers.py``) the dataset needs to rebuild key/value strings.
``processor`` stays None: which tokenizer class reads a checkpoint directory is the caller's business (the reference
loads HF processors there; any tokenizer with ``tokenize`` / ``convert_tokens_to_ids`` / ``pad_token_id`` works with
``peneo_amd.data``); ``image_processor`` names the build's own 224 x 224 normaliser for the LayoutLMv3 entries.  The
LayoutLMv2 / LayoutXLM entries need detectron2 and are registered as unavailable so that asking for them fails with a
clear message."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Callable, Optional

from .configuration_peneo import LayoutLMv3Config, LiltConfig
from .modeling_layoutlmv3 import LayoutLMv3Model
from .modeling_lilt import LiltModel
from .tokenizer_fetchers import (fetcher_LayoutLMv2Tokenizer, fetcher_LayoutLMv3Tokenizer, fetcher_RobertaTokenizer,
                                 fetcher_XLMTokenizer)


def _image_processor(*args, **kwargs):
    """``PEneoImageProcessor`` (late import: ``peneo_amd.data`` imports this package)."""
    from ..data.collator import PEneoImageProcessor
    return PEneoImageProcessor(*args, **kwargs)


@dataclass
class BackboneInfo:
    model: Any
    config: Any
    hf_name: str
    processor: Optional[Any] = None
    image_processor: Optional[Any] = None
    max_token_len: Optional[int] = 512
    add_cls_token: Optional[bool] = False
    add_sep_token: Optional[bool] = False
    has_visual_embeds: Optional[bool] = False
    tokenizer_fetcher: Optional[Callable] = None


class _Unavailable:
    def __init__(self, what: str):
        self.what = what

    def __call__(self, *a, **k):
        raise NotImplementedError(f"{self.what} is not part of the MI355X hot path (needs detectron2; SURVEY §2 row 8)")

    from_dict = __call__


# ! same order as the reference: the weight-generation script depends on it
BACKBONE_MAPPING = {
    "lilt-infoxlm-base": BackboneInfo(model=LiltModel, config=LiltConfig, hf_name="SCUT-DLVCLab/lilt-infoxlm-base",
                                      max_token_len=511, add_cls_token=True, add_sep_token=False, has_visual_embeds=False,
                                      tokenizer_fetcher=fetcher_XLMTokenizer),
    "lilt-roberta-en-base": BackboneInfo(model=LiltModel, config=LiltConfig, hf_name="SCUT-DLVCLab/lilt-roberta-en-base",
                                         max_token_len=511, add_cls_token=True, add_sep_token=False,
                                         has_visual_embeds=False, tokenizer_fetcher=fetcher_RobertaTokenizer),
    "layoutxlm-base": BackboneInfo(model=_Unavailable("LayoutXLM"), config=_Unavailable("LayoutLMv2Config"),
                                   hf_name="microsoft/layoutxlm-base", max_token_len=511, add_cls_token=True,
                                   add_sep_token=False, has_visual_embeds=True, tokenizer_fetcher=fetcher_XLMTokenizer),
    "layoutlmv2-base-uncased": BackboneInfo(model=_Unavailable("LayoutLMv2"), config=_Unavailable("LayoutLMv2Config"),
                                            hf_name="microsoft/layoutlmv2-base-uncased", max_token_len=511,
                                            add_cls_token=True, add_sep_token=False, has_visual_embeds=True,
                                            tokenizer_fetcher=fetcher_LayoutLMv2Tokenizer),
    "layoutlmv3-base-chinese": BackboneInfo(model=LayoutLMv3Model, config=LayoutLMv3Config,
                                            hf_name="microsoft/layoutlmv3-base-chinese", max_token_len=510,
                                            add_cls_token=True, add_sep_token=True, has_visual_embeds=True,
                                            image_processor=_image_processor, tokenizer_fetcher=fetcher_XLMTokenizer),
    "layoutlmv3-base": BackboneInfo(model=LayoutLMv3Model, config=LayoutLMv3Config, hf_name="microsoft/layoutlmv3-base",
                                    max_token_len=510, add_cls_token=True, add_sep_token=True, has_visual_embeds=True,
                                    image_processor=_image_processor, tokenizer_fetcher=fetcher_LayoutLMv3Tokenizer),
}
