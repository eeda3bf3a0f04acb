"""Import the upstream PEneo reference (read-only at /root/reference) in-process.

Test infrastructure only.  Used by ``make_golden.py`` (in the build container) to
produce the committed golden vectors; nothing on the GPU box imports this module
because /root/reference does not exist there.

The reference pins transformers==4.40.1 and imports ``timm``; this container has
transformers 5.x and no timm.  The shims below restate the handful of removed
helpers (own code, written from the documented 4.40.1 behaviour):

* ``timm.models.layers.to_2tuple``
* ``transformers.modeling_utils.find_pruneable_heads_and_indices`` (never called
  on the PEneo path, only imported)
* ``prune_linear_layer`` / ``apply_chunking_to_forward`` re-exports
* ``RobertaTokenizerFast`` module alias
* ``PreTrainedModel.get_extended_attention_mask`` (4.40.1 positional signature:
  ``(mask, input_shape, device)`` -> ``(1 - mask[:, None, None, :]) * finfo.min``)
* ``PreTrainedModel.get_head_mask(None, n) -> [None] * n``
* ``PreTrainedModel.init_weights`` -> ``post_init``
"""
import importlib.machinery
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("PENEO_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "model"))


def _fake_module(name: str) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, loader=None)
    sys.modules[name] = m
    return m


def install_shims() -> None:
    import torch
    import transformers
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu
    from transformers import PreTrainedModel

    # 1. timm.models.layers.to_2tuple
    if "timm" not in sys.modules:
        timm = _fake_module("timm")
        models = _fake_module("timm.models")
        layers = _fake_module("timm.models.layers")
        layers.to_2tuple = lambda x: x if isinstance(x, tuple) else (x, x)
        timm.models = models
        models.layers = layers

    # 2. find_pruneable_heads_and_indices (imported, never called on this path)
    if not hasattr(mu, "find_pruneable_heads_and_indices"):
        def find_pruneable_heads_and_indices(heads, n_heads, head_size, already_pruned_heads):
            mask = torch.ones(n_heads, head_size)
            heads = set(heads) - already_pruned_heads
            for head in heads:
                head = head - sum(1 if h < head else 0 for h in already_pruned_heads)
                mask[head] = 0
            mask = mask.view(-1).contiguous().eq(1)
            index = torch.arange(len(mask))[mask].long()
            return heads, index
        mu.find_pruneable_heads_and_indices = find_pruneable_heads_and_indices

    # 3-4. helpers that moved to pytorch_utils
    for name in ("prune_linear_layer", "apply_chunking_to_forward"):
        if not hasattr(mu, name):
            setattr(mu, name, getattr(pu, name))
        if not hasattr(transformers, name):
            setattr(transformers, name, getattr(pu, name))

    # 5. tokenization_roberta_fast module alias
    modname = "transformers.models.roberta.tokenization_roberta_fast"
    try:
        importlib.import_module(modname)
    except Exception:
        m = _fake_module(modname)
        try:
            from transformers import RobertaTokenizerFast
        except Exception:  # pragma: no cover
            RobertaTokenizerFast = object
        m.RobertaTokenizerFast = RobertaTokenizerFast

    # 6. attention-mask / head-mask helpers with 4.40.1 semantics
    def get_extended_attention_mask(self, attention_mask, input_shape=None, device=None, dtype=None):
        dtype = self.dtype
        ext = attention_mask[:, None, None, :].to(dtype)
        return (1.0 - ext) * torch.finfo(dtype).min

    def get_head_mask(self, head_mask, num_hidden_layers, is_attention_chunked=False):
        assert head_mask is None
        return [None] * num_hidden_layers

    PreTrainedModel.get_extended_attention_mask = get_extended_attention_mask
    PreTrainedModel.get_head_mask = get_head_mask

    # 7. init_weights() -> post_init()
    _orig_init_weights = PreTrainedModel.init_weights

    def init_weights(self):
        if not hasattr(self, "all_tied_weights_keys"):
            self.post_init()
        else:
            _orig_init_weights(self)

    PreTrainedModel.init_weights = init_weights


_REF = None


def import_reference():
    """Return the reference ``model`` package (PEneoConfig, PEneoModel, ...)."""
    global _REF
    if _REF is not None:
        return _REF
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True
    install_shims()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import model as ref_model  # noqa: E402  (the reference's package is literally `model`)
    _REF = ref_model
    return ref_model
