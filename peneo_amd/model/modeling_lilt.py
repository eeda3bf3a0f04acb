"""LiLT backbone (BASELINE config 5) — placeholder wired after the LayoutLMv3 path is green."""
from __future__ import annotations

import torch.nn as nn

from .configuration_peneo import LiltConfig


class LiltModel(nn.Module):
    config_class = LiltConfig

    def __init__(self, config: LiltConfig):
        super().__init__()
        raise NotImplementedError("LiLT backbone kernels are being wired (SURVEY §7 step 9)")
