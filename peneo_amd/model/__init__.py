from .configuration_peneo import LayoutLMv3Config, LiltConfig, PEneoConfig  # noqa: F401
from .modeling_peneo import PEneoModel  # noqa: F401
from .peneo_decoder import HandshakingKernel, HandshakingTaggingScheme, PEneoDecoder, PEneoOutput  # noqa: F401
from .backbone_mapping import BACKBONE_MAPPING, BackboneInfo  # noqa: F401
