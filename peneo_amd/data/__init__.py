from .synthetic import synthetic_rfund_batch, spots_to_shaking_tag  # noqa: F401
from .collator import DataCollatorForPEneo, PEneoImageProcessor  # noqa: F401
from .rfund import RFUNDDataset, SIBRDataset  # noqa: F401
