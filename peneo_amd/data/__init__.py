from .synthetic import synthetic_rfund_batch, spots_to_shaking_tag  # noqa: F401
