from .decode import decode_peneo, parse_matrix_spots, sample_decode_peneo  # noqa: F401
