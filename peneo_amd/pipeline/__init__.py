from .decode import decode_peneo, parse_matrix_spots, sample_decode_peneo  # noqa: F401
from .evaluation import calculate_detail_KVPE_metric, calculate_KVPE_metric  # noqa: F401
from .evaluate import make_compute_metrics, prediction_loop  # noqa: F401
