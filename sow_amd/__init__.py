"""sow_amd -- MI355X-native implementation of the SoW (Sum-of-Weights) low-rank linear hot path of
antoine311200/sow: SoWLinear forward/backward, the periodic accumulate-and-refactor step, the TT
helpers and TT optimizers.  Arithmetic runs in libsow_amd.so (hand-written HIP for gfx950, C ABI in
include/sow_amd.h); this package is the host-side mirror of the reference's Python surface.
"""
from .layer import SoWLinear, SoWParameter  # noqa: F401
from .prepare import SoWConfig, accumulate, load_sow, prepare_sow, reset_optimizer  # noqa: F401
from .tt import TensorTrain  # noqa: F401
from .optimizer import TTAdam, TTSGD, FactorAdamW  # noqa: F401
from .tensor_linear import TensorTrainLinear  # noqa: F401
from .dp import FactorBucket, factor_parameters  # noqa: F401
from .group import group_siblings, ungroup_siblings  # noqa: F401
from .utils import qr_weight, svd_weight, pad_matrix, unpad_matrix, closest_factorization  # noqa: F401

__version__ = "0.1.0"
