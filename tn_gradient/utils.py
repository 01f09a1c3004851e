from sow_amd.utils import (  # noqa: F401
    __colorized_str__, closest_factorization, generate_rank_k, left_unfolding, pad_matrix, perturbe_random, qr_weight,
    randhaar, randuptri, right_unfolding, svd_weight, unfolding, unpad_matrix,
)
