from sow_amd.utils import closest_factorization, pad_matrix, qr_weight, svd_weight, unpad_matrix  # noqa: F401
