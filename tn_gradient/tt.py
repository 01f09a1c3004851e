from sow_amd.tt import TensorTrain  # noqa: F401
