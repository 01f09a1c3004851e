from sow_amd.tensor_linear import ComposedLinear, TensorTrainLinear  # noqa: F401
