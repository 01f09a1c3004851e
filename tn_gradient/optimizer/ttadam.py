from sow_amd.optimizer import TTAdam, TTRAdam  # noqa: F401
